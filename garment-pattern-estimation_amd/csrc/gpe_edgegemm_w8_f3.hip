// F3 instances of the two-waves-per-SIMD edge kernel (dense forward 200 -> 150 + max / min tracking; activation rows fp32 or fp16):
// their own translation unit because they are fastest under max-ilp WITHOUT the slot fence (table in gpe_edgegemm_w8.hip).
#define W8_SLOT_FENCE 0
#include "gpe_edgegemm_w8_kernel.h"

int gpe_w8_launch_f3(const RgParams& p, int stats_nblk, hipStream_t s)
{
    return p.out_half ? w8_launch<10, 13, A_DENSE, E_EDGE_FWD, 2, false>(p, stats_nblk, s)
                      : w8_launch<10, 13, A_DENSE, E_EDGE_FWD, 1, false>(p, stats_nblk, s);
}
