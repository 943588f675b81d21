// B3 instances of the two-waves-per-SIMD edge kernel (in-place backward 150 -> 200; dz3 given, or formed from the stored fp16
// activation): their own translation unit because they are fastest under max-ilp WITH the slot fence (table in gpe_edgegemm_w8.hip).
#define W8_SLOT_FENCE 1
#include "gpe_edgegemm_w8_kernel.h"

int gpe_w8_launch_b3(const RgParams& p, int stats_nblk, hipStream_t s)
{
    return p.lz_g ? w8_launch<13, 10, A_DENSE, E_BWD_INPLACE, -1, true>(p, stats_nblk, s)
                  : w8_launch<13, 10, A_DENSE, E_BWD_INPLACE, -1, false>(p, stats_nblk, s);
}
