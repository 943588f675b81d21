// The dense-A instances of the single-role fused edge GEMM (forward with bias + ReLU + statistics [+ aggregation], in-place
// backward): same template as gpe_edgegemm_sr.hip, a translation unit of its own because these two variants run faster under
// LLVM's max-ILP scheduling strategy (build.py compiles this file with -mllvm -amdgpu-sched-strategy=max-ilp) while the gather
// variants do not.  /root/reference/nn/net_blocks.py:43-47,124-135.
#include "gpe_edgegemm_sr_kernel.h"

int gpe_sr_dispatch_dense(int emode, int NT, int KCH, const RgParams& p, int stats_nblk, hipStream_t s)
{
    if (emode == E_EDGE_FWD) return sr_dispatch<A_DENSE, E_EDGE_FWD>(NT, KCH, p, stats_nblk, s);
    if (emode == E_BWD_INPLACE) return sr_dispatch<A_DENSE, E_BWD_INPLACE>(NT, KCH, p, stats_nblk, s);
    return GPE_EINVAL;
}
