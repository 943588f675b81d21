// "bf16x6": the three-term bf16 policy of the split-precision single-role edge kernel (gpe_edgegemm_split_kernel.h).
#include "gpe_edgegemm_split_kernel.h"

template <int AMODE, int EMODE>
static int x6_dispatch(int NT, int KCH, const RgParams& p, int stats_nblk, hipStream_t s)
{
    // Only variants that fit the 512-register budget of a lone wave WITHOUT scratch spills are on the menu (checked with
    // scripts/kernel_resources.py): 10 N-tiles always; 13 N-tiles keep 252 + 24 weight registers resident, which leaves too
    // little for the 16-row staging / epilogue state — those shapes stay on the exact-fp32 kernel (caller falls through).
    if (NT == 10 && KCH == 13) return x6_launch<SplitBf16x3, 2, 2, 13, AMODE, EMODE>(p, stats_nblk, s);
    if (NT == 10 && KCH == 10) return x6_launch<SplitBf16x3, 2, 2, 10, AMODE, EMODE>(p, stats_nblk, s);
    if (NT == 13 && KCH == 10 && EMODE == E_EDGE_FWD) return x6_launch<SplitBf16x3, 3, 1, 10, AMODE, EMODE>(p, stats_nblk, s);
    return GPE_ENOTSUP_SHAPE;
}

// Returns 1 and launches when the shape is on this kernel's menu, 0 when the caller should try the next kernel,
// < 0 on a launch error.  `p` comes with the generic tiling (R = (64/k)*k); x6_prepare re-tiles it.
int gpe_edgegemm_x6_try(const RgParams& p_in, int amode, int emode, int stats_nblk, hipStream_t s)
{
    RgParams p;
    GpeFold fold;
    if (!x6_prepare(p_in, amode, emode, stats_nblk, p, fold)) return 0;
    const int NT = (p.N <= 160) ? 10 : 13;
    const int KCH = (p.K <= 160) ? 10 : 13;
    int rc = GPE_EINVAL;
    if (amode == A_GATHER && emode == E_EDGE_FWD) rc = x6_dispatch<A_GATHER, E_EDGE_FWD>(NT, KCH, p, stats_nblk, s);
    else if (amode == A_DENSE && emode == E_EDGE_FWD) rc = x6_dispatch<A_DENSE, E_EDGE_FWD>(NT, KCH, p, stats_nblk, s);
    else if (amode == A_DENSE && emode == E_BWD_INPLACE) rc = x6_dispatch<A_DENSE, E_BWD_INPLACE>(NT, KCH, p, stats_nblk, s);
    else if (amode == A_DENSE && emode == E_BWD_GATHER) rc = x6_dispatch<A_DENSE, E_BWD_GATHER>(NT, KCH, p, stats_nblk, s);
    if (rc == GPE_ENOTSUP_SHAPE) return 0;
    if (rc == GPE_OK) rc = gpe_edge_pseudo_fold(p, fold, s);
    return rc == GPE_OK ? 1 : rc;
}
