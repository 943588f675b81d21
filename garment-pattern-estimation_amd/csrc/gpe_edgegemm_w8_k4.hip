// Four-row (pseudo-)point instances of the two-waves-per-SIMD edge kernel (gpe_edgegemm_w8_kernel.h, KK = 4): BASELINE cfg 4's
// neighbourhood k = 20 (GarmentSegmentPattern3D, /root/reference/models/att/att.yaml with k_neighbors 20) runs every per-point
// launch as five pseudo-points of four rows (gpe_edge_pseudo_setup: P rows through RgParams::pmagic, per-pseudo-point maxima /
// sums into the caller's workspace, folded afterwards) and the in-place backward, which needs nothing per point, simply tiled
// by four.  Rounds 2 - 4 ran these on the single-role kernel.
#include "gpe_edgegemm_w8_kernel.h"

int gpe_w8_dispatch_k4(int amode, int emode, int NT, int KCH, const RgParams& p, int stats_nblk, hipStream_t s)
{
    if (amode == A_GATHER && emode == E_EDGE_FWD && NT == 13 && KCH == 13 && !p.agg)
        return w8_launch<13, 13, A_GATHER, E_EDGE_FWD, 0, false, 4>(p, stats_nblk, s);
    if (amode == A_DENSE && emode == E_EDGE_FWD && NT == 10 && KCH == 13 && p.agg)
        return w8_launch<10, 13, A_DENSE, E_EDGE_FWD, 1, false, 4>(p, stats_nblk, s);
    if (amode == A_DENSE && emode == E_BWD_INPLACE && NT == 13 && KCH == 10)
        return w8_launch<13, 10, A_DENSE, E_BWD_INPLACE, -1, false, 4>(p, stats_nblk, s);
    if (amode == A_DENSE && emode == E_BWD_GATHER && NT == 13 && KCH == 13)
        return w8_launch<13, 13, A_DENSE, E_BWD_GATHER, -1, false, 4>(p, stats_nblk, s);
    return GPE_ENOTSUP_SHAPE;
}
