// Instances + dispatch of the two-waves-per-SIMD f16x3 edge kernel (gpe_edgegemm_w8_kernel.h): the four launches of an EdgeConv
// layer at the shipped widths with k = 16 —
//   F2  gather forward 200 -> 200            (13 output tiles x 13 K chunks)
//   F3  dense forward 200 -> 150 + max / min (10 x 13; activation rows as fp32 or, for the lazy-dz3 backward, fp16)
//   B3  in-place backward 150 -> 200         (13 x 10; dz3 given, or formed from the stored fp16 activation: LAZY)
//   B2  gathered backward 200 -> 200         (13 x 13)
// ... and the same four for k = 5, the neighbourhood size of the shipped YAMLs (gpe_edgegemm_w8_k5.hip: three points per wave,
// 60-row tiles; fp32 activation rows and the eager dz3 pass — the lazy path is k = 16 only).
// Everything else stays on the single-role kernel (gpe_edgegemm_h3.hip).  GPE_W8 = four digits "F2 F3 B3 B2" (1 = this kernel,
// 0 = the single-role one) or a single 0 / 1 for all four: A/B measurements inside one session.
//
// Three translation units, because the kernels want different compile flags (build.py EXTRA_FLAGS; measured in one session each,
// us per launch at cfg 2, profiles/r05_a_w8_schedules.md):
//                                   F2    F3    B3    B2
//   default scheduler, no fence     392   349   397   449
//   default scheduler, slot fence   371   350   398   424      <- this file (F2, B2)
//   max-ilp, slot fence             382   343   389   471      <- gpe_edgegemm_w8_b3.hip
//   max-ilp, no fence               378   334   401   473      <- gpe_edgegemm_w8_f3.hip
//   single-role kernel (round 4)    443   462   524   499
#include "gpe_edgegemm_w8_kernel.h"

int gpe_w8_launch_f3(const RgParams& p, int stats_nblk, hipStream_t s);     // gpe_edgegemm_w8_f3.hip
int gpe_w8_dispatch_k5(int amode, int emode, int NT, int KCH, const RgParams& p, int stats_nblk, hipStream_t s);   // gpe_edgegemm_w8_k5.hip
int gpe_w8_dispatch_k4(int amode, int emode, int NT, int KCH, const RgParams& p, int stats_nblk, hipStream_t s);   // gpe_edgegemm_w8_k4.hip
int gpe_w8_launch_b3(const RgParams& p, int stats_nblk, hipStream_t s);     // gpe_edgegemm_w8_b3.hip

static int w8_enabled(int kind)
{
    static int tab[4] = {-1, 0, 0, 0};
    if (tab[0] < 0) {
        const char* e = gpe_dbg_env_str("GPE_W8");
        for (int i = 3; i >= 0; --i) {
            int v = 1;
            if (e && strlen(e) == 4) v = e[i] != '0';
            else if (e && strlen(e) == 1) v = e[0] != '0';
            tab[i] = v;
        }
    }
    return tab[kind];
}

// `p` is the re-tiled copy x6_prepare made (R = 64 for k = 16) with the scale words in place.  Returns GPE_ENOTSUP_SHAPE when the
// launch is not on this kernel's menu.
int gpe_edgegemm_w8_dispatch(int amode, int emode, int NT, int KCH, const RgParams& p, int stats_nblk, hipStream_t s)
{
    // k = 16 (64-row tiles), k = 5 (60-row tiles: gpe_edgegemm_w8_k5.hip) or four-row (pseudo-)points (k = 20 as 5 x 4, and the rows
    // of any k > 16 that need nothing per point: gpe_edgegemm_w8_k4.hip); whole points only
    if (!((p.k == 16 && p.R == 64 && !p.pmagic) || (p.k == 5 && p.R == 60 && !p.pmagic) || (p.k == 4 && p.R == 64)) || p.M % p.k)
        return GPE_ENOTSUP_SHAPE;
    // the K-partials of the split tiles lie in the first 384 / 256 bytes of a plane row: bytes every commit rewrites
    const int kbytes = 2 * ((p.K + 3) & ~3);
    if (kbytes < (KCH == 13 ? 384 : 256)) return GPE_ENOTSUP_SHAPE;
    if (p.k != 16) {
        const int kind = (emode == E_EDGE_FWD) ? (amode == A_GATHER ? 0 : 1) : (emode == E_BWD_INPLACE ? 2 : 3);
        if (p.out_half || p.lz_g) return GPE_EINVAL;
        if (!w8_enabled(kind)) return GPE_ENOTSUP_SHAPE;
        return p.k == 5 ? gpe_w8_dispatch_k5(amode, emode, NT, KCH, p, stats_nblk, s) : gpe_w8_dispatch_k4(amode, emode, NT, KCH, p, stats_nblk, s);
    }
    if (p.out_half && !(emode == E_EDGE_FWD && amode == A_DENSE && p.agg)) return GPE_EINVAL;
    if (p.lz_g && !(emode == E_BWD_INPLACE && amode == A_DENSE)) return GPE_EINVAL;
    if (amode == A_GATHER && emode == E_EDGE_FWD && NT == 13 && KCH == 13 && !p.agg && w8_enabled(0))
        return w8_launch<13, 13, A_GATHER, E_EDGE_FWD, 0, false>(p, stats_nblk, s);
    if (amode == A_DENSE && emode == E_EDGE_FWD && NT == 10 && KCH == 13 && p.agg && w8_enabled(1))
        return gpe_w8_launch_f3(p, stats_nblk, s);
    if (amode == A_DENSE && emode == E_BWD_INPLACE && NT == 13 && KCH == 10 && w8_enabled(2))
        return gpe_w8_launch_b3(p, stats_nblk, s);
    if (amode == A_DENSE && emode == E_BWD_GATHER && NT == 13 && KCH == 13 && w8_enabled(3))
        return w8_launch<13, 13, A_DENSE, E_BWD_GATHER, -1, false>(p, stats_nblk, s);
    return GPE_ENOTSUP_SHAPE;
}
