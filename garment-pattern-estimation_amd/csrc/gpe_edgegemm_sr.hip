// Single-role, software-pipelined fused row GEMM for the per-edge MLP of DynamicEdgeConv on gfx950 — the EXACT-fp32 path
// (/root/reference/nn/net_blocks.py:43-47,124-135 forward; its input-gradient half in backward).
//
// Why this shape (profiles/r01_d_coissue_ubench.md): a wave that streams v_mfma_f32_16x16x4_f32 owns its SIMD — a second
// wave on the same SIMD gets no issue slots, so a producer/consumer pair runs at T_consumer + T_producer.  What a wave CAN
// do is issue its OWN memory instructions between its MFMAs: they cost a few issue cycles each (~10 % of the MFMA time in
// total) and their latency hides under the following MFMAs.  So: ONE persistent 256-thread workgroup per CU, one wave per
// SIMD (512 VGPRs each), every wave does everything for its share, and the K loop is hand-pipelined in chunks of 16 k:
//
//   chunk 0        issue ALL global loads of this iteration: the activation rows the epilogue of the PREVIOUS tile needs
//                  (backward) and the <= 16 rows this wave stages for the NEXT tile (dense rows or gathered Q rows)
//   chunks 1..     epilogue of the previous tile, a few rows per chunk, from the C buffer in LDS (bias+ReLU, fp64 BN
//                  statistics, whole-row stores, max/min over each point's messages; or BN/ReLU backward with the stored
//                  activation and per-point sums)
//   last chunks    commit the staged rows to the other A buffer in LDS (ReLU(P_i+Q_j) applied here for the gather)
//   every chunk    prefetch the next chunk's A fragments (ds_read_b128), then 4*AQ*4 + 4*BQ MFMAs
//
// Wave w owns N-tiles [AQ*w, AQ*w+AQ) for all 64 rows plus rows 16w..16w+15 of the BQ left-over N-tiles (so all four
// SIMDs issue the same number of MFMAs), with its weights resident in VGPRs as MFMA B fragments for the whole kernel.
// It stages and finishes rows [w*R/4, (w+1)*R/4) of every tile: R = 4*npw*k rows = whole points, npw*k <= 16.
// LDS: A[2] + C (159 KB at the shipped sizes), two barriers per tile.
#include "gpe_edgegemm_sr_kernel.h"

// dense-A variants live in gpe_edgegemm_sr_dense.hip (compiled with another scheduling strategy)
int gpe_sr_dispatch_dense(int emode, int NT, int KCH, const RgParams& p, int stats_nblk, hipStream_t s);

// ---- k > 16: merging the per-pseudo-point results ------------------------------------------------------------------------
// A point with k > 16 neighbours is processed as f pseudo-points of kq = k / f rows (kq <= 16, so every wave still owns whole
// pseudo-points).  The kernels then write one max / min / argmax / argmin row, or one dP row, per PSEUDO-point into a scratch
// image; these two kernels fold the f rows of each point, in pseudo-point order (first maximum wins, fixed summation order).
__global__ void gpe_sr_fold_agg_kernel(const float* __restrict__ tmx, const float* __restrict__ tmn,
                                       const uint8_t* __restrict__ tamx, const uint8_t* __restrict__ tamn, long npts, int f,
                                       int kq, int C, int ld, float* __restrict__ mx, float* __restrict__ mn,
                                       uint8_t* __restrict__ amx, uint8_t* __restrict__ amn)
{
    const int cq = (C + 3) >> 2;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npts * cq) return;
    const long pt = t / cq;
    const int c = (int)(t - pt * cq) << 2;
    float bx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, bn[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    int ix[4] = {0, 0, 0, 0}, in_[4] = {0, 0, 0, 0};
    for (int q = 0; q < f; ++q) {
        const long o = (pt * f + q) * ld + c;
        const float4 vx = ld4(tmx + o), vn = ld4(tmn + o);
        const uchar4 ax = *reinterpret_cast<const uchar4*>(tamx + o), an = *reinterpret_cast<const uchar4*>(tamn + o);
        const float x4[4] = {vx.x, vx.y, vx.z, vx.w}, n4[4] = {vn.x, vn.y, vn.z, vn.w};
        const int ax4[4] = {ax.x, ax.y, ax.z, ax.w}, an4[4] = {an.x, an.y, an.z, an.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (x4[u] > bx[u]) { bx[u] = x4[u]; ix[u] = q * kq + ax4[u]; }
            if (n4[u] < bn[u]) { bn[u] = n4[u]; in_[u] = q * kq + an4[u]; }
        }
    }
    const long o = pt * ld + c;
    st4(mx + o, make_float4(bx[0], bx[1], bx[2], bx[3]));
    st4(mn + o, make_float4(bn[0], bn[1], bn[2], bn[3]));
    *reinterpret_cast<uchar4*>(amx + o) = make_uchar4(ix[0], ix[1], ix[2], ix[3]);
    *reinterpret_cast<uchar4*>(amn + o) = make_uchar4(in_[0], in_[1], in_[2], in_[3]);
}

__global__ void gpe_sr_fold_sum_kernel(const float* __restrict__ t, long npts, int f, int C, int ld, float* __restrict__ y)
{
    const int cq = (C + 3) >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts * cq) return;
    const long pt = i / cq;
    const int c = (int)(i - pt * cq) << 2;
    float4 s = ld4(t + (pt * f) * ld + c);
    for (int q = 1; q < f; ++q) {
        const float4 v = ld4(t + (pt * f + q) * ld + c);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    st4(y + pt * ld + c, s);
}

// Shared by the single-role kernels (exact fp32 here, the split-precision ones of gpe_edgegemm_split_kernel.h): re-tiles a
// k > 16 launch.  Returns 0 when the shape cannot run that way (the caller falls through to the producer/consumer kernel).
int gpe_edge_pseudo_setup(RgParams& p, bool per_point, int emode, GpeFold& fd)
{
    fd = GpeFold{};
    fd.f = 1;
    if (p.k <= SR_PB) return 1;
    if (!per_point) { p.k = 4; return 1; }
    const long npts = p.M / p.k;
    int best = 0;
    for (int kq = SR_PB; kq >= SR_PB / SR_NPW; --kq)
        if (p.k % kq == 0 && (SR_PB / kq) * kq > (best ? (SR_PB / best) * best : 0)) best = kq;
    if (!best || npts * (p.k / best) >= (1L << 31) || (p.oldagg & 3) || (p.lddp & 3)) return 0;
    fd.f = p.k / best; fd.kq = best; fd.npts = npts;
    const long nps = npts * fd.f;                                     // pseudo-points
    const bool want_agg = emode == E_EDGE_FWD && p.agg, want_dp = emode == E_BWD_GATHER;
    const size_t agg_f = want_agg ? (size_t)nps * p.oldagg : 0, dp_f = want_dp ? (size_t)nps * p.lddp : 0;
    const size_t bytes = (2 * agg_f + dp_f) * sizeof(float) + 2 * agg_f + 256;
    char* ws = (bytes > 256 && bytes <= p.ws.pseudo_bytes) ? p.ws.pseudo : nullptr;
    if (bytes > 256 && !ws) return 0;                                 // no workspace: the producer/consumer kernel runs it
    if (want_agg) {
        fd.mx = p.mx; fd.mn = p.mn; fd.amx = p.oamx; fd.amn = p.oamn;
        p.mx = (float*)ws; p.mn = p.mx + agg_f;
        p.oamx = (uint8_t*)(p.mn + agg_f); p.oamn = p.oamx + agg_f;
    }
    if (want_dp) { fd.dp = p.dP; p.dP = (float*)ws; }
    p.k = best;
    p.pmagic = (unsigned)(((1ull << 32) + fd.f - 1) / fd.f);          // x / f == umulhi(x, pmagic) for x < 2^31 / f
    return 1;
}

// bytes of the pseudo-point part of an edge workspace for k neighbours, widths <= Cmax (0 for k <= 16)
size_t gpe_edge_pseudo_bytes(long npts, int k, int Cmax)
{
    if (k <= SR_PB) return 0;
    int best = 0;
    for (int kq = SR_PB; kq >= SR_PB / SR_NPW; --kq)
        if (k % kq == 0 && (SR_PB / kq) * kq > (best ? (SR_PB / best) * best : 0)) best = kq;
    if (!best) return 0;
    const size_t nps = (size_t)npts * (k / best), ld = (size_t)((Cmax + 3) & ~3);
    // forward with aggregation: mx, mn (floats) + amx, amn (bytes); gathered backward: dP (floats) — the larger of the two
    return nps * ld * (2 * sizeof(float) + 2) + 256;
}

// after the launch: folds the per-pseudo-point rows of the scratch image into the caller's per-point outputs
int gpe_edge_pseudo_fold(const RgParams& p, const GpeFold& fd, hipStream_t s)
{
    if (fd.f <= 1) return GPE_OK;
    const long th = fd.npts * ((p.N + 3) >> 2);
    if (fd.mx) {
        hipLaunchKernelGGL(gpe_sr_fold_agg_kernel, dim3((unsigned)gpe_cdiv(th, 256)), dim3(256), 0, s, p.mx, p.mn, p.oamx, p.oamn,
                           fd.npts, fd.f, fd.kq, p.N, p.oldagg, fd.mx, fd.mn, fd.amx, fd.amn);
        GPE_CHECK_LAUNCH();
    }
    if (fd.dp) {
        hipLaunchKernelGGL(gpe_sr_fold_sum_kernel, dim3((unsigned)gpe_cdiv(th, 256)), dim3(256), 0, s, p.dP, fd.npts, fd.f, p.N,
                           p.lddp, fd.dp);
        GPE_CHECK_LAUNCH();
    }
    return GPE_OK;
}

// Returns 1 and launches when the shape is on this kernel's menu, 0 when the caller should try the next kernel,
// < 0 on a launch error.  `p` comes with the generic tiling (R = (64/k)*k); this kernel re-tiles so that every wave
// owns whole points: R = 4 * npw * k with npw * k <= 16.
int gpe_edgegemm_sr_try(const RgParams& p_in, int amode, int emode, int stats_nblk, hipStream_t s)
{
    RgParams p = p_in;
    if (p.N <= 96 || p.N > 208 || p.K <= 96 || p.K > 208) return 0;
    if (emode != E_EDGE_FWD && (p.N & 3)) return 0;      // the backward epilogues use aligned 16-B coefficient loads
    if (amode == A_GATHER && (p.K & 3)) return 0;
    if (amode == A_DENSE && (p.a.inner > 0 || (p.a.stride_outer & 3) || p.a.stride_outer < ((p.K + 3) & ~3) ||
                             (((uintptr_t)p.a.base) & 15)))
        return 0;                                        // dense rows must be aligned + padded for plain 16-B loads
    if (p.k < 1) return 0;
    const bool per_point = amode == A_GATHER || emode == E_BWD_GATHER || (emode == E_EDGE_FWD && p.agg);
    // k > 16: rows that need nothing per point can be tiled any way (4 rows per "point": 64-row tiles); the per-point
    // variants split a point into f pseudo-points of kq rows and fold the per-pseudo-point results afterwards
    GpeFold fold;
    if (!gpe_edge_pseudo_setup(p, per_point, emode, fold)) return 0;
    const int npw = SR_PB / p.k;                         // points per wave per tile
    if (per_point && npw > SR_NPW) return 0;
    p.R = 4 * npw * p.k;
    p.num_tiles = gpe_cdiv(p.M, p.R);
    p.pin_tpc = 0;
    if ((amode == A_GATHER || emode == E_BWD_GATHER) && p.pin_clouds > 0 && gpe_pin_clouds(p.pin_clouds) &&
        p.pin_clouds % GPE_NXCD == 0) {
        // gather variants only (dense streaming tiles have nothing to keep in L2): tiles must not straddle clouds and
        // the launcher must keep gridDim.x a multiple of 8 with gridDim.x / 8 <= tiles per cloud
        const long rows_per_cloud = p.M / p.pin_clouds;
        const int gx = gpe_num_cus();
        if (rows_per_cloud % p.R == 0 && gx % GPE_NXCD == 0 && gx <= p.num_tiles &&
            (stats_nblk <= 0 || gx <= stats_nblk) && rows_per_cloud / p.R >= gx / GPE_NXCD)
            p.pin_tpc = (int)(rows_per_cloud / p.R);
    }
    const int NT = (p.N <= 160) ? 10 : 13;
    const int KCH = (p.K <= 160) ? 10 : 13;
    // dummy image of the straight-line instances: 64 rows x 512 floats (row pitches here are <= 256 floats)
    p.dummy = (p.ldo <= 512 && p.oldagg <= 512 && p.lddp <= 512) ? p.ws.dummy : nullptr;
    int rc = GPE_EINVAL;
    if (amode == A_GATHER && emode == E_EDGE_FWD) rc = sr_dispatch<A_GATHER, E_EDGE_FWD>(NT, KCH, p, stats_nblk, s);
    else if (amode == A_DENSE && (emode == E_EDGE_FWD || emode == E_BWD_INPLACE))
        rc = gpe_sr_dispatch_dense(emode, NT, KCH, p, stats_nblk, s);
    else if (amode == A_DENSE && emode == E_BWD_GATHER) rc = sr_dispatch<A_DENSE, E_BWD_GATHER>(NT, KCH, p, stats_nblk, s);
    if (rc == GPE_ENOTSUP_SHAPE) return 0;
    if (rc == GPE_OK) rc = gpe_edge_pseudo_fold(p, fold, s);
    return rc == GPE_OK ? 1 : rc;
}
