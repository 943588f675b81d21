// Per-cloud exact kNN for gfx950 (replaces torch_cluster.knn as reached from
// /root/reference/nn/net_blocks.py:127-135,174 through PyG's DynamicEdgeConv).
//
// One workgroup = 64 query points of one cloud against all N candidates of that cloud.
//   phase 1 (VALU): 64x64 tile of squared distances, 4x4 register micro-tile per lane, operands staged
//                   channel-major in LDS so every LDS read is a conflict-free / broadcast ds_read_b128;
//                   arithmetic is exactly oracle/knn_ref.c's: acc = fmaf(q_c - p_c, q_c - p_c, acc), c ascending.
//   phase 2 (wave ballot/shuffle): each wave owns 16 of the 64 queries.  A query's running top-k list is
//                   DISTRIBUTED OVER THE LANES of the wave (lane s holds the s-th best (dist, idx)); a tile's 64
//                   candidate distances are compared against the k-th best with one v_cmp + ballot, and only the
//                   (rare) survivors are inserted with a lane-shift.  No per-lane sorted arrays, no scratch.
// Ordering rule: ascending (dist, candidate index); an equal-distance candidate never displaces an earlier one.
#include "gpe_common.h"
#include <math.h>

#define KNN_TQ 64
#define KNN_TC 64
#define KNN_CCH 32          // channels staged per step for the candidate tile
#define KNN_LD 68           // row stride (floats) of channel-major LDS tiles: 16-B aligned, rows shifted by 1 slot

__global__ __launch_bounds__(256) void gpe_knn_kernel(const float* __restrict__ x, int N, int C, int ldx, int k,
                                                      int32_t* __restrict__ idx, int32_t* __restrict__ idx_glob, int Cq /* C rounded up to CCH */,
                                                      int B, int tiles, int pin)
{
    extern __shared__ __align__(16) float smem[];
    // both operand tiles are staged per 32-channel chunk (35 KB of LDS per workgroup -> 4 workgroups per CU; a resident
    // 150-channel query tile cost 43 KB and left the barrier-heavy loop with 2 waves per SIMD, 44 % of wave time parked)
    float* qT = smem;                          // [CCH][KNN_LD]  query chunk, channel-major
    float* cT = qT + KNN_CCH * KNN_LD;         // [CCH][KNN_LD]  candidate chunk
    float* dist = cT + KNN_CCH * KNN_LD;       // [64][KNN_LD]   distance tile (query-major)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // 1-D grid.  pin: all 64-query tiles of cloud c run on XCD c % 8 (gpe_common.h), so the cloud's candidate table
    // (N x C floats, re-read by every tile) is fetched from HBM by one L2 instead of eight
    int b, qt;
    if (pin) {
        const int xcd = blockIdx.x & (GPE_NXCD - 1), slot = blockIdx.x >> 3;
        const int jc = slot / tiles;
        b = xcd + GPE_NXCD * jc;
        qt = slot - jc * tiles;
        if (b >= B) return;
    } else {
        b = blockIdx.x / tiles;
        qt = blockIdx.x - b * tiles;
    }
    const int q0 = qt * KNN_TQ;
    const float* cloud = x + (size_t)b * N * ldx;

    // lane-distributed top-k lists for the 16 queries this wave selects for
    float ld_[16];
    int li_[16];
    float thr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { ld_[i] = INFINITY; li_[i] = -1; thr[i] = INFINITY; }

    const int tq = tid & 15;        // query micro-row: queries 4*tq .. 4*tq+3
    const int tc = tid >> 4;        // candidate micro-col: candidates 4*tc .. 4*tc+3

    // candidate chunks are register-prefetched one (tile, channel-chunk) step ahead, so the global-load latency sits
    // under the previous chunk's arithmetic instead of between two barriers
    const int nchunk = Cq / KNN_CCH;
    const int nsteps = ((N + KNN_TC - 1) / KNN_TC) * nchunk;
    float pre[(KNN_TC * KNN_CCH) / 256], preq[(KNN_TQ * KNN_CCH) / 256];
    auto prefetch = [&](int step) {
        const int c0n = (step / nchunk) * KNN_TC, chn = (step % nchunk) * KNN_CCH;
#pragma unroll
        for (int i = 0; i < (KNN_TC * KNN_CCH) / 256; ++i) {
            const int e = tid + 256 * i;
            const int p = e / KNN_CCH, c = e - p * KNN_CCH;
            const int cr = (chn + c < C) ? chn + c : C - 1;
            const int pr = (c0n + p < N) ? c0n + p : N - 1;           // clamped (unconditional load); masked below
            const int qr = (q0 + p < N) ? q0 + p : N - 1;
            const float v = cloud[(size_t)pr * ldx + cr];
            const float w = cloud[(size_t)qr * ldx + cr];
            pre[i] = (c0n + p < N && chn + c < C) ? v : 0.f;
            preq[i] = (q0 + p < N && chn + c < C) ? w : 0.f;
        }
    };
    prefetch(0);
    int step = 0;

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    for (int c0 = 0; c0 < N; c0 += KNN_TC) {
        // packed accumulators: acc2[a][h] = {acc[a][2h], acc[a][2h+1]}  (v_pk_fma_f32: per-element IEEE fma, so the
        // chain is bit-identical to the scalar fmaf of oracle/knn_ref.c)
        f32x2 acc2[4][2];
#pragma unroll
        for (int a = 0; a < 4; ++a) { acc2[a][0] = (f32x2){0.f, 0.f}; acc2[a][1] = (f32x2){0.f, 0.f}; }

        for (int ch = 0; ch < Cq; ch += KNN_CCH, ++step) {
            __syncthreads();   // previous chunk (and, on the first pass, the previous tile's dist reads) done
#pragma unroll
            for (int i = 0; i < (KNN_TC * KNN_CCH) / 256; ++i) {
                const int e = tid + 256 * i;
                const int p = e / KNN_CCH, c = e - p * KNN_CCH;
                cT[c * KNN_LD + p] = pre[i];
                qT[c * KNN_LD + p] = preq[i];
            }
            __syncthreads();
            if (step + 1 < nsteps) prefetch(step + 1);
            const int cend = (C - ch < KNN_CCH) ? (C - ch) : KNN_CCH;   // skip the zero-padded channels
#pragma unroll 4
            for (int c = 0; c < cend; ++c) {
                const float4 qv = *reinterpret_cast<const float4*>(&qT[c * KNN_LD + 4 * tq]);
                const float4 pv = *reinterpret_cast<const float4*>(&cT[c * KNN_LD + 4 * tc]);
                const float qa[4] = {qv.x, qv.y, qv.z, qv.w};
                const f32x2 p01 = (f32x2){pv.x, pv.y}, p23 = (f32x2){pv.z, pv.w};
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const f32x2 qq = (f32x2){qa[a], qa[a]};
                    const f32x2 d0 = qq - p01, d1 = qq - p23;
                    acc2[a][0] = __builtin_elementwise_fma(d0, d0, acc2[a][0]);
                    acc2[a][1] = __builtin_elementwise_fma(d1, d1, acc2[a][1]);
                }
            }
        }
        float acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            acc[a][0] = acc2[a][0].x; acc[a][1] = acc2[a][0].y; acc[a][2] = acc2[a][1].x; acc[a][3] = acc2[a][1].y;
        }
        // padded channels contribute fmaf(0,0,acc) = acc exactly, so chunking does not change the chain

#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float4 o = make_float4(acc[a][0], acc[a][1], acc[a][2], acc[a][3]);
            *reinterpret_cast<float4*>(&dist[(4 * tq + a) * KNN_LD + 4 * tc]) = o;
        }
        __syncthreads();

        // ---- selection: wave `wave` owns queries 16*wave .. 16*wave+15; lane = candidate of this tile -------
        const int cand = c0 + lane;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float d = dist[(16 * wave + i) * KNN_LD + lane];
            if (cand >= N) d = INFINITY;
            if (c0 == 0) {
                // first tile: the list is empty, so instead of 64 one-at-a-time insertions rank all 64 candidates at
                // once — rank = #candidates that precede this one in (dist, index) order — and scatter the k best to
                // their list lanes with one ds_permute each
                const int db = __float_as_int(d);
                int rank = 0;
#pragma unroll
                for (int s2 = 0; s2 < 64; ++s2) {
                    const float ds = __int_as_float(__builtin_amdgcn_readlane(db, s2));
                    rank += (ds < d || (ds == d && s2 < lane)) ? 1 : 0;
                }
                const int dperm = __builtin_amdgcn_ds_permute(rank << 2, db);
                const int iperm = __builtin_amdgcn_ds_permute(rank << 2, lane);
                ld_[i] = (lane < k) ? __int_as_float(dperm) : INFINITY;
                li_[i] = (lane < k) ? iperm : -1;
                thr[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ld_[i]), k - 1));
                continue;
            }
            unsigned long long m = __ballot(d < thr[i]);
            while (m) {
                const int src = __builtin_ctzll(m);
                m &= m - 1;
                // all cross-lane traffic below is v_readlane / DPP wave_shr (VALU latency), not ds_bpermute
                const float dn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), src));
                if (!(dn < thr[i])) continue;               // the threshold may have tightened meanwhile
                const int jn = c0 + src;
                // number of list entries that stay in front: all with dist <= dn (they have lower indices)
                const int pos = __builtin_popcountll(__ballot(ld_[i] <= dn));
                const int ldb = __float_as_int(ld_[i]);
                const int updb = __builtin_amdgcn_update_dpp(ldb, ldb, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                const int upi = __builtin_amdgcn_update_dpp(li_[i], li_[i], 0x138, 0xf, 0xf, false);
                if (lane == pos) { ld_[i] = dn; li_[i] = jn; }
                else if (lane > pos) { ld_[i] = __int_as_float(updb); li_[i] = upi; }
                thr[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ld_[i]), k - 1));
            }
        }
    }

    // ---- write the k indices of each query ---------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int q = q0 + 16 * wave + i;
        if (q < N && lane < k) {
            const size_t o = ((size_t)b * N + q) * k + lane;
            idx[o] = li_[i];
            if (idx_glob) idx_glob[o] = b * N + li_[i];
        }
    }
}

extern "C" int gpe_knn(const float* x, int B, int N, int C, int ldx, int k, int32_t* idx, int32_t* idx_glob,
                       void* stream)
{
    if (!x || !idx || B < 0 || N <= 0 || C <= 0 || ldx < C || k <= 0 || k > 64 || k > N || (long)B * N * k >= (1L << 31)) return GPE_EINVAL;
    if (B == 0) return GPE_OK;
    const int Cq = gpe_round_up(C, KNN_CCH);
    const size_t lds = ((size_t)2 * KNN_CCH * KNN_LD + 64 * KNN_LD) * sizeof(float);
    GPE_ENSURE_MAX_LDS((gpe_knn_kernel));
    const int tiles = gpe_cdiv(N, KNN_TQ);
    const int pin = gpe_pin_clouds(B) ? 1 : 0;
    const long nblocks = pin ? (long)GPE_NXCD * gpe_cdiv(B, GPE_NXCD) * tiles : (long)B * tiles;
    if (nblocks >= (1L << 31)) return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_knn_kernel, dim3((unsigned)nblocks), dim3(256), lds, (hipStream_t)stream, x, N, C, ldx, k, idx,
                       idx_glob, Cq, B, tiles, pin);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}
