// Per-cloud exact kNN for gfx950 (replaces torch_cluster.knn as reached from
// /root/reference/nn/net_blocks.py:127-135,174 through PyG's DynamicEdgeConv).
//
// One workgroup = 64 query points of one cloud against all N candidates of that cloud, 64 candidates at a time.  Wave w
// OWNS queries 16w..16w+15 end to end (distances and selection), so the only workgroup barriers are the two around the
// operand staging of a 32-channel step.
//   staging : both operand tiles are copied POINT-MAJOR ([64 rows][36 floats], the row-major global layout with a
//             4-float pad) — a straight float4 / float2 / float copy, no transposition.  The pad makes every ds_read_b128
//             of the distance loop conflict-free (quad index 9*row + c/4: the <= 4 distinct rows of a b128 lane group
//             fall into distinct 4-bank groups).
//   phase 1 : (VALU) 16 x 64 distances per wave, 4 x 4 register micro-tile per lane, 4 channels per LDS read; arithmetic
//             is exactly oracle/knn_ref.c's: acc = fmaf(q_c - p_c, q_c - p_c, acc), c ascending.  Packed fp32
//             instructions buy nothing on this part (v_pk_fma_f32 issues at half the rate of v_fma_f32 at 4 waves per
//             SIMD: profiles/r02_d_knn.md), the floor is 2 lane-ops per (query, candidate, channel).
//   phase 2 : (wave ballot / readlane) a query's running top-k list is DISTRIBUTED OVER THE LANES of its wave (lane s
//             holds the s-th best (dist, idx)); a tile's 64 candidate distances are compared against the k-th best with
//             one v_cmp + ballot.  ALL survivors of a tile are merged in one pass: each survivor costs one readlane and
//             three compares (how far it pushes the list entries behind it, where it lands among the list entries and
//             among the other survivors), then list entries and survivors are scattered to their new slots through a
//             512-byte per-wave LDS strip.  No per-lane sorted arrays, no scratch memory.
// Ordering rule: ascending (dist, candidate index); an equal-distance candidate never displaces an earlier one.
#include "gpe_common.h"
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define KNN_TQ 64
#define KNN_TC 64
#define KNN_CCH 32          // channels staged per step
#define KNN_LD 36           // row stride (floats) of the point-major operand tiles
#define KNN_LDD 68          // row stride of a wave's 16 x 64 distance strip

__device__ __forceinline__ float knn_readlane_f(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// The distance chain as explicit scalar instructions: the SLP vectoriser otherwise pairs neighbouring candidates into
// v_pk_* (no faster on gfx950) and pays v_mov shuffles plus dependent pk chains for it.  IEEE sub + fused multiply-add,
// identical to fmaf(q - p, q - p, acc).
__device__ __forceinline__ float knn_sub(float a, float b)
{
    float d;
    asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float knn_sqacc(float d, float acc)
{
    asm("v_fma_f32 %0, %1, %1, %0" : "+v"(acc) : "v"(d));
    return acc;
}

// One query's selection step for one candidate tile.  lane = candidate (distance d, index cand); (ldv, liv) = this lane's
// entry of the query's sorted list (lanes >= k: +inf / -1), thr = its k-th best distance.
// Squared distances are >= +0, so their bit patterns order like the values and (dist bits, lane) is one 64-bit key for
// the (dist, index) order among the candidates of a tile.
__device__ __forceinline__ void knn_select(bool first, float d, int lane, int cand, int k, unsigned long long* mW, float& ldv,
                                           int& liv, float& thr)
{
    const int db = __float_as_int(d);
    const unsigned long long key = ((unsigned long long)(unsigned)db << 32) | (unsigned)lane;
    if (first) {
        // first tile: the list is empty, so rank all 64 candidates at once — rank = #candidates that precede this one in
        // (dist, index) order — and scatter the k best to their list lanes with one ds_permute each
        int rank = 0;
#pragma unroll 8
        for (int s2 = 0; s2 < 64; ++s2) {
            const unsigned long long keyn = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(db, s2) << 32) | (unsigned)s2;
            rank += (keyn < key) ? 1 : 0;
        }
        const int dperm = __builtin_amdgcn_ds_permute(rank << 2, db);
        const int iperm = __builtin_amdgcn_ds_permute(rank << 2, cand);
        ldv = (lane < k) ? __int_as_float(dperm) : INFINITY;
        liv = (lane < k) ? iperm : -1;
        thr = knn_readlane_f(ldv, k - 1);
        return;
    }
    const unsigned long long m = __ballot(d < thr);
    if (m == 0) return;
    // ---- merge every survivor of the tile in one pass ----
    int shift = 0;        // list lanes: survivors that go in front of my entry
    int rank = 0;         // survivor lanes: survivors in front of me
    int pos = 0;          // survivor lanes: list entries in front of me
    unsigned long long mm = m;
    do {
        const int src = __builtin_ctzll(mm);
        mm &= mm - 1;
        const int dnb = __builtin_amdgcn_readlane(db, src);
        const float dn = __int_as_float(dnb);
        // a list entry stays in front of an equal-distance survivor (it has the lower index)
        shift += (dn < ldv) ? 1 : 0;
        const unsigned long long keyn = ((unsigned long long)(unsigned)dnb << 32) | (unsigned)src;
        rank += (keyn < key) ? 1 : 0;
        const int front = __builtin_popcountll(__ballot(ldv <= dn));
        pos = (lane == src) ? front : pos;
    } while (mm);
    asm volatile("" ::: "memory");
    if (lane < k) {
        const int np = lane + shift;
        if (np < k) mW[np] = ((unsigned long long)(unsigned)__float_as_int(ldv) << 32) | (unsigned)liv;
    }
    if ((m >> lane) & 1ull) {
        const int np = pos + rank;
        if (np < k) mW[np] = ((unsigned long long)(unsigned)db << 32) | (unsigned)cand;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long got = mW[lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the next query's scatter reuses the strip
    ldv = (lane < k) ? __int_as_float((int)(got >> 32)) : INFINITY;
    liv = (lane < k) ? (int)(unsigned)got : -1;
    thr = knn_readlane_f(ldv, k - 1);
}

// PROBE != 0 compiles the phase switches of scripts/knn_probe.py in (GPE_KNN_PROBE bits: 1 skip the selection after the
// first tile, 2 skip the staging after the first step, 4 skip the distance arithmetic); the shipped kernels have PROBE = 0.
// VEC = floats per staging load (host: rows and channel count are multiples of VEC floats, base pointer VEC*4-aligned).
// SMALLC (host: C <= 4): a chunk is one 4-float column group, so a thread stages ONE vector per operand tile instead of up
// to 32 / VEC — the scalar-load variant otherwise carries 16 staging registers it never fills and spills 16 others.
template <int VEC, int PROBE, bool SMALLC = false>
__global__ __launch_bounds__(256, 4) void gpe_knn_kernel(const float* __restrict__ x, int N, int C, int ldx, int k,
                                                         int32_t* __restrict__ idx, int32_t* __restrict__ idx_glob, int B,
                                                         int tiles, int pin, int probe, int nsplit,
                                                         unsigned long long* __restrict__ part)
{
    extern __shared__ __align__(16) float smem[];
    float* const qS = smem;                                // [64][KNN_LD]   query rows of this step's channels
    float* const cS = qS + KNN_TQ * KNN_LD;                // [64][KNN_LD]   candidate rows
    float* const dS = cS + KNN_TC * KNN_LD;                // [4 waves][16][KNN_LDD]  distance strips
    unsigned long long* const mS = reinterpret_cast<unsigned long long*>(dS + 4 * 16 * KNN_LDD);   // [4 waves][64] merge strips

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid.  pin: all 64-query tiles of cloud c run on XCD c % 8 (gpe_common.h), so the cloud's candidate table
    // (N x C floats, re-read by every tile) is fetched from HBM by one L2 instead of eight
    // nsplit > 1 (host: the tables of the clouds in flight on an XCD would not fit its L2): the candidate range of a query
    // tile is cut into nsplit pieces, one workgroup each; every piece writes its own sorted k-list and gpe_knn_merge_kernel
    // merges them.  All pieces of a cloud are consecutive work items, so nsplit x fewer clouds are streamed at a time.
    int b, item;
    const int ipc = tiles * nsplit;                        // work items per cloud
    if (pin) {
        const int xcd = blockIdx.x & (GPE_NXCD - 1), slot = blockIdx.x >> 3;
        const int jc = slot / ipc;
        b = xcd + GPE_NXCD * jc;
        item = slot - jc * ipc;
        if (b >= B) return;
    } else {
        b = blockIdx.x / ipc;
        item = blockIdx.x - b * ipc;
    }
    const int qt = item / nsplit, piece = item - qt * nsplit;
    const int q0 = qt * KNN_TQ;
    const int tps = (tiles + nsplit - 1) / nsplit;         // candidate tiles per piece (tiles == candidate tiles: TQ == TC)
    const int c_first = piece * tps * KNN_TC;
    const int c_stop = ((piece + 1) * tps * KNN_TC < N) ? (piece + 1) * tps * KNN_TC : N;
    const float* cloud = x + (size_t)b * N * ldx;
    float* const dW = dS + wave * 16 * KNN_LDD;
    unsigned long long* const mW = mS + wave * 64;

    // lane-distributed top-k lists of the 16 queries this wave owns
    float ld_[16];
    int li_[16];
    float thr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { ld_[i] = INFINITY; li_[i] = -1; thr[i] = INFINITY; }

    const int tq = lane & 3;         // query micro-row: queries 16*wave + 4*tq .. +3
    const int tc = lane >> 2;        // candidate micro-col: candidates 4*tc .. 4*tc+3

    // ---- staging: registers one step ahead, so the global-load latency sits under the previous step's arithmetic -------
    // a step = (candidate tile, 32-channel chunk); a chunk is staged chw floats wide (multiple of 4, <= 32)
    const int nchunk = (C + KNN_CCH - 1) / KNN_CCH;
    const int chw = (C < KNN_CCH) ? ((C + 3) & ~3) : KNN_CCH;
    const int vpr = chw / VEC;                             // staging vectors per row
    const int rvpr = (65536 + vpr - 1) / vpr;              // e / vpr == (e * rvpr) >> 16 for e < 2048
    const int nvec = KNN_TC * vpr;                         // vectors per operand tile (<= 2048 / VEC)
    constexpr int NPF = SMALLC ? 1 : (KNN_TC * KNN_CCH) / (256 * VEC);
    float pre_c[NPF][VEC], pre_q[NPF][VEC];
    int pf_c0 = c_first, pf_ch = 0;                        // tile / chunk the NEXT prefetch loads
    auto prefetch = [&]() {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int e = tid + 256 * i;
            if (e < nvec) {
                const int row = (e * rvpr) >> 16, cv = e - row * vpr;
                const int ch = pf_ch + cv * VEC;
                const bool on = ch < C;                    // C % VEC == 0: a vector is valid or padding as a whole
                const int chc = on ? ch : 0;
                const int pr = (pf_c0 + row < N) ? pf_c0 + row : N - 1;   // clamped rows: masked in the selection /
                const int qr = (q0 + row < N) ? q0 + row : N - 1;         // never written back
                const float* pc = cloud + (size_t)pr * ldx + chc;
                const float* pq = cloud + (size_t)qr * ldx + chc;
                if constexpr (VEC == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(pc), w = *reinterpret_cast<const float4*>(pq);
                    pre_c[i][0] = v.x; pre_c[i][1] = v.y; pre_c[i][2] = v.z; pre_c[i][3] = v.w;
                    pre_q[i][0] = w.x; pre_q[i][1] = w.y; pre_q[i][2] = w.z; pre_q[i][3] = w.w;
                } else if constexpr (VEC == 2) {
                    const float2 v = *reinterpret_cast<const float2*>(pc), w = *reinterpret_cast<const float2*>(pq);
                    pre_c[i][0] = v.x; pre_c[i][1] = v.y;
                    pre_q[i][0] = w.x; pre_q[i][1] = w.y;
                } else {
                    pre_c[i][0] = *pc; pre_q[i][0] = *pq;
                }
                if (!on) {
#pragma unroll
                    for (int t = 0; t < VEC; ++t) { pre_c[i][t] = 0.f; pre_q[i][t] = 0.f; }
                }
            }
        }
        pf_ch += KNN_CCH;
        if (pf_ch >= C) { pf_ch = 0; pf_c0 += KNN_TC; }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int e = tid + 256 * i;
            if (e < nvec) {
                const int row = (e * rvpr) >> 16, cv = e - row * vpr;
                float* dc = &cS[row * KNN_LD + cv * VEC];
                float* dq = &qS[row * KNN_LD + cv * VEC];
                if constexpr (VEC == 4) {
                    *reinterpret_cast<float4*>(dc) = make_float4(pre_c[i][0], pre_c[i][1], pre_c[i][2], pre_c[i][3]);
                    *reinterpret_cast<float4*>(dq) = make_float4(pre_q[i][0], pre_q[i][1], pre_q[i][2], pre_q[i][3]);
                } else if constexpr (VEC == 2) {
                    *reinterpret_cast<float2*>(dc) = make_float2(pre_c[i][0], pre_c[i][1]);
                    *reinterpret_cast<float2*>(dq) = make_float2(pre_q[i][0], pre_q[i][1]);
                } else {
                    *dc = pre_c[i][0]; *dq = pre_q[i][0];
                }
            }
        }
    };
    prefetch();
    const int nsteps = ((c_stop - c_first + KNN_TC - 1) / KNN_TC) * nchunk;
    int step = 0;

    const float* const qrow = &qS[(16 * wave + 4 * tq) * KNN_LD];
    const float* const crow = &cS[(4 * tc) * KNN_LD];

    for (int c0 = c_first; c0 < c_stop; c0 += KNN_TC) {
        float acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) acc[a][bb] = 0.f;

        for (int ch = 0; ch < C; ch += KNN_CCH, ++step) {
            __syncthreads();   // every wave is done with the previous step's operand tiles
            if (!(PROBE && (probe & 2) && step > 0)) commit();
            __syncthreads();
            if (step + 1 < nsteps && !(PROBE && (probe & 2))) prefetch();
            int nquad = ((C - ch < KNN_CCH) ? (C - ch + 3) : KNN_CCH) >> 2;   // padded channels inside a quad are zeros
            if (PROBE && (probe & 4)) nquad = 0;
            for (int cq = 0; cq < nquad; ++cq) {
                float4 pv[4];
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) pv[bb] = *reinterpret_cast<const float4*>(&crow[bb * KNN_LD + 4 * cq]);
                float4 qn = *reinterpret_cast<const float4*>(&qrow[4 * cq]);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const float4 qv = qn;
                    if (a < 3) qn = *reinterpret_cast<const float4*>(&qrow[(a + 1) * KNN_LD + 4 * cq]);
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const float qa = (cc == 0) ? qv.x : (cc == 1) ? qv.y : (cc == 2) ? qv.z : qv.w;
                        float d[4];
#pragma unroll
                        for (int bb = 0; bb < 4; ++bb) {
                            const float pb = (cc == 0) ? pv[bb].x : (cc == 1) ? pv[bb].y : (cc == 2) ? pv[bb].z : pv[bb].w;
                            d[bb] = knn_sub(qa, pb);
                        }
#pragma unroll
                        for (int bb = 0; bb < 4; ++bb) acc[a][bb] = knn_sqacc(d[bb], acc[a][bb]);
                    }
                }
            }
        }
        // padded channels contribute fmaf(0,0,acc) = acc exactly, so chunking does not change the chain

        // this wave's 16 x 64 strip, query-major.  LDS operations of one wave execute in order; the asm statements only
        // keep the compiler from moving the strip reads across the strip writes (different lanes, same memory)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int a = 0; a < 4; ++a)
            *reinterpret_cast<float4*>(&dW[(4 * tq + a) * KNN_LDD + 4 * tc]) = make_float4(acc[a][0], acc[a][1], acc[a][2], acc[a][3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

        // ---- selection: lane = candidate of this tile ---------------------------------------------------------
        if (PROBE && (probe & 1) && c0 > c_first) continue;
        const int cand = c0 + lane;
        const bool tail = c0 + KNN_TC > N;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float d = dW[i * KNN_LDD + lane];
            if (tail && cand >= N) d = INFINITY;
            float ldv = ld_[i], t = thr[i];
            int liv = li_[i];
            knn_select(c0 == c_first, d, lane, cand, k, mW, ldv, liv, t);
            ld_[i] = ldv; li_[i] = liv; thr[i] = t;
            __builtin_amdgcn_sched_barrier(0);     // one query at a time: interleaving the 16 merges only spills
        }
    }

    // ---- write the k indices of each query ---------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int q = q0 + 16 * wave + i;
        if (q < N && lane < k) {
            if (nsplit > 1) {          // this piece's sorted list (entries past the piece's candidates: +inf)
                part[(((size_t)b * N + q) * nsplit + piece) * k + lane] =
                    ((unsigned long long)(unsigned)__float_as_int(ld_[i]) << 32) | (unsigned)li_[i];
            } else {
                const size_t o = ((size_t)b * N + q) * k + lane;
                idx[o] = li_[i];
                if (idx_glob) idx_glob[o] = b * N + li_[i];
            }
        }
    }
}

// merge of the nsplit sorted k-lists of a query: one thread per entry.  The lists are sorted and all keys are distinct
// (distinct candidate indices), so the output slot of an entry is its own position plus, for every other list, the number of
// entries below it (a binary search over k sorted keys).
__global__ __launch_bounds__(256) void gpe_knn_merge_kernel(const unsigned long long* __restrict__ part, long nq, int N, int k,
                                                            int nsplit, int32_t* __restrict__ idx, int32_t* __restrict__ idx_glob)
{
    const int n = nsplit * k;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq * n) return;
    const long q = t / n;
    const int pos = (int)(t - q * n);
    const int piece = pos / k, i = pos - piece * k;
    const unsigned long long* base = part + q * n;
    const unsigned long long key = base[pos];
    int rank = i;
    for (int p2 = 0; p2 < nsplit; ++p2) {
        if (p2 == piece) continue;
        const unsigned long long* lst = base + p2 * k;
        int lo = 0, hi = k;                              // first entry of lst that is not below key
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (lst[mid] < key) lo = mid + 1; else hi = mid;
        }
        rank += lo;
    }
    if (rank < k) {
        const unsigned lo32 = (unsigned)key;
        idx[q * k + rank] = (int32_t)lo32;
        if (idx_glob) idx_glob[q * k + rank] = (int32_t)((q / N) * N + lo32);
    }
}

template <int VEC>
static void knn_launch(long nblocks, size_t lds, hipStream_t s, int probe, const float* x, int N, int C, int ldx, int k,
                       int32_t* idx, int32_t* idx_glob, int B, int tiles, int pin, int nsplit, unsigned long long* part)
{
    if (probe)
        hipLaunchKernelGGL((gpe_knn_kernel<VEC, 1>), dim3((unsigned)nblocks), dim3(256), lds, s, x, N, C, ldx, k, idx, idx_glob, B,
                           tiles, pin, probe, nsplit, part);
    else if (VEC == 1 && C <= 4)
        hipLaunchKernelGGL((gpe_knn_kernel<1, 0, true>), dim3((unsigned)nblocks), dim3(256), lds, s, x, N, C, ldx, k, idx, idx_glob,
                           B, tiles, pin, 0, nsplit, part);
    else
        hipLaunchKernelGGL((gpe_knn_kernel<VEC, 0>), dim3((unsigned)nblocks), dim3(256), lds, s, x, N, C, ldx, k, idx, idx_glob, B,
                           tiles, pin, 0, nsplit, part);
}

extern "C" int gpe_knn(const float* x, int B, int N, int C, int ldx, int k, int32_t* idx, int32_t* idx_glob,
                       void* stream)
{
    if (!x || !idx || B < 0 || N <= 0 || C <= 0 || ldx < C || k <= 0 || k > 64 || k > N || (long)B * N * k >= (1L << 31)) return GPE_EINVAL;
    if (B == 0) return GPE_OK;
    const size_t lds = ((size_t)2 * KNN_TQ * KNN_LD + 4 * 16 * KNN_LDD) * sizeof(float) + 4 * 64 * sizeof(unsigned long long);
    static const int probe = getenv("GPE_KNN_PROBE") ? atoi(getenv("GPE_KNN_PROBE")) : 0;
    const int tiles = gpe_cdiv(N, KNN_TQ);
    static const int dbg_pin = getenv("GPE_KNN_PIN") ? atoi(getenv("GPE_KNN_PIN")) : -1;      // measurement overrides
    static const int dbg_vec = getenv("GPE_KNN_VEC") ? atoi(getenv("GPE_KNN_VEC")) : 0;
    const int pin = (dbg_pin >= 0) ? (dbg_pin && B >= GPE_NXCD) : (gpe_pin_clouds(B) ? 1 : 0);
    // Candidate split.  With every workgroup resident (4 per CU) an XCD works on 128 items at a time = 128 / (tiles * nsplit)
    // clouds, whose tables (N x ldx floats each) are streamed once per item: they must fit the XCD's 4 MiB L2 together or the
    // cyclic stream evicts every line before its next use (measured at cfg 2, layer 2: 4 x 1.25 MB -> 396-475 MB fetched for
    // 39 MB; 3 tables -> 36 MB).  nsplit pieces per query tile put nsplit x fewer clouds in flight.
    static const int dbg_split = getenv("GPE_KNN_SPLIT") ? atoi(getenv("GPE_KNN_SPLIT")) : 0;
    int nsplit = 1;
    if (pin) {
        const double table = (double)N * ldx * sizeof(float), l2_budget = 3.2 * 1024 * 1024;
        const int resident = 4 * gpe_num_cus() / GPE_NXCD;                 // items in flight per XCD
        for (;;) {
            const double clouds = (double)resident / ((double)tiles * nsplit);
            if (table * (clouds > 1.0 ? clouds : 1.0) <= l2_budget) break;  // the tables in flight fit
            if (clouds <= 1.0) break;                                      // one table alone is too big: no split helps
            if (nsplit >= 4 || 2 * nsplit * k > 64 || 2 * nsplit > tiles) break;
            nsplit *= 2;
        }
    }
    if (dbg_split > 0 && dbg_split * k <= 64 && dbg_split <= tiles) nsplit = dbg_split;
    unsigned long long* part = nullptr;
    if (nsplit > 1) {
        part = (unsigned long long*)gpe_scratch(1, (size_t)B * N * nsplit * k * sizeof(unsigned long long));
        if (!part) nsplit = 1;                                             // no scratch: one piece, more HBM traffic
    }
    const long nblocks = (pin ? (long)GPE_NXCD * gpe_cdiv(B, GPE_NXCD) * tiles : (long)B * tiles) * nsplit;
    if (nblocks >= (1L << 31)) return GPE_EINVAL;
    // widest staging copy the rows allow (a C < 32 chunk is staged ((C + 3) & ~3) floats wide, so it must divide too)
    const uintptr_t xa = (uintptr_t)x;
    int vec = (C % 4 == 0 && ldx % 4 == 0 && xa % 16 == 0) ? 4 : (C % 2 == 0 && ldx % 2 == 0 && xa % 8 == 0) ? 2 : 1;
    if (dbg_vec > 0 && dbg_vec < vec) vec = dbg_vec;
    hipStream_t s = (hipStream_t)stream;
    if (vec == 4) knn_launch<4>(nblocks, lds, s, probe, x, N, C, ldx, k, idx, idx_glob, B, tiles, pin, nsplit, part);
    else if (vec == 2) knn_launch<2>(nblocks, lds, s, probe, x, N, C, ldx, k, idx, idx_glob, B, tiles, pin, nsplit, part);
    else knn_launch<1>(nblocks, lds, s, probe, x, N, C, ldx, k, idx, idx_glob, B, tiles, pin, nsplit, part);
    GPE_CHECK_LAUNCH();
    if (nsplit > 1) {
        hipLaunchKernelGGL(gpe_knn_merge_kernel, dim3((unsigned)gpe_cdiv((long)B * N * nsplit * k, 256)), dim3(256), 0, s, part,
                           (long)B * N, N, k, nsplit, idx, idx_glob);
        GPE_CHECK_LAUNCH();
    }
    return GPE_OK;
}
