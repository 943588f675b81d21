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

// C == 3 on a spatially sorted cloud with tile pruning (gpe_knn3.hip): 1 launched, 0 not on its menu
int gpe_knn3_try(const float* x, int B, int N, int ldx, int k, int32_t* idx, int32_t* idx_glob, int32_t* order_out, void* ws, long ws_bytes,
                 hipStream_t s);

// all-exact path: every distance by the defined chain (C < 16, k > 48, or GPE_KNN_EXACT=1)
__global__ void gpe_knn_identity_order_kernel(int* __restrict__ order, long nq, int N)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nq) order[i] = (int)(i % N);
}

static int knn_exact(const float* x, int B, int N, int C, int ldx, int k, int32_t* idx, int32_t* idx_glob, void* ws,
                     long ws_bytes, void* stream, int32_t* order_out = nullptr)
{
    if (C == 3) {
        const int rc = gpe_knn3_try(x, B, N, ldx, k, idx, idx_glob, order_out, ws, ws_bytes, (hipStream_t)stream);
        if (rc != 0) return rc < 0 ? rc : GPE_OK;
    }
    if (order_out) {                                   // no curve order on this path: the identity is a valid (locality-free) answer
        hipLaunchKernelGGL(gpe_knn_identity_order_kernel, dim3((unsigned)gpe_cdiv((long)B * N, 256)), dim3(256), 0, (hipStream_t)stream,
                           order_out, (long)B * N, N);
        GPE_CHECK_LAUNCH();
    }
    const size_t lds = ((size_t)2 * KNN_TQ * KNN_LD + 4 * 16 * KNN_LDD) * sizeof(float) + 4 * 64 * sizeof(unsigned long long);
    static const int probe = gpe_dbg_env("GPE_KNN_PROBE", 0);
    const int tiles = gpe_cdiv(N, KNN_TQ);
    static const int dbg_pin = gpe_dbg_env("GPE_KNN_PIN", -1);      // measurement overrides
    static const int dbg_vec = gpe_dbg_env("GPE_KNN_VEC", 0);
    const int pin = (dbg_pin >= 0) ? (dbg_pin && B >= GPE_NXCD) : (gpe_pin_clouds(B) ? 1 : 0);
    // Candidate split.  With every workgroup resident (4 per CU) an XCD works on 128 items at a time = 128 / (tiles * nsplit)
    // clouds, whose tables (N x ldx floats each) are streamed once per item: they must fit the XCD's 4 MiB L2 together or the
    // cyclic stream evicts every line before its next use (measured at cfg 2, layer 2: 4 x 1.25 MB -> 396-475 MB fetched for
    // 39 MB; 3 tables -> 36 MB).  nsplit pieces per query tile put nsplit x fewer clouds in flight.
    static const int dbg_split = gpe_dbg_env("GPE_KNN_SPLIT", 0);
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
        const size_t need = (size_t)B * N * nsplit * k * sizeof(unsigned long long);
        part = (ws && !(((uintptr_t)ws) & 15) && (size_t)ws_bytes >= need) ? (unsigned long long*)ws : nullptr;
        if (!part) nsplit = 1;                                             // no workspace: one piece, more HBM traffic
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

// =====================================================================================================================
// C >= 16: the distance arithmetic on the fp32 matrix pipe, bit-exact result through a bounded exact recheck
// =====================================================================================================================
// The sub + fma chain above costs 2 VALU lane-ops per (query, candidate, channel); d~ = |q|^2 + |p|^2 - 2 q.p costs one fp32
// MFMA multiply-add (v_mfma_f32_16x16x4_f32: exact products, fp32 accumulation) — half the issue slots.  d~ is NOT the
// defined distance (oracle/knn_ref.c: the fmaf chain of (q_c - p_c)^2), so it only FILTERS:
//   1. gpe_knn_norms_kernel    |x|^2 per point, max per cloud;
//   2. gpe_knn_mfma_kernel     per query the K2 = min(64, N, 2k + 8) best candidates by (d~, index), streamed like above;
//                              a candidate is inserted only below min(K2-th best, k-th best + 2E): nothing further out can
//                              be needed (the k-th best only decreases);
//   3. gpe_knn_rerank_kernel   one wave per query: every candidate of the exact top-k has d~ <= T = d~[k-th] + 2E (proof
//                              below), list neighbours further than 2E apart are in their exact order already, so only runs
//                              of entries closer than 2E are re-evaluated with the exact chain and sorted by (d, index).
//                              If K2 entries lie below T the list may have lost a needed candidate: that query is redone
//                              exactly by its wave (lattices, duplicated points: correct, just slower).
// E bounds |d~ - d| + |d_chain - d| for the real-valued d = sum (q_c - p_c)^2 of the fp32 inputs (u = 2^-24):
//   d_chain: every term is >= 0, C + 2 roundings deep                        ->  <= (C + 2) u d <= 2 (C + 2) u (|q|^2 + |p|^2)
//   d~     : |q|^2, |p|^2 chains (C u each), the dot product (at most two roundings per element: 2 C u |q||p|, doubled by the
//            factor 2), three more roundings                                 ->  <= (3 C + 4) u (|q|^2 + |p|^2)
// so E = (5 C + 8) u (|q|^2 + max_p |p|^2) suffices; the kernels use (6 C + 16) u and 1 % on top.
// Why T suffices: let p be in the exact top-k with d~(p) > kth~ + 2E.  The k list entries r with d~(r) <= kth~ have
// d_chain(r) <= d~(r) + E <= kth~ + E < d~(p) - E <= d_chain(p): k candidates strictly closer than p — contradiction.
#define KNN_MF_MINC 16
#ifndef KNN_MF_CCH
#define KNN_MF_CCH 32          // channels staged per step by the matrix-pipe filter
#endif
#define KNN_MF_LD (KNN_MF_CCH + 4)
#define KNN_MF_WGS ((KNN_MF_CCH <= 32) ? 4 : (KNN_MF_CCH <= 48) ? 3 : 2)

#define KNN_NORM_ROWS 16
__global__ __launch_bounds__(256) void gpe_knn_norms_kernel(const float* __restrict__ x, long rows, int N, int C, int ldx,
                                                            float* __restrict__ norms, int* __restrict__ cmax)
{
    // wave = KNN_NORM_ROWS consecutive rows, all of their loads of a 64-channel slab in flight together (a row at a time was a
    // chain of dependent round trips)
    const int lane = threadIdx.x & 63;
    const long r0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * KNN_NORM_ROWS;
    if (r0 >= rows) return;
    const long last = rows - 1;
    float s[KNN_NORM_ROWS];
#pragma unroll
    for (int u = 0; u < KNN_NORM_ROWS; ++u) s[u] = 0.f;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        const int cc = (c < C) ? c : 0;
        float v[KNN_NORM_ROWS];
#pragma unroll
        for (int u = 0; u < KNN_NORM_ROWS; ++u) v[u] = x[((r0 + u < last) ? r0 + u : last) * ldx + cc];
#pragma unroll
        for (int u = 0; u < KNN_NORM_ROWS; ++u) s[u] = (c < C) ? __builtin_fmaf(v[u], v[u], s[u]) : s[u];
    }
#pragma unroll
    for (int u = 0; u < KNN_NORM_ROWS; ++u) {
        float t = s[u];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ o) << 2, __float_as_int(t)));
        if (lane == 0 && r0 + u <= last) norms[r0 + u] = t;
    }
}

// max_p |p|^2 of every cloud (one workgroup per cloud; per-wave atomics on 32 addresses cost 40 us)
__global__ __launch_bounds__(256) void gpe_knn_cmax_kernel(const float* __restrict__ norms, int N, int* __restrict__ cmax)
{
    __shared__ float red[256];
    const float* p = norms + (size_t)blockIdx.x * N;
    float m = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) m = fmaxf(m, p[i]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) cmax[blockIdx.x] = __float_as_int(red[0]);
}

// the selection step of knn_select with a list of K2 entries of which only the first kk matter for the insertion bound
__device__ __forceinline__ float knn_select_mf(bool first, float d, int lane, int cand, int K2, int kk, float m2e,
                                               float thr, unsigned long long* mW, float& ldv, int& liv)
{
    const int db = __float_as_int(d);
    const unsigned long long key = ((unsigned long long)(unsigned)db << 32) | (unsigned)lane;
    if (first) {
        int rank = 0;
#pragma unroll 8
        for (int s2 = 0; s2 < 64; ++s2) {
            const unsigned long long keyn = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(db, s2) << 32) | (unsigned)s2;
            rank += (keyn < key) ? 1 : 0;
        }
        const int dperm = __builtin_amdgcn_ds_permute(rank << 2, db);
        const int iperm = __builtin_amdgcn_ds_permute(rank << 2, cand);
        ldv = (lane < K2) ? __int_as_float(dperm) : INFINITY;
        liv = (lane < K2) ? iperm : -1;
        return fminf(knn_readlane_f(ldv, K2 - 1), knn_readlane_f(ldv, kk - 1) + m2e);
    }
    const unsigned long long m = __ballot(d < thr);
    if (m == 0) return thr;
    int shift = 0, rank = 0, pos = 0;
    unsigned long long mm = m;
    do {
        const int src = __builtin_ctzll(mm);
        mm &= mm - 1;
        const int dnb = __builtin_amdgcn_readlane(db, src);
        const float dn = __int_as_float(dnb);
        shift += (dn < ldv) ? 1 : 0;
        const unsigned long long keyn = ((unsigned long long)(unsigned)dnb << 32) | (unsigned)src;
        rank += (keyn < key) ? 1 : 0;
        const int front = __builtin_popcountll(__ballot(ldv <= dn));
        pos = (lane == src) ? front : pos;
    } while (mm);
    asm volatile("" ::: "memory");
    if (lane < K2) {
        const int np = lane + shift;
        if (np < K2) mW[np] = ((unsigned long long)(unsigned)__float_as_int(ldv) << 32) | (unsigned)liv;
    }
    if ((m >> lane) & 1ull) {
        const int np = pos + rank;
        if (np < K2) mW[np] = ((unsigned long long)(unsigned)db << 32) | (unsigned)cand;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long got = mW[lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ldv = (lane < K2) ? __int_as_float((int)(got >> 32)) : INFINITY;
    liv = (lane < K2) ? (int)(unsigned)got : -1;
    // the insertion bound of this query from now on: min(K2-th best, kk-th best + 2E)
    return fminf(knn_readlane_f(ldv, K2 - 1), knn_readlane_f(ldv, kk - 1) + m2e);
}

// Same work decomposition as gpe_knn_kernel (64 queries x 64 candidates per step, wave w owns queries 16w..16w+15, operand
// tiles point-major in LDS).  Matrix roles: A = the tile's 64 candidates (4 row blocks), B = the wave's 16 queries, so lane
// (j = lane % 16, g = lane / 16) ends up with the dot products of query j against candidates 16 mt + 4g .. + 3 — one float4
// of the wave's query-major distance strip per row block.  In MFMA t of a 16-channel block lane group g supplies channel
// 4g + t of both operands (one ds_read_b128 per operand row and block).
template <int VEC>
__global__ __launch_bounds__(256, KNN_MF_WGS) void gpe_knn_mfma_kernel(const float* __restrict__ x, int N, int C, int ldx, int kk,
                                                              int K2, const float* __restrict__ norms,
                                                              const int* __restrict__ cmax, float ce, int B, int tiles,
                                                              int pin, int nsplit, unsigned long long* __restrict__ part,
                                                              int probe)
{
    extern __shared__ __align__(16) float smem[];
    float* const qS = smem;
    float* const cS = qS + KNN_TQ * KNN_MF_LD;
    float* const dS = cS + KNN_TC * KNN_MF_LD;
    unsigned long long* const mS = reinterpret_cast<unsigned long long*>(dS + 4 * 16 * KNN_LDD);
    float* const npS = reinterpret_cast<float*>(mS + 4 * 64);       // [64] |p|^2 of the candidate tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b, item;
    const int ipc = tiles * nsplit;
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
    const int tps = (tiles + nsplit - 1) / nsplit;
    const int c_first = piece * tps * KNN_TC;
    const int c_stop = ((piece + 1) * tps * KNN_TC < N) ? (piece + 1) * tps * KNN_TC : N;
    const float* cloud = x + (size_t)b * N * ldx;
    const float* cnorm = norms + (size_t)b * N;
    float* const dW = dS + wave * 16 * KNN_LDD;
    unsigned long long* const mW = mS + wave * 64;

    float ld_[16];
    int li_[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { ld_[i] = INFINITY; li_[i] = -1; }
    float thrq = INFINITY;           // lane L: the insertion bound of query L % 16 (what its tile minimum is tested against)

    const int j = lane & 15, g = lane >> 4;
    const int myq = (q0 + 16 * wave + j < N) ? q0 + 16 * wave + j : N - 1;
    const float nq = cnorm[myq];
    const float cm = __int_as_float(cmax[b]);
    // 2E of each of the wave's 16 queries (uniform per query; lane i of the first 16 holds query i's norm)
    float m2e_of;
    {
        const int qi = (q0 + 16 * wave + (lane & 15) < N) ? q0 + 16 * wave + (lane & 15) : N - 1;
        m2e_of = 2.02f * ce * (cnorm[qi] + cm);
    }

    // ---- staging (as in gpe_knn_kernel) ----------------------------------------------------------------------------
    const int nchunk = (C + KNN_MF_CCH - 1) / KNN_MF_CCH;
    const int chw = (C < KNN_MF_CCH) ? ((C + 15) & ~15) : KNN_MF_CCH;        // whole 16-channel blocks
    const int vpr = chw / VEC;
    const int rvpr = (65536 + vpr - 1) / vpr;
    const int nvec = KNN_TC * vpr;
    constexpr int NPF = (KNN_TC * KNN_MF_CCH) / (256 * VEC);
    float pre_c[NPF][VEC], pre_q[NPF][VEC];
    float pre_n = 0.f;
    int pf_c0 = c_first, pf_ch = 0;
    auto prefetch = [&]() {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int e = tid + 256 * i;
            if (e < nvec) {
                const int row = (e * rvpr) >> 16, cv = e - row * vpr;
                const int ch = pf_ch + cv * VEC;
                const bool on = ch < C;                    // a vector may straddle C (row pitch padded): zeroed per element below
                const int chc = on ? ch : 0;
                const int pr = (pf_c0 + row < N) ? pf_c0 + row : N - 1;
                const int qr = (q0 + row < N) ? q0 + row : N - 1;
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
#pragma unroll
                for (int t = 0; t < VEC; ++t)
                    if (ch + t >= C) { pre_c[i][t] = 0.f; pre_q[i][t] = 0.f; }
            }
        }
        if (pf_ch == 0 && tid < KNN_TC) pre_n = cnorm[(pf_c0 + tid < N) ? pf_c0 + tid : N - 1];
        pf_ch += KNN_MF_CCH;
        if (pf_ch >= C) { pf_ch = 0; pf_c0 += KNN_TC; }
    };
    auto commit = [&](bool first_chunk) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int e = tid + 256 * i;
            if (e < nvec) {
                const int row = (e * rvpr) >> 16, cv = e - row * vpr;
                float* dc = &cS[row * KNN_MF_LD + cv * VEC];
                float* dq = &qS[row * KNN_MF_LD + cv * VEC];
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
        if (first_chunk && tid < KNN_TC) npS[tid] = pre_n;
    };
    prefetch();
    const int nsteps = ((c_stop - c_first + KNN_TC - 1) / KNN_TC) * nchunk;
    int step = 0;

    const float* const brow = &qS[(16 * wave + j) * KNN_MF_LD + 4 * g];
    const float* const arow = &cS[j * KNN_MF_LD + 4 * g];

    for (int c0 = c_first; c0 < c_stop; c0 += KNN_TC) {
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ch = 0; ch < C; ch += KNN_MF_CCH, ++step) {
            __syncthreads();
            if (!((probe & 2) && step > 0)) commit(ch == 0);
            __syncthreads();
            if (step + 1 < nsteps && !(probe & 2)) prefetch();
            int nblk = ((C - ch < KNN_MF_CCH) ? (C - ch + 15) : KNN_MF_CCH) >> 4;   // channels past C are staged as zeros
            if (probe & 4) nblk = 0;
            for (int blk = 0; blk < nblk; ++blk) {
                const float4 bq = *reinterpret_cast<const float4*>(&brow[16 * blk]);
                float4 aq[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) aq[mt] = *reinterpret_cast<const float4*>(&arow[16 * mt * KNN_MF_LD + 16 * blk]);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float bv = (t == 0) ? bq.x : (t == 1) ? bq.y : (t == 2) ? bq.z : bq.w;
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const float av = (t == 0) ? aq[mt].x : (t == 1) ? aq[mt].y : (t == 2) ? aq[mt].z : aq[mt].w;
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[mt], 0, 0, 0);
                    }
                }
            }
        }
        // d~ = |q|^2 + |p|^2 - 2 q.p, clamped at +0 (the bit pattern must order like the value)
        const bool tail = c0 + KNN_TC > N;
        float4 dq[4];
        float dmin = INFINITY;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const float4 np = *reinterpret_cast<const float4*>(&npS[16 * mt + 4 * g]);
            float4 d;
            d.x = __builtin_fmaf(-2.f, acc[mt][0], nq + np.x); d.y = __builtin_fmaf(-2.f, acc[mt][1], nq + np.y);
            d.z = __builtin_fmaf(-2.f, acc[mt][2], nq + np.z); d.w = __builtin_fmaf(-2.f, acc[mt][3], nq + np.w);
            d.x = d.x > 0.f ? d.x : 0.f; d.y = d.y > 0.f ? d.y : 0.f; d.z = d.z > 0.f ? d.z : 0.f; d.w = d.w > 0.f ? d.w : 0.f;
            if (tail) {                                  // candidates past the cloud never qualify
                const int cb = c0 + 16 * mt + 4 * g;
                d.x = (cb + 0 < N) ? d.x : INFINITY; d.y = (cb + 1 < N) ? d.y : INFINITY;
                d.z = (cb + 2 < N) ? d.z : INFINITY; d.w = (cb + 3 < N) ? d.w : INFINITY;
            }
            dq[mt] = d;
            dmin = fminf(fminf(dmin, fminf(d.x, d.y)), fminf(d.z, d.w));
        }
        // which of the wave's 16 queries have a candidate below their insertion bound in this tile?  (lane (j, g) tested the
        // 16 candidates it holds of query j.)  The others skip the tile without touching the strip.
        unsigned qm = 0xffffu;
        if (c0 != c_first) {
            const unsigned long long hm = __ballot(dmin < thrq);
            qm = (unsigned)((hm | (hm >> 16) | (hm >> 32) | (hm >> 48)) & 0xffffull);
        }
        if ((probe & 1) && c0 > c_first) qm = 0;
        if (qm == 0) continue;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<float4*>(&dW[j * KNN_LDD + 16 * mt + 4 * g]) = dq[mt];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

        const int cand = c0 + lane;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (!((qm >> i) & 1u)) continue;
            const float d = dW[i * KNN_LDD + lane];
            float ldv = ld_[i];
            int liv = li_[i];
            const float t = knn_select_mf(c0 == c_first, d, lane, cand, K2, kk, knn_readlane_f(m2e_of, i),
                                          knn_readlane_f(thrq, i), mW, ldv, liv);
            ld_[i] = ldv; li_[i] = liv;
            thrq = ((lane & 15) == i) ? t : thrq;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int q = q0 + 16 * wave + i;
        if (q < N && lane < K2)
            part[(((size_t)b * N + q) * nsplit + piece) * K2 + lane] =
                ((unsigned long long)(unsigned)__float_as_int(ld_[i]) << 32) | (unsigned)li_[i];
    }
}

// =====================================================================================================================
// The same filter on the fp16 matrix pipe (round 4): d~ from f16x3 products — every operand row x_c * 2^sh = h + l in two fp16
// terms (its own power of two sh brings the row's largest magnitude into [2^14, 2^15): exact), q.p from three
// v_mfma_f32_16x16x32_f16 (hq.hp + hq.lp + lq.hp, fp32 accumulate), undone per (candidate, query) pair by the two exact inverse
// scales.  Why it may replace the exact-product filter: the filter only has to stay inside E.  Per element
// |x_c 2^sh - h - l| <= 2^-22 |x_c 2^sh| (+ 2^-28 of the row maximum when l underflows), the dropped lq.lp term is another
// 2^-22, so the product sum is off by < 2^-20 |q||p| <= 2^-21 (|q|^2 + |p|^2), doubled by the factor 2: 16 u (|q|^2 + |p|^2),
// u = 2^-24 — the host adds 16 to the (6 C + 16) u of the bound above; the MFMA's internal fp32 accumulation is covered by
// the two-roundings-per-element allowance already in it.  The exact recheck (gpe_knn_rerank_kernel) is unchanged.
// What it buys (MI355X): 60 short fp16 MFMAs per 64 x 64 x 160 tile and wave instead of 160 fp32 ones that own their SIMD
// while they run (0.26 of the 0.76 ms); the wave's 16 queries stay RESIDENT in registers as B fragments, so a
// step stages only the candidate tile (half the loads and LDS writes; pre-split planes: no conversion work in the loop); 64
// channels per step on two LDS buffers: one barrier per step instead of two per 32 channels; and 60 KB of LDS = two workgroups
// per CU = two clouds in flight per XCD, whose plane tables (2 x 1.3 MB at the shipped size) stay in the 4 MiB L2 — the
// fp32 filter streamed four (729 MB fetched for a 39 MB table set, profiles/r03_i_hbm_traffic.json).
typedef _Float16 knn_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned knn_u32x4 __attribute__((ext_vector_type(4)));
#define KNN_H3_CCH 64                     // channels per staged step: two 32-k MFMA blocks
#define KNN_H3_PITCH 160                  // bytes per plane row in LDS: 8 data chunks of 16 B + 2 pad (chunk count = 2 mod 4:
                                          // the ds_read_b128 that walks down a column is conflict-free, gpe_edgegemm_split_kernel.h)
#define KNN_H3_MAXC 256

// planes of the feature table: pl[row] = [h plane: CP halves | l plane: CP halves] (CP = C rounded up to 32, zero pad),
// isc[row] = 2^-sh.  Wave = KNN_NORM_ROWS rows; pass 1 the rows' largest magnitudes, pass 2 the split (the rows come from L2).
// `order` (may be NULL): plane row r of a cloud holds point order[r] of that cloud — the filter then works in the caller's locality
// order throughout (queries and candidates) and translates back when it writes its lists.
__global__ __launch_bounds__(256) void gpe_knn_planes_kernel(const float* __restrict__ x, long rows, int C, int ldx, int CP,
                                                             _Float16* __restrict__ pl, float* __restrict__ isc,
                                                             const int* __restrict__ order, int N)
{
    const int lane = threadIdx.x & 63;
    const long r0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * KNN_NORM_ROWS;
    if (r0 >= rows) return;
    const long last = rows - 1;
    long src[KNN_NORM_ROWS];                              // source row of plane row r0 + u
#pragma unroll
    for (int u = 0; u < KNN_NORM_ROWS; ++u) {
        const long r = (r0 + u < last) ? r0 + u : last;
        src[u] = order ? (r / N) * N + order[r] : r;
    }
    float m[KNN_NORM_ROWS];
#pragma unroll
    for (int u = 0; u < KNN_NORM_ROWS; ++u) m[u] = 0.f;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        const int cc = (c < C) ? c : 0;
        float v[KNN_NORM_ROWS];
#pragma unroll
        for (int u = 0; u < KNN_NORM_ROWS; ++u) v[u] = x[src[u] * ldx + cc];
#pragma unroll
        for (int u = 0; u < KNN_NORM_ROWS; ++u) m[u] = (c < C) ? fmaxf(m[u], fabsf(v[u])) : m[u];
    }
    float sc[KNN_NORM_ROWS];
#pragma unroll
    for (int u = 0; u < KNN_NORM_ROWS; ++u) {
        float t = m[u];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t = fmaxf(t, __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ o) << 2, __float_as_int(t))));
        float s_, inv_;
        gpe_h3_scale_of(__float_as_uint(t), s_, inv_);
        sc[u] = s_;
        if (lane == 0 && r0 + u <= last) isc[r0 + u] = inv_;
    }
    for (int c0 = 0; c0 < CP; c0 += 64) {
        const int c = c0 + lane;
        if (c >= CP) break;
        const int cc = (c < C) ? c : 0;
        float v[KNN_NORM_ROWS];
#pragma unroll
        for (int u = 0; u < KNN_NORM_ROWS; ++u) v[u] = x[src[u] * ldx + cc];
#pragma unroll
        for (int u = 0; u < KNN_NORM_ROWS; ++u) {
            if (r0 + u <= last) {
                const float xs = (c < C) ? v[u] * sc[u] : 0.f;
                const _Float16 h = (_Float16)xs;                      // RNE
                const _Float16 l = (_Float16)(xs - (float)h);         // the difference is exact in fp32
                _Float16* row = pl + (r0 + u) * 2 * (long)CP;
                row[c] = h;
                row[CP + c] = l;
            }
        }
    }
}

template <int NBMAX>
__global__ __launch_bounds__(256, 2) void gpe_knn_h3_kernel(const _Float16* __restrict__ pl, const float* __restrict__ isc, int N,
                                                            int CP, int kk, int K2, const float* __restrict__ norms,
                                                            const int* __restrict__ cmax, float ce, int B, int tiles, int pin,
                                                            int nsplit, unsigned long long* __restrict__ part, int probe,
                                                            const int* __restrict__ ord)
{
    // ord (may be NULL): the planes are in the caller's locality order (plane row r = point ord[r] of the cloud); norms / lists are in
    // point numbering.  The scan then starts one tile before the queries' own tile (see tile_c0 below).
    const int rot = ord != nullptr;
    extern __shared__ __align__(16) float smem[];
    constexpr int PLANE_B = KNN_TC * KNN_H3_PITCH;
    constexpr int BUF_B = 2 * PLANE_B;
    char* const cB = reinterpret_cast<char*>(smem);                  // [2 buffers][2 planes][64 rows][KNN_H3_PITCH]
    float* const dS = reinterpret_cast<float*>(cB + 2 * BUF_B);
    unsigned long long* const mS = reinterpret_cast<unsigned long long*>(dS + 4 * 16 * KNN_LDD);
    float* const npS = reinterpret_cast<float*>(mS + 4 * 64);        // [2 tile parities][64] |p|^2 of the candidate tile
    float* const isS = npS + 2 * KNN_TC;                             // [2][64] inverse row scales of the candidate tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b, item;
    const int ipc = tiles * nsplit;
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
    const int tps = (tiles + nsplit - 1) / nsplit;
    const int c_first = piece * tps * KNN_TC;
    const int c_stop = ((piece + 1) * tps * KNN_TC < N) ? (piece + 1) * tps * KNN_TC : N;
    const _Float16* cloud = pl + (size_t)b * N * 2 * CP;
    const float* cnorm = norms + (size_t)b * N;
    const float* cisc = isc + (size_t)b * N;
    const int* cord = ord ? ord + (size_t)b * N : nullptr;
    auto pt = [&](int r) -> int { return cord ? cord[r] : r; };      // plane row -> point
    float* const dW = dS + wave * 16 * KNN_LDD;
    unsigned long long* const mW = mS + wave * 64;

    float ld_[16];
    int li_[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { ld_[i] = INFINITY; li_[i] = -1; }
    float thrq = INFINITY;

    const int j = lane & 15, g = lane >> 4;
    const int myq = (q0 + 16 * wave + j < N) ? q0 + 16 * wave + j : N - 1;
    const float nq = cnorm[pt(myq)];
    const float fq = -2.f * cisc[myq];                    // exact (a power of two)
    const float cm = __int_as_float(cmax[b]);
    float m2e_of;
    {
        const int qi = (q0 + 16 * wave + (lane & 15) < N) ? q0 + 16 * wave + (lane & 15) : N - 1;
        m2e_of = 2.02f * ce * (cnorm[pt(qi)] + cm);
    }
    // ---- the wave's 16 queries: resident B fragments (lane (j, g): query j, halves 32 blk + 8 g .. + 7 of both planes) ----
    const int NB = CP >> 5;
    knn_u32x4 qh[NBMAX], ql[NBMAX];
    {
        const _Float16* qrow = cloud + (size_t)myq * 2 * CP;
#pragma unroll
        for (int blk = 0; blk < NBMAX; ++blk) {
            const int bb = (blk < NB) ? blk : 0;
            qh[blk] = *reinterpret_cast<const knn_u32x4*>(qrow + 32 * bb + 8 * g);
            ql[blk] = *reinterpret_cast<const knn_u32x4*>(qrow + CP + 32 * bb + 8 * g);
        }
    }
    // ---- staging: a step = 64 candidates x 64 channels of both planes = 1024 sixteen-byte pieces, four per thread ----------
    const int nchunk = (CP + KNN_H3_CCH - 1) / KNN_H3_CCH;
    knn_u32x4 pre[4];
    float pre_n = 0.f, pre_s = 0.f;
    int pre_first = 0, pre_tp = 0;
    // visit order of the candidate tiles of this piece: rotated so that the scan STARTS one tile before the queries' own tile (rot:
    // the rows are in a locality order, e.g. the Morton order of the points — the first three tiles then hold most of the true
    // neighbours and the insertion bound is tight for the rest of the scan); rot = 0: ascending from the piece's first tile
    const int ntile_p = (c_stop - c_first + KNN_TC - 1) / KNN_TC;
    int v_start = 0;
    if (rot && ntile_p > 0) {
        v_start = qt - 1 - c_first / KNN_TC;
        v_start = v_start < 0 ? 0 : (v_start >= ntile_p ? ntile_p - 1 : v_start);
    }
    auto tile_c0 = [&](int v) -> int { int t = v_start + v; t = t >= ntile_p ? t - ntile_p : t; return c_first + t * KNN_TC; };
    int pf_v = 0;
    int pf_c0 = tile_c0(0), pf_ci = 0, pf_tp = 0;
    auto prefetch = [&]() {
        const int ch0 = pf_ci * KNN_H3_CCH;
        const int nv = ((CP - ch0) >> 3) < 8 ? ((CP - ch0) >> 3) : 8;     // valid 16-byte pieces per plane row of this chunk
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const int plane = e >> 9, row = (e >> 3) & 63, c16 = e & 7;
            const int pr = (pf_c0 + row < N) ? pf_c0 + row : N - 1;
            const int cc = (c16 < nv) ? c16 : 0;                            // pieces past CP are never read back: any valid address
            pre[i] = *reinterpret_cast<const knn_u32x4*>(cloud + (size_t)pr * 2 * CP + plane * CP + ch0 + 8 * cc);
        }
        pre_first = pf_ci == 0;
        pre_tp = pf_tp;
        if (pre_first && tid < KNN_TC) {
            const int pr = (pf_c0 + tid < N) ? pf_c0 + tid : N - 1;
            pre_n = cnorm[pt(pr)];
            pre_s = cisc[pr];
        }
        if (++pf_ci == nchunk) { pf_ci = 0; ++pf_v; pf_c0 = tile_c0(pf_v < ntile_p ? pf_v : 0); pf_tp ^= 1; }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const int plane = e >> 9, row = (e >> 3) & 63, c16 = e & 7;
            *reinterpret_cast<knn_u32x4*>(cB + buf * BUF_B + plane * PLANE_B + row * KNN_H3_PITCH + 16 * c16) = pre[i];
        }
        if (pre_first && tid < KNN_TC) { npS[pre_tp * KNN_TC + tid] = pre_n; isS[pre_tp * KNN_TC + tid] = pre_s; }
    };
    const int ntile = (c_stop - c_first + KNN_TC - 1) / KNN_TC;
    const int nsteps = ntile * nchunk;                    // uniform; 0 when a candidate piece lies past the cloud (the empty lists
    if (nsteps > 0) {                                     // are still written below)
        prefetch();
        commit(0);
        if (nsteps > 1) prefetch();
    }
    __syncthreads();

    int step = 0, buf = 0, tp = 0;
    for (int v = 0; v < ntile; ++v, tp ^= 1) {
        const int c0 = tile_c0(v);
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // chunk index at compile time (the resident query fragments are register arrays): NBMAX blocks = NCHMAX chunks at most
        constexpr int NCHMAX = (NBMAX + 1) / 2;
#pragma unroll
        for (int ci = 0; ci < NCHMAX; ++ci) {
            if (ci >= nchunk) break;                      // uniform
            if (step + 1 < nsteps) {
                commit(buf ^ 1);                          // every wave left that buffer at the barrier below
                if (step + 2 < nsteps) prefetch();
            }
            int nblk = NB - 2 * ci;
            nblk = nblk > 2 ? 2 : nblk;
            if (probe & 4) nblk = 0;
            const char* tile = cB + buf * BUF_B + j * KNN_H3_PITCH + 16 * g;
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                if (2 * ci + b2 < NBMAX && b2 < nblk) {
                    const knn_u32x4 bh = qh[(2 * ci + b2 < NBMAX) ? 2 * ci + b2 : 0], bl = ql[(2 * ci + b2 < NBMAX) ? 2 * ci + b2 : 0];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const char* src = tile + 16 * mt * KNN_H3_PITCH + 64 * b2;
                        const knn_u32x4 ah = *reinterpret_cast<const knn_u32x4*>(src);
                        const knn_u32x4 al = *reinterpret_cast<const knn_u32x4*>(src + PLANE_B);
                        // small terms first
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(knn_f16x8, al), __builtin_bit_cast(knn_f16x8, bh), acc[mt], 0, 0, 0);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(knn_f16x8, ah), __builtin_bit_cast(knn_f16x8, bl), acc[mt], 0, 0, 0);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(knn_f16x8, ah), __builtin_bit_cast(knn_f16x8, bh), acc[mt], 0, 0, 0);
                    }
                }
            }
            if (ci == nchunk - 1) {
                // d~ = |q|^2 + |p|^2 - 2 q.p with the two row scales undone, clamped at +0 (the bit pattern must order like the value)
                const bool tail = c0 + KNN_TC > N;
                float4 dq[4];
                float dmin = INFINITY;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const float4 np = *reinterpret_cast<const float4*>(&npS[tp * KNN_TC + 16 * mt + 4 * g]);
                    const float4 is = *reinterpret_cast<const float4*>(&isS[tp * KNN_TC + 16 * mt + 4 * g]);
                    float4 d;
                    d.x = __builtin_fmaf(fq * is.x, acc[mt][0], nq + np.x); d.y = __builtin_fmaf(fq * is.y, acc[mt][1], nq + np.y);
                    d.z = __builtin_fmaf(fq * is.z, acc[mt][2], nq + np.z); d.w = __builtin_fmaf(fq * is.w, acc[mt][3], nq + np.w);
                    d.x = d.x > 0.f ? d.x : 0.f; d.y = d.y > 0.f ? d.y : 0.f; d.z = d.z > 0.f ? d.z : 0.f; d.w = d.w > 0.f ? d.w : 0.f;
                    if (tail) {                                  // candidates past the cloud never qualify
                        const int cb = c0 + 16 * mt + 4 * g;
                        d.x = (cb + 0 < N) ? d.x : INFINITY; d.y = (cb + 1 < N) ? d.y : INFINITY;
                        d.z = (cb + 2 < N) ? d.z : INFINITY; d.w = (cb + 3 < N) ? d.w : INFINITY;
                    }
                    dq[mt] = d;
                    dmin = fminf(fminf(dmin, fminf(d.x, d.y)), fminf(d.z, d.w));
                }
                unsigned qm = 0xffffu;
                if (v != 0) {
                    const unsigned long long hm = __ballot(dmin < thrq);
                    qm = (unsigned)((hm | (hm >> 16) | (hm >> 32) | (hm >> 48)) & 0xffffull);
                }
                if ((probe & 1) && v > 0) qm = 0;
                if (qm != 0) {
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<float4*>(&dW[j * KNN_LDD + 16 * mt + 4 * g]) = dq[mt];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const int cand = c0 + lane;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        if (!((qm >> i) & 1u)) continue;
                        const float d = dW[i * KNN_LDD + lane];
                        float ldv = ld_[i];
                        int liv = li_[i];
                        const float t = knn_select_mf(v == 0, d, lane, cand, K2, kk, knn_readlane_f(m2e_of, i),
                                                      knn_readlane_f(thrq, i), mW, ldv, liv);
                        ld_[i] = ldv; li_[i] = liv;
                        thrq = ((lane & 15) == i) ? t : thrq;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __syncthreads();                              // `buf` is free for the commit of step + 2; buf ^ 1 is complete
            buf ^= 1;
            ++step;
        }
    }
    // lists out, in POINT numbering (the rerank reads rows and norms by point): query plane row -> point, candidate rows -> points
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int q = q0 + 16 * wave + i;
        if (q < N && lane < K2) {
            const int li = (li_[i] >= 0) ? pt(li_[i]) : li_[i];
            part[(((size_t)b * N + pt(q)) * nsplit + piece) * K2 + lane] =
                ((unsigned long long)(unsigned)__float_as_int(ld_[i]) << 32) | (unsigned)li;
        }
    }
}

// The threshold-scan filter (round 6) lives in gpe_knn_ft.hip; what gpe_knn needs of it:
#define KNN_FT_NBMAX 5                    // 32-channel blocks: C <= 160
#define KNN_FT_MAXK 32
int gpe_knn_ft_launch(int wide, long nblocks, hipStream_t s, const _Float16* planes, const float* iscale, int N, int CP, int k,
                      const float* norms, const int* cmax, float ce, int B, int qtiles, int pin, unsigned long long* part,
                      const int* rot, int probe);

// exact chain distance of one (query row, candidate row) pair per lane — oracle/knn_ref.c's arithmetic
// (16-byte loads when both rows allow it: a lane walks its own row, so the loads of the next channels must be in flight under
// the dependent fma chain — one dword load per step was a full L2 round trip per channel, 12 us per re-evaluated candidate)
__device__ __forceinline__ float knn_exact_chain(const float* __restrict__ q, const float* __restrict__ p, int C, bool vec4)
{
    float acc = 0.f;
    int c = 0;
    if (vec4) {
        // 32 channels per batch: all 16 loads are issued before the first dependent fma
        for (; c + 32 <= C; c += 32) {
            float4 qv[8], pv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                qv[u] = *reinterpret_cast<const float4*>(q + c + 4 * u);
                pv[u] = *reinterpret_cast<const float4*>(p + c + 4 * u);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc = knn_sqacc(knn_sub(qv[u].x, pv[u].x), acc);
                acc = knn_sqacc(knn_sub(qv[u].y, pv[u].y), acc);
                acc = knn_sqacc(knn_sub(qv[u].z, pv[u].z), acc);
                acc = knn_sqacc(knn_sub(qv[u].w, pv[u].w), acc);
            }
        }
        for (; c + 4 <= C; c += 4) {
            const float4 qv = *reinterpret_cast<const float4*>(q + c), pv = *reinterpret_cast<const float4*>(p + c);
            acc = knn_sqacc(knn_sub(qv.x, pv.x), acc);
            acc = knn_sqacc(knn_sub(qv.y, pv.y), acc);
            acc = knn_sqacc(knn_sub(qv.z, pv.z), acc);
            acc = knn_sqacc(knn_sub(qv.w, pv.w), acc);
        }
    }
#pragma unroll 4
    for (; c < C; ++c) acc = knn_sqacc(knn_sub(q[c], p[c]), acc);
    return acc;
}

// one query, by one wave (mW: 64 keys of LDS scratch of this wave)
__device__ __forceinline__ void knn_rerank_query(const float* __restrict__ x, long q, int N, int C, int ldx, int k, int K2, int nsplit,
                                                 const unsigned long long* __restrict__ part, const float* __restrict__ norms,
                                                 const int* __restrict__ cmax, float ce, int32_t* __restrict__ idx,
                                                 int32_t* __restrict__ idx_glob, int unsorted, unsigned long long* mW)
{
    // unsorted (gpe_knn_ft_kernel's output): 64 keys per query in no order, ~0 = empty slot, 64 valid keys = redo this query exactly
    const int lane = threadIdx.x & 63;
    const int b = (int)(q / N);
    const float* cloud = x + (size_t)b * N * ldx;
    const float* qrow = x + q * ldx;
    const bool vec4 = (ldx % 4 == 0) && ((uintptr_t)x % 16 == 0);      // every row starts 16-byte aligned
    const int nk = unsorted ? 64 : nsplit * K2;                       // <= 64
    // ---- merge the pieces' sorted lists: rank of every key among all of them (keys are distinct: distinct candidates) ----
    unsigned long long key = (lane < nk) ? part[q * nk + lane] : ~0ull;
    const int nv = __builtin_popcountll(__ballot(key != ~0ull));      // unsorted: the valid keys are a prefix of the 64
    const bool full = unsorted && nv == 64;
    if (nsplit > 1 || unsorted) {
        int rank = 0;
        const unsigned klo = (unsigned)key, khi = (unsigned)(key >> 32);
        const int ns = unsorted ? nv : nk;                            // (the empty slots stay where they are: behind every valid key)
        for (int s2 = 0; s2 < ns; ++s2) {
            const unsigned long long kn = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)khi, s2) << 32) |
                                          (unsigned)__builtin_amdgcn_readlane((int)klo, s2);
            rank += (kn < key || (kn == key && s2 < lane)) ? 1 : 0;
        }
        if (unsorted && lane >= nv) rank = lane;
        if (lane < nk) mW[rank] = key;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        key = (lane < nk) ? mW[lane] : ~0ull;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float da = __int_as_float((int)(key >> 32));                      // approximate distance (lanes >= nk: NaN pattern, unused)
    int id = (int)(unsigned)key;
    const float m2e = 2.02f * ce * (norms[q] + __int_as_float(cmax[b]));
    const float T = knn_readlane_f(da, k - 1) + m2e;
    const unsigned long long below = __ballot(lane < nk && da <= T);
    const int m = __builtin_popcountll(below);                        // a prefix of the sorted list
    const bool overflow = unsorted ? full : ((m >= K2) && (K2 < N));
    if (overflow) {
        // the list may have lost a needed candidate: this query exactly, by this wave (candidates 64 at a time)
        float ldv = INFINITY, thr = INFINITY;
        int liv = -1;
        for (int c0 = 0; c0 < N; c0 += 64) {
            const int cand = c0 + lane;
            float d = INFINITY;
            if (cand < N) d = knn_exact_chain(qrow, cloud + (size_t)cand * ldx, C, vec4);
            knn_select(c0 == 0, d, lane, cand, k, mW, ldv, liv, thr);
        }
        if (lane < k) {
            idx[q * k + lane] = liv;
            if (idx_glob) idx_glob[q * k + lane] = b * N + liv;
        }
        return;
    }
    // ---- runs of entries closer than 2E to their neighbour: exact distances, sorted inside the run by (d, index) ----------
    const bool in_s = lane < m;
    const float dnext = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, __float_as_int(da)));
    const bool near_next = (lane + 1 < m) && (dnext - da <= m2e);
    const unsigned long long nn = __ballot(near_next);
    const bool near_prev = (lane > 0) && ((nn >> (lane - 1)) & 1ull);
    const bool flagged = in_s && (near_next || near_prev);
    const unsigned long long fl = __ballot(flagged);
    int newpos = lane;
#ifdef KNN_DEBUG_DUMP
    float dbg_dex0 = -1.f, dbg_dex1 = -1.f; int dbg_r0 = -1, dbg_r1 = -1, dbg_s0 = -1;
#endif
    if (fl) {
        const unsigned long long starts = __ballot(in_s && !near_prev);
        const unsigned long long le = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);
        const int start = 63 - __builtin_clzll((starts & le) | 1ull);
        float dex = da;
        if (flagged) dex = knn_exact_chain(qrow, cloud + (size_t)id * ldx, C, vec4);
        int r = 0;
        for (int s2 = 0; s2 < m; ++s2) {
            const float dj = knn_readlane_f(dex, s2);
            const int ij = __builtin_amdgcn_readlane(id, s2);
            const int sj = __builtin_amdgcn_readlane(start, s2);
            r += (sj == start && (dj < dex || (dj == dex && ij < id))) ? 1 : 0;
        }
        if (flagged) newpos = start + r;
#ifdef KNN_DEBUG_DUMP
        dbg_dex0 = knn_readlane_f(dex, 0); dbg_dex1 = knn_readlane_f(dex, 1);
        dbg_r0 = __builtin_amdgcn_readlane(r, 0); dbg_r1 = __builtin_amdgcn_readlane(r, 1); dbg_s0 = __builtin_amdgcn_readlane(start, 0);
#endif
    }
    if (in_s && newpos < k) {
        idx[q * k + newpos] = id;
        if (idx_glob) idx_glob[q * k + newpos] = b * N + id;
    }
#ifdef KNN_DEBUG_DUMP
    if (idx_glob && lane < k) {
        int v = 0;
        if (lane == 0) v = m;
        if (lane == 1) v = (int)(unsigned)fl;
        if (lane == 2) v = __float_as_int(m2e);
        if (lane == 3) v = __float_as_int(knn_readlane_f(da, 0));
        if (lane == 4) v = __float_as_int(knn_readlane_f(da, 1));
        if (lane == 5) v = __float_as_int(T);
        if (lane == 6) v = __float_as_int(dbg_dex0);
        if (lane == 7) v = __float_as_int(dbg_dex1);
        if (lane == 8) v = dbg_r0;
        if (lane == 9) v = dbg_r1;
        if (lane == 10) v = dbg_s0;
        if (lane == 11) v = __builtin_amdgcn_readlane(id, 0);
        if (lane == 12) v = __builtin_amdgcn_readlane(newpos, 0);
        idx_glob[q * k + lane] = v;
    }
#endif
}

__global__ __launch_bounds__(256) void gpe_knn_rerank_kernel(const float* __restrict__ x, long nq_total, int N, int C, int ldx,
                                                             int k, int K2, int nsplit,
                                                             const unsigned long long* __restrict__ part,
                                                             const float* __restrict__ norms, const int* __restrict__ cmax,
                                                             float ce, int32_t* __restrict__ idx,
                                                             int32_t* __restrict__ idx_glob, int unsorted)
{
    __shared__ unsigned long long strip[4][64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long q = (long)blockIdx.x * 4 + wave;
    if (q >= nq_total) return;
    knn_rerank_query(x, q, N, C, ldx, k, K2, nsplit, part, norms, cmax, ce, idx, idx_glob, unsorted, strip[wave]);
}

// The recheck behind the threshold scan, TWO queries per wave (lanes 0..31 / 32..63): the scan's last tighten leaves ~18 keys per
// query, and a wave that walks one query's near-ties keeps ~12 of its 64 lanes busy through the 150-channel exact chains.  A pair
// whose lists do not both fit 32 lanes (or that must be re-done exactly) goes through knn_rerank_query, one query after the other.
// Same rules, same arithmetic: T = d~[k-th] + 2E, runs of keys closer than 2E re-ranked by (exact distance, index).
__global__ __launch_bounds__(256) void gpe_knn_rerank2_kernel(const float* __restrict__ x, long nq_total, int N, int C, int ldx, int k,
                                                              const unsigned long long* __restrict__ part,
                                                              const float* __restrict__ norms, const int* __restrict__ cmax,
                                                              float ce, int32_t* __restrict__ idx, int32_t* __restrict__ idx_glob)
{
    __shared__ unsigned long long strip[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long q0 = ((long)blockIdx.x * 4 + wave) * 2;
    if (q0 >= nq_total) return;
    unsigned long long* const mW = strip[wave];
    const int h = lane >> 5, l = lane & 31;
    const bool two = q0 + 1 < nq_total;
    const long q = (h && two) ? q0 + 1 : q0;                         // (an odd last query: the upper half repeats it and writes nothing)
    // the first 32 slots of both lists, and how many valid keys each list holds (a prefix of its 64 slots)
    unsigned long long key = part[q * 64 + l];
    const unsigned long long hi32 = part[q * 64 + 32 + l];
    const unsigned long long v_lo = __ballot(key != ~0ull), v_hi = __ballot(hi32 != ~0ull);
    const bool packable = (v_hi == 0ull) && k <= 32;
    if (!packable) {
        knn_rerank_query(x, q0, N, C, ldx, k, 64, 1, part, norms, cmax, ce, idx, idx_glob, 1, mW);
        if (two) knn_rerank_query(x, q0 + 1, N, C, ldx, k, 64, 1, part, norms, cmax, ce, idx, idx_glob, 1, mW);
        return;
    }
    const int nv = __builtin_popcount((unsigned)(v_lo >> (32 * h)));
    const int nvmax = max(__builtin_popcount((unsigned)v_lo), __builtin_popcount((unsigned)(v_lo >> 32)));
    const int base = 32 * h;
    // ---- rank sort of each half's valid prefix ----
    {
        int rank = 0;
        const int klo = (int)(unsigned)key, khi = (int)(unsigned)(key >> 32);
        for (int s2 = 0; s2 < nvmax; ++s2) {
            const unsigned long long kn = ((unsigned long long)(unsigned)__shfl(khi, base + s2) << 32) | (unsigned)__shfl(klo, base + s2);
            rank += (s2 < nv && (kn < key || (kn == key && s2 < l))) ? 1 : 0;
        }
        if (l >= nv) rank = l;
        mW[base + rank] = key;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        key = mW[lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const float da = __int_as_float((int)(key >> 32));               // approximate distance (empty slots: NaN pattern, never compared true)
    const int id = (int)(unsigned)key;
    const int b = (int)(q / N);
    const float* cloud = x + (size_t)b * N * ldx;
    const float* qrow = x + q * ldx;
    const bool vec4 = (ldx % 4 == 0) && ((uintptr_t)x % 16 == 0);
    const float m2e = 2.02f * ce * (norms[q] + __int_as_float(cmax[b]));
    const float T = __shfl(da, base + k - 1) + m2e;
    const unsigned long long below = __ballot(l < nv && da <= T);
    const int m = __builtin_popcount((unsigned)(below >> base));     // a prefix of the half's sorted list
    const int mmax = max(__builtin_popcount((unsigned)below), __builtin_popcount((unsigned)(below >> 32)));
    // ---- runs of entries closer than 2E to their neighbour: exact distances, sorted inside the run by (d, index) ----
    const bool in_s = l < m;
    const float dnext = __shfl(da, (lane + 1) & 63);
    const bool near_next = (l + 1 < m) && (dnext - da <= m2e);
    const unsigned long long nn = __ballot(near_next);
    const bool near_prev = (l > 0) && ((nn >> (lane - 1)) & 1ull);
    const bool flagged = in_s && (near_next || near_prev);
    const unsigned long long fl = __ballot(flagged);
    int newpos = l;
    if (fl) {
        const unsigned long long starts = __ballot(in_s && !near_prev);
        const unsigned long long le = ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull)) & (0xffffffffull << base);
        const int start = 63 - __builtin_clzll((starts & le) | (1ull << base));      // absolute lane of the run's first entry
        float dex = da;
        if (flagged) dex = knn_exact_chain(qrow, cloud + (size_t)id * ldx, C, vec4);
        int r = 0;
        for (int s2 = 0; s2 < mmax; ++s2) {
            const float dj = __shfl(dex, base + s2);
            const int ij = __shfl(id, base + s2);
            const int sj = __shfl(start, base + s2);
            r += (s2 < m && sj == start && (dj < dex || (dj == dex && ij < id))) ? 1 : 0;
        }
        if (flagged) newpos = (start - base) + r;
    }
    if (in_s && newpos < k && (h == 0 || two)) {
        idx[q * k + newpos] = id;
        if (idx_glob) idx_glob[q * k + newpos] = b * N + id;
    }
}

// what the filter path keeps per query: the K2 best candidates by the matrix-pipe distance
static int knn_k2(int k, int N)
{
    int K2 = (2 * k < 32) ? 2 * k : 32;
    if (K2 < k + 8) K2 = k + 8;
    if (K2 > 64) K2 = 64;
    if (K2 > N) K2 = N;
    return K2;
}
// bytes of the caller's workspace: partial k-lists of up to 4 candidate pieces (all-exact path: k entries each; filter path:
// K2 entries), the squared norms and the per-cloud maxima
extern "C" long gpe_knn_ws_bytes(int B, int N, int C, int k)
{
    if (B < 0 || N <= 0 || C <= 0 || k <= 0 || k > 64) return GPE_EINVAL;
    const size_t nq = (size_t)B * N;
    const size_t lists = nq * 64 * sizeof(unsigned long long) + 256;      // nsplit * K2 <= 64 and nsplit * k <= 64 by construction
    const size_t norm_bytes = (nq * sizeof(float) + 255) & ~(size_t)255;
    // fp16-pipe filter (16 <= C <= 256): two fp16 planes of the table (C rounded up to 32) + one inverse scale per row
    const size_t CP = ((size_t)C + 31) & ~(size_t)31;
    const size_t plane_bytes = (C >= KNN_MF_MINC && C <= KNN_H3_MAXC) ? ((nq * 2 * CP * sizeof(_Float16) + 255) & ~(size_t)255) + norm_bytes : 0;
    return (long)(lists + norm_bytes + ((size_t)B * sizeof(int) + 255 & ~(size_t)255) + plane_bytes + 256);
}

extern "C" int gpe_knn(const float* x, int B, int N, int C, int ldx, int k, int32_t* idx, int32_t* idx_glob, const int32_t* order_in,
                       int32_t* order_out, void* ws, long ws_bytes, void* stream)
{
    if (!x || !idx || B < 0 || N <= 0 || C <= 0 || ldx < C || k <= 0 || k > 64 || k > N || (long)B * N * k >= (1L << 31)) return GPE_EINVAL;
    if (B == 0) return GPE_OK;
    static const int force_exact = gpe_dbg_env("GPE_KNN_EXACT", 0);
    if (C < KNN_MF_MINC || k > 48 || force_exact) return knn_exact(x, B, N, C, ldx, k, idx, idx_glob, ws, ws_bytes, stream, order_out);
    if (order_out) {                                   // (only the xyz search produces an order; a filter-path caller gets the identity)
        hipLaunchKernelGGL(gpe_knn_identity_order_kernel, dim3((unsigned)gpe_cdiv((long)B * N, 256)), dim3(256), 0, (hipStream_t)stream,
                           order_out, (long)B * N, N);
        GPE_CHECK_LAUNCH();
    }
    // ---- matrix-pipe filter + exact recheck ----
    const int K2 = knn_k2(k, N);
    const int tiles = gpe_cdiv(N, KNN_TQ);
    static const int dbg_pin = gpe_dbg_env("GPE_KNN_PIN", -1);
    static const int dbg_vec = gpe_dbg_env("GPE_KNN_VEC", 0);
    static const int dbg_split = gpe_dbg_env("GPE_KNN_SPLIT", 0);
    static const int mprobe = gpe_dbg_env("GPE_KNN_PROBE", 0);   // timing aid (wrong results)
    const int pin = (dbg_pin >= 0) ? (dbg_pin && B >= GPE_NXCD) : (gpe_pin_clouds(B) ? 1 : 0);
    // No candidate split here.  knn_exact cuts the candidate range in pieces so that the tables in flight fit an L2; for this
    // path every piece would pay its own first-tile ranking and its own list build-up (selection work x 1.7 at two pieces) plus
    // a 64-key merge per query in the rerank: measured at cfg 2, layer 2: 0.99 ms with two pieces, 0.86 ms with one (the extra
    // ~360 MB of L2 misses per launch are 0.5 TB/s of HBM traffic under a kernel that is not memory-bound).  GPE_KNN_SPLIT
    // still forces pieces (the merge code stays tested).
    int nsplit = 1;
    if (dbg_split > 0 && dbg_split * K2 <= 64 && dbg_split <= tiles) nsplit = dbg_split;
    const size_t nq = (size_t)B * N;
    const size_t part_bytes = (nq * 64 * sizeof(unsigned long long) + 255) & ~(size_t)255;   // nsplit * K2 <= 64; the threshold scan: 64 keys per query
    const size_t norm_bytes = (nq * sizeof(float) + 255) & ~(size_t)255;
    const size_t cmax_bytes = ((size_t)B * sizeof(int) + 255) & ~(size_t)255;
    // the fp16-pipe filter (default for C <= 256; GPE_KNN_F32FILTER=1 keeps the exact-product filter for A/B measurements)
    static const int f32filter = gpe_dbg_env("GPE_KNN_F32FILTER", 0);
    const int CP = (C + 31) & ~31;
    const size_t pl_bytes = (nq * 2 * (size_t)CP * sizeof(_Float16) + 255) & ~(size_t)255;
    bool h3 = !f32filter && C <= KNN_H3_MAXC;
    size_t need = part_bytes + norm_bytes + cmax_bytes + 256;
    if (h3 && (!ws || (size_t)ws_bytes < need + pl_bytes + norm_bytes)) h3 = false;      // workspace sized by an older query
    if (h3) need += pl_bytes + norm_bytes;
    char* scratch = (ws && !(((uintptr_t)ws) & 15) && (size_t)ws_bytes >= need) ? (char*)ws : nullptr;
    if (!scratch) return knn_exact(x, B, N, C, ldx, k, idx, idx_glob, nullptr, 0, stream);   // no workspace: the all-exact kernel
    unsigned long long* part = (unsigned long long*)scratch;
    float* norms = (float*)(scratch + part_bytes);
    int* cmax = (int*)(scratch + part_bytes + norm_bytes);
    _Float16* planes = (_Float16*)(scratch + part_bytes + norm_bytes + cmax_bytes);
    float* iscale = (float*)(scratch + part_bytes + norm_bytes + cmax_bytes + pl_bytes);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gpe_knn_norms_kernel, dim3((unsigned)gpe_cdiv((long)nq, 4 * KNN_NORM_ROWS)), dim3(256), 0, s, x, (long)nq, N, C, ldx, norms,
                       cmax);
    GPE_CHECK_LAUNCH();
    hipLaunchKernelGGL(gpe_knn_cmax_kernel, dim3(B), dim3(256), 0, s, norms, N, cmax);
    GPE_CHECK_LAUNCH();
    // (6C + 16) * 2^-24, see the bound above; + 16 * 2^-24 for the two-term fp16 products of the fp16-pipe filter
    const float ce = (6.f * C + (h3 ? 32.f : 16.f)) * 5.9604645e-8f;
    const size_t lds = ((size_t)2 * KNN_TQ * KNN_MF_LD + 4 * 16 * KNN_LDD) * sizeof(float) + 4 * 64 * sizeof(unsigned long long) +
                       KNN_TC * sizeof(float);
    const long nblocks = (pin ? (long)GPE_NXCD * gpe_cdiv(B, GPE_NXCD) * tiles : (long)B * tiles) * nsplit;
    if (nblocks >= (1L << 31)) return GPE_EINVAL;
    const uintptr_t xa = (uintptr_t)x;
    // widest staging copy: a vector may run into the row's pad columns (their contents are replaced by zeros), so only the
    // pitch and the base address have to allow it
    int vec = (ldx % 4 == 0 && xa % 16 == 0 && ((C + 3) & ~3) <= ldx) ? 4
            : (ldx % 2 == 0 && xa % 8 == 0 && ((C + 1) & ~1) <= ldx) ? 2 : 1;
    if (dbg_vec > 0 && dbg_vec < vec) vec = dbg_vec;
    if (h3) {
        hipLaunchKernelGGL(gpe_knn_planes_kernel, dim3((unsigned)gpe_cdiv((long)nq, 4 * KNN_NORM_ROWS)), dim3(256), 0, s, x, (long)nq, C,
                           ldx, CP, planes, iscale, gpe_dbg_env("GPE_KNN_NOORDER", 0) ? nullptr : order_in, N);
        GPE_CHECK_LAUNCH();
        const size_t lds3 = (size_t)2 * 2 * KNN_TC * KNN_H3_PITCH + (size_t)4 * 16 * KNN_LDD * sizeof(float) +
                            4 * 64 * sizeof(unsigned long long) + 4 * KNN_TC * sizeof(float);
        const int NB = CP >> 5;
        static const int no_order = gpe_dbg_env("GPE_KNN_NOORDER", 0);     // A/B: ignore the caller's order (planes built above with it: keep both off)
        const int32_t* rot = no_order ? nullptr : order_in;
        static const int ft_on = gpe_dbg_env("GPE_KNN_FT", 1);              // A/B: 0 = the ordered-list filter (gpe_knn_h3_kernel)
        if (ft_on && NB <= KNN_FT_NBMAX && k <= KNN_FT_MAXK && nsplit == 1) {
            // the threshold scan: 128 queries per workgroup when that still fills the chip, 64 otherwise
            const int cus = gpe_num_cus();
            const bool wide = ft_on == 8 || (ft_on != 4 && (long)B * gpe_cdiv(N, 128) >= cus);
            const int tq = wide ? 128 : 64;
            const int qtiles = gpe_cdiv(N, tq);
            const long nb = pin ? (long)GPE_NXCD * gpe_cdiv(B, GPE_NXCD) * qtiles : (long)B * qtiles;
            const int rc = gpe_knn_ft_launch(wide ? 1 : 0, nb, s, planes, iscale, N, CP, k, norms, cmax, ce, B, qtiles, pin, part, rot, mprobe);
            if (rc != GPE_OK) return rc;
            static const int rr2 = gpe_dbg_env("GPE_KNN_RR2", 1);                // A/B: 0 = one query per wave
            if (rr2)
                hipLaunchKernelGGL(gpe_knn_rerank2_kernel, dim3((unsigned)gpe_cdiv((long)nq, 8)), dim3(256), 0, s, x, (long)nq, N, C, ldx, k,
                                   part, norms, cmax, ce, idx, idx_glob);
            else
                hipLaunchKernelGGL(gpe_knn_rerank_kernel, dim3((unsigned)gpe_cdiv((long)nq, 4)), dim3(256), 0, s, x, (long)nq, N, C, ldx, k, K2,
                                   nsplit, part, norms, cmax, ce, idx, idx_glob, 1);
            GPE_CHECK_LAUNCH();
            return GPE_OK;
        }
        if (NB <= 2)
            hipLaunchKernelGGL((gpe_knn_h3_kernel<2>), dim3((unsigned)nblocks), dim3(256), lds3, s, planes, iscale, N, CP, k, K2, norms,
                               cmax, ce, B, tiles, pin, nsplit, part, mprobe, rot);
        else if (NB <= 5)
            hipLaunchKernelGGL((gpe_knn_h3_kernel<5>), dim3((unsigned)nblocks), dim3(256), lds3, s, planes, iscale, N, CP, k, K2, norms,
                               cmax, ce, B, tiles, pin, nsplit, part, mprobe, rot);
        else
            hipLaunchKernelGGL((gpe_knn_h3_kernel<8>), dim3((unsigned)nblocks), dim3(256), lds3, s, planes, iscale, N, CP, k, K2, norms,
                               cmax, ce, B, tiles, pin, nsplit, part, mprobe, rot);
    } else if (vec == 4)
        hipLaunchKernelGGL((gpe_knn_mfma_kernel<4>), dim3((unsigned)nblocks), dim3(256), lds, s, x, N, C, ldx, k, K2, norms, cmax, ce,
                           B, tiles, pin, nsplit, part, mprobe);
    else if (vec == 2)
        hipLaunchKernelGGL((gpe_knn_mfma_kernel<2>), dim3((unsigned)nblocks), dim3(256), lds, s, x, N, C, ldx, k, K2, norms, cmax, ce,
                           B, tiles, pin, nsplit, part, mprobe);
    else
        hipLaunchKernelGGL((gpe_knn_mfma_kernel<1>), dim3((unsigned)nblocks), dim3(256), lds, s, x, N, C, ldx, k, K2, norms, cmax, ce,
                           B, tiles, pin, nsplit, part, mprobe);
    GPE_CHECK_LAUNCH();
    hipLaunchKernelGGL(gpe_knn_rerank_kernel, dim3((unsigned)gpe_cdiv((long)nq, 4)), dim3(256), 0, s, x, (long)nq, N, C, ldx, k, K2,
                       nsplit, part, norms, cmax, ce, idx, idx_glob, 0);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}
