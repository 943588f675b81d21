// Exact kNN for 3-channel clouds (the xyz search of the first EdgeConv layer, /root/reference/nn/net_blocks.py:127-135,174 through
// PyG's DynamicEdgeConv -> torch_cluster.knn) on gfx950: the all-pairs scan of gpe_knn.hip minus the candidate tiles that cannot
// matter.
//
// The definition is oracle/knn_ref.c's: d(q, p) = fmaf chain of (q_c - p_c)^2, c ascending, the k smallest (d, index) pairs in
// ascending order, self included.  What changes is only WHICH candidates a query looks at:
//   1. gpe_knn3_sort_kernel   (one workgroup per cloud) orders the cloud's points along a 16 x 16 x 16 Morton curve of their
//                             bounding box (LDS counting sort) and writes them as float4 (x, y, z, original index) plus the
//                             bounding box of every run of 64 sorted points (a "tile").
//   2. gpe_knn3_query_kernel  one wave = K3_QW = 4 consecutive sorted points (spatial neighbours).  It sorts the cloud's tiles by a lower
//                             bound of their distance to the box of its queries, scans them in that order — lane = candidate,
//                             the queries' coordinates wave-uniform, no LDS staging and no workgroup barrier — and stops at the
//                             first tile whose bound exceeds the LARGEST k-th distance among its queries: neither that tile nor any
//                             later one can hold a candidate that would be inserted.  Inside a visited tile a query whose own
//                             point-to-box bound exceeds its k-th distance skips the tile (a Gaussian cloud of 2048 points: 24 of
//                             32 tiles per wave, but 9.5 per query; a surface: 5).
// Exactness of the stop: for a query q of the wave and a point p of a tile, the real-valued squared distance is >= the squared
// box-to-box gap G.  The chain rounds at most four times (sub per axis is correctly rounded, three fma): d_chain >= G (1 - 2^-22);
// the computed bound rounds at most eight times: lb <= G (1 + 2^-21).  The kernel compares lb (1 - 2^-17) > max k-th distance,
// which implies d_chain > k-th distance of every query: the candidate would not enter any list (insertion needs (d, index) below
// the k-th pair).  Visiting order never matters for the result: lists are kept by the full (d, index) key.
// Selection: the first tile (the wave's own: it holds the queries themselves) is sorted by a 64-lane bitonic network per query;
// later tiles are merged survivor by survivor exactly like gpe_knn.hip's knn_select, with 64-bit keys.
// Measured / motivation: DESIGN.md 5.9 and 9(c) — the all-pairs kernel spends 322 us on 0.4 GFLOP at cfg 2 and 5.7 ms at N = 8192.
#include "gpe_common.h"
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define K3_MAXN 8192
#define K3_MINN 128
#define K3_CELLS 4096

__device__ __forceinline__ float k3_sub_sv(float a_uniform, float b)
{
    float d;
    asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "s"(a_uniform), "v"(b));
    return d;
}
__device__ __forceinline__ float k3_sqacc(float d, float acc)
{
    asm("v_fma_f32 %0, %1, %1, %0" : "+v"(acc) : "v"(d));
    return acc;
}
__device__ __forceinline__ unsigned long long k3_readlane64(unsigned long long v, int l)
{
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l) << 32) |
           (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
}
__device__ __forceinline__ unsigned long long k3_xor64(unsigned long long v, int lane, int j)
{
    const int a = ((lane ^ j) << 2);
    return ((unsigned long long)(unsigned)__builtin_amdgcn_ds_bpermute(a, (int)(unsigned)(v >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_ds_bpermute(a, (int)(unsigned)v);
}
// ascending bitonic sort of one DISTINCT 64-bit key per lane
__device__ __forceinline__ unsigned long long k3_sort64(unsigned long long key, int lane)
{
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const unsigned long long other = k3_xor64(key, lane, j);
            const bool takemin = (((lane & k2) == 0) == ((lane & j) == 0));
            const bool lt = other < key;
            key = (takemin == lt) ? other : key;
        }
    }
    return key;
}

__device__ __forceinline__ unsigned k3_spread(unsigned v) { return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6); }
__device__ __forceinline__ unsigned k3_code(float px, float py, float pz, const float (&lo)[3], const float (&inv)[3])
{
    int cx = (int)((px - lo[0]) * inv[0]), cy = (int)((py - lo[1]) * inv[1]), cz = (int)((pz - lo[2]) * inv[2]);
    cx = cx < 0 ? 0 : (cx > 15 ? 15 : cx);
    cy = cy < 0 ? 0 : (cy > 15 ? 15 : cy);
    cz = cz < 0 ? 0 : (cz > 15 ? 15 : cz);
    return k3_spread((unsigned)cx) | (k3_spread((unsigned)cy) << 1) | (k3_spread((unsigned)cz) << 2);
}

// xs [B][N] float4 (x, y, z, bits of the original index), tb [B][tiles][8] = {lo x, lo y, lo z, -, hi x, hi y, hi z, -}
__global__ __launch_bounds__(1024) void gpe_knn3_sort_kernel(const float* __restrict__ x, int N, int ldx, int* __restrict__ order_out, float4* __restrict__ xs,
                                                             float* __restrict__ tb, int tiles)
{
    extern __shared__ __align__(16) float k3_smem[];
    float4* const sp = reinterpret_cast<float4*>(k3_smem);                    // [N] sorted points
    unsigned* const hist = reinterpret_cast<unsigned*>(sp + N);              // [K3_CELLS]
    float* const red = reinterpret_cast<float*>(hist + K3_CELLS);            // [6][16]
    unsigned* const wsum = reinterpret_cast<unsigned*>(red + 96);            // [16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const float* cloud = x + (size_t)b * N * ldx;

    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < N; i += 1024) {
        const float* p = cloud + (size_t)i * ldx;
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], p[a]); hi[a] = fmaxf(hi[a], p[a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
        }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[a * 16 + wave] = lo[a]; red[(3 + a) * 16 + wave] = hi[a]; }
    }
    for (int c = tid; c < K3_CELLS; c += 1024) hist[c] = 0u;
    __syncthreads();
    float inv[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = red[a * 16], h = red[(3 + a) * 16];
        for (int w = 1; w < 16; ++w) { l = fminf(l, red[a * 16 + w]); h = fmaxf(h, red[(3 + a) * 16 + w]); }
        lo[a] = l;
        const float r = h - l;
        inv[a] = (r > 0.f && r < INFINITY) ? 16.f / r : 0.f;
    }
    for (int i = tid; i < N; i += 1024) {
        const float* p = cloud + (size_t)i * ldx;
        atomicAdd(&hist[k3_code(p[0], p[1], p[2], lo, inv)], 1u);
    }
    __syncthreads();
    // exclusive scan of the 4096 counts: thread t owns cells 4t .. 4t + 3
    {
        const unsigned c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
        const unsigned s = c0 + c1 + c2 + c3;
        unsigned inc = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned base = 0;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        const unsigned ex = base + inc - s;
        hist[4 * tid] = ex; hist[4 * tid + 1] = ex + c0; hist[4 * tid + 2] = ex + c0 + c1; hist[4 * tid + 3] = ex + c0 + c1 + c2;
    }
    __syncthreads();
    // (the order inside a cell depends on the atomics' arrival order: it changes which tile a point belongs to, never the result)
    for (int i = tid; i < N; i += 1024) {
        const float* p = cloud + (size_t)i * ldx;
        const float p0 = p[0], p1 = p[1], p2 = p[2];
        const unsigned pos = atomicAdd(&hist[k3_code(p0, p1, p2, lo, inv)], 1u);
        sp[pos] = make_float4(p0, p1, p2, __int_as_float(i));
    }
    __syncthreads();
    float4* out = xs + (size_t)b * N;
    for (int i = tid; i < N; i += 1024) out[i] = sp[i];
    // the curve order itself, for the caller (gpe_knn order_out): sorted position -> original index.  The layer-2 search of an
    // EdgeConv stack takes it as a LOCALITY order of its rows (gpe_knn.hip: rotated tile visits)
    if (order_out)
        for (int i = tid; i < N; i += 1024) order_out[(size_t)b * N + i] = __float_as_int(sp[i].w);
    for (int t = wave; t < tiles; t += 16) {
        const int i = 64 * t + lane;
        const float4 p = sp[i < N ? i : N - 1];                  // (N - 1 lies in the last tile: the clamp adds no foreign point)
        float l0 = p.x, l1 = p.y, l2 = p.z, h0 = p.x, h1 = p.y, h2 = p.z;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            l0 = fminf(l0, __shfl_xor(l0, o)); l1 = fminf(l1, __shfl_xor(l1, o)); l2 = fminf(l2, __shfl_xor(l2, o));
            h0 = fmaxf(h0, __shfl_xor(h0, o)); h1 = fmaxf(h1, __shfl_xor(h1, o)); h2 = fmaxf(h2, __shfl_xor(h2, o));
        }
        if (lane == 0) {
            float4* d = reinterpret_cast<float4*>(tb + ((size_t)b * tiles + t) * 8);
            d[0] = make_float4(l0, l1, l2, 0.f);
            d[1] = make_float4(h0, h1, h2, 0.f);
        }
    }
}

#define K3_INVALID_HI 0xffffffffull

// one query's selection step for one candidate tile: `key` = this lane's candidate ((distance bits << 32) | index; lanes without a
// candidate: (0xffffffff << 32) | lane), lk = this lane's entry of the query's sorted list (lanes >= k: all ones), kth = its k-th key
template <bool FIRST>
__device__ __forceinline__ void k3_select(unsigned long long key, int lane, int k, unsigned long long* mW, unsigned long long& lk,
                                          unsigned long long& kth)
{
    if (FIRST) {
        const unsigned long long s = k3_sort64(key, lane);
        lk = (lane < k) ? s : ~0ull;
        kth = k3_readlane64(lk, k - 1);
        return;
    }
    const unsigned long long m = __ballot(key < kth && (unsigned)(key >> 32) != 0xffffffffu);    // (lanes without a candidate never enter)
    if (m == 0) return;
    int shift = 0, rank = 0, pos = 0;
    unsigned long long mm = m;
    do {
        const int src = __builtin_ctzll(mm);
        mm &= mm - 1;
        const unsigned long long kn = k3_readlane64(key, src);
        shift += (kn < lk) ? 1 : 0;                           // list lanes: survivors that go in front of my entry
        rank += (kn < key) ? 1 : 0;                           // survivor lanes: survivors in front of me
        const int front = __builtin_popcountll(__ballot(lk < kn));
        pos = (lane == src) ? front : pos;                    // survivor lanes: list entries in front of me
    } while (mm);
    asm volatile("" ::: "memory");
    if (lane < k) {
        const int np = lane + shift;
        if (np < k) mW[np] = lk;
    }
    if ((m >> lane) & 1ull) {
        const int np = pos + rank;
        if (np < k) mW[np] = key;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long got = mW[lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the next query's scatter reuses the strip
    lk = (lane < k) ? got : ~0ull;
    kth = k3_readlane64(lk, k - 1);
}

// K3_QW queries per wave (consecutive sorted points), 4 waves per workgroup.  The kernel is a chain of dependent lane exchanges
// per wave; more, shorter waves hide it better than the reuse of a candidate load across many queries pays, and the box of a few
// neighbours prunes more tiles.  Measured at cfg 2 (32 x 2048 Gaussian points, k = 16; whole gpe_knn family, ms per step, the
// layer-2 search is 0.80 of it): 16 queries 1.058, 8: 1.007, 4: 0.938, 2: 0.931, 1: 0.921 (all-pairs kernel: 1.10); at N = 8192 x 64
// clouds 14.6 - 14.9 for all of them (all-pairs: 18.1).
#ifndef K3_QW
#define K3_QW 4
#endif
static_assert(64 % K3_QW == 0 && K3_QW <= 16, "a wave's queries must lie in ONE tile of 64 sorted points (the pruning argument), in registers");
__global__ __launch_bounds__(256) void gpe_knn3_query_kernel(const float4* __restrict__ xs, const float* __restrict__ tb, int N, int k,
                                                             int tiles, int wgs, int32_t* __restrict__ idx, int32_t* __restrict__ idx_glob)
{
    __shared__ unsigned long long mS[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / wgs, wg = blockIdx.x - b * wgs;              // 4 K3_QW sorted points per workgroup
    const int q0 = 4 * K3_QW * wg + K3_QW * wave;
    if (q0 >= N) return;                                                     // (no workgroup barrier below)
    const float4* cs = xs + (size_t)b * N;
    const float* cb = tb + (size_t)b * tiles * 8;
    unsigned long long* const mW = mS[wave];

    float qx[K3_QW], qy[K3_QW], qz[K3_QW];
    int qi[K3_QW];
    float bl0 = INFINITY, bl1 = INFINITY, bl2 = INFINITY, bh0 = -INFINITY, bh1 = -INFINITY, bh2 = -INFINITY;
#pragma unroll
    for (int i = 0; i < K3_QW; ++i) {
        const int s = (q0 + i < N) ? q0 + i : N - 1;
        const float4 v = cs[s];                                              // wave-uniform address
        qx[i] = v.x; qy[i] = v.y; qz[i] = v.z; qi[i] = __float_as_int(v.w);
        bl0 = fminf(bl0, v.x); bl1 = fminf(bl1, v.y); bl2 = fminf(bl2, v.z);
        bh0 = fmaxf(bh0, v.x); bh1 = fmaxf(bh1, v.y); bh2 = fmaxf(bh2, v.z);
    }
    unsigned long long lk[K3_QW], kth[K3_QW];
#pragma unroll
    for (int i = 0; i < K3_QW; ++i) { lk[i] = ~0ull; kth[i] = ~0ull; }
    unsigned maxthr = 0xffffffffu;                       // largest k-th distance (bit pattern) among the wave's queries

    auto fetch = [&](int tt) -> float4 {
        const int ci = 64 * tt + lane;
        return cs[ci < N ? ci : N - 1];
    };
    unsigned lbq[K3_QW];                                 // lane = tile of the current group: bound of query i against that tile
    auto visit = [&](int tt, int tlane, const float4 p, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const bool cv = 64 * tt + lane < N;
        const unsigned plo = cv ? (unsigned)__float_as_int(p.w) : (unsigned)lane;
        unsigned mt = 0u;
#pragma unroll
        for (int i = 0; i < K3_QW; ++i) {
            // the tile cannot hold anything below this query's k-th pair (same argument as the wave-level stop, point to box)
            const bool skip = !FIRST && (unsigned)__builtin_amdgcn_readlane((int)lbq[i], tlane) > (unsigned)(kth[i] >> 32);
            if (!skip) {
                float d = k3_sqacc(k3_sub_sv(qx[i], p.x), 0.f);
                d = k3_sqacc(k3_sub_sv(qy[i], p.y), d);
                d = k3_sqacc(k3_sub_sv(qz[i], p.z), d);
                const unsigned long long key = ((unsigned long long)(cv ? (unsigned)__float_as_int(d) : 0xffffffffu) << 32) | plo;
                k3_select<FIRST>(key, lane, k, mW, lk[i], kth[i]);
            }
            const unsigned th = (unsigned)(kth[i] >> 32);
            mt = th > mt ? th : mt;
            if (!FIRST) __builtin_amdgcn_sched_barrier(0);    // merges one query at a time (interleaved they only spill); the
        }                                                     // first tile's sorting networks are left to interleave
        maxthr = mt;
    };

    const int t0 = q0 >> 6;
    visit(t0, 0, fetch(t0), std::integral_constant<bool, true>{});
    const int ngrp = (tiles + 63) >> 6, g0 = t0 >> 6;
    for (int gi = 0; gi < ngrp; ++gi) {
        const int g = (gi == 0) ? g0 : (gi <= g0 ? gi - 1 : gi);            // the group of the own tile first
        const int t = 64 * g + lane;
        const bool tv = t < tiles && t != t0;
        unsigned long long tk = (K3_INVALID_HI << 32) | (unsigned)lane;
        {
            const int tc = (t < tiles) ? t : tiles - 1;
            const float4 tl = *reinterpret_cast<const float4*>(cb + (size_t)tc * 8);
            const float4 th = *reinterpret_cast<const float4*>(cb + (size_t)tc * 8 + 4);
#pragma unroll
            for (int i = 0; i < K3_QW; ++i) {
                const float a0 = fmaxf(fmaxf(tl.x - qx[i], qx[i] - th.x), 0.f);
                const float a1 = fmaxf(fmaxf(tl.y - qy[i], qy[i] - th.y), 0.f);
                const float a2 = fmaxf(fmaxf(tl.z - qz[i], qz[i] - th.z), 0.f);
                float l = a0 * a0;
                l = l + a1 * a1;
                l = l + a2 * a2;
                l *= 0.99999237f;
                const unsigned lb_ = (unsigned)__float_as_int(l);
                lbq[i] = lb_ <= 0x7f800000u ? lb_ : 0u;
            }
            if (tv) {
                const float g0_ = fmaxf(fmaxf(tl.x - bh0, bl0 - th.x), 0.f);
                const float g1_ = fmaxf(fmaxf(tl.y - bh1, bl1 - th.y), 0.f);
                const float g2_ = fmaxf(fmaxf(tl.z - bh2, bl2 - th.z), 0.f);
                float lb = g0_ * g0_;
                lb = lb + g1_ * g1_;
                lb = lb + g2_ * g2_;
                lb *= 0.99999237f;                                           // 1 - 2^-17
                const unsigned lbb = (unsigned)__float_as_int(lb);
                tk = ((unsigned long long)(lbb <= 0x7f800000u ? lbb : 0u) << 32) | (unsigned)lane;     // NaN boxes: never pruned
            }
        }
        tk = k3_sort64(tk, lane);
        // candidates of the NEXT tile in the order are loaded before the current one is merged (a pruned tile's load is wasted)
        float4 pn = fetch(64 * g + (int)(unsigned)k3_readlane64(tk, 0));
        for (int r = 0; r < 64; ++r) {
            const unsigned long long e = k3_readlane64(tk, r);
            const unsigned lbb = (unsigned)(e >> 32);
            if (lbb == 0xffffffffu || lbb > maxthr) break;                  // sorted: every later tile of the group is pruned too
            const float4 p = pn;
            pn = fetch(64 * g + (int)(unsigned)k3_readlane64(tk, r < 63 ? r + 1 : r));
            visit(64 * g + (int)(unsigned)e, (int)(unsigned)e, p, std::integral_constant<bool, false>{});
        }
    }
#pragma unroll
    for (int i = 0; i < K3_QW; ++i) {
        if (q0 + i < N && lane < k) {
            const size_t o = ((size_t)b * N + qi[i]) * k + lane;
            const int nb = (int)(unsigned)lk[i];
            idx[o] = nb;
            if (idx_glob) idx_glob[o] = b * N + nb;
        }
    }
}

// 1 = launched, 0 = not on this path's menu (the caller runs the all-pairs kernel), < 0 error.  ws: >= B*N*16 + B*tiles*32 bytes.
// GPE_KNN_SORTED=0 keeps the all-pairs kernel (A/B measurements, tests of the old path).
int gpe_knn3_try(const float* x, int B, int N, int ldx, int k, int32_t* idx, int32_t* idx_glob, int32_t* order_out, void* ws,
                 long ws_bytes, hipStream_t s)
{
    static const int off = gpe_dbg_env("GPE_KNN_SORTED", 1) == 0;
    if (off || N < K3_MINN || N > K3_MAXN || k > 64 || k > N) return 0;
    const int tiles = gpe_cdiv(N, 64);
    const size_t xs_bytes = (size_t)B * N * sizeof(float4), tb_bytes = (size_t)B * tiles * 8 * sizeof(float);
    if (!ws || (((uintptr_t)ws) & 15) || (size_t)ws_bytes < xs_bytes + tb_bytes) return 0;
    if ((long)B * gpe_cdiv(N, 4 * K3_QW) >= (1L << 31)) return 0;
    float4* xs = reinterpret_cast<float4*>(ws);
    float* tb = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + xs_bytes);
    const size_t lds = (size_t)N * sizeof(float4) + K3_CELLS * sizeof(unsigned) + 96 * sizeof(float) + 16 * sizeof(unsigned);
    GPE_ENSURE_MAX_LDS(gpe_knn3_sort_kernel);
    hipLaunchKernelGGL(gpe_knn3_sort_kernel, dim3(B), dim3(1024), lds, s, x, N, ldx, order_out, xs, tb, tiles);
    GPE_CHECK_LAUNCH();
    const int wgs = gpe_cdiv(N, 4 * K3_QW);
    hipLaunchKernelGGL(gpe_knn3_query_kernel, dim3((unsigned)((long)B * wgs)), dim3(256), 0, s, xs, tb, N, k, tiles, wgs, idx, idx_glob);
    GPE_CHECK_LAUNCH();
    return 1;
}
