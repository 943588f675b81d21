// The layer-2 (wide-feature) graph search's filter as a threshold scan — the kernel behind gpe_knn (gpe_knn.hip) for 16 <= C <= 160,
// k <= 32 (replaces torch_cluster.knn as reached from /root/reference/nn/net_blocks.py:127-135,174 through PyG's DynamicEdgeConv).
// Its own translation unit: gpe_knn.hip is compiled under the max-ILP scheduling strategy its ordered-list kernels were tuned with.
#include "gpe_common.h"
#include <math.h>
#include <stdint.h>

#define KNN_TC 64
typedef _Float16 knn_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned knn_u32x4 __attribute__((ext_vector_type(4)));

// =====================================================================================================================
// Round 6: the fp16-pipe filter as a THRESHOLD scan (gpe_knn_ft_kernel; C <= 160, k <= 32).  What gpe_knn_h3_kernel spends its time
// on is not arithmetic: a barrier per 64-channel step (96 per workgroup at the shipped size, each one a load round trip:
// 230 us of "skeleton") and an ordered list per query that every tile with a candidate below the bound re-sorts through LDS
// (~800 cycles per call, 265 us) — profiles/r06_c_knn_layer2.md.  Here
//   * a step is a WHOLE candidate tile (64 rows x both planes x all channels, <= 42 KB, two LDS buffers): one barrier per tile, the
//     loads of tile v + 2 in flight under the whole of tile v; 16 NW queries per workgroup (NW = 8: half the staged bytes per query);
//   * a query keeps ONE number in the loop, its bound thr (lane (j, g) holds query j's), and an UNORDERED survivor list of <= 64
//     (distance, row) keys in LDS: a candidate with d~ < thr is appended behind a per-query LDS counter — no ranking, no shifting;
//   * thr only ever comes from `tighten`: when a list would overflow, the wave finds, for its 16 queries at once, a t with
//     #{v in list + current tile : v <= t} >= k by bisection on the distance bit patterns (the values stay in the MFMA register
//     layout: 16 list entries + 16 tile distances per lane, two cross-lane adds per probe), sets thr = min(thr, t + 2E) and rebuilds
//     the lists from the registers.  The first tile fills the lists (thr = inf), the second always tightens; in the curve order the
//     bound is final after the third tile and the rest of the scan appends ~10 more candidates per query.
// Invariant (what the recheck needs): every candidate with d~ <= (k-th smallest d~ of the cloud) + 2E is in the list — thr never
// drops below (k-th smallest d~ seen so far) + 2E.  A query whose list still overflows after a tighten (more than 64 candidates
// within 2E of its k-th: duplicates, lattices) is marked and re-done exactly by the recheck, like an overflowing h3 list.
// Output: 64 keys per query, unordered, ~0 = empty; 64 valid keys = "redo exactly".
// =====================================================================================================================
#define KNN_FT_CAP 64
#define KNN_FT_LSTR 65                    // keys per list row in LDS (odd: the 16 queries of a wave spread over the banks)
#define KNN_FT_NBMAX 5                    // 32-channel blocks: C <= 160
#define KNN_FT_MAXK 32

__device__ __forceinline__ void knn_ft_append(int* cnt, unsigned long long* list, unsigned long long key)
{
    const int p = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (p < KNN_FT_CAP) list[p] = key;
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void gpe_knn_ft_kernel(const _Float16* __restrict__ pl, const float* __restrict__ isc, int N,
                                                             int CP, int kk, const float* __restrict__ norms,
                                                             const int* __restrict__ cmax, float ce, int B, int qtiles, int pin,
                                                             unsigned long long* __restrict__ part, const int* __restrict__ ord, int probe)
{
    constexpr int NT = 64 * NW;
    constexpr int TQ = 16 * NW;
    constexpr int PRE = (KNN_TC * KNN_FT_NBMAX * 8 + NT - 1) / NT;      // 16-byte pieces of a tile per thread (a row = CP / 4 pieces)
    extern __shared__ __align__(16) float smem[];
    const int pitch = 4 * CP + 32;                                        // bytes per row: [h plane | l plane] + 2 pad chunks (chunk count = 2 mod 4)
    const int bufB = KNN_TC * pitch;
    char* const cB = reinterpret_cast<char*>(smem);                       // [2 buffers][64 rows][pitch]
    unsigned long long* const listS = reinterpret_cast<unsigned long long*>(cB + 2 * bufB);   // [NW][16][KNN_FT_LSTR]
    int* const cntS = reinterpret_cast<int*>(listS + NW * 16 * KNN_FT_LSTR);                   // [NW][16]
    float* const npS = reinterpret_cast<float*>(cntS + NW * 16);          // [2 tile parities][64] |p|^2 of the candidate tile
    float* const isS = npS + 2 * KNN_TC;                                  // [2][64] inverse row scales

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b, qt;
    if (pin) {
        const int xcd = blockIdx.x & (GPE_NXCD - 1), slot = blockIdx.x >> 3;
        const int jc = slot / qtiles;
        b = xcd + GPE_NXCD * jc;
        qt = slot - jc * qtiles;
        if (b >= B) return;
    } else {
        b = blockIdx.x / qtiles;
        qt = blockIdx.x - b * qtiles;
    }
    const int q0 = qt * TQ;
    const _Float16* cloud = pl + (size_t)b * N * 2 * CP;
    const float* cnorm = norms + (size_t)b * N;
    const float* cisc = isc + (size_t)b * N;
    const int* cord = ord ? ord + (size_t)b * N : nullptr;
    auto pt = [&](int r) -> int { return cord ? cord[r] : r; };          // plane row -> point
    unsigned long long* const listW = listS + wave * 16 * KNN_FT_LSTR;
    int* const cntW = cntS + wave * 16;

    const int j = lane & 15, g = lane >> 4;
    const int myq = (q0 + 16 * wave + j < N) ? q0 + 16 * wave + j : N - 1;
    const float nq = cnorm[pt(myq)];
    const float fq = -2.f * cisc[myq];                                    // exact (a power of two)
    const float m2e = 2.02f * ce * (nq + __int_as_float(cmax[b]));
    // ---- the wave's 16 queries: resident B fragments (lane (j, g): query j, halves 32 blk + 8 g .. + 7 of both planes) ----
    const int NB = CP >> 5;
    knn_u32x4 qh[KNN_FT_NBMAX], ql[KNN_FT_NBMAX];
    {
        const _Float16* qrow = cloud + (size_t)myq * 2 * CP;
#pragma unroll
        for (int blk = 0; blk < KNN_FT_NBMAX; ++blk) {
            const int bb = (blk < NB) ? blk : 0;
            qh[blk] = *reinterpret_cast<const knn_u32x4*>(qrow + 32 * bb + 8 * g);
            ql[blk] = *reinterpret_cast<const knn_u32x4*>(qrow + CP + 32 * bb + 8 * g);
        }
    }
    // ---- staging: a step = one candidate tile, whole rows (4 CP bytes = CP / 4 sixteen-byte pieces per row) -----------------
    const int ppr = CP >> 2;
    const int total = KNN_TC * ppr;
    int prow[PRE], poff[PRE];
#pragma unroll
    for (int i = 0; i < PRE; ++i) {
        const int e = tid + NT * i;
        const int r = (e < total) ? e / ppr : -1;
        prow[i] = r;
        poff[i] = (e < total) ? 16 * (e - r * ppr) : 0;
    }
    knn_u32x4 pre[PRE];
    float pre_n = 0.f, pre_s = 0.f;
    const int ntile = (N + KNN_TC - 1) / KNN_TC;
    // visit order: rotated so that the scan starts one tile before the queries' own tiles when the rows are in a locality order
    int v_start = 0;
    if (ord) {
        v_start = q0 / KNN_TC - 1;
        v_start = v_start < 0 ? 0 : (v_start >= ntile ? ntile - 1 : v_start);
    }
    auto tile_c0 = [&](int v) -> int { int t = v_start + v; t = t >= ntile ? t - ntile : t; return t * KNN_TC; };
    auto prefetch = [&](int c0) {
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
            if (prow[i] >= 0) {
                const int pr = (c0 + prow[i] < N) ? c0 + prow[i] : N - 1;
                pre[i] = *reinterpret_cast<const knn_u32x4*>(reinterpret_cast<const char*>(cloud) + (size_t)pr * 4 * CP + poff[i]);
            }
        }
        if (tid < KNN_TC) {
            const int pr = (c0 + tid < N) ? c0 + tid : N - 1;
            pre_n = cnorm[pt(pr)];
            pre_s = cisc[pr];
        }
    };
    auto commit = [&](int buf, int tp) {
#pragma unroll
        for (int i = 0; i < PRE; ++i)
            if (prow[i] >= 0) *reinterpret_cast<knn_u32x4*>(cB + buf * bufB + prow[i] * pitch + poff[i]) = pre[i];
        if (tid < KNN_TC) { npS[tp * KNN_TC + tid] = pre_n; isS[tp * KNN_TC + tid] = pre_s; }
    };
    prefetch(tile_c0(0));
    commit(0, 0);
    if (ntile > 1) prefetch(tile_c0(1));
    if (lane < 16) cntW[lane] = 0;
    __syncthreads();

    float thr = INFINITY;
    int ovf = 0;
    float dq[16];                                         // the tile's distances: lane (j, g), entry r = candidate 16 (r / 4) + 4 g + r % 4
    // ---- tighten: all 16 queries of the wave at once (lanes of query j: g = 0..3, list entries 16 g .. 16 g + 15).
    // S = the list + the `lost` candidates of the current tile = every candidate seen so far below thr ----
    auto tighten = [&](int c, unsigned lost, int c0) {
        const int cv = c < KNN_FT_CAP ? c : KNN_FT_CAP;
        unsigned long long L[16];
        int Lb[16], Tb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) L[r] = (16 * g + r < cv) ? listW[j * KNN_FT_LSTR + 16 * g + r] : ~0ull;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            Lb[r] = (L[r] == ~0ull) ? 0x7f800000 : (int)(L[r] >> 32);
            Tb[r] = ((lost >> r) & 1u) ? __float_as_int(dq[r]) : 0x7f800000;
        }
        int lo = -1;
        int hi = __float_as_int(thr);                     // #{v in S : v <= thr} >= k once k candidates have been seen (the invariant)
        hi = hi > 0x7f800000 ? 0x7f800000 : hi;
        while (true) {
            const bool open = hi - lo > 1;
            if (!__ballot(open)) break;
            const int mid = lo + ((hi - lo) >> 1);
            int cc = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) cc += (Lb[r] <= mid) ? 1 : 0;
            if (lost) {
#pragma unroll
                for (int r = 0; r < 16; ++r) cc += (Tb[r] <= mid) ? 1 : 0;
            }
            cc += __shfl_xor(cc, 16);
            cc += __shfl_xor(cc, 32);
            if (open) {
                if (cc >= kk) { hi = mid; if (cc <= kk + 2) lo = mid - 1; }      // close enough: stop refining this query
                else lo = mid;
            }
        }
        if (!ovf) {
            // (t + 2E, and never t itself: a cloud of identical points has E = 0)
            const float tn = (hi >= 0x7f800000) ? INFINITY : fmaxf(__int_as_float(hi) + m2e, __int_as_float(hi + 1));
            thr = fminf(thr, tn);
            if (g == 0) __hip_atomic_store(cntW + j, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            int n = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) n += (__int_as_float(Lb[r]) < thr) ? 1 : 0;          // (empty slots: inf)
#pragma unroll
            for (int r = 0; r < 16; ++r) n += (__int_as_float(Tb[r]) < thr) ? 1 : 0;
            int p = __hip_atomic_fetch_add(cntW + j, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (__int_as_float(Lb[r]) < thr) { if (p < KNN_FT_CAP) listW[j * KNN_FT_LSTR + p] = L[r]; ++p; }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (__int_as_float(Tb[r]) < thr) {
                    if (p < KNN_FT_CAP)
                        listW[j * KNN_FT_LSTR + p] = ((unsigned long long)(unsigned)Tb[r] << 32) | (unsigned)(c0 + 16 * (r >> 2) + 4 * g + (r & 3));
                    ++p;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // still too many within 2E of the k-th: the recheck does this query exactly
            if (__hip_atomic_load(cntW + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > KNN_FT_CAP) ovf = 1;
        }
    };
#ifdef KNN_FT_TIMING
    unsigned long long tS[6] = {0, 0, 0, 0, 0, 0};
#define FT_T(x) const unsigned long long x = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define FT_T(x)
#endif
    for (int v = 0; v < ntile; ++v) {
        const int buf = v & 1, tp = v & 1;
        const int c0 = tile_c0(v);
        FT_T(t0);
        if (v + 1 < ntile) {
            commit(buf ^ 1, tp ^ 1);                      // every wave left that buffer at the barrier below
            if (v + 2 < ntile) prefetch(tile_c0(v + 2));
        }
        FT_T(t1);
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const char* tile = cB + buf * bufB + j * pitch + 16 * g;
#pragma unroll
        for (int blk = 0; blk < KNN_FT_NBMAX; ++blk) {
            if (blk < NB && !(probe & 4)) {               // uniform
                const knn_u32x4 bh = qh[blk], bl = ql[blk];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const char* src = tile + 16 * mt * pitch + 64 * blk;
                    const knn_u32x4 ah = *reinterpret_cast<const knn_u32x4*>(src);
                    const knn_u32x4 al = *reinterpret_cast<const knn_u32x4*>(src + 2 * CP);
                    // small terms first
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(knn_f16x8, al), __builtin_bit_cast(knn_f16x8, bh), acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(knn_f16x8, ah), __builtin_bit_cast(knn_f16x8, bl), acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(knn_f16x8, ah), __builtin_bit_cast(knn_f16x8, bh), acc[mt], 0, 0, 0);
                }
            }
        }
        // d~ = |q|^2 + |p|^2 - 2 q.p with the two row scales undone, clamped at +0 (the bit pattern must order like the value)
        const bool tail = c0 + KNN_TC > N;
        float dmin = INFINITY;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const float4 np = *reinterpret_cast<const float4*>(&npS[tp * KNN_TC + 16 * mt + 4 * g]);
            const float4 is = *reinterpret_cast<const float4*>(&isS[tp * KNN_TC + 16 * mt + 4 * g]);
            float4 d;
            d.x = __builtin_fmaf(fq * is.x, acc[mt][0], nq + np.x); d.y = __builtin_fmaf(fq * is.y, acc[mt][1], nq + np.y);
            d.z = __builtin_fmaf(fq * is.z, acc[mt][2], nq + np.z); d.w = __builtin_fmaf(fq * is.w, acc[mt][3], nq + np.w);
            d.x = d.x > 0.f ? d.x : 0.f; d.y = d.y > 0.f ? d.y : 0.f; d.z = d.z > 0.f ? d.z : 0.f; d.w = d.w > 0.f ? d.w : 0.f;
            if (tail) {                                   // candidates past the cloud never qualify
                const int cb = c0 + 16 * mt + 4 * g;
                d.x = (cb + 0 < N) ? d.x : INFINITY; d.y = (cb + 1 < N) ? d.y : INFINITY;
                d.z = (cb + 2 < N) ? d.z : INFINITY; d.w = (cb + 3 < N) ? d.w : INFINITY;
            }
            dq[4 * mt + 0] = d.x; dq[4 * mt + 1] = d.y; dq[4 * mt + 2] = d.z; dq[4 * mt + 3] = d.w;
            dmin = fminf(fminf(dmin, fminf(d.x, d.y)), fminf(d.z, d.w));
        }
        const bool hit = (dmin < thr) && !ovf && !((probe & 1) && v > 2);
        FT_T(t3);
#ifdef KNN_FT_TIMING
        if (__ballot(hit)) tS[5] += 1;
#endif
        if (__ballot(hit)) {
            unsigned lost = 0u;                           // this lane's candidates below the bound that found the list full
            if (hit) {
                // one counter round trip per lane and tile: count, reserve, write
                int n = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) n += (dq[r] < thr) ? 1 : 0;
                int p = __hip_atomic_fetch_add(cntW + j, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (dq[r] < thr) {
                        if (p < KNN_FT_CAP)
                            listW[j * KNN_FT_LSTR + p] = ((unsigned long long)(unsigned)__float_as_int(dq[r]) << 32) |
                                                         (unsigned)(c0 + 16 * (r >> 2) + 4 * g + (r & 3));
                        else
                            lost |= 1u << r;
                        ++p;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int c = __hip_atomic_load(cntW + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef KNN_FT_TIMING
            if (__ballot(c > KNN_FT_CAP && !ovf)) tS[4] += 1;
#endif
            if (__ballot(c > KNN_FT_CAP && !ovf)) tighten(c, lost, c0);
        }
        FT_T(t4);
        __syncthreads();                                  // the other buffer is complete; this one is free for the commit of v + 2
        FT_T(t5);
#ifdef KNN_FT_TIMING
        tS[0] += t1 - t0; tS[1] += t3 - t1; tS[2] += t4 - t3; tS[3] += t5 - t4;
#endif
    }
    // one last tighten: what goes to the recheck is what lies within 2E of (about) the k-th — the recheck ranks only those
    {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int c = __hip_atomic_load(cntW + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        tighten(c, 0u, 0);
    }
    // lists out, unordered, in POINT numbering (query plane row -> point, candidate rows -> points); ~0 = empty, 64 valid keys = redo
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
        const int q = q0 + 16 * wave + i;
        if (q >= N) break;
        int ci = __hip_atomic_load(cntW + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ci = ci < KNN_FT_CAP ? ci : KNN_FT_CAP;
        const int oi = __builtin_amdgcn_readlane(ovf, i);
        unsigned long long e = ~0ull;
        if (lane < ci) {
            e = listW[i * KNN_FT_LSTR + lane];
            e = (e & 0xffffffff00000000ull) | (unsigned)pt((int)(unsigned)e);
        }
        if (oi) e = (unsigned long long)(unsigned)lane;   // 64 valid keys
#ifdef KNN_FT_TIMING
        if (i == 0) { e = ~0ull; for (int u = 0; u < 6; ++u) if (lane == u) e = (tS[u] << 32) | (unsigned)u; }
#endif
        part[((size_t)b * N + pt(q)) * KNN_FT_CAP + lane] = e;
    }
}


// host side (called by gpe_knn): wide = 128 queries per workgroup (8 waves), else 64 (4 waves)
int gpe_knn_ft_launch(int wide, long nblocks, hipStream_t s, const _Float16* planes, const float* iscale, int N, int CP, int k,
                      const float* norms, const int* cmax, float ce, int B, int qtiles, int pin, unsigned long long* part,
                      const int* rot, int probe)
{
    const int tq = wide ? 128 : 64;
    const size_t ldsf = (size_t)2 * KNN_TC * (4 * CP + 32) + (size_t)(tq / 16) * 16 * KNN_FT_LSTR * sizeof(unsigned long long) +
                        (size_t)tq * sizeof(int) + 4 * KNN_TC * sizeof(float);
    if (wide) {
        GPE_ENSURE_MAX_LDS(gpe_knn_ft_kernel<8>);
        hipLaunchKernelGGL((gpe_knn_ft_kernel<8>), dim3((unsigned)nblocks), dim3(512), ldsf, s, planes, iscale, N, CP, k, norms, cmax, ce, B,
                           qtiles, pin, part, rot, probe);
    } else {
        GPE_ENSURE_MAX_LDS(gpe_knn_ft_kernel<4>);
        hipLaunchKernelGGL((gpe_knn_ft_kernel<4>), dim3((unsigned)nblocks), dim3(256), ldsf, s, planes, iscale, N, CP, k, norms, cmax, ce, B,
                           qtiles, pin, part, rot, probe);
    }
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}
