// Producer/consumer fused row GEMM for the per-edge MLP of DynamicEdgeConv on gfx950
// (/root/reference/nn/net_blocks.py:43-47,124-135 forward; its input-gradient half in backward).
//
// Measured background (scripts/ablate_edge.py, profiles/r01_*): these kernels move 1.7-2.3 GB per launch next to
// 63-84 GFLOP of fp32 MFMA work.  With one wave per SIMD the memory phases (gather/staging, epilogue loads+stores) do
// not hide under the matrix pipe, they ADD to it (pure MFMA stream 0.57-0.70 ms, single-role kernel 1.3-1.6 ms).  So
// the work is split by ROLE across the two waves that share each SIMD — ONE persistent 512-thread workgroup per CU:
//
//   consumers (waves 0-3)  hold their slice of the packed weight (AQ N-tiles x all K) in VGPRs as ready-made MFMA B
//                          fragments for the whole kernel and do nothing but ds_read_b128 A fragments + MFMAs
//                          (v_mfma_f32_16x16x4_f32) on the 64-row tile in LDS, then drop their accumulators in C;
//   producers (waves 4-7)  do every memory operation: prefetch the neighbour rows, gather/stage the NEXT tile into the
//                          other A buffer, run the epilogue of the PREVIOUS tile from C (bias+ReLU, fp64 BN statistics,
//                          whole-row stores, max/min over each point's messages; or BN/ReLU backward with the stored
//                          activation and per-point sums), and compute the BQ left-over N-tiles (rows 16w..16w+15
//                          each) so that every SIMD issues exactly NT MFMAs per k-step.
//
// LDS: A[2] + C (3 x 64 x 212 floats = 159 KB of the CU's 160 KB at the shipped sizes).  Two barriers per tile.  The
// kernel's VGPR allocation is the max of both roles and must stay <= 256 (2 waves/SIMD): consumers 156 (weights) + 48
// (accumulators) + fragments.
#include "gpe_rowgemm.h"
#include <math.h>

// ---- split-bf16 ("bf16x3") arithmetic ------------------------------------------------------------------------------
// x = hi + lo + O(2^-18 |x|) with hi = bf16(x), lo = bf16(x - hi);  a*b ~= ah*bh + ah*bl + al*bh on the bf16 matrix pipe
// (v_mfma_f32_16x16x32_bf16, fp32 accumulate): three MFMAs at 16x the fp32-MFMA rate each, AND a bf16 stream leaves the
// SIMD's other wave free to issue (see the header).  LDS/VGPR footprint is unchanged: 2 x 2 bytes per element.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void eg_split4(const float4 v, uint2& hi, uint2& lo)
{
    const f32x2_t a = {v.x, v.y}, b = {v.z, v.w};
    const unsigned h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2_t));   // v_cvt_pk_bf16_f32 (RNE)
    const unsigned h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2_t));
    const f32x2_t ra = {v.x - __uint_as_float(h0 << 16), v.y - __uint_as_float(h0 & 0xffff0000u)};
    const f32x2_t rb = {v.z - __uint_as_float(h1 << 16), v.w - __uint_as_float(h1 & 0xffff0000u)};
    hi = make_uint2(h0, h1);
    lo = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(ra, bf16x2_t)),
                    __builtin_bit_cast(unsigned, __builtin_convertvector(rb, bf16x2_t)));
}

// One N-tile of the packed weight as bf16x3 B fragments: KCH/2 slabs of 32 k (+ a 16-k tail when KCH is odd).  Slab s of
// lane (j, g) holds k = 32s + 4g + {0..3} and 32s + 16 + 4g + {0..3} (packed chunks 2s and 2s+1); the A tile is laid
// out in LDS with the same k order (eg_kpos), so any consistent order is as good as the natural one.
template <int KCH>
struct EgWFrag {
    uint4 h[KCH / 2 > 0 ? KCH / 2 : 1], l[KCH / 2 > 0 ? KCH / 2 : 1];
    uint2 ht, lt;
    __device__ __forceinline__ void load(const float* wp, int Npad, int col, int g)
    {
        const bool on = col < Npad;
#pragma unroll
        for (int sl = 0; sl < KCH / 2; ++sl) {
            const float4 f0 = on ? ld4(wp + (((long)((2 * sl) * 4 + g)) * Npad + col) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 f1 = on ? ld4(wp + (((long)((2 * sl + 1) * 4 + g)) * Npad + col) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            uint2 h0, l0, h1, l1;
            eg_split4(f0, h0, l0); eg_split4(f1, h1, l1);
            h[sl] = make_uint4(h0.x, h0.y, h1.x, h1.y);
            l[sl] = make_uint4(l0.x, l0.y, l1.x, l1.y);
        }
        ht = lt = make_uint2(0u, 0u);
        if (KCH & 1) {
            const float4 ft = on ? ld4(wp + (((long)((KCH - 1) * 4 + g)) * Npad + col) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            eg_split4(ft, ht, lt);
        }
    }
};

// bf16 index inside a row of the LDS A tile of column quad c (c % 4 == 0): 32-slabs are stored so that lane group g of
// the MFMA finds its 8 k-values in 16 contiguous bytes
template <int KCH>
__device__ __forceinline__ int eg_kpos(int c)
{
    if (c >= 32 * (KCH / 2)) return c;                                   // 16-k tail: natural order
    return (c & ~31) + 8 * ((c >> 2) & 3) + 4 * ((c >> 4) & 1);
}

__device__ __forceinline__ f32x4 eg_mfma32(const uint4 a, const uint4 b, const f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 eg_mfma16(const uint2 a, const uint2 b, const f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_t, a), __builtin_bit_cast(s16x4_t, b), c, 0, 0, 0);
}

template <int AQ, int BQ, int KCH, int AMODE, int EMODE, int MATH>
__global__ __launch_bounds__(512, 2) void gpe_edgegemm_kernel(RgParams p, int stats_nblk)
{
    constexpr int NT = 4 * AQ + BQ;
    constexpr int LDA = 16 * KCH + 4, LDC = 16 * NT + 4;
    constexpr int PB = 16;                               // rows a producer wave keeps in flight
    extern __shared__ __align__(16) float smem[];
    float* const Abuf0 = smem;
    float* const Abuf1 = smem + RG_BM * LDA;
    float* const Cs = smem + 2 * RG_BM * LDA;            // [64][LDC]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w4 = wave & 3;
    const bool consumer = wave < 4;
    const int j = lane & 15, g = lane >> 4;
    const int PT = p.R / p.k;                            // whole points per tile

    // zero both A buffers once (pad rows R..63 and pad columns K..16*KCH are never written again)
    for (int e = tid; e < 2 * RG_BM * LDA; e += 512) smem[e] = 0.f;
    __syncthreads();

    if (consumer) {
        // =================================================================================================
        float4 wA[MATH == 0 ? AQ : 1][MATH == 0 ? KCH : 1];
        EgWFrag<KCH> wF[MATH == 1 ? AQ : 1];
#pragma unroll
        for (int i = 0; i < AQ; ++i) {
            const int col = 16 * (AQ * w4 + i) + j;
            if constexpr (MATH == 0) {
#pragma unroll
                for (int kc = 0; kc < KCH; ++kc)
                    wA[i][kc] = (col < p.Npad) ? ld4(p.wp + (((long)(kc * 4 + g)) * p.Npad + col) * 4)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                wF[i].load(p.wp, p.Npad, col, g);
            }
        }
        __syncthreads();                                 // prologue: tile 0 staged by the producers
        int buf = 0;
        for (int tile = blockIdx.x; tile < p.num_tiles; tile += gridDim.x) {
            const float* As = buf ? Abuf1 : Abuf0;
            f32x4 acc[4][AQ];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int i = 0; i < AQ; ++i) acc[mt][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (MATH == 0) {
                // A fragments are read ONE CHUNK AHEAD of the MFMAs that use them: LDS latency (inflated by the
                // producers' staging/epilogue traffic) then sits under 48 MFMAs instead of stalling the matrix pipe
                float4 an[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) an[mt] = ld4(&As[(16 * mt + j) * LDA + 4 * g]);
                if (!(p.dbg & 16))
#pragma unroll
                for (int kc = 0; kc < KCH; ++kc) {
                    float a[4][4];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        a[mt][0] = an[mt].x; a[mt][1] = an[mt].y; a[mt][2] = an[mt].z; a[mt][3] = an[mt].w;
                    }
                    if (kc + 1 < KCH) {
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) an[mt] = ld4(&As[(16 * mt + j) * LDA + 16 * (kc + 1) + 4 * g]);
                    }
                    __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of this chunk's MFMAs
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int i = 0; i < AQ; ++i) {
                            const float bv = (t == 0) ? wA[i][kc].x : (t == 1) ? wA[i][kc].y : (t == 2) ? wA[i][kc].z
                                                                                                       : wA[i][kc].w;
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt)
                                acc[mt][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][t], bv, acc[mt][i], 0, 0, 0);
                        }
                }
            } else {
                // bf16x3: row (16 mt + j), hi plane at byte 0 and lo plane at byte 32 KCH of the row
                constexpr int S32 = KCH / 2;
                const char* rowp = reinterpret_cast<const char*>(As) + (size_t)j * (4 * LDA) + 16 * g;
                if (!(p.dbg & 16)) {
#pragma unroll
                    for (int sl = 0; sl < S32; ++sl) {
                        // no software prefetch here: the bf16 stream is short next to the producers' work, and the 32
                        // extra VGPRs would spill at AQ = 3, KCH = 13
                        uint4 ah[4], al[4];
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) {
                            ah[mt] = *reinterpret_cast<const uint4*>(rowp + (size_t)mt * (64 * LDA) + 64 * sl);
                            al[mt] = *reinterpret_cast<const uint4*>(rowp + (size_t)mt * (64 * LDA) + 32 * KCH + 64 * sl);
                        }
                        // small terms first; consecutive MFMAs always hit different accumulators
#pragma unroll
                        for (int i = 0; i < AQ; ++i)
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt) acc[mt][i] = eg_mfma32(al[mt], wF[i].h[sl], acc[mt][i]);
#pragma unroll
                        for (int i = 0; i < AQ; ++i)
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt) acc[mt][i] = eg_mfma32(ah[mt], wF[i].l[sl], acc[mt][i]);
#pragma unroll
                        for (int i = 0; i < AQ; ++i)
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt) acc[mt][i] = eg_mfma32(ah[mt], wF[i].h[sl], acc[mt][i]);
                    }
                    if (KCH & 1) {
                        const char* tp = reinterpret_cast<const char*>(As) + (size_t)j * (4 * LDA) + 64 * S32 + 8 * g;
                        uint2 th[4], tl[4];
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) {
                            th[mt] = *reinterpret_cast<const uint2*>(tp + (size_t)mt * (64 * LDA));
                            tl[mt] = *reinterpret_cast<const uint2*>(tp + (size_t)mt * (64 * LDA) + 32 * KCH);
                        }
                        // Shape change on the same accumulator: safe HERE because 4*AQ-1 >= 7 other MFMAs (>= 100
                        // cycles) sit between the last 16x16x32 and the first 16x16x16 that touch a given accumulator —
                        // issued back to back the narrower one reads stale registers (see the producers' left-over tile)
#pragma unroll
                        for (int i = 0; i < AQ; ++i)
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt) acc[mt][i] = eg_mfma16(tl[mt], wF[i].ht, acc[mt][i]);
#pragma unroll
                        for (int i = 0; i < AQ; ++i)
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt) acc[mt][i] = eg_mfma16(th[mt], wF[i].lt, acc[mt][i]);
#pragma unroll
                        for (int i = 0; i < AQ; ++i)
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt) acc[mt][i] = eg_mfma16(th[mt], wF[i].ht, acc[mt][i]);
                    }
                }
            }
            __syncthreads();                             // (1) producers are done with C (epilogue of tile-1)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int i = 0; i < AQ; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Cs[(16 * mt + 4 * g + r) * LDC + 16 * (AQ * w4 + i) + j] = acc[mt][i][r];
            __syncthreads();                             // (2) C complete
            buf ^= 1;
        }
        __syncthreads();                                 // tail (T1): producers finished the last epilogue
        __syncthreads();                                 // tail (T2): statistics combined in LDS
    } else {
        // =================================================================================================
        float4 wB[(MATH == 0 && BQ > 0) ? BQ : 1][MATH == 0 ? KCH : 1];
        EgWFrag<KCH> wG[(MATH == 1 && BQ > 0) ? BQ : 1];
#pragma unroll
        for (int b = 0; b < BQ; ++b) {
            const int col = 16 * (4 * AQ + b) + j;
            if constexpr (MATH == 0) {
#pragma unroll
                for (int kc = 0; kc < KCH; ++kc)
                    wB[b][kc] = (col < p.Npad) ? ld4(p.wp + (((long)(kc * 4 + g)) * p.Npad + col) * 4)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                wG[b].load(p.wp, p.Npad, col, g);
            }
        }
        // producers issue few, latency-critical memory instructions: let them win issue arbitration against the
        // MFMA-issuing consumer wave on the same SIMD (MI355X_MICROARCH.md "Two waves per SIMD")
        __builtin_amdgcn_s_setprio(3);
        const int c = lane << 2;                         // this lane's column quad
        const int kpos = eg_kpos<KCH>(c);                // bf16x3: where that quad sits in an LDS A row
        const bool k_on = c < p.K;                       // staging lanes
        const bool n_on = c < p.N;                       // epilogue lanes
        double stS[4] = {0, 0, 0, 0}, stQ[4] = {0, 0, 0, 0};
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 cs4 = bias4, c14 = bias4, k24 = bias4, mu4 = bias4;
        if (n_on) {
            if (EMODE == E_EDGE_FWD) {
                if (p.bias) {
                    bias4.x = p.bias[c];
                    if (c + 1 < p.N) bias4.y = p.bias[c + 1];
                    if (c + 2 < p.N) bias4.z = p.bias[c + 2];
                    if (c + 3 < p.N) bias4.w = p.bias[c + 3];
                }
            } else {                                     // N % 4 == 0 guaranteed by the dispatcher
                cs4 = ld4(p.coef_out + c); c14 = ld4(p.coef_out + p.N + c);
                k24 = ld4(p.coef_out + 2 * p.N + c); mu4 = ld4(p.coef_out + 3 * p.N + c);
            }
        }

        // ---- stage one tile into As: wave w4 handles points w4, w4+4, ... (rows of a point are contiguous) ---------
        auto stage = [&](int tile, float* As, int jgv) {
            const long row0 = (long)tile * p.R;
            const int rv = (int)((p.M - row0 < p.R) ? (p.M - row0) : p.R);
            const int pts = rv / p.k;
            if (k_on) {
                for (int pt = w4; pt < pts; pt += 4) {
                    float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (AMODE == A_GATHER) pv = ld4(p.pq + ((long)tile * PT + pt) * p.ldpq + c);
                    for (int s0 = 0; s0 < p.k; s0 += PB) {
                        float4 v[PB];
#pragma unroll
                        for (int u = 0; u < PB; ++u) {
                            const int s = (s0 + u < p.k) ? s0 + u : p.k - 1;     // clamp: unconditional loads
                            const int r = pt * p.k + s;
                            if (AMODE == A_GATHER) {
                                // neighbour row from the lane-distributed prefetch (load_jg, all 64 lanes, outside any
                                // exec-masked region): read as p.jg[row0 + r] it compiles to a vector load plus
                                // s_waitcnt vmcnt(0) in front of EVERY gathered row — 16 serial round trips per point
                                const long jj = __builtin_amdgcn_readlane(jgv, r);
                                v[u] = ld4(p.pq + jj * p.ldpq + p.H + c);
                            } else {
                                // rows are 16-B aligned and padded to a multiple of 4 columns (checked by the
                                // dispatcher): one plain 16-B load, no tail path, so all PB loads stay in flight
                                v[u] = ld4(p.a.base + (row0 + r) * p.a.stride_outer + c);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < PB; ++u) {
                            if (s0 + u < p.k) {
                                float4 o = v[u];
                                if (AMODE == A_GATHER) {
                                    o.x = fmaxf(o.x + pv.x, 0.f); o.y = fmaxf(o.y + pv.y, 0.f);
                                    o.z = fmaxf(o.z + pv.z, 0.f); o.w = fmaxf(o.w + pv.w, 0.f);
                                }
                                if constexpr (MATH == 0) {
                                    st4(&As[(pt * p.k + s0 + u) * LDA + c], o);
                                } else {
                                    uint2 hi, lo;
                                    eg_split4(o, hi, lo);
                                    char* rp = reinterpret_cast<char*>(As) + (size_t)(pt * p.k + s0 + u) * (4 * LDA) + 2 * kpos;
                                    *reinterpret_cast<uint2*>(rp) = hi;
                                    *reinterpret_cast<uint2*>(rp + 32 * KCH) = lo;
                                }
                            }
                        }
                    }
                }
                // a partial last tile leaves stale rows [rv, R) from the previous occupant of this buffer
                if (rv < p.R)
                    for (int r = rv + w4; r < p.R; r += 4) st4(&As[r * LDA + c], make_float4(0.f, 0.f, 0.f, 0.f));
            }
        };
        auto load_jg = [&](int tile) -> int {            // lane L holds the neighbour row of edge row L of `tile`
            int v = 0;
            if (AMODE == A_GATHER || EMODE == E_BWD_GATHER) {
                const long gr = (long)tile * p.R + lane;
                v = p.jg[(tile < p.num_tiles && gr < p.M) ? gr : 0];
                // pin the load HERE, under the full exec mask: the value is consumed by v_readlane inside `if (k_on)`
                // regions, and the compiler is free to sink a plain load next to that use — where the lanes past K are
                // switched off and their copy of the row index would never be loaded (wild gather address, GPU fault)
                asm volatile("" : "+v"(v));
            }
            return v;
        };

        // ---- epilogue of one finished tile from C -------------------------------------------------------------------
        auto epilogue = [&](int tile, int jgv) {
            const long row0 = (long)tile * p.R;
            const int rv = (int)((p.M - row0 < p.R) ? (p.M - row0) : p.R);
            const int pts = rv / p.k;
            if (!n_on) return;
            float s32[4] = {0.f, 0.f, 0.f, 0.f}, q32[4] = {0.f, 0.f, 0.f, 0.f};
            for (int pt = w4; pt < pts; pt += 4) {
                const long gpt = (long)tile * PT + pt;
                if (EMODE == E_EDGE_FWD) {
                    float vmx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    float vmn[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
                    int imx[4] = {0, 0, 0, 0}, imn[4] = {0, 0, 0, 0};
                    for (int s = 0; s < p.k; ++s) {
                        const int r = pt * p.k + s;
                        const float4 z = ld4(&Cs[r * LDC + c]);
                        float v[4] = {fmaxf(z.x + bias4.x, 0.f), fmaxf(z.y + bias4.y, 0.f), fmaxf(z.z + bias4.z, 0.f),
                                      fmaxf(z.w + bias4.w, 0.f)};
                        st4(p.out + (row0 + r) * p.ldo + c, make_float4(v[0], v[1], v[2], v[3]));
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            s32[t] += v[t];
                            q32[t] = __builtin_fmaf(v[t], v[t], q32[t]);
                            if (v[t] > vmx[t]) { vmx[t] = v[t]; imx[t] = s; }
                            if (v[t] < vmn[t]) { vmn[t] = v[t]; imn[t] = s; }
                        }
                    }
                    if (p.agg) {
                        const long o = gpt * p.oldagg + c;
                        st4(p.mx + o, make_float4(vmx[0], vmx[1], vmx[2], vmx[3]));
                        st4(p.mn + o, make_float4(vmn[0], vmn[1], vmn[2], vmn[3]));
                        *reinterpret_cast<uchar4*>(p.oamx + o) = make_uchar4(imx[0], imx[1], imx[2], imx[3]);
                        *reinterpret_cast<uchar4*>(p.oamn + o) = make_uchar4(imn[0], imn[1], imn[2], imn[3]);
                    }
                } else {
                    float4 pv = make_float4(0.f, 0.f, 0.f, 0.f), dp = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (EMODE == E_BWD_GATHER) pv = ld4(p.pq + gpt * p.ldpq + c);
                    for (int s0 = 0; s0 < p.k; s0 += PB) {
                        float4 act[PB];
#pragma unroll
                        for (int u = 0; u < PB; ++u) {
                            const int s = (s0 + u < p.k) ? s0 + u : p.k - 1;
                            const int r = pt * p.k + s;
                            if (EMODE == E_BWD_INPLACE) act[u] = ld4(p.out + (row0 + r) * p.ldo + c);
                            else {
                                const long jj = __builtin_amdgcn_readlane(jgv, r);
                                act[u] = ld4(p.pq + jj * p.ldpq + p.H + c);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < PB; ++u) {
                            if (s0 + u < p.k) {
                                const int r = pt * p.k + s0 + u;
                                float4 av = act[u];
                                if (EMODE == E_BWD_GATHER) {
                                    av.x = fmaxf(av.x + pv.x, 0.f); av.y = fmaxf(av.y + pv.y, 0.f);
                                    av.z = fmaxf(av.z + pv.z, 0.f); av.w = fmaxf(av.w + pv.w, 0.f);
                                }
                                const float4 uu = ld4(&Cs[r * LDC + c]);
                                float4 o;
                                o.x = (av.x > 0.f) ? uu.x * cs4.x - c14.x - (av.x - mu4.x) * k24.x : 0.f;
                                o.y = (av.y > 0.f) ? uu.y * cs4.y - c14.y - (av.y - mu4.y) * k24.y : 0.f;
                                o.z = (av.z > 0.f) ? uu.z * cs4.z - c14.z - (av.z - mu4.z) * k24.z : 0.f;
                                o.w = (av.w > 0.f) ? uu.w * cs4.w - c14.w - (av.w - mu4.w) * k24.w : 0.f;
                                st4(p.out + (row0 + r) * p.ldo + c, o);
                                dp.x += o.x; dp.y += o.y; dp.z += o.z; dp.w += o.w;
                            }
                        }
                    }
                    if (EMODE == E_BWD_GATHER) st4(p.dP + gpt * p.lddp + c, dp);
                }
            }
            if (EMODE == E_EDGE_FWD) {
#pragma unroll
                for (int t = 0; t < 4; ++t) { stS[t] += (double)s32[t]; stQ[t] += (double)q32[t]; }
            }
        };

        // ---- main loop ------------------------------------------------------------------------------------------------
        int tile = blockIdx.x;
        if (tile < p.num_tiles && !(p.dbg & 1)) stage(tile, Abuf0, load_jg(tile));
        __syncthreads();                                 // prologue barrier (matches the consumers')
        int buf = 0;
        int prev = -1;
        for (; tile < p.num_tiles; tile += gridDim.x) {
            const int next = tile + gridDim.x;
            const float* As = buf ? Abuf1 : Abuf0;
            float* An = buf ? Abuf0 : Abuf1;
            const int jg_next = (AMODE == A_GATHER && next < p.num_tiles) ? load_jg(next) : 0;
            const int jg_prev = (EMODE == E_BWD_GATHER && prev >= 0) ? load_jg(prev) : 0;

            // left-over N-tiles: rows 16*w4 .. 16*w4+15 of this tile
            f32x4 accB[BQ > 0 ? BQ : 1];
#pragma unroll
            for (int b = 0; b < (BQ > 0 ? BQ : 1); ++b) accB[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (BQ > 0 && !(p.dbg & 32)) {
                if constexpr (MATH == 0) {
#pragma unroll
                    for (int kc = 0; kc < KCH; ++kc) {
                        const float4 t4 = ld4(&As[(16 * w4 + j) * LDA + 16 * kc + 4 * g]);
                        const float aw[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int b = 0; b < BQ; ++b) {
                                const float bv = (t == 0) ? wB[b][kc].x : (t == 1) ? wB[b][kc].y : (t == 2) ? wB[b][kc].z
                                                                                                           : wB[b][kc].w;
                                accB[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[t], bv, accB[b], 0, 0, 0);
                            }
                    }
                } else {
                    constexpr int S32 = KCH / 2;
                    const char* rowp = reinterpret_cast<const char*>(As) + (size_t)(16 * w4 + j) * (4 * LDA);
#pragma unroll
                    for (int sl = 0; sl < S32; ++sl) {
                        const uint4 ah = *reinterpret_cast<const uint4*>(rowp + 64 * sl + 16 * g);
                        const uint4 al = *reinterpret_cast<const uint4*>(rowp + 32 * KCH + 64 * sl + 16 * g);
#pragma unroll
                        for (int b = 0; b < BQ; ++b) accB[b] = eg_mfma32(al, wG[b].h[sl], accB[b]);
#pragma unroll
                        for (int b = 0; b < BQ; ++b) accB[b] = eg_mfma32(ah, wG[b].l[sl], accB[b]);
#pragma unroll
                        for (int b = 0; b < BQ; ++b) accB[b] = eg_mfma32(ah, wG[b].h[sl], accB[b]);
                    }
                    if (KCH & 1) {
                        // The 16-k tail gets its OWN accumulators: a v_mfma_f32_16x16x16_bf16 issued right behind a
                        // v_mfma_f32_16x16x32_bf16 with the same vDst as SrcC intermittently picks up two of the four
                        // accumulator registers before the wider instruction has written them (seen as rows 4g+{0,1}
                        // of this tile losing one product whenever the LDS wait in between happened to be short).
                        const uint2 th = *reinterpret_cast<const uint2*>(rowp + 64 * S32 + 8 * g);
                        const uint2 tl = *reinterpret_cast<const uint2*>(rowp + 32 * KCH + 64 * S32 + 8 * g);
                        f32x4 accT[BQ > 0 ? BQ : 1];
#pragma unroll
                        for (int b = 0; b < BQ; ++b) accT[b] = eg_mfma16(tl, wG[b].ht, (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
                        for (int b = 0; b < BQ; ++b) accT[b] = eg_mfma16(th, wG[b].lt, accT[b]);
#pragma unroll
                        for (int b = 0; b < BQ; ++b) accT[b] = eg_mfma16(th, wG[b].ht, accT[b]);
#pragma unroll
                        for (int b = 0; b < BQ; ++b) accB[b] += accT[b];
                    }
                }
            }
            if (prev >= 0 && !(p.dbg & 2)) epilogue(prev, jg_prev);      // overlaps the consumers' MFMAs of `tile`
            if (next < p.num_tiles && !(p.dbg & 1)) stage(next, An, jg_next);
            __syncthreads();                             // (1)
#pragma unroll
            for (int b = 0; b < BQ; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) Cs[(16 * w4 + 4 * g + r) * LDC + 16 * (4 * AQ + b) + j] = accB[b][r];
            __syncthreads();                             // (2)
            prev = tile;
            buf ^= 1;
        }
        if (prev >= 0 && !(p.dbg & 2)) epilogue(prev, (EMODE == E_BWD_GATHER) ? load_jg(prev) : 0);
        __syncthreads();                                 // tail (T1)
        if (EMODE == E_EDGE_FWD && p.stats_part && n_on) {
            double* red = reinterpret_cast<double*>(smem);      // [4 producer waves][2][16*NT]
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                red[(w4 * 2 + 0) * (16 * NT) + c + t] = stS[t];
                red[(w4 * 2 + 1) * (16 * NT) + c + t] = stQ[t];
            }
        }
        __syncthreads();                                 // tail (T2)
    }

    if (EMODE == E_EDGE_FWD && p.stats_part && tid < p.N) {
        const double* red = reinterpret_cast<const double*>(smem);
        constexpr int NC = 16 * NT;
        const double ss = (red[0 * NC + tid] + red[2 * NC + tid]) + (red[4 * NC + tid] + red[6 * NC + tid]);
        const double qq = (red[1 * NC + tid] + red[3 * NC + tid]) + (red[5 * NC + tid] + red[7 * NC + tid]);
        for (int b = blockIdx.x; b < stats_nblk; b += gridDim.x) {
            double* dst = p.stats_part + (size_t)b * 2 * p.N;
            dst[tid] = (b == (int)blockIdx.x) ? ss : 0.0;
            dst[p.N + tid] = (b == (int)blockIdx.x) ? qq : 0.0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
static int eg_num_cus() { return gpe_num_cus(); }

template <int AQ, int BQ, int KCH, int AMODE, int EMODE, int MATH>
static int eg_launch(const RgParams& p, int stats_nblk, hipStream_t s)
{
    constexpr int NT = 4 * AQ + BQ;
    constexpr int LDA = 16 * KCH + 4, LDC = 16 * NT + 4;
    const size_t lds = (size_t)RG_BM * (2 * LDA + LDC) * sizeof(float);
    GPE_ENSURE_MAX_LDS((gpe_edgegemm_kernel<AQ, BQ, KCH, AMODE, EMODE, MATH>));
    int gx = eg_num_cus();
    if (gx > p.num_tiles) gx = p.num_tiles;
    if (stats_nblk > 0 && gx > stats_nblk) gx = stats_nblk;
    hipLaunchKernelGGL((gpe_edgegemm_kernel<AQ, BQ, KCH, AMODE, EMODE, MATH>), dim3(gx), dim3(512), lds, s, p, stats_nblk);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

template <int AMODE, int EMODE, int MATH>
static int eg_dispatch_m(int NT, int KCH, const RgParams& p, int stats_nblk, hipStream_t s)
{
    if (NT == 13 && KCH == 13) return eg_launch<3, 1, 13, AMODE, EMODE, MATH>(p, stats_nblk, s);
    if (NT == 13 && KCH == 10) return eg_launch<3, 1, 10, AMODE, EMODE, MATH>(p, stats_nblk, s);
    if (NT == 10 && KCH == 13) return eg_launch<2, 2, 13, AMODE, EMODE, MATH>(p, stats_nblk, s);
    if (NT == 10 && KCH == 10) return eg_launch<2, 2, 10, AMODE, EMODE, MATH>(p, stats_nblk, s);
    return GPE_EINVAL;
}

static int g_eg_math = 0;            // 0: exact fp32 MFMA, 1: bf16x3, 2: bf16x6 where it fits, 3: f16x3 (gpe_math_set)
void gpe_edgegemm_set_math(int m) { g_eg_math = m; }

template <int AMODE, int EMODE>
static int eg_dispatch(int NT, int KCH, const RgParams& p, int stats_nblk, hipStream_t s, int math)
{
    return math == 1 ? eg_dispatch_m<AMODE, EMODE, 1>(NT, KCH, p, stats_nblk, s)
                     : eg_dispatch_m<AMODE, EMODE, 0>(NT, KCH, p, stats_nblk, s);
}

// Returns 1 and launches when the shape is on the register-stationary menu, 0 when the caller should use the generic
// LDS-streamed kernel, < 0 on a launch error.
int gpe_edgegemm_sr_try(const RgParams& p, int amode, int emode, int stats_nblk, hipStream_t s);   // gpe_edgegemm_sr.hip
int gpe_edgegemm_x6_try(const RgParams& p, int amode, int emode, int stats_nblk, hipStream_t s);   // gpe_edgegemm_x6.hip
int gpe_edgegemm_h3_try(const RgParams& p, int amode, int emode, int stats_nblk, hipStream_t s);   // gpe_edgegemm_h3.hip

int gpe_edgegemm_try(const RgParams& p, int amode, int emode, int stats_nblk, hipStream_t s)
{
    // (r02: a "forward-only bf16x3" mode was measured and dropped — 1785 garments/s, but first-layer weight gradients are
    // residuals of cancelling sums that amplify ANY 1e-5 perturbation of the stored activations ~1e3 times (1.5e-2 of
    // max|grad|): nothing short of ~24-bit operands is parity-grade, forward or backward.)
    const int math = g_eg_math;
    if (g_eg_math == 2) {                                // bf16x6: three-term split-bf16 single-role kernel where it fits
        const int r = gpe_edgegemm_x6_try(p, amode, emode, stats_nblk, s);
        if (r != 0) return r;
    }
    if (g_eg_math == 3) {                                // f16x3: two-term split-fp16 single-role kernel, every shipped shape
        const int r = gpe_edgegemm_h3_try(p, amode, emode, stats_nblk, s);
        if (r != 0) return r;
    }
    // fp16 activation rows / a lazily formed dz3 exist only in the f16x3 single-role kernels: nothing below may touch such buffers
    if (p.out_half || p.lz_g) return GPE_EINVAL;
    if ((math == 0 || g_eg_math >= 2) && !(p.dbg & 64)) {   // exact fp32: the single-role software-pipelined kernel
        const int r = gpe_edgegemm_sr_try(p, amode, emode, stats_nblk, s);
        if (r != 0) return r;
    }
    if (p.N <= 96 || p.N > 208 || p.K <= 96 || p.K > 208) return 0;
    if (emode != E_EDGE_FWD && (p.N & 3)) return 0;      // the backward epilogues use aligned 16-B coefficient loads
    if (amode == A_GATHER && (p.K & 3)) return 0;
    if (amode == A_DENSE && (p.a.inner > 0 || (p.a.stride_outer & 3) || p.a.stride_outer < ((p.K + 3) & ~3) ||
                             (((uintptr_t)p.a.base) & 15)))
        return 0;                                        // dense rows must be aligned + padded for plain 16-B loads
    if (p.R > 64 || p.k > 64) return 0;
    const int NT = (p.N <= 160) ? 10 : 13;
    const int KCH = (p.K <= 160) ? 10 : 13;
    int rc = GPE_EINVAL;
    if (amode == A_GATHER && emode == E_EDGE_FWD) rc = eg_dispatch<A_GATHER, E_EDGE_FWD>(NT, KCH, p, stats_nblk, s, math);
    else if (amode == A_DENSE && emode == E_EDGE_FWD) rc = eg_dispatch<A_DENSE, E_EDGE_FWD>(NT, KCH, p, stats_nblk, s, math);
    else if (amode == A_DENSE && emode == E_BWD_INPLACE)
        rc = eg_dispatch<A_DENSE, E_BWD_INPLACE>(NT, KCH, p, stats_nblk, s, math);
    else if (amode == A_DENSE && emode == E_BWD_GATHER)
        rc = eg_dispatch<A_DENSE, E_BWD_GATHER>(NT, KCH, p, stats_nblk, s, math);
    return rc == GPE_OK ? 1 : rc;
}
