// Wavefront execution of stacked recurrences (nn.LSTM / nn.GRU under the reference's decoders,
// /root/reference/nn/net_blocks.py:336-497) on gfx950.
//
// A stack of L layers over T steps is L*T dependent cells when run layer by layer, but cell (l, t) only needs (l, t-1)
// and (l-1, t): all cells of one anti-diagonal d = l + t are independent.  The two entry points below walk the T + L - 1
// diagonals and launch every cell of a diagonal TOGETHER (blockIdx.z = cell):
//   forward : one fused launch per diagonal — gates = h_{l,t-1}.W_hh^T (+ h_{l-1,t}.W_ih^T for layers > 0, a second K
//             segment) + addend, then the LSTM / GRU cell update in the epilogue (gate-interleaved packed weights, so
//             the gates of a unit sit in one accumulator row);
//   backward: two launches per diagonal — split-K products dh = dG_{l,t+1}.W_hh + dG_{l+1,t}.W_ih over all cells and
//             K slabs at once, then the pointwise cell backward of all cells.
// The shipped decoders (T = 23, L = 2 and T = 14, L = 3) drop from 88 + 176 dependent launches to 40 + 80.
// Arithmetic: exact fp32 MFMA (v_mfma_f32_16x16x4_f32), same staging scheme as gpe_smallgemm.hip: a workgroup stages a
// whole K slab (<= 256) of 64 rows and of its packed weight block, one barrier pair per slab.
#include "gpe_rowgemm.h"
#include <math.h>
#include <stdlib.h>

extern "C" int gpe_math_get(void);
extern "C" int gpe_debug_get(void);
extern "C" long gpe_packed_size(int N, int K);

// gpe_rnn_persist.hip: the whole stack as ONE persistent launch (LSTM, <= 256 units, one workgroup per CU); 1 = launched,
// 0 = not eligible, < 0 = error
long gpe_rnn_persist_ws_bytes(int gates, int L, int T, int Bn, int H, int bwd);
int gpe_rnn_persist_fwd(int L, int T, int Bn, int H, const float* xproj0, long xp0_sb, long xp0_st, const void* const* whh,
                        const void* const* wih, const void* const* bias, float* hs, long hs_sl, long hs_sb, long hs_st, float* cs,
                        long cs_sl, long cs_st, float* saved, long sv_sl, long sv_st, bool h3, const void* const* whh_amax,
                        const void* const* wih_amax, void* ws, long ws_bytes, hipStream_t s);
int gpe_rnn_persist_bwd(int L, int T, int Bn, int H, const float* dtop, long dt_sb, long dt_st, const float* d_hN,
                        const float* d_cN, const void* const* whh_t, const void* const* wih_t, int KP, const float* cs, long cs_sl,
                        long cs_st, const float* saved, long sv_sl, long sv_st, float* dgx, long dg_sl, long dg_sb, long dg_st,
                        float* carry, bool h3, const void* const* whh_amax, const void* const* wih_amax, void* ws, long ws_bytes,
                        hipStream_t s);

// gpe_rnn_persist_mt.hip: the same for stacks with several row tiles per workgroup (f16x3 only; waves own row tiles)
long gpe_rnn_pm_ws_bytes(int gates, int L, int T, int Bn, int H, int bwd);
int gpe_rnn_pm_fwd(int L, int T, int Bn, int H, const float* xproj0, long xp0_sb, long xp0_st, const void* const* whh,
                   const void* const* wih, const void* const* bias, float* hs, long hs_sl, long hs_sb, long hs_st, float* cs,
                   long cs_sl, long cs_st, float* saved, long sv_sl, long sv_st, const void* const* whh_amax,
                   const void* const* wih_amax, void* ws, long ws_bytes, hipStream_t s);

#define WV_MAXCELL 4

struct WvFwdCell {
    const float* a0; long a0_stride; const float* w0;       // h_{l,t-1} rows, gate-packed W_hh_l
    const float* a1; long a1_stride; const float* w1;       // h_{l-1,t} rows, gate-packed W_ih_l (NULL for layer 0)
    const unsigned* s0; const unsigned* s1;                 // H3: amax words of W_hh_l / W_ih_l (w0 / w1 are then plane packs)
    const float* xproj; long xp_stride;                     // addend rows [..][G*H] (stride 0: one bias row)
    const float* c_prev; float* c_out;                      // LSTM
    const float* bhn;                                       // GRU
    float* saved; float* h_out; long h_stride;              // saved gates [Bn][4H]; h rows
};
struct WvFwdParams { int Bn, H, Npad, ncell; WvFwdCell cell[WV_MAXCELL]; };

__device__ __forceinline__ float wv_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// stage rows [row0, row0+64) x K columns of `a` (row pitch `stride`, 16-B aligned, padded to 4) and the packed weight block
// of this workgroup (columns n0 .. n0+16*NT of a [K/4][Npad][4] packed matrix) into LDS, then run the slab's MFMAs
template <int NT, int KS>
__device__ __forceinline__ void wv_segment(const float* __restrict__ a, long stride, const float* __restrict__ wp, int K,
                                           int Npad, int row0, int rv, int n0, float* As, float* Ws, int lda,
                                           f32x4 (&acc)[NT], bool split = false)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    for (int ks = 0; ks < K; ks += KS) {
        const int kslab = (K - ks < KS) ? (K - ks) : KS;
        const int kp = (kslab + 15) & ~15;
        __syncthreads();
        {
            const int c = lane << 2;
            if (c < kp) {
                const int nvalid = kslab - c;
                const int cc = (nvalid > 0) ? c : 0;
                float4 v[RG_BM / 4];
#pragma unroll
                for (int q = 0; q < RG_BM / 4; ++q) {
                    const int r = wave + 4 * q;
                    v[q] = ld4(a + (long)(row0 + (r < rv ? r : rv - 1)) * stride + ks + cc);
                }
#pragma unroll
                for (int q = 0; q < RG_BM / 4; ++q) {
                    const int r = wave + 4 * q;
                    float4 o = v[q];
                    if (r >= rv || nvalid <= 0) o = make_float4(0.f, 0.f, 0.f, 0.f);
                    else {
                        if (nvalid < 2) o.y = 0.f;
                        if (nvalid < 3) o.z = 0.f;
                        if (nvalid < 4) o.w = 0.f;
                    }
                    st4(&As[r * lda + c], o);
                }
            }
        }
        {
            const int planes = (kp >> 4) * 4;
            constexpr int per_plane = 16 * NT;
            const int chunk0 = ks >> 4;
            const int total = planes * per_plane;
            constexpr int WB = 8;
            for (int e0 = tid; e0 < total; e0 += 256 * WB) {
                float4 v[WB];
#pragma unroll
                for (int u = 0; u < WB; ++u) {
                    const int e = e0 + 256 * u;
                    const int ec = (e < total) ? e : total - 1;
                    const int pl = ec / per_plane, n = ec - pl * per_plane;
                    const int nn = (n0 + n < Npad) ? n0 + n : Npad - 1;
                    v[u] = ld4(wp + (((long)(chunk0 * 4 + pl)) * Npad + nn) * 4);
                }
#pragma unroll
                for (int u = 0; u < WB; ++u) {
                    const int e = e0 + 256 * u;
                    if (e < total) {
                        const int n = e % per_plane;
                        st4(&Ws[e * 4], (n0 + n < Npad) ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f));
                    }
                }
            }
        }
        __syncthreads();
        const int nchunks = kp >> 4;
        // split: see wv_mma (defined below) — row tile (wave & 1), K half (wave >> 1)
        const int rt = split ? (wave & 1) : wave;
        const int half = (nchunks + 1) >> 1;
        const int kc0 = (split && (wave >> 1)) ? half : 0;
        const int kc1 = (split && !(wave >> 1)) ? half : nchunks;
        for (int kc = kc0; kc < kc1; ++kc) {
            const float4 a4 = ld4(&As[(16 * rt + j) * lda + kc * 16 + 4 * g]);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
            float4 b4[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) b4[n] = ld4(&Ws[(((kc * 4 + g) * 16 * NT) + 16 * n + j) * 4]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const float bv = (t == 0) ? b4[n].x : (t == 1) ? b4[n].y : (t == 2) ? b4[n].z : b4[n].w;
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv, acc[n], 0, 0, 0);
                }
            }
        }
    }
}

// ---- the same slab staging, split into fetch (global -> registers) / commit (registers -> LDS) / multiply, so that the
// forward kernel can keep the NEXT slab's loads in flight under the current slab's MFMAs (a cell is 2-4 slabs: with the
// loads exposed every slab cost ~2.5 us of L2 latency on top of ~2 us of matrix work) ----------------------------------------
template <int NT, int KS>
struct WvRegs {
    float4 a[RG_BM / 4];
    float4 w[(KS * NT + 63) / 64];           // (KS/4 planes) x (16*NT columns) float4 over 256 threads
};

// H3 (f16x3 arithmetic, round 4): `wp` is the fp16-PLANE pack of the weight (gpe_pack_multi kinds 8 / 10: [plane][K / 8][Npad][8
// halves]); a slab's share of it has the same number of 16-byte pieces as the fp32 slab, in the order [plane][k-group][column],
// and lands in LDS in that order — the B fragments of v_mfma_f32_16x16x32_f16 are then single ds_read_b128.  The A slab stays
// fp32 in LDS (zero-filled to a multiple of 32 columns) and is split at fragment-read time (each element is read by ONE wave).
template <int NT, int KS, bool H3 = false>
__device__ __forceinline__ void wv_fetch(WvRegs<NT, KS>& R, const float* __restrict__ a, long stride,
                                         const float* __restrict__ wp, int K, int Npad, int row0, int rv, int n0, int ks)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kslab = (K - ks < KS) ? (K - ks) : KS;
    const int kp = (kslab + 15) & ~15;
    const int c = lane << 2;
    const int cc = (c < kslab) ? c : 0;                       // clamped: every load below is unconditional
#pragma unroll
    for (int q = 0; q < RG_BM / 4; ++q) {
        const int r = wave + 4 * q;
        R.a[q] = ld4(a + (long)(row0 + (r < rv ? r : rv - 1)) * stride + ks + cc);
    }
    constexpr int per_plane = 16 * NT;
    if constexpr (H3) {
        const int kg_slab = ((kslab + 31) & ~31) >> 3;                    // k-groups of 8 in this slab (whole 32-k steps)
        const int KG = ((K + 31) & ~31) >> 3;                             // ... in the whole K extent of the pack
        const int half = kg_slab * per_plane, total = 2 * half;
#pragma unroll
        for (int u = 0; u < (KS * NT + 63) / 64; ++u) {
            const int e = tid + 256 * u;
            const int ec = (e < total) ? e : total - 1;
            const int plane = ec >= half, r2 = ec - plane * half;
            const int kg = r2 / per_plane, n = r2 - kg * per_plane;
            const int nn = (n0 + n < Npad) ? n0 + n : Npad - 1;
            R.w[u] = ld4(wp + (((long)plane * KG + (ks >> 3) + kg) * Npad + nn) * 4);      // 16-byte pieces = "4 floats"
        }
        return;
    }
    const int total = (kp >> 2) * per_plane;
    const int chunk0 = ks >> 4;
#pragma unroll
    for (int u = 0; u < (KS * NT + 63) / 64; ++u) {
        const int e = tid + 256 * u;
        const int ec = (e < total) ? e : total - 1;
        const int pl = ec / per_plane, n = ec - pl * per_plane;
        const int nn = (n0 + n < Npad) ? n0 + n : Npad - 1;
        R.w[u] = ld4(wp + (((long)(chunk0 * 4 + pl)) * Npad + nn) * 4);
    }
}

template <int NT, int KS, bool H3 = false>
__device__ __forceinline__ void wv_commit(const WvRegs<NT, KS>& R, int K, int Npad, int rv, int n0, int ks, float* As,
                                          float* Ws, int lda)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kslab = (K - ks < KS) ? (K - ks) : KS;
    const int kp = H3 ? ((kslab + 31) & ~31) : ((kslab + 15) & ~15);
    const int c = lane << 2;
    if (c < kp) {
        const int nvalid = kslab - c;
#pragma unroll
        for (int q = 0; q < RG_BM / 4; ++q) {
            const int r = wave + 4 * q;
            float4 o = R.a[q];
            if (r >= rv || nvalid <= 0) o = make_float4(0.f, 0.f, 0.f, 0.f);
            else {
                if (nvalid < 2) o.y = 0.f;
                if (nvalid < 3) o.z = 0.f;
                if (nvalid < 4) o.w = 0.f;
            }
            st4(&As[r * lda + c], o);
        }
    }
    constexpr int per_plane = 16 * NT;
    const int total = H3 ? 2 * (kp >> 3) * per_plane : (kp >> 2) * per_plane;
#pragma unroll
    for (int u = 0; u < (KS * NT + 63) / 64; ++u) {
        const int e = tid + 256 * u;
        if (e < total) {
            const int n = e % per_plane;
            st4(&Ws[e * 4], (n0 + n < Npad) ? R.w[u] : make_float4(0.f, 0.f, 0.f, 0.f));
        }
    }
}

// f16x3 products of one staged slab: A rows fp32 in LDS, scaled by the power of two `sA` and split into two fp16 terms right
// after the read; B = the weight's planes in LDS; three MFMAs per 16 x 16 x 32 block, small terms first.  `split` as in wv_mma
// (a <= 32-row tile: wave w = row tile (w & 1), K half (w >> 1)).
typedef _Float16 wv_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 wv_f16x2 __attribute__((ext_vector_type(2)));
typedef float wv_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned wv_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wv_split2(float a, float b, float s, unsigned& h, unsigned& l)
{
    const wv_f32x2 v = {a * s, b * s};
    const wv_f16x2 hh = __builtin_convertvector(v, wv_f16x2);
    const wv_f32x2 r = v - __builtin_convertvector(hh, wv_f32x2);
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, wv_f16x2));
}
template <int NT>
__device__ __forceinline__ void wv_mma_h3(const float* As, const float* Ws, int lda, int kp, float sA, f32x4 (&acc)[NT],
                                          bool split = false)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int nsteps = kp >> 5;                           // kp is a multiple of 32 here
    const int rt = split ? (wave & 1) : wave;
    const int half = (nsteps + 1) >> 1;
    const int s0 = (split && (wave >> 1)) ? half : 0;
    const int s1 = (split && !(wave >> 1)) ? half : nsteps;
    const int plane_f = (kp >> 3) * 16 * NT * 4;          // floats per plane of the staged slab
    for (int st = s0; st < s1; ++st) {
        const float* ar = &As[(16 * rt + j) * lda + 32 * st + 8 * g];
        const float4 a0 = ld4(ar), a1 = ld4(ar + 4);
        wv_u32x4 ah, al;
        { unsigned h, l; wv_split2(a0.x, a0.y, sA, h, l); ah[0] = h; al[0] = l; }
        { unsigned h, l; wv_split2(a0.z, a0.w, sA, h, l); ah[1] = h; al[1] = l; }
        { unsigned h, l; wv_split2(a1.x, a1.y, sA, h, l); ah[2] = h; al[2] = l; }
        { unsigned h, l; wv_split2(a1.z, a1.w, sA, h, l); ah[3] = h; al[3] = l; }
        const float* wb = &Ws[(((4 * st + g) * 16 * NT) + j) * 4];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const wv_u32x4 bh = *reinterpret_cast<const wv_u32x4*>(wb + 64 * n);
            const wv_u32x4 bl = *reinterpret_cast<const wv_u32x4*>(wb + 64 * n + plane_f);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wv_f16x8, al), __builtin_bit_cast(wv_f16x8, bh), acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wv_f16x8, ah), __builtin_bit_cast(wv_f16x8, bl), acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wv_f16x8, ah), __builtin_bit_cast(wv_f16x8, bh), acc[n], 0, 0, 0);
        }
    }
}

// split (a tile with <= 32 valid rows — the pattern decoders run Bn = 32): waves 2, 3 would only multiply zero rows, so
// instead wave w takes row tile (w & 1) and HALF of the slab's K chunks (w >> 1); the halves meet in the C tile, where
// rows 32..63 hold the second half's partial sums of rows 0..31 (the epilogues add them).
template <int NT>
__device__ __forceinline__ void wv_mma(const float* As, const float* Ws, int lda, int kp, f32x4 (&acc)[NT], bool split = false)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int nchunks = kp >> 4;
    const int rt = split ? (wave & 1) : wave;
    const int half = (nchunks + 1) >> 1;
    const int kc0 = (split && (wave >> 1)) ? half : 0;
    const int kc1 = (split && !(wave >> 1)) ? half : nchunks;
    for (int kc = kc0; kc < kc1; ++kc) {
        const float4 a4 = ld4(&As[(16 * rt + j) * lda + kc * 16 + 4 * g]);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
        float4 b4[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) b4[n] = ld4(&Ws[(((kc * 4 + g) * 16 * NT) + 16 * n + j) * 4]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float bv = (t == 0) ? b4[n].x : (t == 1) ? b4[n].y : (t == 2) ? b4[n].z : b4[n].w;
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv, acc[n], 0, 0, 0);
            }
        }
    }
}

// The recurrent state enters the fp16 pipe scaled by 2^12: |h| < 1 for every state an LSTM / GRU cell produces (o * tanh(c);
// a convex combination of tanh values), start states up to |h0| < 16 stay finite, and a state down to 3e-5 keeps both terms
// normal (smaller ones keep an absolute error < 1.5e-8).
#define WV_H3_SA 4096.f
#define WV_H3_INV_SA (1.f / 4096.f)

// G = 4: LSTM (i,f,g,o)   G = 3: GRU (r,z,n).   H3: f16x3 arithmetic (gpe_math_set(4)) — weights as fp16 plane packs with their
// amax words, the state rows split on the fly, three fp16 MFMAs per product block, fp32 accumulate (wv_mma_h3).
template <int G, int KS, bool H3 = false>
__global__ __launch_bounds__(256, (KS <= 128 ? 2 : 1)) void gpe_rnn_wave_fwd_kernel(WvFwdParams p)
{
    extern __shared__ __align__(16) float smem[];
    const int kp_max = H3 ? (((p.H < KS ? p.H : KS) + 31) & ~31) : (((p.H < KS ? p.H : KS) + 15) & ~15);
    const int lda = kp_max + 4;
    constexpr int ldc = 16 * 2 * G + 4;              // GRU keeps the input-side and recurrent-side products apart
    const int a_floats = RG_BM * (lda > ldc ? lda : ldc);
    float* As = smem;
    float* Cs = smem;
    float* Ws = smem + a_floats;
    const WvFwdCell& c = p.cell[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * RG_BM;
    const int rv = (p.Bn - row0 < RG_BM) ? (p.Bn - row0) : RG_BM;
    const int n0 = blockIdx.y * (16 * G);

    const bool split = rv <= 32;                      // wave-pair K split (wv_mma)
    float h3_inv0 = 1.f, h3_inv1 = 1.f, h3_ratio = 1.f;
    if constexpr (H3) {
        float sw0, sw1 = 1.f;
        gpe_h3_scale_of(c.s0[0], sw0, h3_inv0);
        if (c.a1) gpe_h3_scale_of(c.s1[0], sw1, h3_inv1);
        h3_ratio = h3_inv0 * sw1;                     // = inv0 / inv1
        h3_inv0 *= WV_H3_INV_SA; h3_inv1 *= WV_H3_INV_SA;
    }
    f32x4 accH[G], accX[G];
#pragma unroll
    for (int n = 0; n < G; ++n) { accH[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; accX[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    // epilogue operands (addend rows, c_{t-1} / h_{t-1}) are fetched FIRST: their L2 latency hides under the slab loop
    // instead of sitting between the last MFMA and the cell update
    const int u = tid & 15;
    const int unit = blockIdx.y * 16 + u;
    const int unitc = (unit < p.H) ? unit : p.H - 1;
    constexpr int IT = RG_BM / 16;
    float e0[IT], e1[IT], e2[IT], e3[IT], e4[IT];
    const float bhn = (G == 3) ? c.bhn[unitc] : 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int r = (tid >> 4) + 16 * it;
        const long gr = row0 + (r < rv ? r : rv - 1);
        const float* xp = c.xproj + gr * c.xp_stride;
        e0[it] = xp[unitc]; e1[it] = xp[p.H + unitc]; e2[it] = xp[2 * p.H + unitc];
        if (G == 4) { e3[it] = xp[3 * p.H + unitc]; e4[it] = c.c_prev[gr * p.H + unitc]; }
        else { e3[it] = c.a0[gr * c.a0_stride + unitc]; e4[it] = 0.f; }
    }
    // slab jobs: segment 0 = h_{l,t-1} x W_hh, segment 1 (layers > 0) = h_{l-1,t} x W_ih, each cdiv(H, KS) slabs
    {
        const int nslab = (p.H + KS - 1) / KS;
        const int njobs = c.a1 ? 2 * nslab : nslab;
        WvRegs<G, KS> R;
        auto job_fetch = [&](int i) {
            const int seg = (i >= nslab) ? 1 : 0, ks = (i - seg * nslab) * KS;
            wv_fetch<G, KS, H3>(R, seg ? c.a1 : c.a0, seg ? c.a1_stride : c.a0_stride, seg ? c.w1 : c.w0, p.H, p.Npad, row0, rv,
                                n0, ks);
        };
        job_fetch(0);
        for (int i = 0; i < njobs; ++i) {
            const int seg = (i >= nslab) ? 1 : 0, ks = (i - seg * nslab) * KS;
            const int kslab = (p.H - ks < KS) ? (p.H - ks) : KS;
            __syncthreads();                               // the previous slab's MFMAs are done with As / Ws
            wv_commit<G, KS, H3>(R, p.H, p.Npad, rv, n0, ks, As, Ws, lda);
            __syncthreads();
            job_fetch(i + 1 < njobs ? i + 1 : i);          // unconditional (the last one re-fetches itself): a load under
                                                           // a branch is waited for at the join
            if constexpr (H3) {
                if (G == 4 && i == nslab) {
                    // the LSTM's single accumulator crosses from W_hh's scale into W_ih's: an exact power-of-two rescale
#pragma unroll
                    for (int n = 0; n < G; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r) accH[n][r] *= h3_ratio;
                }
                if (G == 4 || seg == 0) wv_mma_h3<G>(As, Ws, lda, (kslab + 31) & ~31, WV_H3_SA, accH, split);
                else wv_mma_h3<G>(As, Ws, lda, (kslab + 31) & ~31, WV_H3_SA, accX, split);
            } else {
                if (G == 4 || seg == 0) wv_mma<G>(As, Ws, lda, (kslab + 15) & ~15, accH, split);
                else wv_mma<G>(As, Ws, lda, (kslab + 15) & ~15, accX, split);
            }
        }
    }
    if constexpr (H3) {
        // undo the operand scales (exact: powers of two): state 2^-12, weight from its amax word.  LSTM: accH ends in the scale of
        // its last segment (W_ih for layers > 0, see the rescale above); GRU: one accumulator per segment
#pragma unroll
        for (int n = 0; n < G; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (G == 4) accH[n][r] *= (c.a1 ? h3_inv1 : h3_inv0);
                else { accH[n][r] *= h3_inv0; accX[n][r] *= h3_inv1; }
            }
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < G; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Cs[(16 * wave + 4 * g + r) * ldc + 16 * n + j] = accH[n][r];
            if (G == 3) Cs[(16 * wave + 4 * g + r) * ldc + 16 * (G + n) + j] = accX[n][r];
        }
    __syncthreads();

    // C element (row, col) of the tile; with the wave-pair K split rows 32.. carry the second half of rows 0..31
    auto cs_at = [&](int r, int col) -> float {
        const float v = Cs[r * ldc + col];
        return split ? v + Cs[(r + 32) * ldc + col] : v;
    };
    if (unit >= p.H) return;
    if (G == 4) {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int r = (tid >> 4) + 16 * it;
            if (r < rv) {
                const long gr = row0 + r;
                const float ig = wv_sigmoid(cs_at(r, u) + e0[it]);
                const float fg = wv_sigmoid(cs_at(r, 16 + u) + e1[it]);
                const float gg = tanhf(cs_at(r, 32 + u) + e2[it]);
                const float og = wv_sigmoid(cs_at(r, 48 + u) + e3[it]);
                const float cn = fg * e4[it] + ig * gg;
                float* go = c.saved + gr * 4 * p.H;
                go[unit] = ig; go[p.H + unit] = fg; go[2 * p.H + unit] = gg; go[3 * p.H + unit] = og;
                c.c_out[gr * p.H + unit] = cn;
                c.h_out[gr * c.h_stride + unit] = og * tanhf(cn);
            }
        }
    } else {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int r = (tid >> 4) + 16 * it;
            if (r < rv) {
                const long gr = row0 + r;
                const float rg = wv_sigmoid(cs_at(r, u) + cs_at(r, 48 + u) + e0[it]);
                const float zg = wv_sigmoid(cs_at(r, 16 + u) + cs_at(r, 64 + u) + e1[it]);
                const float hn = cs_at(r, 32 + u) + bhn;
                const float ng = tanhf(cs_at(r, 80 + u) + e2[it] + rg * hn);
                float* go = c.saved + gr * 4 * p.H;
                go[unit] = rg; go[p.H + unit] = zg; go[2 * p.H + unit] = ng; go[3 * p.H + unit] = hn;
                c.h_out[gr * c.h_stride + unit] = (1.f - zg) * ng + zg * e3[it];
            }
        }
    }
}

// K slab of the wavefront kernels: 128 -> 67 KB of LDS per workgroup, two workgroups per CU (a diagonal of the shipped
// panel decoder is 576 workgroups: with 256-wide slabs (132 KB, one per CU) it runs in three rounds).  GPE_WV_KS=256
// selects the wide slab (A/B measurements).
static int wv_ks()
{
    static int ks = 0;
    if (!ks) {
        const char* e = gpe_dbg_env_str("GPE_WV_KS");
        const int v = e ? atoi(e) : 0;
        ks = (v == 256) ? 256 : 128;
    }
    return ks;
}

template <int G, int KS, bool H3 = false>
static int wv_fwd_launch_ks(const WvFwdParams& p, hipStream_t s)
{
    const int kp_max = gpe_round_up(p.H < KS ? p.H : KS, H3 ? 32 : 16);
    const int lda = kp_max + 4, ldc = 16 * 2 * G + 4;
    const size_t lds = ((size_t)RG_BM * (lda > ldc ? lda : ldc) + (size_t)kp_max * 16 * G) * sizeof(float);
    if (lds > 160 * 1024) return GPE_EINVAL;
    GPE_ENSURE_MAX_LDS((gpe_rnn_wave_fwd_kernel<G, KS, H3>));
    hipLaunchKernelGGL((gpe_rnn_wave_fwd_kernel<G, KS, H3>), dim3(gpe_cdiv(p.Bn, RG_BM), gpe_cdiv(p.H, 16), p.ncell), dim3(256),
                       lds, s, p);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

template <int G>
static int wv_fwd_launch(const WvFwdParams& p, bool h3, hipStream_t s)
{
    // a single row tile (the pattern decoders) is a pure latency chain: the wide slab halves its barrier pairs
    // (a 96-wide slab — 50 KB, three workgroups per CU, the 576-workgroup panel diagonal in one round — measured slower:
    // 686 vs 641 us per panel-decoder forward; three slabs per segment instead of two)
    const bool wide = wv_ks() == 256 || p.Bn <= RG_BM;
    if (h3) return wide ? wv_fwd_launch_ks<G, 256, true>(p, s) : wv_fwd_launch_ks<G, 128, true>(p, s);
    return wide ? wv_fwd_launch_ks<G, 256>(p, s) : wv_fwd_launch_ks<G, 128>(p, s);
}

extern "C" int gpe_rnn_seq_fwd(int gates, int L, int T, int Bn, int H, const float* xproj0, long xp0_sb, long xp0_st,
                               const void* const* whh, const void* const* wih, const void* const* bias,
                               const void* const* bhn, float* hs, long hs_sl, long hs_sb, long hs_st, float* cs, long cs_sl, long cs_st,
                               float* saved, long sv_sl, long sv_st, const void* const* whh_pl, const void* const* wih_pl,
                               const void* const* whh_amax, const void* const* wih_amax, void* ws, long ws_bytes, void* stream)
{
    if ((gates != 3 && gates != 4) || L <= 0 || T <= 0 || Bn <= 0 || H <= 0 || !xproj0 || !whh || !hs || !saved ||
        (L > 1 && (!wih || !bias)) || (gates == 4 && !cs) || (gates == 3 && !bhn) || (hs_sb & 3) || (hs_st & 3))
        return GPE_EINVAL;
    const int G = gates;
    // f16x3: the gate products on the fp16 pipe when the arithmetic mode asks for it and the caller supplies the plane packs and
    // amax words of every weight (gpe_pack_multi kinds 9 + 8); else the exact fp32 instruction
    static const int dbg_f32 = gpe_dbg_env("GPE_RNN_F32", 0);        // A/B measurements: keep the exact kernels
    bool h3 = !dbg_f32 && gpe_math_get() == 4 && whh_pl && whh_amax && (L == 1 || (wih_pl && wih_amax));
    for (int l = 0; h3 && l < L; ++l)
        if (!whh_pl[l] || !whh_amax[l] || (l > 0 && (!wih_pl[l] || !wih_amax[l]))) h3 = false;
    if (G == 4) {
        // one persistent launch for the whole stack when it fits the chip (gpe_rnn_persist.hip)
        const int rc = gpe_rnn_persist_fwd(L, T, Bn, H, xproj0, xp0_sb, xp0_st, h3 ? whh_pl : whh, h3 ? wih_pl : wih, bias, hs, hs_sl,
                                           hs_sb, hs_st, cs, cs_sl, cs_st, saved, sv_sl, sv_st, h3, whh_amax, wih_amax, ws, ws_bytes,
                                           (hipStream_t)stream);
        if (rc != 0) return rc < 0 ? rc : GPE_OK;
        if (h3) {
            const int rc2 = gpe_rnn_pm_fwd(L, T, Bn, H, xproj0, xp0_sb, xp0_st, whh_pl, wih_pl, bias, hs, hs_sl, hs_sb, hs_st, cs, cs_sl,
                                           cs_st, saved, sv_sl, sv_st, whh_amax, wih_amax, ws, ws_bytes, (hipStream_t)stream);
            if (rc2 != 0) return rc2 < 0 ? rc2 : GPE_OK;
        }
    }
    for (int d = 0; d <= T + L - 2; ++d) {
        const int l_lo = (d - (T - 1) > 0) ? d - (T - 1) : 0;
        const int l_hi = (d < L - 1) ? d : L - 1;
        for (int l0 = l_lo; l0 <= l_hi; l0 += WV_MAXCELL) {
            WvFwdParams p = {};
            p.Bn = Bn; p.H = H; p.Npad = 16 * G * gpe_cdiv(H, 16);
            int n = 0;
            // highest layer first: workgroups are dispatched in blockIdx.z order and a 3-cell panel diagonal is 576
            // workgroups on 512 slots — the second round should be the cheap layer-0 cells (one K segment instead of two)
            const int l_top = (l_hi < l0 + WV_MAXCELL - 1) ? l_hi : l0 + WV_MAXCELL - 1;
            for (int l = l_top; l >= l0; --l, ++n) {
                const int t = d - l;
                WvFwdCell& c = p.cell[n];
                c.a0 = hs + l * hs_sl + (long)t * hs_st;             // h_{l,t-1} lives at time slot t
                c.a0_stride = hs_sb;
                c.w0 = (const float*)(h3 ? whh_pl[l] : whh[l]);
                c.s0 = h3 ? (const unsigned*)whh_amax[l] : nullptr;
                if (l > 0) {
                    c.a1 = hs + (l - 1) * hs_sl + (long)(t + 1) * hs_st;   // h_{l-1,t} at slot t+1
                    c.a1_stride = hs_sb;
                    c.w1 = (const float*)(h3 ? wih_pl[l] : wih[l]);
                    c.s1 = h3 ? (const unsigned*)wih_amax[l] : nullptr;
                    c.xproj = (const float*)bias[l];
                    c.xp_stride = 0;
                } else {
                    c.xproj = xproj0 + (long)t * xp0_st;
                    c.xp_stride = xp0_sb;
                }
                if (G == 4) {
                    c.c_prev = cs + l * cs_sl + (long)t * cs_st;
                    c.c_out = cs + l * cs_sl + (long)(t + 1) * cs_st;
                } else
                    c.bhn = (const float*)bhn[l];
                c.saved = saved + l * sv_sl + (long)t * sv_st;
                c.h_out = hs + l * hs_sl + (long)(t + 1) * hs_st;
                c.h_stride = hs_sb;
            }
            p.ncell = n;
            const int rc = (G == 4) ? wv_fwd_launch<4>(p, h3, (hipStream_t)stream) : wv_fwd_launch<3>(p, h3, (hipStream_t)stream);
            if (rc != GPE_OK) return rc;
        }
    }
    return GPE_OK;
}

// =====================================================================================================================
// backward
// =====================================================================================================================
struct WvBwdCell {
    // split-K products: segment s multiplies rows a[s] [Bn][K] (pitch as[s]) with packed w[s] ([H out] x [K]); NULL = absent
    const float* a[2]; long as[2]; const float* w[2];
    float* part;                                   // [2 * nz][Bn][H] partial products of this cell
    // pointwise
    const float* dh_out; long dho_stride;          // top-layer output gradient rows or NULL
    const float* dh_extra;                         // d h_T of this layer (t == T-1) or NULL          [Bn][H]
    const float* carry_in; float* carry_out;       // dc (LSTM) / z-gated dh (GRU) of step t+1 -> of step t  [Bn][H]
    const float* saved;                            // activated gates of the cell [Bn][4H]
    const float* c; const float* c_prev;           // LSTM: c_t, c_{t-1} [Bn][H]
    const float* h_prev; long hp_stride;           // GRU
    float* dgx; float* dgh; long dg_stride;        // pre-activation gradients of the cell, rows [Bn] pitch dg_stride
    int nseg_mask;                                 // bit s: segment s present
};
// nz = K-slab GROUPS of the launch (partials per segment); a workgroup multiplies `jslabs` consecutive slabs of its group
// fuse (LSTM): the workgroup that completes an output block — the last of the block's split-K partials to arrive, by a counter per
// (cell, row tile, column block) at cnt — runs the pointwise cell backward of that block itself (no second launch per diagonal);
// Hp = row pitch of the partial images (H rounded up to 4: 16-byte write-through stores)
struct WvBwdParams { int Bn, H, K, Kpad_n, nz, jslabs, ncell, fuse, Hp; unsigned* cnt; WvBwdCell cell[WV_MAXCELL]; };

// grid (row tiles, cdiv(H, 64), ncell * 2 * nz): block z -> (cell, segment, K slab)
// (The same products on the fp16 pipe were built and measured in round 4 — dG rows scaled by a per-cell amax word that the
// pointwise kernel filled by atomicMax, transposed plane packs — and dropped: gpe_rnn_seq_bwd 1.00 -> 1.09 ms per step at cfg 2,
// the atomics + the word memset + the split of 16 short slabs per workgroup cost more than the MFMAs they replace;
// profiles/r04_e_recurrences.md.)
// pointwise LSTM cell backward of one element (shared by the stand-alone kernel and the fused epilogue)
__device__ __forceinline__ void wv_lstm_cell_bwd(const WvBwdCell& c, int H, long b, int u, long e, float dh)
{
    const float* sv = c.saved + b * 4 * H;
    float* gx = c.dgx + b * c.dg_stride;
    const float ig = sv[u], fg = sv[H + u], gg = sv[2 * H + u], og = sv[3 * H + u];
    const float tc = tanhf(c.c[e]);
    float dc = dh * og * (1.f - tc * tc);
    if (c.carry_in) dc += c.carry_in[e];
    gx[u] = dc * gg * ig * (1.f - ig);
    gx[H + u] = dc * c.c_prev[e] * fg * (1.f - fg);
    gx[2 * H + u] = dc * ig * (1.f - gg * gg);
    gx[3 * H + u] = dh * tc * og * (1.f - og);
    c.carry_out[e] = dc * fg;
}

typedef unsigned wv_st4 __attribute__((ext_vector_type(4)));

template <int KS>
__global__ __launch_bounds__(256) void gpe_rnn_wave_splitk_kernel(WvBwdParams p)
{
    extern __shared__ __align__(16) float smem[];
    __shared__ unsigned last_sh;
    constexpr int NT = 4;
    const int lda = KS + 4;
    constexpr int ldc = 16 * NT + 4;
    float* As = smem;
    float* Cs = smem;
    float* Ws = smem + RG_BM * lda;
    const int zz = blockIdx.z;
    const int ci = zz / (2 * p.nz), rem = zz - ci * 2 * p.nz;
    const int seg = rem / p.nz, z = rem - seg * p.nz;
    const WvBwdCell& c = p.cell[ci];
    const int nsegs = (c.nseg_mask & 1) + ((c.nseg_mask >> 1) & 1);
    const bool present = (c.nseg_mask >> seg) & 1;
    // fused: a cell without any product (the top layer's last step) is finished by the workgroups of (segment 0, slab group 0)
    if (!present && !(p.fuse && nsegs == 0 && seg == 0 && z == 0)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * RG_BM;
    const int rv = (p.Bn - row0 < RG_BM) ? (p.Bn - row0) : RG_BM;
    const int n0 = blockIdx.y * (16 * NT);
    const int ncols = (p.H - n0 < 16 * NT) ? (p.H - n0) : 16 * NT;
    const int Hp = p.fuse ? p.Hp : p.H;
    if (present) {
        f32x4 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bool split = rv <= 32;
        // this workgroup's slabs ks = (z * jslabs + i) * KS, i < jslabs, while ks < K: the next slab's loads stay in flight
        // under the current slab's MFMAs (same pipeline as the forward)
        {
            const int ks0 = z * p.jslabs * KS;
            int njobs = (p.K - ks0 + KS - 1) / KS;
            if (njobs > p.jslabs) njobs = p.jslabs;
            WvRegs<NT, KS> R;
            auto job_fetch = [&](int i) {
                const int ks = ks0 + i * KS;
                wv_fetch<NT, KS>(R, c.a[seg] + ks, c.as[seg], c.w[seg] + (long)(ks >> 4) * 4 * p.Kpad_n * 4, p.K - ks, p.Kpad_n, row0, rv,
                                 n0, 0);
            };
            job_fetch(0);
            for (int i = 0; i < njobs; ++i) {
                const int ks = ks0 + i * KS;
                const int kslab = (p.K - ks < KS) ? (p.K - ks) : KS;
                __syncthreads();
                wv_commit<NT, KS>(R, p.K - ks, p.Kpad_n, rv, n0, 0, As, Ws, lda);
                __syncthreads();
                job_fetch(i + 1 < njobs ? i + 1 : i);
                wv_mma<NT>(As, Ws, lda, (kslab + 15) & ~15, acc, split);
            }
        }
        __syncthreads();
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(16 * wave + 4 * g + r) * ldc + 16 * n + j] = acc[n][r];
        __syncthreads();
        const int cq = lane << 2;
        float* dst0 = c.part + ((long)(seg * p.nz + z) * p.Bn) * Hp;
        if (p.fuse) {
            // write-through (sc1) 16-byte stores: the block's last arriver reads them inside this launch.  Hp is a multiple of 4, so
            // a quad that starts inside the row may run into its pad columns
            if (cq < ((ncols + 3) & ~3)) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst0, 0, (unsigned)((long)p.Bn * Hp * 4), 0x00020000);
                for (int r = wave; r < rv; r += 4) {
                    float4 v = ld4(&Cs[r * ldc + cq]);
                    if (split) {
                        const float4 v2 = ld4(&Cs[(r + 32) * ldc + cq]);
                        v.x += v2.x; v.y += v2.y; v.z += v2.z; v.w += v2.w;
                    }
                    const wv_st4 o = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
                    __builtin_amdgcn_raw_buffer_store_b128(o, rs, (int)(((long)(row0 + r) * Hp + n0 + cq) * 4), 0, 16);
                }
            }
        } else if (cq < ncols) {
            for (int r = wave; r < rv; r += 4) {
                float4 v = ld4(&Cs[r * ldc + cq]);
                if (split) {
                    const float4 v2 = ld4(&Cs[(r + 32) * ldc + cq]);
                    v.x += v2.x; v.y += v2.y; v.z += v2.z; v.w += v2.w;
                }
                float* dst = dst0 + (long)(row0 + r) * p.H + n0 + cq;
                const float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) if (cq + t < ncols) dst[t] = o[t];
            }
        }
    }
    if (!p.fuse) return;
    // ---- fused cell backward: count this workgroup in; the block's last arriver owns the pointwise pass (MI355X_MICROARCH.md
    // "splitk-seam": sc1 partial stores, every storing wave drains, ONE relaxed agent-scope ticket, sc1 partial loads) ----
    const int nparts = nsegs * p.nz;
    if (nparts > 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0)
            last_sh = __hip_atomic_fetch_add(p.cnt + ((long)ci * gridDim.x + blockIdx.x) * gridDim.y + blockIdx.y, 1u, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (last_sh != (unsigned)(nparts - 1)) return;
    } else if (nparts == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(c.part, 0, (unsigned)((long)2 * p.nz * p.Bn * Hp * 4), 0x00020000);
    // 64 rows x 16 column quads; a thread: row tid / 16 + 16 it, quad tid % 16
    const int q4 = (tid & 15) << 2;
    if (q4 < ncols) {
#pragma unroll
        for (int it = 0; it < RG_BM / 16; ++it) {
            const int r = (tid >> 4) + 16 * it;
            if (r >= rv) break;
            const long b = row0 + r;
            float dh4[4] = {0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < 2; ++s)
                if ((c.nseg_mask >> s) & 1)
                    for (int zq = 0; zq < p.nz; ++zq) {
                        const wv_st4 v = __builtin_amdgcn_raw_buffer_load_b128(rp, (int)((((long)(s * p.nz + zq) * p.Bn + b) * Hp + n0 + q4) * 4), 0, 16);
                        dh4[0] += __uint_as_float(v[0]); dh4[1] += __uint_as_float(v[1]);
                        dh4[2] += __uint_as_float(v[2]); dh4[3] += __uint_as_float(v[3]);
                    }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int u = n0 + q4 + t;
                if (u >= p.H) break;
                const long e = b * p.H + u;
                float dh = dh4[t];
                if (c.dh_out) dh += c.dh_out[b * c.dho_stride + u];
                if (c.dh_extra) dh += c.dh_extra[e];
                wv_lstm_cell_bwd(c, p.H, b, u, e, dh);
            }
        }
    }
}

template <int G>
__global__ void gpe_rnn_wave_cell_bwd_kernel(WvBwdParams p)
{
    const WvBwdCell& c = p.cell[blockIdx.y];
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)p.Bn * p.H) return;
    const int H = p.H;
    const long b = e / H;
    const int u = (int)(e - b * H);
    float dh = c.dh_out ? c.dh_out[b * c.dho_stride + u] : 0.f;
    if (c.dh_extra) dh += c.dh_extra[e];
    for (int s = 0; s < 2; ++s)
        if ((c.nseg_mask >> s) & 1)
            for (int z = 0; z < p.nz; ++z) dh += c.part[((long)(s * p.nz + z) * p.Bn) * H + e];
    const float* sv = c.saved + b * 4 * H;
    float* gx = c.dgx + b * c.dg_stride;
    if (G == 4) {
        wv_lstm_cell_bwd(c, H, b, u, e, dh);
    } else {
        if (c.carry_in) dh += c.carry_in[e];
        const float rg = sv[u], zg = sv[H + u], ng = sv[2 * H + u], hn = sv[3 * H + u];
        const float hp = c.h_prev[b * c.hp_stride + u];
        const float dn_pre = dh * (1.f - zg) * (1.f - ng * ng);
        const float dz_pre = dh * (hp - ng) * zg * (1.f - zg);
        const float dr_pre = dn_pre * hn * rg * (1.f - rg);
        float* gh = c.dgh + b * c.dg_stride;
        gx[u] = dr_pre; gx[H + u] = dz_pre; gx[2 * H + u] = dn_pre;
        gh[u] = dr_pre; gh[H + u] = dz_pre; gh[2 * H + u] = dn_pre * rg;
        c.carry_out[e] = dh * zg;
    }
}

extern "C" long gpe_rnn_seq_fwd_ws(int gates, int L, int T, int Bn, int H)
{
    if ((gates != 3 && gates != 4) || L <= 0 || T <= 0 || Bn <= 0 || H <= 0) return GPE_EINVAL;
    const long a = gpe_rnn_persist_ws_bytes(gates, L, T, Bn, H, 0), b = gpe_rnn_pm_ws_bytes(gates, L, T, Bn, H, 0);
    return a > b ? a : b;
}

extern "C" long gpe_rnn_seq_bwd_ws(int gates, int L, int T, int Bn, int H)
{
    if ((gates != 3 && gates != 4) || L <= 0 || T <= 0 || Bn <= 0 || H <= 0) return GPE_EINVAL;
    const int nz = gpe_cdiv(gates * H, 128);              // sized for the narrow slab (the wide one needs half)
    const int ncell = L < WV_MAXCELL ? L : WV_MAXCELL;
    // partial images (row pitch H rounded up to 4: the fused kernel stores them in 16-byte pieces) + one arrival counter per
    // (diagonal, cell, row tile, column block) of the fused split-K + cell-backward launches
    const long diag = (long)ncell * 2 * nz * Bn * gpe_round_up(H, 4) + (long)(T + L) * WV_MAXCELL * gpe_cdiv(Bn, RG_BM) * gpe_cdiv(H, 64) + 64;
    const long pers = gpe_rnn_persist_ws_bytes(gates, L, T, Bn, H, 1) / 4;     // arrival counters of the persistent kernel
    return diag > pers ? diag : pers;
}

// dgx / dgh: [L][Bn][T][G*H] (for LSTM pass the same buffer twice); carry: [2][L][Bn][H] scratch;
// whh_t / wih_t: host arrays [L] of device pointers to the plain TRANSPOSED packs (gpe_pack_weight(.., transpose = 1));
// d_hN / d_cN: gradients of the final states [L][Bn][H] or NULL.  On return carry[0 or 1] holds dc_0 / the z-gated dh_0 of
// every layer at carry + (T & 1 ? .. ) — see carry_final below: the slot index written last is returned through *carry_slot.
extern "C" int gpe_rnn_seq_bwd(int gates, int L, int T, int Bn, int H, const float* dtop, long dt_sb, long dt_st,
                               const float* d_hN, const float* d_cN, const void* const* whh_t, const void* const* wih_t,
                               const float* hs, long hs_sl, long hs_sb, long hs_st, const float* cs, long cs_sl, long cs_st,
                               const float* saved, long sv_sl, long sv_st, float* dgx, float* dgh, long dg_sl, long dg_sb,
                               long dg_st, float* part, float* carry, const void* const* whh_tpl, const void* const* wih_tpl,
                               const void* const* whh_amax, const void* const* wih_amax, void* stream)
{
    if ((gates != 3 && gates != 4) || L <= 0 || T <= 0 || Bn <= 0 || H <= 0 || !whh_t || !hs || !saved || !dgx || !dgh ||
        !part || !carry || (L > 1 && !wih_t) || (gates == 4 && !cs) || (dg_sb & 3) || (dg_st & 3))
        return GPE_EINVAL;
    const int G = gates, K = G * H;
    if (G == 4) {
        // one persistent launch for the whole stack when it fits the chip (gpe_rnn_persist.hip); f16x3: the transposed plane packs
        // (gpe_pack_multi kind 10) and amax words of every weight
        bool h3 = gpe_math_get() == 4 && whh_tpl && whh_amax && (L == 1 || (wih_tpl && wih_amax));
        for (int l = 0; h3 && l < L; ++l)
            if (!whh_tpl[l] || !whh_amax[l] || (l > 0 && (!wih_tpl[l] || !wih_amax[l]))) h3 = false;
        const int KP = h3 ? gpe_round_up(K, 32) : (int)(gpe_packed_size(H, K) / gpe_round_up(H, 16));
        const int rc = gpe_rnn_persist_bwd(L, T, Bn, H, dtop, dt_sb, dt_st, d_hN, d_cN, h3 ? whh_tpl : whh_t, h3 ? wih_tpl : wih_t, KP,
                                           cs, cs_sl, cs_st, saved, sv_sl, sv_st, dgx, dg_sl, dg_sb, dg_st, carry, h3, whh_amax, wih_amax,
                                           part, gpe_rnn_persist_ws_bytes(4, L, T, Bn, H, 1), (hipStream_t)stream);
        if (rc != 0) return rc < 0 ? rc : GPE_OK;
    }
    const int KS = wv_ks();
    // big batches: four slabs per workgroup (a 3-cell panel diagonal: 2304 single-slab workgroups in 4.5 rounds -> 480 in one,
    // a quarter of the partial images; measured 804 / 717 / 621 / 689 us per backward at 1 / 2 / 4 / 8 slabs); a single row
    // tile (the pattern decoders) stays at one slab per workgroup — it is a latency chain, not a throughput problem
    static const int dbg_bj = gpe_dbg_env("GPE_WV_BJ", 0);             // measurement override
    int jslabs = (Bn > 3 * RG_BM) ? (gpe_cdiv(K, KS) < 4 ? gpe_cdiv(K, KS) : 4) : 1;
    if (dbg_bj > 0 && Bn > 3 * RG_BM) jslabs = dbg_bj < gpe_cdiv(K, KS) ? dbg_bj : gpe_cdiv(K, KS);
    const int nz = gpe_cdiv(gpe_cdiv(K, KS), jslabs);
    const long BH = (long)Bn * H;
    hipStream_t s = (hipStream_t)stream;
    // LSTM, gpe_debug_set(65536): the pointwise cell backward runs in the split-K launch itself (the last partial of an output block
    // to arrive does it): one launch per diagonal instead of two.  Built, bit-compatible (same tests), and
    // MEASURED SLOWER — cfg 2, one session, twice each: gpe_rnn_seq_bwd 0.838 ms with two launches, 1.115 / 1.117 ms fused (step
    // 9.15 -> 9.46 / 9.50 ms): the block's last arriver runs drain + ticket + write-through partial reads + a latency-bound
    // 64 x 64 pointwise pass on ONE workgroup per block at the tail of every diagonal, which costs more than the 9 us memory-parallel
    // launch it replaces.  Off by default.
    const int fuse = (G == 4) && L <= WV_MAXCELL && (gpe_debug_get() & 65536);
    const int Hp = fuse ? gpe_round_up(H, 4) : H;
    const long BHp = (long)Bn * Hp;
    const int gxr = gpe_cdiv(Bn, RG_BM), gyc = gpe_cdiv(H, 64);
    const int ncell_max = L < WV_MAXCELL ? L : WV_MAXCELL;
    unsigned* cnt_base = reinterpret_cast<unsigned*>(part + (((long)ncell_max * 2 * nz * BHp + 15) & ~15L));
    const long cnt_per_launch = (long)WV_MAXCELL * gxr * gyc;
    int launch_no = 0;
    if (fuse && hipMemsetAsync(cnt_base, 0, (size_t)(T + L) * cnt_per_launch * sizeof(unsigned), s) != hipSuccess) return GPE_ELAUNCH;
    const size_t lds = ((size_t)RG_BM * (KS + 4) + (size_t)KS * 64) * sizeof(float);
    if (KS == 256) GPE_ENSURE_MAX_LDS_N((gpe_rnn_wave_splitk_kernel<256>), 160 * 1024 - 64);     // (4 bytes of static __shared__ beside the dynamic image)
    else GPE_ENSURE_MAX_LDS_N((gpe_rnn_wave_splitk_kernel<128>), 160 * 1024 - 64);
    for (int d = T + L - 2; d >= 0; --d) {
        const int l_lo = (d - (T - 1) > 0) ? d - (T - 1) : 0;
        const int l_hi = (d < L - 1) ? d : L - 1;
        for (int l0 = l_lo; l0 <= l_hi; l0 += WV_MAXCELL) {
            WvBwdParams p = {};
            p.Bn = Bn; p.H = H; p.K = K; p.Kpad_n = gpe_round_up(H, 16); p.nz = nz; p.jslabs = jslabs;
            p.fuse = fuse; p.Hp = Hp; p.cnt = cnt_base + (long)launch_no * cnt_per_launch;
            ++launch_no;
            int n = 0, any_seg = 0;
            for (int l = l0; l <= l_hi && n < WV_MAXCELL; ++l, ++n) {
                const int t = d - l;
                WvBwdCell& c = p.cell[n];
                c.part = part + (long)n * 2 * nz * BHp;
                if (t < T - 1) {                 // recurrent path: dGh_{l,t+1} . W_hh_l
                    c.a[0] = dgh + l * dg_sl + (long)(t + 1) * dg_st; c.as[0] = dg_sb; c.w[0] = (const float*)whh_t[l];
                    c.nseg_mask |= 1;
                }
                if (l < L - 1) {                 // from the layer above: dGx_{l+1,t} . W_ih_{l+1}
                    c.a[1] = dgx + (l + 1) * dg_sl + (long)t * dg_st; c.as[1] = dg_sb; c.w[1] = (const float*)wih_t[l + 1];
                    c.nseg_mask |= 2;
                }
                any_seg |= c.nseg_mask;
                if (l == L - 1 && dtop) { c.dh_out = dtop + (long)t * dt_st; c.dho_stride = dt_sb; }
                if (t == T - 1) {
                    if (d_hN) c.dh_extra = d_hN + (long)l * BH;
                    c.carry_in = (G == 4 && d_cN) ? d_cN + (long)l * BH : nullptr;
                } else
                    c.carry_in = carry + ((long)((t + 1) & 1) * L + l) * BH;
                c.carry_out = carry + ((long)(t & 1) * L + l) * BH;
                c.saved = saved + l * sv_sl + (long)t * sv_st;
                if (G == 4) {
                    c.c = cs + l * cs_sl + (long)(t + 1) * cs_st;
                    c.c_prev = cs + l * cs_sl + (long)t * cs_st;
                } else {
                    c.h_prev = hs + l * hs_sl + (long)t * hs_st; c.hp_stride = hs_sb;
                }
                c.dgx = dgx + l * dg_sl + (long)t * dg_st;
                c.dgh = dgh + l * dg_sl + (long)t * dg_st;
                c.dg_stride = dg_sb;
            }
            p.ncell = n;
            if (any_seg || fuse) {
                const dim3 grid(gpe_cdiv(Bn, RG_BM), gpe_cdiv(H, 64), n * 2 * nz);
                if (KS == 256) hipLaunchKernelGGL(gpe_rnn_wave_splitk_kernel<256>, grid, dim3(256), lds, s, p);
                else hipLaunchKernelGGL(gpe_rnn_wave_splitk_kernel<128>, grid, dim3(256), lds, s, p);
                GPE_CHECK_LAUNCH();
            }
            if (fuse) continue;
            if (G == 4)
                hipLaunchKernelGGL((gpe_rnn_wave_cell_bwd_kernel<4>), dim3(gpe_cdiv(BH, 256), n), dim3(256), 0, s, p);
            else
                hipLaunchKernelGGL((gpe_rnn_wave_cell_bwd_kernel<3>), dim3(gpe_cdiv(BH, 256), n), dim3(256), 0, s, p);
            GPE_CHECK_LAUNCH();
        }
    }
    return GPE_OK;
}
