// Row-tile GEMM family for gfx950:  Y[rows, N] = f(A[rows, K]) . W[N, K]^T  with fused producers / epilogues.
//
//   * nn.Linear / LSTM gate projections                  (/root/reference/nn/net_blocks.py:45,158,373-376,397)
//   * the per-edge MLP of DynamicEdgeConv, forward       (nn/net_blocks.py:43-47,124-135)
//   * its backward (input-gradient half; the weight-gradient half is gpe_redgemm.hip)
//
// Structure (one 256-thread workgroup = 4 waves, one per SIMD; two workgroups per CU):
//   - a 64-row A tile is BUILT in LDS by the producer (dense copy | gather relu(P_i+Q_j) | dz3 on the fly);
//   - the weight arrives pre-packed (gpe_pack_weight) in 16-wide K chunks that are copied to LDS with coalesced
//     16-B loads, register-prefetched one chunk ahead;
//   - wave w multiplies rows 16w..16w+15 of the tile against all NT 16-column tiles with
//     v_mfma_f32_16x16x4_f32 (exact fp32, 32-cycle issue) — every operand fetch is a ds_read_b128 giving the
//     4 k-values of 4 consecutive MFMAs, and the packed layout [k/4][n][4] makes the B reads conflict-free;
//   - the accumulator tile goes back through LDS so that stores are whole-row coalesced and the fused
//     epilogues (BN statistics in fp64, ReLU, max/min over the k messages of a point, BN/ReLU backward,
//     per-point sums) see complete rows.
// fp32 MFMA runs at the fp32 vector rate (157 TF chip peak), so these kernels are MFMA-issue bound by design;
// everything else (LDS traffic, gathers from L2/MALL, stores) hides underneath.
#include "gpe_common.h"

#include "gpe_rowgemm.h"

// ---------------------------------------------------------------------------------------------------------
// A-tile producers: fill As[64][lda] columns [0, kp) for the K slab [ks, ks+kslab)
// ---------------------------------------------------------------------------------------------------------
template <int AMODE>
__device__ __forceinline__ void rg_build_a(const RgParams& p, float* As, int lda, long row0, int rv, int ks,
                                           int kslab, int kp)
{
    const int tid = threadIdx.x;
    const int c = (tid & 63) << 2;                // column quad inside the slab (kp <= 256)
    if (c >= kp) return;
    const int nvalid = kslab - c;
    for (int r = tid >> 6; r < RG_BM; r += 4) {   // wave-uniform row
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < rv && nvalid > 0) {
            const long gr = row0 + r;
            if (AMODE == A_DENSE) {
                const float* src = gpe_row_ptr(p.a, gr) + ks + c;
                v = ld4_guard(src, nvalid, gpe_aligned16(src));
            } else {   // A_GATHER: relu(P_i + Q_j)
                const long i = (long)gpe_udiv((unsigned)gr, (unsigned)p.k, p.rcp_k);
                const long j = p.jg[gr];
                const float* pp = p.pq + i * p.ldpq + ks + c;
                const float* qq = p.pq + j * p.ldpq + p.H + ks + c;
                const float4 a = ld4_guard(pp, nvalid, gpe_aligned16(pp));
                const float4 b = ld4_guard(qq, nvalid, gpe_aligned16(qq));
                v.x = fmaxf(a.x + b.x, 0.f); v.y = fmaxf(a.y + b.y, 0.f);
                v.z = fmaxf(a.z + b.z, 0.f); v.w = fmaxf(a.w + b.w, 0.f);
            }
        }
        st4(&As[r * lda + c], v);
    }
}

// ---------------------------------------------------------------------------------------------------------
// MFMA over one K slab held in As
// ---------------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void rg_mma_slab(const RgParams& p, const float* As, int lda, int kp, int chunk0,
                                            int n0, float* Wl, f32x4 (&acc)[NT])
{
    constexpr int NQ = 64 * NT;                 // float4 per packed chunk of this N block
    constexpr int PRE = (NQ + 255) / 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int nchunks = kp >> 4;
    float4 pre[PRE];

    auto gload = [&](int kc) {
#pragma unroll
        for (int q = 0; q < PRE; ++q) {
            const int e = tid + 256 * q;
            pre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < NQ) {
                const int plane = e / (16 * NT), n = e - plane * (16 * NT);
                if (n0 + n < p.Npad)
                    pre[q] = ld4(p.wp + ((((long)(chunk0 + kc) << 2) + plane) * p.Npad + n0 + n) * 4);
            }
        }
    };
    gload(0);
    for (int kc = 0; kc < nchunks; ++kc) {
        __syncthreads();                        // Wl free again; on kc==0 also: A slab complete
#pragma unroll
        for (int q = 0; q < PRE; ++q) {
            const int e = tid + 256 * q;
            if (e < NQ) st4(&Wl[e * 4], pre[q]);
        }
        __syncthreads();
        if (kc + 1 < nchunks) gload(kc + 1);

        const float4 a4 = ld4(&As[(16 * wave + j) * lda + kc * 16 + 4 * g]);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
        float4 b4[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) b4[n] = ld4(&Wl[((g * 16 * NT) + 16 * n + j) * 4]);
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

// ---------------------------------------------------------------------------------------------------------
template <int NT, int AMODE, int EMODE>
__global__ __launch_bounds__(256, 2) void gpe_rowgemm_kernel(RgParams p)
{
    extern __shared__ __align__(16) float smem[];
    const int kp_max = ((p.K < RG_KSLAB ? p.K : RG_KSLAB) + 15) & ~15;
    const int lda = kp_max + 4;
    constexpr int ldc = 16 * NT + 4;
    const int a_floats = RG_BM * (lda > ldc ? lda : ldc);
    float* As = smem;
    float* Cs = smem;                    // aliases As after the MFMA loop
    float* Wl = smem + a_floats;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.y * (16 * NT);
    const int ncols = (p.N - n0 < 16 * NT) ? (p.N - n0) : 16 * NT;      // valid output cols of this block

    // fp64 running BN statistics of column `tid` across all tiles of this workgroup (E_EDGE_FWD)
    double st_sum = 0.0, st_sq = 0.0;

    for (int tile = blockIdx.x; tile < p.num_tiles; tile += gridDim.x) {
        const long row0 = (long)tile * p.R;
        const int rv = (int)((p.M - row0 < p.R) ? (p.M - row0) : p.R);

        f32x4 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int ks = 0; ks < p.K; ks += RG_KSLAB) {
            const int kslab = (p.K - ks < RG_KSLAB) ? (p.K - ks) : RG_KSLAB;
            const int kp = (kslab + 15) & ~15;
            __syncthreads();             // previous slab / previous tile's epilogue done with the A region
            rg_build_a<AMODE>(p, As, lda, row0, rv, ks, kslab, kp);
            rg_mma_slab<NT>(p, As, lda, kp, ks >> 4, n0, Wl, acc);   // begins with a barrier
        }
        __syncthreads();                 // all waves done reading As -> reuse as Cs
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(16 * wave + 4 * g + r) * ldc + 16 * n + j] = acc[n][r];
        __syncthreads();

        rg_epilogue<EMODE>(p, Cs, ldc, row0, rv, n0, ncols, st_sum, st_sq);
    }

    if (EMODE == E_EDGE_FWD && p.stats_part && tid < ncols) {
        double* dst = p.stats_part + (size_t)blockIdx.x * 2 * p.N;
        dst[n0 + tid] = st_sum;
        dst[p.N + n0 + tid] = st_sq;
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
static size_t rg_lds_bytes(int NT, int K)
{
    const int kp_max = gpe_round_up(K < RG_KSLAB ? K : RG_KSLAB, 16);
    const int lda = kp_max + 4, ldc = 16 * NT + 4;
    return ((size_t)RG_BM * (lda > ldc ? lda : ldc) + (size_t)64 * NT * 4) * sizeof(float);
}

template <int NT, int AMODE, int EMODE>
static int rg_launch(const RgParams& p, dim3 grid, hipStream_t stream)
{
    const size_t lds = rg_lds_bytes(NT, p.K);
    if (lds > 160 * 1024) return GPE_EINVAL;
    GPE_ENSURE_MAX_LDS((gpe_rowgemm_kernel<NT, AMODE, EMODE>));
    hipLaunchKernelGGL((gpe_rowgemm_kernel<NT, AMODE, EMODE>), grid, dim3(256), lds, stream, p);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

template <int AMODE, int EMODE>
static int rg_dispatch_nt(int NT, const RgParams& p, dim3 grid, hipStream_t stream)
{
    switch (NT) {
        case 4: return rg_launch<4, AMODE, EMODE>(p, grid, stream);
        case 7: return rg_launch<7, AMODE, EMODE>(p, grid, stream);
        case 10: return rg_launch<10, AMODE, EMODE>(p, grid, stream);
        case 13: return rg_launch<13, AMODE, EMODE>(p, grid, stream);
        case 16: return rg_launch<16, AMODE, EMODE>(p, grid, stream);
    }
    return GPE_EINVAL;
}

static int rg_pick_nt_single(int N)   // smallest instantiated NT whose block covers all N columns
{
    const int need = gpe_cdiv(N, 16);
    const int opts[5] = {4, 7, 10, 13, 16};
    for (int i = 0; i < 5; ++i) if (opts[i] >= need) return opts[i];
    return -1;
}

// single-stage kernel for latency-bound shapes (gpe_smallgemm.hip)
int gpe_smallgemm_linear(const RgParams& r, hipStream_t s);
// register-stationary fast path for the shipped edge-MLP sizes (gpe_edgegemm.hip): 1 launched, 0 not on its menu
int gpe_edgegemm_try(const RgParams& p, int amode, int emode, int stats_nblk, hipStream_t s);

void gpe_edgegemm_set_math(int m);
void gpe_redgemm_set_math(int m);
size_t gpe_edge_pseudo_bytes(long npts, int k, int Cmax);   // gpe_edgegemm_sr.hip

int gpe_gemm_x6_linear(const GpeRows& a, const float* wp, int Npad, int Kq, const float* bias, const GpeRows& addend, float* y,
                       long y_so, long y_si, int y_inner, long M, int N, int K, int act, hipStream_t s);      // gpe_gemm_x6.hip
extern "C" long gpe_packed_size(int N, int K);
static int g_gpe_dbg = 0;
extern "C" int gpe_debug_set(int flags) { g_gpe_dbg = flags; return 0; }
extern "C" int gpe_debug_get(void) { return g_gpe_dbg; }
static int g_gpe_math = 0;
extern "C" int gpe_edge_lazy_dz3_ok(int B, int N, int k, int F, int Cprev);
extern "C" int gpe_math_get(void) { return g_gpe_math; }
extern "C" int gpe_math_set(int mode)
{
    if (mode < 0 || mode > 4) return GPE_EINVAL;
    const int prev = g_gpe_math;
    g_gpe_math = mode;
    // row GEMMs (forward, input-gradient half): 0 exact fp32, 1 two-term split-bf16 (modes 1, 2), 2 three-term split-bf16
    // (mode 3), 3 two-term split-fp16 on tensor-normalised operands (mode 4)
    gpe_edgegemm_set_math(mode == 4 ? 3 : mode == 3 ? 2 : (mode != 0 ? 1 : 0));
    // the weight-gradient reduce-GEMM: split-bf16 only in mode 1; mode 4: split-fp16 where the operand scales are known
    gpe_redgemm_set_math(mode == 1 ? 1 : mode == 4 ? 2 : 0);
    return prev;
}

#define GPE_STATS_BLOCKS 512
extern "C" int gpe_stats_blocks(void) { return GPE_STATS_BLOCKS; }

// workspace of the edge entry points: fixed part + the pseudo-point rows of a k > 16 launch (ldmax = the largest row pitch, in
// floats, of a per-point output the caller will pass: ldagg of gpe_edge_mlp_fwd, lddp of gpe_edge_mlp_bwd)
extern "C" long gpe_edge_ws_bytes(int B, int N, int k, int ldmax)
{
    if (B <= 0 || N <= 0 || k <= 0 || ldmax <= 0) return GPE_EINVAL;
    return (long)(GPE_WS_DUMMY_BYTES + GPE_WS_H3_BYTES) + (long)gpe_edge_pseudo_bytes((long)B * N, k, ldmax);
}
static long g_h3_min_rows = GPE_H3_MIN_ROWS_DEFAULT;
long gpe_h3_min_rows() { return g_h3_min_rows; }
extern "C" long gpe_f16x3_min_rows(void) { return g_h3_min_rows; }
extern "C" long gpe_f16x3_min_rows_set(long rows)
{
    if (rows < 0) return GPE_EINVAL;
    const long prev = g_h3_min_rows;
    g_h3_min_rows = rows;
    return prev;
}
extern "C" int gpe_edge_pq_amax(const float* pq, int ldpq, int H, long rows, uint32_t* amax, void* ws, long ws_bytes,
                                void* stream)
{
    const GpeEdgeWs w = gpe_edge_ws(ws, ws_bytes);
    if (!pq || !amax || !w.h3 || rows <= 0 || H <= 0 || (H & 3) || (ldpq & 3) || ldpq < 2 * H) return GPE_EINVAL;
    return gpe_h3_pq_passes(amax, reinterpret_cast<float*>(w.h3 + 2), pq, rows, H, ldpq, (hipStream_t)stream);
}
extern "C" int gpe_absmax(const float* x, int ldx, long rows, int cols, uint32_t* amax, void* stream)
{
    if (!x || !amax || rows < 0 || cols <= 0 || ldx < cols || (ldx & 3) || (((uintptr_t)x) & 15)) return GPE_EINVAL;
    return gpe_h3_absmax(amax, x, rows, cols, ldx, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------
// K <= 8 Linear (the layer-1 [P|Q] projection of the raw N x 3 positions: 65 536 x 400 outputs from 3 inputs).  Nothing
// here is a GEMM: 105 MB of output against 0.8 MB of input, so the kernel is a streaming store — lane = one output
// column quad with its K x 4 weights in registers, wave = one row at a time (row-uniform input loads).
//   y = act(bias + sum_k a_k * W[k][:] (+ addend)),  k ascending fused multiply-adds.
// ---------------------------------------------------------------------------------------------------------
#define RG_SMALLK 8
__global__ __launch_bounds__(256) void gpe_linear_smallk_kernel(RgParams p)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = (blockIdx.y * 64 + lane) << 2;
    if (c >= p.N) return;                                  // N % 4 == 0 (dispatcher): whole quads only
    float w[RG_SMALLK][4];
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < RG_SMALLK; ++k) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // packed weights: float4 ((k/16)*4 + (k/4)%4, n) holds W[4*(k/4) .. +3][n]
            w[k][t] = (k < p.K) ? p.wp[((long)(((k >> 4) << 2) + ((k >> 2) & 3)) * p.Npad + c + t) * 4 + (k & 3)] : 0.f;
        }
    }
    if (p.bias) {
#pragma unroll
        for (int t = 0; t < 4; ++t) b4[t] = p.bias[c + t];
    }
    const long nw = (long)gridDim.x * 4;
    for (long r = (long)blockIdx.x * 4 + wave; r < p.M; r += nw) {
        const float* ar = gpe_row_ptr(p.a, r);
        float av[RG_SMALLK];
#pragma unroll
        for (int k = 0; k < RG_SMALLK; ++k) av[k] = ar[k < p.K ? k : 0];
        float o[4] = {b4[0], b4[1], b4[2], b4[3]};
#pragma unroll
        for (int k = 0; k < RG_SMALLK; ++k) {
            if (k < p.K) {
#pragma unroll
                for (int t = 0; t < 4; ++t) o[t] = __builtin_fmaf(av[k], w[k][t], o[t]);
            }
        }
        if (p.addend.base) {
            const float4 ad = ld4(gpe_row_ptr(p.addend, r) + c);
            o[0] += ad.x; o[1] += ad.y; o[2] += ad.z; o[3] += ad.w;
        }
        if (p.act == 1) {
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = fmaxf(o[t], 0.f);
        }
        float* dst;
        if (p.y_inner <= 0) dst = p.y + r * p.y_so + c;
        else { const long oo = r / p.y_inner; dst = p.y + oo * p.y_so + (r - oo * p.y_inner) * p.y_si + c; }
        st4(dst, make_float4(o[0], o[1], o[2], o[3]));
    }
}

static bool rg_rows_aligned16(const float* base, long so, long si, int inner)
{
    return !(((uintptr_t)base) & 15) && !(so & 3) && (inner <= 0 || !(si & 3));
}

extern "C" int gpe_linear(const float* a, long a_so, long a_si, int a_inner, const float* wp, const float* bias,
                          const float* addend, long ad_so, long ad_si, int ad_inner, float* y, long y_so,
                          long y_si, int y_inner, int M, int N, int K, int act, void* stream)
{
    if (!a || !wp || !y || M < 0 || N <= 0 || K <= 0 || (act != 0 && act != 1)) return GPE_EINVAL;
    if (M == 0) return GPE_OK;
    RgParams p = {};
    p.M = M; p.N = N; p.K = K; p.R = RG_BM; p.num_tiles = gpe_cdiv(M, RG_BM);
    p.a = GpeRows{a, a_so, a_si, a_inner};
    p.wp = wp; p.Npad = gpe_round_up(N, 16); p.bias = bias;
    p.addend = GpeRows{addend, ad_so, ad_si, ad_inner};
    p.y = y; p.y_so = y_so; p.y_si = y_si; p.y_inner = y_inner; p.act = act;
    // f16x3 mode: the big dense products (the layer-2 [P|Q] projection and its input gradient, 65536 rows x 400 x 150) on the bf16
    // pipe, three-term splits, six MFMAs per product (gpe_gemm_x6.hip); gpe_debug_set(16384) keeps the exact kernels (A/B runs)
    if (g_gpe_math == 4 && !(g_gpe_dbg & 16384)) {
        const int rc = gpe_gemm_x6_linear(p.a, wp, p.Npad, (int)(gpe_packed_size(N, K) / p.Npad / 4), bias, p.addend, y, y_so, y_si, y_inner, M, N, K,
                                          act, (hipStream_t)stream);
        if (rc != 0) return rc < 0 ? rc : GPE_OK;
    }
    if (K <= RG_SMALLK && !(N & 3) && M >= 4096 && rg_rows_aligned16(y, y_so, y_si, y_inner) &&
        (!addend || rg_rows_aligned16(addend, ad_so, ad_si, ad_inner))) {
        long gx = gpe_cdiv(M, 4 * 8);                      // >= 8 rows per wave
        if (gx > 4096) gx = 4096;
        hipLaunchKernelGGL(gpe_linear_smallk_kernel, dim3((unsigned)gx, gpe_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, p);
        GPE_CHECK_LAUNCH();
        return GPE_OK;
    }
    // latency-bound regime (the streaming kernel could not even put one workgroup on half the CUs): single-stage kernel
    if ((long)p.num_tiles * gpe_cdiv(N, 208) < 128) return gpe_smallgemm_linear(p, (hipStream_t)stream);
    // widest column block that still gives the chip >= 256 workgroups, else the narrowest
    int NT = 4;
    const int opts[5] = {16, 13, 10, 7, 4};
    for (int i = 0; i < 5; ++i) {
        const int nblk = gpe_cdiv(N, 16 * opts[i]);
        if ((long)nblk * p.num_tiles >= 256 || opts[i] == 4) { NT = opts[i]; break; }
    }
    // never use a block much wider than N
    const int single = rg_pick_nt_single(N);
    if (single > 0 && single < NT) NT = single;
    // several column blocks: the width with the fewest padded columns (N = 400: two 208-wide blocks, 67 KB of LDS and
    // two workgroups per CU, instead of 256 + 144 at 82 KB and one)
    if (N > 256) {
        int best = NT, best_pad = gpe_cdiv(N, 16 * NT) * 16 * NT;
        const int wide[3] = {16, 13, 10};
        for (int i = 0; i < 3; ++i) {
            const int pad = gpe_cdiv(N, 16 * wide[i]) * 16 * wide[i];
            if (pad < best_pad && (long)gpe_cdiv(N, 16 * wide[i]) * p.num_tiles >= 256) { best = wide[i]; best_pad = pad; }
        }
        NT = best;
    }
    dim3 grid(p.num_tiles, gpe_cdiv(N, 16 * NT));
    return rg_dispatch_nt<A_DENSE, E_LINEAR>(NT, p, grid, (hipStream_t)stream);
}

// ---- lazy dz3 (round 4): the 1.3 GB in-place pass of gpe_edge_dz3 folded into its two consumers -----------------------------------
// 1 when gpe_edge_mlp_bwd (act_mode 0) and gpe_edge_redgemm (v_mode 1) can form dz3 from the stored activation themselves for a
// block F -> Cprev of B clouds x N points x k: f16x3 arithmetic, k = 16, both widths on the two-plane kernels' menu, above the
// size gate.  Host only.
extern "C" int gpe_edge_lazy_dz3_ok(int B, int N, int k, int F, int Cprev)
{
    static const int dbg_off = gpe_dbg_env("GPE_LAZY_DZ3", 1) == 0;   // A/B measurements
    if (dbg_off || (g_gpe_dbg & 512) || g_gpe_math != 4 || k != 16 || B <= 0 || N <= 0) return 0;    // gpe_debug_set(512): eager dz3
    if (F <= 96 || F > 208 || Cprev <= 96 || Cprev > 208 || (Cprev & 3)) return 0;
    if (gpe_cdiv(Cprev, 16) != 13 || (gpe_cdiv(F, 16) != 10 && gpe_cdiv(F, 16) != 13)) return 0;   // the fp16 reduce-GEMM's instantiated shapes
    const long rows = (long)B * N * k;
    return rows >= gpe_h3_min_rows() && rows < (1L << 31) && rows / 32 >= 4L * 2 * gpe_num_cus();
}

// bound of |dz3| for its f16x3 scale: max_i,c |s_c g_ic| (measured by gpe_edge_bwd_point_sums while it reads g: amax_sg) +
// max_c (|c1_c| + (amax(a3) + |mean_c|) |k2_c|), rounded up.  One workgroup over the F coefficient columns.
__global__ __launch_bounds__(256) void gpe_dz3_bound_kernel(const unsigned* __restrict__ amax_sg, const float* __restrict__ coef, int F,
                                                            const unsigned* __restrict__ amax_a3, unsigned* __restrict__ out)
{
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float a3 = __uint_as_float(amax_a3[0]) * 1.001f;           // (the stored activation may be fp16-rounded: + 2^-11)
    float tail = 0.f;
    for (int c = threadIdx.x; c < F; c += 256)
        tail = fmaxf(tail, fabsf(coef[F + c]) + (a3 + fabsf(coef[3 * F + c])) * fabsf(coef[2 * F + c]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tail = fmaxf(tail, __shfl_xor(tail, o));
    if (lane == 0) red[wave] = tail;
    __syncthreads();
    if (threadIdx.x == 0) {
        tail = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const float b = (__uint_as_float(amax_sg[0]) + tail) * 1.000001f;
        out[0] = __float_as_uint(b) & 0x7fffffffu;
    }
}
extern "C" int gpe_edge_dz3_bound(const uint32_t* amax_sg, const float* coef, int F, const uint32_t* amax_a3, uint32_t* amax_out,
                                  void* stream)
{
    if (!amax_sg || !coef || !amax_a3 || !amax_out || F <= 0) return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_dz3_bound_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, amax_sg, coef, F, amax_a3, amax_out);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

extern "C" int gpe_edge_mlp_fwd(int a_mode, const float* pq, int ldpq, const int32_t* jg, const float* a_in,
                                int lda, int B, int N, int k, int Cin, int Cout, const float* wp,
                                const float* bias, float* out, int ldo, double* stats_part, int agg, float* mx,
                                float* mn, uint8_t* amx, uint8_t* amn, int ldagg, const uint32_t* amax_a,
                                uint32_t* amax_out, void* ws, long ws_bytes, int out_half, void* stream)
{
    if (!wp || !out || B <= 0 || N <= 0 || k <= 0 || k > 64 || Cin <= 0 || Cout <= 0 || (ldo & 3) || ldo < Cout)
        return GPE_EINVAL;
    // out_half: `out` is a _Float16 [E][ldo] tensor (the aggregated block's activation when its backward will form dz3 lazily):
    // only the dense f16x3 forward of a max-aggregated block above the size gate stores it (gpe_edge_lazy_dz3_ok)
    const int w_ready = (out_half >> 1) & 1;           // bit 1: the packed weight's f16x3 scale is in the workspace (gpe_pack_fold)
    out_half &= 1;
    if (out_half && (a_mode != 1 || !agg || !gpe_edge_lazy_dz3_ok(B, N, k, Cout, Cin))) return GPE_EINVAL;
    // the gather producer builds one <= 256-wide K slab; dense rows stream any K in 256-wide slabs
    if (a_mode == 0 && (!pq || !jg || (ldpq & 3) || (Cin & 3) || Cin > RG_KSLAB)) return GPE_EINVAL;
    if (a_mode == 1 && (!a_in || lda < Cin)) return GPE_EINVAL;
    if (agg && (!mx || !mn || !amx || !amn || ldagg < Cout)) return GPE_EINVAL;
    {
        // the activation rows must not alias an input or a per-point output (gpe_common.h gpe_overlap)
        const size_t E_ = (size_t)B * N * k, P_ = (size_t)B * N, ob = E_ * ldo * (out_half ? 2 : 4);
        if (gpe_overlap(out, ob, a_mode == 1 ? a_in : nullptr, E_ * lda * 4) || gpe_overlap(out, ob, a_mode == 0 ? pq : nullptr, P_ * ldpq * 4) ||
            gpe_overlap(out, ob, agg ? mx : nullptr, P_ * ldagg * 4) || gpe_overlap(out, ob, agg ? mn : nullptr, P_ * ldagg * 4))
            return GPE_EINVAL;
    }
    // one column block up to 256 outputs; wider layers (dense MLPs of the attention / MLP-decoder variants) run as
    // several 208-column blocks (grid.y), each with its own slice of the statistics / epilogue
    const int NT = (Cout > 256) ? 13 : rg_pick_nt_single(Cout);
    if (NT < 0) return GPE_EINVAL;
    const int ny = gpe_cdiv(Cout, 16 * NT);
    if (ny > 1 && a_mode == 0) return GPE_EINVAL;
    RgParams p = {};
    p.M = (long)B * N * k; p.N = Cout; p.K = Cin;
    p.R = (RG_BM / k) * k; p.num_tiles = gpe_cdiv(p.M, p.R);
    p.a = GpeRows{a_in, lda, 0, 0};
    p.pq = pq; p.ldpq = ldpq; p.H = Cin; p.jg = jg; p.k = k; p.rcp_k = 1.0 / k;
    p.wp = wp; p.Npad = gpe_round_up(Cout, 16); p.bias = bias;
    p.out = out; p.ldo = ldo; p.stats_part = stats_part;
    p.agg = agg; p.mx = mx; p.mn = mn; p.oamx = amx; p.oamn = amn; p.oldagg = ldagg;
    p.dbg = g_gpe_dbg; p.pin_clouds = B;
    p.user_amax_a = amax_a; p.user_amax_out = amax_out; p.ws = gpe_edge_ws(ws, ws_bytes);
    p.out_half = out_half;
    p.w_ready = w_ready;
    p.rev = gpe_walk_rev(a_mode == 1);                 // F2 walks up, F3 down (gpe_common.h)
    int tracked = 0;
    p.tracked = &tracked;
    int rc = gpe_edgegemm_try(p, a_mode == 0 ? A_GATHER : A_DENSE, E_EDGE_FWD,
                              stats_part ? GPE_STATS_BLOCKS : 0, (hipStream_t)stream);
    if (rc == 0) {
        dim3 grid(GPE_STATS_BLOCKS, ny);
        rc = (a_mode == 0) ? rg_dispatch_nt<A_GATHER, E_EDGE_FWD>(NT, p, grid, (hipStream_t)stream)
                           : rg_dispatch_nt<A_DENSE, E_EDGE_FWD>(NT, p, grid, (hipStream_t)stream);
    } else
        rc = rc == 1 ? GPE_OK : rc;
    // the caller asked for the largest magnitude written and the kernel that ran did not track it: one streaming pass
    if (rc == GPE_OK && out_half && !tracked) return GPE_EINVAL;       // (cannot happen: the f16x3 kernels track what they store)
    if (rc == GPE_OK && amax_out && !tracked) rc = gpe_h3_absmax(amax_out, out, p.M, Cout, ldo, (hipStream_t)stream);
    return rc;
}

extern "C" int gpe_edge_mlp_bwd(const float* a, int lda, int act_mode, const float* pq, int ldpq,
                                const int32_t* jg, int B, int N, int k, int Cin, int Cout, const float* wp,
                                const float* coef_out, float* dz_out, int ldo, float* dP, int lddp,
                                const uint32_t* amax_a, uint32_t* amax_out, void* ws, long ws_bytes, const float* lz_g, int lz_ldg,
                                const uint8_t* lz_amx, const uint8_t* lz_amn, int lz_ldagg, const float* lz_coef, void* stream)
{
    if (!a || !wp || !coef_out || !dz_out || B <= 0 || N <= 0 || k <= 0 || k > 64 || Cin <= 0 || Cout <= 0 ||
        (ldo & 3) || ldo < Cout || lda < Cin)
        return GPE_EINVAL;
    if (act_mode == 1 && (!pq || !jg || !dP || (ldpq & 3) || (Cout & 3) || Cout > 256)) return GPE_EINVAL;
    {
        // dz_out is read (the stored activation) and overwritten in place by design.  It may also BE `a` (the gathered backward run
        // in place over dz_1's rows, same pitch: a tile's rows are staged two tiles before they are finished, by the same workgroup,
        // with the tile barriers in between); any other overlap with `a`, and any with the [P|Q] table or dP, is refused
        const size_t E_ = (size_t)B * N * k, P_ = (size_t)B * N, ob = E_ * ldo * 4;
        const bool in_place = act_mode == 1 && !lz_g && (const void*)a == (const void*)dz_out && lda == ldo;
        if ((!in_place && gpe_overlap(dz_out, ob, a, E_ * lda * (lz_g ? 2 : 4))) || gpe_overlap(dz_out, ob, act_mode == 1 ? pq : nullptr, P_ * ldpq * 4) ||
            gpe_overlap(dz_out, ob, act_mode == 1 ? dP : nullptr, P_ * lddp * 4))
            return GPE_EINVAL;
    }
    const int NT = (Cout > 256) ? 13 : rg_pick_nt_single(Cout);       // wide dense layers: several column blocks
    if (NT < 0) return GPE_EINVAL;
    const int ny = gpe_cdiv(Cout, 16 * NT);
    RgParams p = {};
    p.M = (long)B * N * k; p.N = Cout; p.K = Cin;
    p.R = (RG_BM / k) * k; p.num_tiles = gpe_cdiv(p.M, p.R);
    p.a = GpeRows{a, lda, 0, 0};
    p.pq = pq; p.ldpq = ldpq; p.H = Cout; p.jg = jg; p.k = k; p.rcp_k = 1.0 / k;
    p.wp = wp; p.Npad = gpe_round_up(Cout, 16);
    p.out = dz_out; p.ldo = ldo; p.coef_out = coef_out; p.dP = dP; p.lddp = lddp;
    hipStream_t s = (hipStream_t)stream;
    p.dbg = g_gpe_dbg; p.pin_clouds = B;
    p.user_amax_a = amax_a; p.user_amax_out = amax_out; p.ws = gpe_edge_ws(ws, ws_bytes);
    p.rev = gpe_walk_rev(1);                           // B3 and B2 walk down: behind a reduce-GEMM that walked up
    if (lz_g) {
        // lazy dz3: `a` is the stored activation of the aggregated block; only the f16x3 k = 16 in-place kernel forms dz3 from it
        // (the caller asked gpe_edge_lazy_dz3_ok first and passes the bound of |dz3| as amax_a)
        if (!lz_amx || !lz_amn || !lz_coef || lz_ldg < Cin || (lz_ldagg & 3) || lz_ldagg < ((Cin + 3) & ~3) || act_mode != 0 || !amax_a ||
            !gpe_edge_lazy_dz3_ok(B, N, k, Cin, Cout) || !p.ws.h3)
            return GPE_EINVAL;
        p.lz_g = lz_g; p.lz_ldg = lz_ldg; p.lz_amx = lz_amx; p.lz_amn = lz_amn; p.lz_ldagg = lz_ldagg; p.lz_coef = lz_coef;
    }
    int tracked = 0;
    p.tracked = &tracked;
    int rc = gpe_edgegemm_try(p, A_DENSE, act_mode == 1 ? E_BWD_GATHER : E_BWD_INPLACE, 0, s);
    if (lz_g && rc != 1) return rc < 0 ? rc : GPE_EINVAL;          // no eager kernel may read a3 as if it were dz3
    if (rc == 0) {
        dim3 grid(p.num_tiles < 2048 ? p.num_tiles : 2048, ny);
        rc = (act_mode == 1) ? rg_dispatch_nt<A_DENSE, E_BWD_GATHER>(NT, p, grid, s)
                             : rg_dispatch_nt<A_DENSE, E_BWD_INPLACE>(NT, p, grid, s);
    } else
        rc = rc == 1 ? GPE_OK : rc;
    if (rc == GPE_OK && amax_out && !tracked) rc = gpe_h3_absmax(amax_out, dz_out, p.M, Cout, ldo, s);
    return rc;
}
