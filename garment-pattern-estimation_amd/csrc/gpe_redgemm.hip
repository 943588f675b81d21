// Reduce-GEMM family for gfx950:  G[Mg, Ng] = sum over rows r of U[r, :]^T V[r, :]   (+ colsum of U)
//
// This is the weight-gradient half of every Linear on the path (nn.Linear backward under
// /root/reference/nn/trainer.py:97 `loss.backward()`), with the row operands produced on the fly:
//   U: dense rows | dz3 rebuilt from (a3, g, argsel, coef)        V: dense rows | relu(P_i+Q_j) gathered
// Because BatchNorm's backward reductions are linear in the same products, G and colsum(U) are also all that
// the BN backward of the previous block needs (gpe_bn_bwd_from_G) — no extra pass over the E edges.
//
// Structure: persistent workgroups (grid.x ~ 2 per CU) walk 32-row tiles; both operand tiles are staged in
// LDS; wave w keeps the accumulators of M-tiles {w, w+4, ...} x all N-tiles in registers across ALL its row
// tiles (v_mfma_f32_16x16x4_f32, reduction dim = rows) and writes ONE partial per workgroup at the end; a
// second kernel sums the partials in a fixed order (fp64) so results are run-to-run deterministic.
#include "gpe_common.h"

#define RD_RT 32

enum { U_DENSE = 1, U_DZ3 = 0 };
enum { V_DENSE = 1, V_GATHER = 0 };

struct RdParams {
    long rows;
    int Mg, Ng, MgPad, NgPad;
    int num_tiles;
    GpeRows u;                                   // U_DENSE
    const float* a3; int lda3; const float* g; int ldg; const uint8_t* amx; const uint8_t* amn; int ldagg;
    const float* coef;                           // U_DZ3: [3][Mg]
    GpeRows v;                                   // V_DENSE
    const float* pq; int ldpq; int H; const int32_t* idx; int npts; int k;   // V_GATHER
    float* part;                                 // [gridDim.x][MgPad][NgPad]
    double* part_cs;                             // [gridDim.x][MgPad]
};

__device__ __forceinline__ float4 rd_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

__device__ __forceinline__ float4 rd_ld4_guard(const float* p, int nvalid, bool vec)
{
    if (nvalid >= 4 && vec) return rd_ld4(p);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nvalid > 0) v.x = p[0];
    if (nvalid > 1) v.y = p[1];
    if (nvalid > 2) v.z = p[2];
    if (nvalid > 3) v.w = p[3];
    return v;
}

template <int MTW, int NT, int UMODE, int VMODE>
__global__ __launch_bounds__(256, (MTW * NT * 4 > 200) ? 1 : 2) void gpe_redgemm_kernel(RdParams p)
{
    constexpr int UC = 64 * MTW;                                  // U columns handled by this block
    constexpr int LDU = (UC % 32 == 0) ? UC + 16 : UC;            // stride == 16 (mod 32): conflict-free b32 reads
    constexpr int VC = 16 * NT;
    constexpr int LDV = (VC % 32 == 0) ? VC + 16 : VC;
    __shared__ __align__(16) float Us[RD_RT * LDU];
    __shared__ __align__(16) float Vs[RD_RT * LDV];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * UC;
    const int ucols = (p.Mg - m0 < UC) ? (p.Mg - m0) : UC;

    f32x4 acc[MTW][NT];
#pragma unroll
    for (int q = 0; q < MTW; ++q)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[q][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    double cs = 0.0;

    for (int tile = blockIdx.x; tile < p.num_tiles; tile += gridDim.x) {
        const long row0 = (long)tile * RD_RT;
        const int rv = (int)((p.rows - row0 < RD_RT) ? (p.rows - row0) : RD_RT);
        __syncthreads();
        // ---- stage U columns [m0, m0+UC) --------------------------------------------------------------
        for (int e = tid; e < RD_RT * (UC / 4); e += 256) {
            const int r = e / (UC / 4), c = (e - r * (UC / 4)) << 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < rv && c < ucols) {
                const long gr = row0 + r;
                const int nvalid = ucols - c;
                if (UMODE == U_DENSE) {
                    const float* src = gpe_row_ptr(p.u, gr) + m0 + c;
                    v = rd_ld4_guard(src, nvalid, gpe_aligned16(src));
                } else {
                    const long i = gr / p.k;
                    const int slot = (int)(gr - i * p.k);
                    const float* ap = p.a3 + gr * p.lda3 + m0 + c;
                    const float* gp = p.g + i * p.ldg + m0 + c;
                    const float4 a = rd_ld4_guard(ap, nvalid, gpe_aligned16(ap));
                    const float4 gg = rd_ld4_guard(gp, nvalid, gpe_aligned16(gp));
                    const float av[4] = {a.x, a.y, a.z, a.w};
                    const float gv[4] = {gg.x, gg.y, gg.z, gg.w};
                    float o[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        o[t] = 0.f;
                        if (t < nvalid) {
                            const int cc = m0 + c + t;
                            const float s = p.coef[cc];
                            const float k1 = p.coef[p.Mg + cc];
                            const float k2 = p.coef[2 * p.Mg + cc];
                            const uint8_t sel = (s >= 0.f) ? p.amx[i * p.ldagg + cc] : p.amn[i * p.ldagg + cc];
                            const float hit = (sel == slot) ? s * gv[t] : 0.f;
                            o[t] = (av[t] > 0.f) ? (hit - k1 - av[t] * k2) : 0.f;
                        }
                    }
                    v = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
            *reinterpret_cast<float4*>(&Us[r * LDU + c]) = v;
        }
        // ---- stage V columns [0, Ng) --------------------------------------------------------------------
        for (int e = tid; e < RD_RT * (VC / 4); e += 256) {
            const int r = e / (VC / 4), c = (e - r * (VC / 4)) << 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < rv && c < p.Ng) {
                const long gr = row0 + r;
                const int nvalid = p.Ng - c;
                if (VMODE == V_DENSE) {
                    const float* src = gpe_row_ptr(p.v, gr) + c;
                    v = rd_ld4_guard(src, nvalid, gpe_aligned16(src));
                } else {
                    const long i = gr / p.k;
                    const long cloud0 = (i / p.npts) * (long)p.npts;
                    const long jj = cloud0 + p.idx[gr];
                    const float* pp = p.pq + i * p.ldpq + c;
                    const float* qq = p.pq + jj * p.ldpq + p.H + c;
                    const float4 a = rd_ld4_guard(pp, nvalid, gpe_aligned16(pp));
                    const float4 b = rd_ld4_guard(qq, nvalid, gpe_aligned16(qq));
                    v = make_float4(fmaxf(a.x + b.x, 0.f), fmaxf(a.y + b.y, 0.f), fmaxf(a.z + b.z, 0.f),
                                    fmaxf(a.w + b.w, 0.f));
                }
            }
            *reinterpret_cast<float4*>(&Vs[r * LDV + c]) = v;
        }
        __syncthreads();
        if (tid < ucols) {
            for (int r = 0; r < rv; ++r) cs += (double)Us[r * LDU + tid];
        }
#pragma unroll
        for (int r0 = 0; r0 < RD_RT; r0 += 4) {
            float a[MTW], b[NT];
#pragma unroll
            for (int q = 0; q < MTW; ++q) a[q] = Us[(r0 + g) * LDU + 16 * (wave + 4 * q) + j];
#pragma unroll
            for (int n = 0; n < NT; ++n) b[n] = Vs[(r0 + g) * LDV + 16 * n + j];
#pragma unroll
            for (int q = 0; q < MTW; ++q)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[q][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[n], acc[q][n], 0, 0, 0);
        }
    }

    // ---- one partial per workgroup ---------------------------------------------------------------------
    float* dst = p.part + (size_t)blockIdx.x * p.MgPad * p.NgPad;
#pragma unroll
    for (int q = 0; q < MTW; ++q) {
        const int mrow0 = m0 + 16 * (wave + 4 * q);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mrow0 + 4 * g + r, nn = 16 * n + j;
                if (m < p.MgPad && nn < p.NgPad) dst[(size_t)m * p.NgPad + nn] = acc[q][n][r];
            }
    }
    if (UC >= 256 ? (tid < ucols) : (tid < UC && tid < ucols))
        p.part_cs[(size_t)blockIdx.x * p.MgPad + m0 + tid] = cs;
}

__global__ void gpe_redgemm_finish(const float* part, const double* part_cs, int nblk, int Mg, int Ng, int MgPad,
                                   int NgPad, float* G, int ldg, float* colsum, int accumulate)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (long)Mg * Ng) {
        const int m = (int)(e / Ng), n = (int)(e - (long)m * Ng);
        double s = 0.0;
        for (int b = 0; b < nblk; ++b) s += (double)part[((size_t)b * MgPad + m) * NgPad + n];
        float* d = G + (size_t)m * ldg + n;
        *d = accumulate ? (*d + (float)s) : (float)s;
    }
    if (colsum && e < Mg) {
        double s = 0.0;
        for (int b = 0; b < nblk; ++b) s += part_cs[(size_t)b * MgPad + e];
        colsum[e] = accumulate ? (colsum[e] + (float)s) : (float)s;
    }
}

// ---------------------------------------------------------------------------------------------------------
static int rd_pick_nt(int Ng)
{
    const int need = gpe_cdiv(Ng, 16);
    const int opts[5] = {1, 4, 10, 13, 16};
    for (int i = 0; i < 5; ++i) if (opts[i] >= need) return opts[i];
    return -1;
}

static void rd_geometry(int Mg, int Ng, long rows, int* MTW, int* gy, int* gx, int* MgPad, int* NgPad)
{
    const int mt = gpe_cdiv(Mg, 16);
    *MTW = (mt <= 8) ? 2 : 4;
    *gy = gpe_cdiv(Mg, 64 * (*MTW));
    *MgPad = (*gy) * 64 * (*MTW);
    *NgPad = gpe_round_up(Ng, 16);
    long cap = (1L << 24) / ((long)(*MgPad) * (*NgPad));
    if (cap < 8) cap = 8;
    long g = 512 / (*gy);
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    *gx = (int)g;
}

extern "C" long gpe_redgemm_ws(int Mg, int Ng)
{
    int MTW, gy, gx, MgPad, NgPad;
    rd_geometry(Mg, Ng, 0, &MTW, &gy, &gx, &MgPad, &NgPad);
    // floats: partial G + fp64 colsum partials (2 floats each), 16-B aligned sections
    return (long)gx * MgPad * NgPad + 2L * gx * MgPad + 8;
}

template <int MTW, int NT, int UMODE, int VMODE>
static int rd_launch(const RdParams& p, dim3 grid, hipStream_t s)
{
    hipLaunchKernelGGL((gpe_redgemm_kernel<MTW, NT, UMODE, VMODE>), grid, dim3(256), 0, s, p);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

template <int UMODE, int VMODE>
static int rd_dispatch(int MTW, int NT, const RdParams& p, dim3 grid, hipStream_t s)
{
#define RD_CASE(M_, N_) if (MTW == M_ && NT == N_) return rd_launch<M_, N_, UMODE, VMODE>(p, grid, s)
    RD_CASE(2, 1); RD_CASE(2, 4); RD_CASE(2, 10); RD_CASE(2, 13); RD_CASE(2, 16);
    RD_CASE(4, 1); RD_CASE(4, 4); RD_CASE(4, 10); RD_CASE(4, 13); RD_CASE(4, 16);
#undef RD_CASE
    return GPE_EINVAL;
}

static int rd_run(RdParams& p, int umode, int vmode, float* G, int ldG, float* colsum, float* part,
                  int accumulate, hipStream_t s)
{
    int MTW, gy, gx, MgPad, NgPad;
    rd_geometry(p.Mg, p.Ng, p.rows, &MTW, &gy, &gx, &MgPad, &NgPad);
    const int NT = rd_pick_nt(p.Ng);
    if (NT < 0) return GPE_EINVAL;
    p.MgPad = MgPad; p.NgPad = NgPad;
    p.num_tiles = gpe_cdiv(p.rows, RD_RT);
    p.part = part;
    size_t off = (size_t)gx * MgPad * NgPad;
    off = (off + 1) & ~(size_t)1;                                  // 8-B align the fp64 section
    p.part_cs = reinterpret_cast<double*>(part + off);
    dim3 grid(gx, gy);
    int rc;
    if (umode == U_DENSE && vmode == V_DENSE) rc = rd_dispatch<U_DENSE, V_DENSE>(MTW, NT, p, grid, s);
    else if (umode == U_DZ3 && vmode == V_DENSE) rc = rd_dispatch<U_DZ3, V_DENSE>(MTW, NT, p, grid, s);
    else if (umode == U_DENSE && vmode == V_GATHER) rc = rd_dispatch<U_DENSE, V_GATHER>(MTW, NT, p, grid, s);
    else return GPE_EINVAL;
    if (rc != GPE_OK) return rc;
    const long total = (long)p.Mg * p.Ng;
    hipLaunchKernelGGL(gpe_redgemm_finish, dim3(gpe_cdiv(total, 256)), dim3(256), 0, s, p.part, p.part_cs, gx,
                       p.Mg, p.Ng, MgPad, NgPad, G, ldG, colsum, accumulate);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

extern "C" int gpe_redgemm(const float* u, long u_so, long u_si, int u_inner, const float* v, long v_so,
                           long v_si, int v_inner, long rows, int Mg, int Ng, float* G, int ldg, float* colsum,
                           float* part, int accumulate, void* stream)
{
    if (!u || !v || !G || !part || rows < 0 || Mg <= 0 || Ng <= 0 || Ng > 256 || ldg < Ng) return GPE_EINVAL;
    RdParams p = {};
    p.rows = rows; p.Mg = Mg; p.Ng = Ng;
    p.u = GpeRows{u, u_so, u_si, u_inner};
    p.v = GpeRows{v, v_so, v_si, v_inner};
    return rd_run(p, U_DENSE, V_DENSE, G, ldg, colsum, part, accumulate, (hipStream_t)stream);
}

extern "C" int gpe_edge_redgemm(int u_mode, const float* u, int ldu, const float* g, int ldg, const uint8_t* amx,
                                const uint8_t* amn, int ldagg, const float* coef, int v_mode, const float* v,
                                int ldv, const float* pq, int ldpq, const int32_t* idx, int B, int N, int k,
                                int Mg, int Ng, float* G, int ldG, float* colsum, float* part, void* stream)
{
    if (!u || !G || !part || B <= 0 || N <= 0 || k <= 0 || Mg <= 0 || Ng <= 0 || Ng > 256 || ldG < Ng)
        return GPE_EINVAL;
    if (u_mode == 0 && (!g || !amx || !amn || !coef)) return GPE_EINVAL;
    if (v_mode == 0 && (!pq || !idx)) return GPE_EINVAL;
    if (v_mode == 1 && !v) return GPE_EINVAL;
    RdParams p = {};
    p.rows = (long)B * N * k; p.Mg = Mg; p.Ng = Ng;
    p.u = GpeRows{u, ldu, 0, 0};
    p.a3 = u; p.lda3 = ldu; p.g = g; p.ldg = ldg; p.amx = amx; p.amn = amn; p.ldagg = ldagg; p.coef = coef;
    p.v = GpeRows{v, ldv, 0, 0};
    p.pq = pq; p.ldpq = ldpq; p.H = Ng; p.idx = idx; p.npts = N; p.k = k;
    return rd_run(p, u_mode == 0 ? U_DZ3 : U_DENSE, v_mode == 0 ? V_GATHER : V_DENSE, G, ldG, colsum, part, 0,
                  (hipStream_t)stream);
}
