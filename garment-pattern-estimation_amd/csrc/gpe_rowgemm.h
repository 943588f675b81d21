// Shared pieces of the row-tile GEMM kernels (gpe_rowgemm.hip: LDS-streamed weights, any shape;
// gpe_edgegemm.hip: register-stationary weights for the shipped edge-MLP sizes).
#pragma once
#include "gpe_common.h"

#define RG_BM 64
#define RG_KSLAB 256

enum { A_DENSE = 0, A_GATHER = 1 };
enum { E_LINEAR = 0, E_EDGE_FWD = 1, E_BWD_INPLACE = 2, E_BWD_GATHER = 3 };

struct RgParams {
    // problem
    long M;                 // logical rows (E for edge kernels)
    int N, K;               // output cols / reduction dim
    int R;                  // rows per tile (<= 64); edge kernels: whole points, R = (64/k)*k
    int num_tiles;
    // A producers
    GpeRows a;              // A_DENSE
    const float* pq; int ldpq; int H;           // A_GATHER / E_BWD_GATHER : P = pq[:, 0:H], Q = pq[:, H:2H]
    const int32_t* jg; int k; double rcp_k;     // GLOBAL neighbour row per edge, neighbours per point, 1/k
    // weight
    const float* wp; int Npad;
    const float* bias;
    // epilogue
    GpeRows addend;         // E_LINEAR (base may be NULL)
    float* y; long y_so, y_si; int y_inner; int act;
    float* out; int ldo;    // edge kernels: activation / dz rows
    double* stats_part;     // E_EDGE_FWD: [gridDim.x][2][N]
    int agg; float* mx; float* mn; uint8_t* oamx; uint8_t* oamn; int oldagg;
    const float* coef_out;  // E_BWD_*: [4][N] = {s, c1, k2, mean}
    float* dP; int lddp;    // E_BWD_GATHER
    int dbg;                // ablation switches for profiling (0 in production): see gpe_debug_set
    // cloud -> XCD pinning of the persistent edge kernels (gpe_common.h): tiles of cloud c are processed by the workgroups
    // of XCD c % 8.  tpc = tiles per cloud (N*k / R, exact); 0 = off.
    int pin_tpc;
    int rev;                                     // walk the tile sequence from the far end (gpe_common.h GpeTileSeq)
    int pin_clouds;         // B when the caller's rows are B equal clouds (edge kernels), else 0
    // k > 16 on the single-role kernels: a point's k rows are handled as f pseudo-points of k/f rows (gpe_edgegemm_sr.hip);
    // the P row of pseudo-point x is then x / f = umulhi(x, pmagic).  0 = pseudo-points are points.
    unsigned pmagic;
    // 64 KB+ scratch image that absorbs the epilogue stores of a wave with nothing valid to finish (straight-line instances of
    // gpe_edgegemm_sr_kernel); NULL = use the branchy instances
    float* dummy;
    // f16x3 (SplitF16x2, gpe_edgegemm_split_kernel.h): bit patterns of the largest magnitude of the A operand and of the packed
    // weight, measured on the device; amax_out (may be NULL) receives (atomicMax) the largest magnitude the kernel writes to `out`
    const unsigned* h3_amax_a;
    const unsigned* h3_amax_w;
    unsigned* amax_out;
    // LAZY dz3 (f16x3, k = 16, in-place backward of the block under the aggregation): the A operand is the stored activation a3
    // and the kernel forms dz3 = (a3 > 0) ? [slot == argsel] * s * g - c1 - (a3 - mean) * k2 : 0 while staging it — gpe_edge_dz3's
    // arithmetic without its 1.3 GB round trip.  lz_g [B*N][lz_ldg] the layer-output gradient, lz_amx / lz_amn [B*N][lz_ldagg] the
    // slots saved by the forward, lz_coef [4][K] = {s, c1, k2, mean} of the BatchNorm behind the aggregation.  NULL = off.
    const float* lz_g; int lz_ldg;
    const uint8_t* lz_amx; const uint8_t* lz_amn; int lz_ldagg;
    const float* lz_coef;
    // fp16 storage of the aggregated block's activation (row g): out_half = the forward stores `out` as _Float16 rows
    // (ldo in halves); the lazy consumers always read their A operand that way (a.stride_outer in halves)
    int out_half;
    int w_ready;                          // f16x3: ws.h3[1] already holds the packed weight's amax and the clears are done (gpe_pack_fold)
    // host side only: what the CALLER passed (include/gpe_hip.h: amax_a / amax_out / ws of the edge entry points)
    const unsigned* user_amax_a;    // amax word of the A operand (gather: of relu(P_i + Q_j)); NULL = measure in-call
    unsigned* user_amax_out;        // receives the largest magnitude written to `out`; NULL = not wanted
    GpeEdgeWs ws;
    int* tracked;                   // set to 1 by a kernel path that filled user_amax_out itself
};

// k > 16 on the single-role kernels (gpe_edgegemm_sr.hip): what gpe_edge_pseudo_setup redirected to scratch, for the fold
struct GpeFold {
    int f, kq;              // pseudo-points per point (1 = nothing to fold), rows per pseudo-point
    long npts;
    float *mx, *mn, *dp;    // the caller's per-point outputs (NULL = not redirected)
    uint8_t *amx, *amn;
};
int gpe_edge_pseudo_setup(RgParams& p, bool per_point, int emode, GpeFold& fd);
int gpe_edge_pseudo_fold(const RgParams& p, const GpeFold& fd, hipStream_t s);

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// streaming 16-B store that does NOT keep its line in the XCD's L2 (sc1: write-through + drop, MI355X_MICROARCH.md "stores of
// each flavour"): for activation rows that are written once and read by a later kernel, so that they do not evict the
// gathered per-cloud table the same kernel keeps re-reading.  (vmcnt is in-order on gfx9: a store the compiler does not
// count only makes its waits more conservative.)
__device__ __forceinline__ void st4_stream(float* p, float4 v)
{
    const f32x4 vv = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(vv) : "memory");
}

// guarded load of 4 consecutive floats p[0..3] of which `nvalid` exist; vec => p is 16-B aligned
__device__ __forceinline__ float4 ld4_guard(const float* p, int nvalid, bool vec)
{
    if (nvalid >= 4 && vec) return ld4(p);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nvalid > 0) v.x = p[0];
    if (nvalid > 1) v.y = p[1];
    if (nvalid > 2) v.z = p[2];
    if (nvalid > 3) v.w = p[3];
    return v;
}

// ---------------------------------------------------------------------------------------------------------
// Fused epilogues working on the accumulator tile staged in LDS (Cs[64][ldc]).
// Thread mapping everywhere: column quad = tid & 63, rows = (tid >> 6) + 4*it  (no integer divisions; the row is
// wave-uniform, so per-row scalars — point index, neighbour row — are computed once per wave).
// ---------------------------------------------------------------------------------------------------------
template <int EMODE>
__device__ __forceinline__ void rg_epilogue(const RgParams& p, float* Cs, int ldc, long row0, int rv, int n0,
                                            int ncols, double& st_sum, double& st_sq)
{
    const int tid = threadIdx.x;
    const int c = (tid & 63) << 2;            // first column of this thread's quad inside the block
    const int rw = tid >> 6;
    if (EMODE == E_LINEAR) {
        if (c < ncols) {
            const int nvalid = ncols - c;
            for (int r = rw; r < rv; r += 4) {
                const long gr = row0 + r;
                const float4 v = ld4(&Cs[r * ldc + c]);
                float o[4] = {v.x, v.y, v.z, v.w};
                const float* ad = nullptr;
                if (p.addend.base) ad = gpe_row_ptr(p.addend, gr) + n0 + c;
                float* dst;
                if (p.y_inner <= 0) dst = p.y + gr * p.y_so + n0 + c;
                else { long oo = gr / p.y_inner; dst = p.y + oo * p.y_so + (gr - oo * p.y_inner) * p.y_si + n0 + c; }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t < nvalid) {
                        float x = o[t];
                        if (p.bias) x += p.bias[n0 + c + t];
                        if (ad) x += ad[t];
                        if (p.act == 1) x = fmaxf(x, 0.f);
                        o[t] = x;
                    }
                }
                if (nvalid >= 4 && gpe_aligned16(dst)) st4(dst, make_float4(o[0], o[1], o[2], o[3]));
                else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) if (t < nvalid) dst[t] = o[t];
                }
            }
        }
    } else if (EMODE == E_EDGE_FWD) {
        // bias + ReLU in place in LDS; BN statistics per column in fp64 (thread tid owns column tid)
        if (tid < ncols) {
            const float bz = p.bias ? p.bias[n0 + tid] : 0.f;
            for (int r = 0; r < rv; ++r) {
                const float v = fmaxf(Cs[r * ldc + tid] + bz, 0.f);
                Cs[r * ldc + tid] = v;
                st_sum += (double)v;
                st_sq += (double)v * (double)v;
            }
        }
        __syncthreads();
        // coalesced row stores (ldo is a multiple of 4; pad columns hold relu(0) = 0)
        if (c < ncols)
            for (int r = rw; r < rv; r += 4) st4(p.out + (row0 + r) * p.ldo + n0 + c, ld4(&Cs[r * ldc + c]));
        if (p.agg) {
            // tiles hold whole points: thread = column, loop over the tile's points
            if (tid < ncols) {
                const int pts = rv / p.k;
                const long pt0 = (long)gpe_udiv((unsigned)row0, (unsigned)p.k, p.rcp_k);
                for (int pt = 0; pt < pts; ++pt) {
                    const float* col = &Cs[(pt * p.k) * ldc + tid];
                    float vmx = col[0], vmn = col[0];
                    int imx = 0, imn = 0;
                    for (int s = 1; s < p.k; ++s) {
                        const float v = col[s * ldc];
                        if (v > vmx) { vmx = v; imx = s; }
                        if (v < vmn) { vmn = v; imn = s; }
                    }
                    const long o = (pt0 + pt) * p.oldagg + n0 + tid;
                    p.mx[o] = vmx; p.mn[o] = vmn;
                    p.oamx[o] = (uint8_t)imx; p.oamn[o] = (uint8_t)imn;
                }
            }
        }
    } else {   // E_BWD_INPLACE / E_BWD_GATHER : dz = (act>0) ? s*u - c1 - (act-mean)*k2 : 0
        if (c < ncols) {
            const int nvalid = ncols - c;
            float cs_[4], c1_[4], k2_[4], mu_[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int cc = n0 + c + (t < nvalid ? t : 0);
                cs_[t] = p.coef_out[cc]; c1_[t] = p.coef_out[p.N + cc];
                k2_[t] = p.coef_out[2 * p.N + cc]; mu_[t] = p.coef_out[3 * p.N + cc];
            }
            for (int r = rw; r < rv; r += 4) {
                const long gr = row0 + r;
                const float4 u4 = ld4(&Cs[r * ldc + c]);
                float* dst = p.out + gr * p.ldo + n0 + c;
                float4 act;
                if (EMODE == E_BWD_INPLACE) {
                    act = ld4(dst);
                } else {
                    const long i = (long)gpe_udiv((unsigned)gr, (unsigned)p.k, p.rcp_k);
                    const long jj = p.jg[gr];
                    const float4 a = ld4(p.pq + i * p.ldpq + n0 + c);
                    const float4 b = ld4(p.pq + jj * p.ldpq + p.H + n0 + c);
                    act = make_float4(fmaxf(a.x + b.x, 0.f), fmaxf(a.y + b.y, 0.f), fmaxf(a.z + b.z, 0.f),
                                      fmaxf(a.w + b.w, 0.f));
                }
                const float uv[4] = {u4.x, u4.y, u4.z, u4.w};
                const float av[4] = {act.x, act.y, act.z, act.w};
                float o[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    o[t] = 0.f;
                    if (t < nvalid && av[t] > 0.f) o[t] = uv[t] * cs_[t] - c1_[t] - (av[t] - mu_[t]) * k2_[t];
                }
                const float4 o4 = make_float4(o[0], o[1], o[2], o[3]);
                st4(dst, o4);
                if (EMODE == E_BWD_GATHER) st4(&Cs[r * ldc + c], o4);
            }
        }
        if (EMODE == E_BWD_GATHER) {
            __syncthreads();
            if (tid < ncols) {
                const int pts = rv / p.k;
                const long pt0 = (long)gpe_udiv((unsigned)row0, (unsigned)p.k, p.rcp_k);
                for (int pt = 0; pt < pts; ++pt) {
                    const float* col = &Cs[(pt * p.k) * ldc + tid];
                    float s = 0.f;
                    for (int t = 0; t < p.k; ++t) s += col[t * ldc];
                    p.dP[(pt0 + pt) * p.lddp + n0 + tid] = s;
                }
            }
        }
    }
}
