// Latency-bound row GEMMs for gfx950: the LSTM recurrence (nn.LSTM under LSTMDecoderModule,
// /root/reference/nn/net_blocks.py:373,388-393) and every other Linear whose grid cannot fill the chip.
//
// The streaming kernel in gpe_rowgemm.hip pays two barriers and one exposed L2 round trip per 16-wide K chunk; with
// only 4 N-tiles per workgroup that is ~2 us of latency per 0.2 us of MFMA work.  Here a workgroup stages a WHOLE
// K slab (<= 256) of both operands at once — 64 A rows and the packed weight block of its 16*NT columns, every load in
// flight together — then runs the slab's MFMAs (v_mfma_f32_16x16x4_f32) straight through: one barrier pair per slab.
//
// EPI_LSTM fuses the LSTM cell into the epilogue.  The recurrent weight is packed GATE-INTERLEAVED
// (gpe_pack_weight_gates): column block b holds [i | f | g | o] x units 16b..16b+15, so the four pre-activations of one
// (row, unit) sit in the same accumulator tile row and the cell update (sigmoid/tanh, c, h) needs no second kernel.
#include "gpe_rowgemm.h"
#include <math.h>

enum { EPI_LINEAR = 0, EPI_LSTM = 1, EPI_GRU = 2 };

struct SgParams {
    int M, N, K;                 // N = 4*H for EPI_LSTM (H hidden units)
    GpeRows a;
    const float* wp; int Npad;   // packed weight (gate-interleaved for EPI_LSTM)
    const float* bias;
    GpeRows addend;
    float* y; long y_so, y_si; int y_inner; int act;
    // EPI_LSTM
    int H;
    const float* xproj; long xp_stride;          // [M][4H] rows (b_ih + b_hh already folded in)
    const float* c_prev; long ldc_prev;
    float* gates; float* c_out; float* h_out; long h_stride;
    const float* bhn;            // EPI_GRU: b_hn [H] (stays inside the r-gated recurrent term)
    int a_padded;                // A rows may be read up to round4(K) (finite pad): enables plain 16-B staging loads
    // split-K (gridDim.z > 1): block z handles K slab z only and writes its partial product to y + z * y_zstride
    long y_zstride;
};

__device__ __forceinline__ float sg_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

template <int NT, int EPI>
__global__ __launch_bounds__(256) void gpe_smallgemm_kernel(SgParams p)
{
    extern __shared__ __align__(16) float smem[];
    const int kp_max = ((p.K < RG_KSLAB ? p.K : RG_KSLAB) + 15) & ~15;
    const int lda = kp_max + 4;
    constexpr int ldc = 16 * NT + 4;
    const int a_floats = RG_BM * (lda > ldc ? lda : ldc);
    float* As = smem;
    float* Cs = smem;                               // aliases As after the last slab
    float* Ws = smem + a_floats;                    // [kp/16][4][16*NT][4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * RG_BM;
    const int rv = (p.M - row0 < RG_BM) ? (p.M - row0) : RG_BM;
    const int n0 = blockIdx.y * (16 * NT);

    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ks_begin = (gridDim.z > 1) ? (int)blockIdx.z * RG_KSLAB : 0;
    const int ks_end = (gridDim.z > 1) ? ((ks_begin + RG_KSLAB < p.K) ? ks_begin + RG_KSLAB : p.K) : p.K;
    for (int ks = ks_begin; ks < ks_end; ks += RG_KSLAB) {
        const int kslab = (p.K - ks < RG_KSLAB) ? (p.K - ks) : RG_KSLAB;
        const int kp = (kslab + 15) & ~15;
        __syncthreads();                            // previous slab's reads finished
        // ---- stage A slab: lane = column quad, rows = wave + 4*it ------------------------------------------
        {
            const int c = lane << 2;
            const bool vec = p.a_padded && p.a.inner <= 0 && !(p.a.stride_outer & 3) && gpe_aligned16(p.a.base);
            if (c < kp) {
                const int nvalid = kslab - c;
                if (vec) {
                    // aligned rows: one plain 16-B load per row, unconditional (row clamped), all 16 in flight
                    const int cc = (nvalid > 0) ? c : 0;
                    float4 v[RG_BM / 4];
#pragma unroll
                    for (int q = 0; q < RG_BM / 4; ++q) {
                        const int r = wave + 4 * q;
                        v[q] = ld4(p.a.base + (long)(row0 + (r < rv ? r : rv - 1)) * p.a.stride_outer + ks + cc);
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
                } else {
                    for (int r = wave; r < RG_BM; r += 4) {
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (r < rv && nvalid > 0) {
                            const float* src = gpe_row_ptr(p.a, row0 + r) + ks + c;
                            v = ld4_guard(src, nvalid, gpe_aligned16(src));
                        }
                        st4(&As[r * lda + c], v);
                    }
                }
            }
        }
        // ---- stage the packed weight block of this slab: contiguous 16-B copies -----------------------------
        {
            const int planes = (kp >> 4) * 4;       // (chunk, k-quad) planes
            constexpr int per_plane = 16 * NT;      // float4 per plane for this column block
            const int chunk0 = ks >> 4;
            const int total = planes * per_plane;
            // WB copies in flight per thread: issued one at a time (load -> LDS store -> next load) this copy was a
            // chain of up to 16 dependent L2 round trips, most of the kernel's run time at LSTM sizes
            constexpr int WB = 8;
            for (int e0 = tid; e0 < total; e0 += 256 * WB) {
                float4 v[WB];
#pragma unroll
                for (int u = 0; u < WB; ++u) {
                    const int e = e0 + 256 * u;
                    const int ec = (e < total) ? e : total - 1;          // clamped: unconditional loads
                    const int pl = ec / per_plane, n = ec - pl * per_plane;
                    const int nn = (n0 + n < p.Npad) ? n0 + n : p.Npad - 1;
                    v[u] = ld4(p.wp + (((long)(chunk0 * 4 + pl)) * p.Npad + nn) * 4);
                }
#pragma unroll
                for (int u = 0; u < WB; ++u) {
                    const int e = e0 + 256 * u;
                    if (e < total) {
                        const int n = e % per_plane;
                        st4(&Ws[e * 4], (n0 + n < p.Npad) ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f));
                    }
                }
            }
        }
        __syncthreads();
        const int nchunks = kp >> 4;
        for (int kc = 0; kc < nchunks; ++kc) {
            const float4 a4 = ld4(&As[(16 * wave + j) * lda + kc * 16 + 4 * g]);
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
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cs[(16 * wave + 4 * g + r) * ldc + 16 * n + j] = acc[n][r];
    __syncthreads();

    if (EPI == EPI_LINEAR) {
        const int ncols = (p.N - n0 < 16 * NT) ? (p.N - n0) : 16 * NT;
        const int c = lane << 2;
        if (c < ncols) {
            const int nvalid = ncols - c;
            for (int r = wave; r < rv; r += 4) {
                const long gr = row0 + r;
                const float4 v = ld4(&Cs[r * ldc + c]);
                float o[4] = {v.x, v.y, v.z, v.w};
                const float* ad = nullptr;
                if (p.addend.base) ad = gpe_row_ptr(p.addend, gr) + n0 + c;
                float* dst;
                if (p.y_inner <= 0) dst = p.y + (long)blockIdx.z * p.y_zstride + gr * p.y_so + n0 + c;
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
    } else if (EPI == EPI_GRU) {
        // NT == 3: the block's 48 columns are [r|z|n] x 16 units of W_hh.h (recurrent part only); xproj holds the input part
        // x.W_ih^T + b_ih (+ b_hr, b_hz).  nn.GRU: r = s(xr + hr), z = s(xz + hz), n = tanh(xn + r*(hn + b_hn)),
        // h' = (1-z)*n + z*h.  Saved for backward: [r | z | n | hn + b_hn] (4H per row).
        const int u = tid & 15;
        const int unit = blockIdx.y * 16 + u;
        if (unit < p.H) {
            constexpr int IT = RG_BM / 16;
            float xr[IT], xz[IT], xn[IT], hp[IT];
            const float bhn = p.bhn[unit];
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int r = (tid >> 4) + 16 * it;
                const long gr = row0 + (r < rv ? r : rv - 1);
                const float* xp = p.xproj + gr * p.xp_stride;
                xr[it] = xp[unit]; xz[it] = xp[p.H + unit]; xn[it] = xp[2 * p.H + unit];
                hp[it] = p.a.base[gr * p.a.stride_outer + unit];
            }
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int r = (tid >> 4) + 16 * it;
                if (r < rv) {
                    const long gr = row0 + r;
                    const float rg = sg_sigmoid(Cs[r * ldc + u] + xr[it]);
                    const float zg = sg_sigmoid(Cs[r * ldc + 16 + u] + xz[it]);
                    const float hn = Cs[r * ldc + 32 + u] + bhn;
                    const float ng = tanhf(xn[it] + rg * hn);
                    float* go = p.gates + gr * 4 * p.H;
                    go[unit] = rg; go[p.H + unit] = zg; go[2 * p.H + unit] = ng; go[3 * p.H + unit] = hn;
                    p.h_out[gr * p.h_stride + unit] = (1.f - zg) * ng + zg * hp[it];
                }
            }
        }
    } else {
        // NT == 4: the block's 64 columns are [i|f|g|o] x 16 units.  thread -> (row = tid/16 + 16*it, unit = tid%16)
        const int u = tid & 15;
        const int unit = blockIdx.y * 16 + u;
        if (unit < p.H) {
            // all four row-iterations' global operands are loaded up front (clamped rows): consumed one iteration at a
            // time they were four serial L2 round trips at the tail of a 13 us kernel
            constexpr int IT = RG_BM / 16;
            float xi[IT], xf[IT], xg[IT], xo[IT], cp[IT];
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int r = (tid >> 4) + 16 * it;
                const long gr = row0 + (r < rv ? r : rv - 1);
                const float* xp = p.xproj + gr * p.xp_stride;
                xi[it] = xp[unit]; xf[it] = xp[p.H + unit]; xg[it] = xp[2 * p.H + unit]; xo[it] = xp[3 * p.H + unit];
                cp[it] = p.c_prev[gr * p.ldc_prev + unit];
            }
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int r = (tid >> 4) + 16 * it;
                if (r < rv) {
                    const long gr = row0 + r;
                    const float zi = Cs[r * ldc + u] + xi[it];
                    const float zf = Cs[r * ldc + 16 + u] + xf[it];
                    const float zg = Cs[r * ldc + 32 + u] + xg[it];
                    const float zo = Cs[r * ldc + 48 + u] + xo[it];
                    const float ig = sg_sigmoid(zi), fg = sg_sigmoid(zf), gg = tanhf(zg), og = sg_sigmoid(zo);
                    const float cn = fg * cp[it] + ig * gg;
                    float* go = p.gates + gr * 4 * p.H;
                    go[unit] = ig; go[p.H + unit] = fg; go[2 * p.H + unit] = gg; go[3 * p.H + unit] = og;
                    p.c_out[gr * p.H + unit] = cn;
                    p.h_out[gr * p.h_stride + unit] = og * tanhf(cn);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
static size_t sg_lds_bytes(int NT, int K)
{
    const int kp_max = gpe_round_up(K < RG_KSLAB ? K : RG_KSLAB, 16);
    const int lda = kp_max + 4, ldc = 16 * NT + 4;
    return ((size_t)RG_BM * (lda > ldc ? lda : ldc) + (size_t)kp_max * 16 * NT) * sizeof(float);
}

template <int NT, int EPI>
static int sg_launch(const SgParams& p, dim3 grid, hipStream_t s)
{
    const size_t lds = sg_lds_bytes(NT, p.K);
    if (lds > 160 * 1024) return GPE_EINVAL;
    GPE_ENSURE_MAX_LDS((gpe_smallgemm_kernel<NT, EPI>));
    hipLaunchKernelGGL((gpe_smallgemm_kernel<NT, EPI>), grid, dim3(256), lds, s, p);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// used by gpe_linear (gpe_rowgemm.hip) when the streaming kernel's grid would be latency-bound
int gpe_smallgemm_linear(const RgParams& r, hipStream_t s)
{
    SgParams p = {};
    p.M = (int)r.M; p.N = r.N; p.K = r.K;
    p.a = r.a; p.wp = r.wp; p.Npad = r.Npad; p.bias = r.bias; p.addend = r.addend;
    p.y = r.y; p.y_so = r.y_so; p.y_si = r.y_si; p.y_inner = r.y_inner; p.act = r.act;
    p.a_padded = !(p.K & 3);
    const int tiles = gpe_cdiv(p.M, RG_BM);
    // 64-column blocks unless that leaves most CUs idle, then 16-column blocks
    if ((long)tiles * gpe_cdiv(p.N, 64) >= 128 || p.N <= 16) {
        if (p.N <= 16) return sg_launch<1, EPI_LINEAR>(p, dim3(tiles, gpe_cdiv(p.N, 16)), s);
        return sg_launch<4, EPI_LINEAR>(p, dim3(tiles, gpe_cdiv(p.N, 64)), s);
    }
    return sg_launch<1, EPI_LINEAR>(p, dim3(tiles, gpe_cdiv(p.N, 16)), s);
}

extern "C" int gpe_lstm_step_fwd(const float* h_prev, long hp_stride, const float* whh_gates_packed,
                                 const float* xproj, long xp_stride, const float* c_prev, long ldc_prev,
                                 float* gates, float* c_out, float* h_out, long h_stride, int Bn, int H,
                                 void* stream)
{
    if (!h_prev || !whh_gates_packed || !xproj || !c_prev || !gates || !c_out || !h_out || Bn <= 0 || H <= 0)
        return GPE_EINVAL;
    SgParams p = {};
    p.M = Bn; p.N = 4 * H; p.K = H;
    p.a = GpeRows{h_prev, hp_stride, 0, 0};
    p.a_padded = !(hp_stride & 3);              // contract: h rows are padded to a multiple of 4 floats (finite pad)
    p.wp = whh_gates_packed; p.Npad = 64 * gpe_cdiv(H, 16);
    p.H = H; p.xproj = xproj; p.xp_stride = xp_stride; p.c_prev = c_prev; p.ldc_prev = ldc_prev;
    p.gates = gates; p.c_out = c_out; p.h_out = h_out; p.h_stride = h_stride;
    return sg_launch<4, EPI_LSTM>(p, dim3(gpe_cdiv(Bn, RG_BM), gpe_cdiv(H, 16)), (hipStream_t)stream);
}

// split-K product for long-K, small-M shapes (the LSTM backward recurrence dh = dG . W_hh, K = 4H): one K slab per
// workgroup so every load of the launch is in flight at once; the ceil(K/256) partial products land in
// y[z][M][N] and are summed by their consumer (gpe_lstm_cell_bwd's n_rec).  Returns the number of partials via *nz.
extern "C" int gpe_linear_splitk(const float* a, long a_so, const float* wp, float* y, int M, int N, int K,
                                 void* stream)
{
    if (!a || !wp || !y || M <= 0 || N <= 0 || K <= 0) return GPE_EINVAL;
    SgParams p = {};
    p.M = M; p.N = N; p.K = K;
    p.a = GpeRows{a, a_so, 0, 0};
    p.wp = wp; p.Npad = gpe_round_up(N, 16);
    p.y = y; p.y_so = N; p.y_zstride = (long)M * N;
    p.a_padded = !(K & 3);
    const int nz = gpe_cdiv(K, RG_KSLAB);
    return sg_launch<4, EPI_LINEAR>(p, dim3(gpe_cdiv(M, RG_BM), gpe_cdiv(N, 64), nz), (hipStream_t)stream);
}

// fused GRU step (nn.GRU recurrence, gate order r,z,n; GRUDecoderModule, /root/reference/nn/net_blocks.py:457-497):
// W_hh packed gate-interleaved with 3 gates (gpe_pack_weight_ngates(.., 3, ..)); xproj rows [Bn][3H] = x.W_ih^T + b_ih
// (+ b_hr, b_hz folded in); saved [Bn][4H] = {r, z, n, W_hn.h + b_hn}.
extern "C" int gpe_gru_step_fwd(const float* h_prev, long hp_stride, const float* whh_gates_packed, const float* xproj,
                                long xp_stride, const float* bhn, float* saved, float* h_out, long h_stride, int Bn,
                                int H, void* stream)
{
    if (!h_prev || !whh_gates_packed || !xproj || !bhn || !saved || !h_out || Bn <= 0 || H <= 0) return GPE_EINVAL;
    SgParams p = {};
    p.M = Bn; p.N = 3 * H; p.K = H;
    p.a = GpeRows{h_prev, hp_stride, 0, 0};
    p.a_padded = !(hp_stride & 3);
    p.wp = whh_gates_packed; p.Npad = 48 * gpe_cdiv(H, 16);
    p.H = H; p.xproj = xproj; p.xp_stride = xp_stride; p.bhn = bhn;
    p.gates = saved; p.h_out = h_out; p.h_stride = h_stride;
    return sg_launch<3, EPI_GRU>(p, dim3(gpe_cdiv(Bn, RG_BM), gpe_cdiv(H, 16)), (hipStream_t)stream);
}
