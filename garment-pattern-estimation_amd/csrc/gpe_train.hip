// The steps either side of the encoder/decoder path, as gfx950 kernels:
//   * pattern loss (MSE shape / rotation / translation + panel loop loss) value and gradient, and the ground-truth
//     pre-processing in front of it (panel-origin matching, greedy panel-order matching)
//       /root/reference/nn/metrics/composed_loss.py:294-334,530-570,656-703  and  nn/metrics/losses.py:19-51
//   * attention pooling over the per-point panel scores (nn/nets.py:263-276) for mean / add / max pooling
//   * global max / add pooling (torch_geometric global_max_pool / global_add_pool, nn/net_blocks.py:145-150)
//   * sum over the k messages of a point (EdgeConv aggr = 'add' / 'mean', nn/net_blocks.py:129)
//   * fused Adam step over one flat parameter arena (torch.optim.Adam, nn/trainer.py:162-172)
//   * input standardisation (nn/data/transforms.py:35-50)
// All of it is bandwidth- or latency-bound small work: no MFMA here.  Reductions run in fp64 with a fixed order, so
// results are run-to-run deterministic; there are no float atomics.
#include "gpe_common.h"
#include <math.h>

// =====================================================================================================================
// pattern loss
// =====================================================================================================================
// Predictions are strided views (the output dict holds slices of one [B,P,L,8] and one [B*P,7] tensor):
//   outlines element (b,p,l,c) at ol + b*ol_sb + p*ol_sp + l*ol_sl + c           (c < 4)
//   rotations (b,p,c) at rot + (b*P+p)*rot_s + c  (c < R);  translations likewise (c < T)
// Ground truth is dense: gt_ol [B,P,L,4], gt_rot [B,P,R], gt_tr [B,P,T], num_edges int32 [B*P].
// flags: bit0 shape, bit1 loop, bit2 rotation, bit3 translation.
//   shape = mean (ol-gt)^2 ; rotation / translation likewise ; loop = sum_panels |sum_{l<n} (ol[l,:2]-pad)|^2 / (2*B*P),
//   panels with n < 3 skipped (nn/metrics/losses.py:36-51).
struct LossParams {
    const float* ol; long ol_sb, ol_sp, ol_sl;
    const float* rot; long rot_s;
    const float* tr; long tr_s;
    const float* gt_ol; const float* gt_rot; const float* gt_tr; const int32_t* num_edges;
    int B, P, L, R, T, flags;
    float pad0, pad1, loop_w;
};

#define LOSS_TPB 256
__global__ __launch_bounds__(LOSS_TPB) void gpe_loss_fwd_kernel(LossParams p, double* __restrict__ part /* [B][4] */,
                                                                float* __restrict__ loop_sums /* [B*P][2] */)
{
    __shared__ double red[4][LOSS_TPB];
    const int b = blockIdx.x, tid = threadIdx.x;
    double s_shape = 0, s_loop = 0, s_rot = 0, s_tr = 0;
    if (p.flags & 1) {
        const int n = p.P * p.L * 4;
        for (int e = tid; e < n; e += LOSS_TPB) {
            const int c = e & 3, l = (e >> 2) % p.L, pp = (e >> 2) / p.L;
            const float d = p.ol[b * p.ol_sb + pp * p.ol_sp + l * p.ol_sl + c] - p.gt_ol[(size_t)b * n + e];
            s_shape += (double)d * d;
        }
    }
    if (p.flags & 2) {
        // thread (panel, coordinate): sequential sum over the panel's first n edges
        for (int e = tid; e < p.P * 2; e += LOSS_TPB) {
            const int pp = e >> 1, c = e & 1;
            const int n = p.num_edges[b * p.P + pp];
            float s = 0.f;
            if (n >= 3) {
                const float pad = c ? p.pad1 : p.pad0;
                const int nn = n < p.L ? n : p.L;
                for (int l = 0; l < nn; ++l) s += p.ol[b * p.ol_sb + pp * p.ol_sp + l * p.ol_sl + c] - pad;
            }
            loop_sums[((size_t)b * p.P + pp) * 2 + c] = s;
            s_loop += (double)s * s;
        }
    }
    if (p.flags & 4) {
        for (int e = tid; e < p.P * p.R; e += LOSS_TPB) {
            const int pp = e / p.R, c = e - pp * p.R;
            const float d = p.rot[((long)b * p.P + pp) * p.rot_s + c] - p.gt_rot[(size_t)b * p.P * p.R + e];
            s_rot += (double)d * d;
        }
    }
    if (p.flags & 8) {
        for (int e = tid; e < p.P * p.T; e += LOSS_TPB) {
            const int pp = e / p.T, c = e - pp * p.T;
            const float d = p.tr[((long)b * p.P + pp) * p.tr_s + c] - p.gt_tr[(size_t)b * p.P * p.T + e];
            s_tr += (double)d * d;
        }
    }
    red[0][tid] = s_shape; red[1][tid] = s_loop; red[2][tid] = s_rot; red[3][tid] = s_tr;
    __syncthreads();
    for (int st = LOSS_TPB / 2; st > 0; st >>= 1) {
        if (tid < st)
#pragma unroll
            for (int q = 0; q < 4; ++q) red[q][tid] += red[q][tid + st];
        __syncthreads();
    }
    if (tid < 4) part[(size_t)b * 4 + tid] = red[tid][0];
}

// out[0] = total, out[1..4] = shape, loop, rotation, translation (each already divided by its element count)
__global__ void gpe_loss_final_kernel(const double* __restrict__ part, LossParams p, float* __restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s[4] = {0, 0, 0, 0};
    for (int b = 0; b < p.B; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] += part[(size_t)b * 4 + q];
    const double nb = (double)p.B * p.P;
    const double shape = s[0] / (nb * p.L * 4), loop = s[1] / (nb * 2), rot = s[2] / (nb * p.R), tr = s[3] / (nb * p.T);
    double total = 0;
    if (p.flags & 1) total += shape;
    if (p.flags & 2) total += (double)p.loop_w * loop;
    if (p.flags & 4) total += rot;
    if (p.flags & 8) total += tr;
    out[0] = (float)total;
    out[1] = (p.flags & 1) ? (float)shape : 0.f;
    out[2] = (p.flags & 2) ? (float)loop : 0.f;
    out[3] = (p.flags & 4) ? (float)rot : 0.f;
    out[4] = (p.flags & 8) ? (float)tr : 0.f;
}

// gradients of the TOTAL loss w.r.t. the three prediction views, scaled by the upstream gradient *gscale (device scalar):
//   g_ol [B,P,L,4] dense, g_rot [B,P,R], g_tr [B,P,T]
__global__ __launch_bounds__(LOSS_TPB) void gpe_loss_bwd_kernel(LossParams p, const float* __restrict__ loop_sums,
                                                                const float* __restrict__ gscale,
                                                                float* __restrict__ g_ol, float* __restrict__ g_rot,
                                                                float* __restrict__ g_tr)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    const float gs = gscale ? gscale[0] : 1.f;
    const double nb = (double)p.B * p.P;
    const float k_shape = (p.flags & 1) ? (float)(2.0 / (nb * p.L * 4)) * gs : 0.f;
    const float k_loop = (p.flags & 2) ? (float)(2.0 * p.loop_w / (nb * 2)) * gs : 0.f;
    const float k_rot = (float)(2.0 / (nb * p.R)) * gs, k_tr = (float)(2.0 / (nb * p.T)) * gs;
    {
        const int n = p.P * p.L * 4;
        for (int e = tid; e < n; e += LOSS_TPB) {
            const int c = e & 3, l = (e >> 2) % p.L, pp = (e >> 2) / p.L;
            float g = 0.f;
            if (p.flags & 1)
                g = k_shape * (p.ol[b * p.ol_sb + pp * p.ol_sp + l * p.ol_sl + c] - p.gt_ol[(size_t)b * n + e]);
            if ((p.flags & 2) && c < 2) {
                const int ne = p.num_edges[b * p.P + pp];
                if (ne >= 3 && l < ne) g += k_loop * loop_sums[((size_t)b * p.P + pp) * 2 + c];
            }
            g_ol[(size_t)b * n + e] = g;
        }
    }
    if (g_rot)
        for (int e = tid; e < p.P * p.R; e += LOSS_TPB) {
            const int pp = e / p.R, c = e - pp * p.R;
            g_rot[(size_t)b * p.P * p.R + e] = (p.flags & 4)
                ? k_rot * (p.rot[((long)b * p.P + pp) * p.rot_s + c] - p.gt_rot[(size_t)b * p.P * p.R + e]) : 0.f;
        }
    if (g_tr)
        for (int e = tid; e < p.P * p.T; e += LOSS_TPB) {
            const int pp = e / p.T, c = e - pp * p.T;
            g_tr[(size_t)b * p.P * p.T + e] = (p.flags & 8)
                ? k_tr * (p.tr[((long)b * p.P + pp) * p.tr_s + c] - p.gt_tr[(size_t)b * p.P * p.T + e]) : 0.f;
        }
}

static int loss_params(LossParams& p, const float* ol, long ol_sb, long ol_sp, long ol_sl, const float* rot, long rot_s,
                       const float* tr, long tr_s, const float* gt_ol, const float* gt_rot, const float* gt_tr,
                       const int32_t* num_edges, int B, int P, int L, int R, int T, int flags, float pad0, float pad1,
                       float loop_w)
{
    if (B <= 0 || P <= 0 || L <= 0 || (flags & ~15)) return GPE_EINVAL;
    if ((flags & 3) && (!ol || !gt_ol)) return GPE_EINVAL;
    if ((flags & 2) && !num_edges) return GPE_EINVAL;
    if ((flags & 4) && (!rot || !gt_rot || R <= 0)) return GPE_EINVAL;
    if ((flags & 8) && (!tr || !gt_tr || T <= 0)) return GPE_EINVAL;
    p.ol = ol; p.ol_sb = ol_sb; p.ol_sp = ol_sp; p.ol_sl = ol_sl;
    p.rot = rot; p.rot_s = rot_s; p.tr = tr; p.tr_s = tr_s;
    p.gt_ol = gt_ol; p.gt_rot = gt_rot; p.gt_tr = gt_tr; p.num_edges = num_edges;
    p.B = B; p.P = P; p.L = L; p.R = R > 0 ? R : 1; p.T = T > 0 ? T : 1; p.flags = flags;
    p.pad0 = pad0; p.pad1 = pad1; p.loop_w = loop_w;
    return GPE_OK;
}

extern "C" int gpe_pattern_loss_fwd(const float* ol, long ol_sb, long ol_sp, long ol_sl, const float* rot, long rot_s,
                                    const float* tr, long tr_s, const float* gt_ol, const float* gt_rot,
                                    const float* gt_tr, const int32_t* num_edges, int B, int P, int L, int R, int T,
                                    int flags, float pad0, float pad1, float loop_w, double* part, float* loop_sums,
                                    float* out5, void* stream)
{
    LossParams p;
    const int rc = loss_params(p, ol, ol_sb, ol_sp, ol_sl, rot, rot_s, tr, tr_s, gt_ol, gt_rot, gt_tr, num_edges, B, P, L,
                               R, T, flags, pad0, pad1, loop_w);
    if (rc != GPE_OK || !part || !loop_sums || !out5) return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_loss_fwd_kernel, dim3(B), dim3(LOSS_TPB), 0, (hipStream_t)stream, p, part, loop_sums);
    hipLaunchKernelGGL(gpe_loss_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, part, p, out5);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

extern "C" int gpe_pattern_loss_bwd(const float* ol, long ol_sb, long ol_sp, long ol_sl, const float* rot, long rot_s,
                                    const float* tr, long tr_s, const float* gt_ol, const float* gt_rot,
                                    const float* gt_tr, const int32_t* num_edges, int B, int P, int L, int R, int T,
                                    int flags, float pad0, float pad1, float loop_w, const float* loop_sums,
                                    const float* gscale, float* g_ol, float* g_rot, float* g_tr, void* stream)
{
    LossParams p;
    const int rc = loss_params(p, ol, ol_sb, ol_sp, ol_sl, rot, rot_s, tr, tr_s, gt_ol, gt_rot, gt_tr, num_edges, B, P, L,
                               R, T, flags, pad0, pad1, loop_w);
    if (rc != GPE_OK || !loop_sums || !g_ol) return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_loss_bwd_kernel, dim3(B), dim3(LOSS_TPB), 0, (hipStream_t)stream, p, loop_sums, gscale, g_ol,
                       g_rot, g_tr);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// panel-origin matching (composed_loss.py:656-703): for every panel try each of its n cyclic edge shifts of the GT loop
// and keep the FIRST one with the smallest squared distance to the prediction.  One wave per panel: lane r evaluates
// shift r (r < n <= 64).  gt_out [B*P][L][D] = the chosen rotation; lead [B*P] = the chosen leading edge.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gpe_origin_match_kernel(const float* __restrict__ ol, long ol_sb, long ol_sp,
                                                               long ol_sl, const float* __restrict__ gt, int D,
                                                               const int32_t* __restrict__ num_edges, int panels, int P,
                                                               int L, float* __restrict__ gt_out,
                                                               int32_t* __restrict__ lead)
{
    const int lane = threadIdx.x & 63;
    const int el = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (el >= panels) return;
    const int b = el / P, pp = el - b * P;
    const float* pr = ol + b * ol_sb + pp * ol_sp;
    const float* g = gt + (size_t)el * L * D;
    int n = num_edges[el];
    if (n > L) n = L;
    if (n < 0) n = 0;
    // shift r: gt_r[l] = gt[(l + r) mod n] for l < n, gt[l] beyond (padding stays in place)
    float dist = INFINITY;
    if (lane < (n > 1 ? n : 1)) {
        float s = 0.f;
        for (int l = 0; l < L; ++l) {
            int src = l;
            if (l < n) { src = l + lane; if (src >= n) src -= n; }
            for (int c = 0; c < D; ++c) {
                const float d = pr[l * ol_sl + c] - g[src * D + c];
                s = __builtin_fmaf(d, d, s);
            }
        }
        dist = s;
    }
    // first minimum over lanes: (dist, lane) lexicographic
    float best = dist;
    int bl = lane;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float od = __shfl_xor(best, off);
        const int ol_ = __shfl_xor(bl, off);
        if (od < best || (od == best && ol_ < bl)) { best = od; bl = ol_; }
    }
    const int r = bl;
    if (lane == 0) lead[el] = r;
    for (int e = lane; e < L * D; e += 64) {
        const int l = e / D, c = e - l * D;
        int src = l;
        if (l < n) { src = l + r; if (src >= n) src -= n; }
        gt_out[(size_t)el * L * D + e] = g[src * D + c];
    }
}

extern "C" int gpe_origin_match(const float* ol, long ol_sb, long ol_sp, long ol_sl, const float* gt_ol, int D,
                                const int32_t* num_edges, int B, int P, int L, float* gt_out, int32_t* lead,
                                void* stream)
{
    if (!ol || !gt_ol || !num_edges || !gt_out || !lead || B <= 0 || P <= 0 || L <= 0 || L > 64 || D <= 0)
        return GPE_EINVAL;
    const int panels = B * P;
    hipLaunchKernelGGL(gpe_origin_match_kernel, dim3(gpe_cdiv(panels, 4)), dim3(256), 0, (hipStream_t)stream, ol, ol_sb,
                       ol_sp, ol_sl, gt_ol, D, num_edges, panels, P, L, gt_out, lead);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// greedy panel-order matching (composed_loss.py:530-570): distance matrix dist[i][j] = |pred_i - gt_j|_2 (P x P), then P
// rounds of "take the global minimum (first in row-major order on ties), fix perm[row] = col, strike row and column".
// One workgroup per pattern; P <= 64.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gpe_order_match_kernel(const float* __restrict__ pf, const float* __restrict__ gf,
                                                              int P, int D, int64_t* __restrict__ perm,
                                                              int32_t* __restrict__ fail)
{
    extern __shared__ float dm[];            // [P*P] + reduction scratch
    __shared__ float rv[256];
    __shared__ int ri[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* pb = pf + (size_t)b * P * D;
    const float* gb = gf + (size_t)b * P * D;
    for (int e = tid; e < P * P; e += 256) {
        const int i = e / P, j = e - i * P;
        float s = 0.f;
        for (int c = 0; c < D; ++c) {
            const float d = pb[i * D + c] - gb[j * D + c];
            s = __builtin_fmaf(d, d, s);
        }
        dm[e] = sqrtf(s);
    }
    if (tid < P) perm[(size_t)b * P + tid] = -1;
    __syncthreads();
    for (int round = 0; round < P; ++round) {
        float best = INFINITY;
        int bi = 0x7fffffff;
        for (int e = tid; e < P * P; e += 256) {
            const float v = dm[e];
            if (v < best || (v == best && e < bi)) { best = v; bi = e; }
        }
        rv[tid] = best; ri[tid] = bi;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) {
                const float ov = rv[tid + st];
                const int oi = ri[tid + st];
                if (ov < rv[tid] || (ov == rv[tid] && oi < ri[tid])) { rv[tid] = ov; ri[tid] = oi; }
            }
            __syncthreads();
        }
        int sel = ri[0];
        if (sel == 0x7fffffff) sel = 0;      // everything +inf already (NaN inputs): torch's argmin returns 0 as well
        const int row = sel / P, col = sel - row * P;
        __syncthreads();
        if (tid == 0) perm[(size_t)b * P + row] = col;
        for (int e = tid; e < P; e += 256) { dm[row * P + e] = INFINITY; dm[e * P + col] = INFINITY; }
        __syncthreads();
    }
    // the reference raises if a finite entry is left (composed_loss.py:567-568)
    int bad = 0;
    for (int e = tid; e < P * P; e += 256) bad |= isfinite(dm[e]) ? 1 : 0;
    if (bad) fail[0] = 1;
    // Degenerate input (NaN / inf predictions: every distance +inf, the rounds above keep hitting entry 0): rows would stay
    // at -1 and the gathers that consume `perm` would index out of bounds (a device-side assert that poisons the context,
    // where the reference raises a clean ValueError).  Keep the output in bounds — unmatched rows take the unused columns
    // in ascending order — and report the failure through `fail` (thread 0 wrote every perm entry itself).
    if (tid == 0) {
        unsigned long long used = 0ull;
        bool open_rows = false;
        for (int r = 0; r < P; ++r) {
            const int64_t c = perm[(size_t)b * P + r];
            if (c >= 0) used |= 1ull << c; else open_rows = true;
        }
        if (open_rows) {
            fail[0] = 1;
            int c = 0;
            for (int r = 0; r < P; ++r) {
                if (perm[(size_t)b * P + r] >= 0) continue;
                while (c < P - 1 && ((used >> c) & 1ull)) ++c;
                perm[(size_t)b * P + r] = c;
                used |= 1ull << c;
            }
        }
    }
}

extern "C" int gpe_order_match(const float* pred_feat, const float* gt_feat, int B, int P, int D, int64_t* perm,
                               int32_t* fail, void* stream)
{
    if (!pred_feat || !gt_feat || !perm || !fail || B <= 0 || P <= 0 || P > 64 || D <= 0) return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_order_match_kernel, dim3(B), dim3(256), (size_t)P * P * sizeof(float), (hipStream_t)stream,
                       pred_feat, gt_feat, P, D, perm, fail);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// =====================================================================================================================
// attention pooling: pooled[b][p][c] = pool_n ( w[b*N+n][p] * feat[b*N+n][c] )      (nn/nets.py:263-276)
//   mode 0 mean (1/N sum), 1 max, 2 add.  Workgroup = (cloud, 256-point slab); thread = a set of (p, c) outputs held in
//   registers; the slab's w / feat rows are staged in LDS.  Partials [B][nslab][P][C] are combined in slab order by the
//   second kernel (deterministic).  max keeps the argmax point for the backward pass.
// =====================================================================================================================
#define AP_ROWS 64
#define AP_TPT 2         // 4 x 4 output tiles per thread: ceil(P/4) * ceil(C/4) <= 256 * AP_TPT
__global__ __launch_bounds__(256) void gpe_attn_pool_part_kernel(const float* __restrict__ w, int ldw,
                                                                 const float* __restrict__ feat, int ldf, int N, int P,
                                                                 int C, int mode, int nslab, float* __restrict__ part,
                                                                 int32_t* __restrict__ part_arg)
{
    // register tiling: a thread owns 4 heads x 4 channels, so a slab row costs two ds_read_b128 per 16 products (one
    // (p, c) output per register with two ds_read_b32 per product was LDS-issue bound: 0.72 ms at cfg 4)
    extern __shared__ __align__(16) float sm[];
    const int Pp = (P + 3) & ~3, Cp = (C + 3) & ~3;
    float* ws = sm;                       // [AP_ROWS][Pp]  (pad columns zero)
    float* fs = sm + AP_ROWS * Pp;        // [AP_ROWS][Cp]
    const int b = blockIdx.y, slab = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = slab * AP_ROWS;
    const int nr = (N - n0 < AP_ROWS) ? (N - n0) : AP_ROWS;
    for (int r = wave; r < nr; r += 4) {
        const long row = (long)b * N + n0 + r;
        for (int c = lane; c < Pp; c += 64) ws[r * Pp + c] = (c < P) ? w[row * ldw + c] : 0.f;
        for (int c = lane; c < Cp; c += 64) fs[r * Cp + c] = (c < C) ? feat[row * ldf + c] : 0.f;
    }
    __syncthreads();
    const int pq = Pp >> 2, cqn = Cp >> 2, tiles = pq * cqn;
    float acc[AP_TPT][4][4];
    int arg[AP_TPT][4][4];
    int tp[AP_TPT], tc[AP_TPT];
#pragma unroll
    for (int q = 0; q < AP_TPT; ++q) {
        const int t = tid + 256 * q;
        const int tt = (t < tiles) ? t : 0;
        tp[q] = tt / cqn; tc[q] = tt - tp[q] * cqn;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc[q][i][k] = (mode == 1) ? -INFINITY : 0.f; arg[q][i][k] = 0; }
    }
    const int ntq = (tiles + 255) >> 8;       // tile slots in use (uniform)
    for (int r = 0; r < nr; ++r) {
#pragma unroll
        for (int q = 0; q < AP_TPT; ++q) {
            if (q < ntq) {
                const float4 wv = *reinterpret_cast<const float4*>(&ws[r * Pp + 4 * tp[q]]);
                const float4 fv = *reinterpret_cast<const float4*>(&fs[r * Cp + 4 * tc[q]]);
                const float wa[4] = {wv.x, wv.y, wv.z, wv.w}, fa[4] = {fv.x, fv.y, fv.z, fv.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float v = wa[i] * fa[k];
                        if (mode == 1) { if (v > acc[q][i][k]) { acc[q][i][k] = v; arg[q][i][k] = n0 + r; } }
                        else acc[q][i][k] += v;
                    }
            }
        }
    }
    const int total = P * C;
#pragma unroll
    for (int q = 0; q < AP_TPT; ++q) {
        if (tid + 256 * q < tiles) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int p_ = 4 * tp[q] + i, c = 4 * tc[q] + k;
                    if (p_ < P && c < C) {
                        const size_t dst = ((size_t)b * nslab + slab) * total + (size_t)p_ * C + c;
                        part[dst] = acc[q][i][k];
                        if (mode == 1) part_arg[dst] = arg[q][i][k];
                    }
                }
        }
    }
}

__global__ void gpe_attn_pool_final_kernel(const float* __restrict__ part, const int32_t* __restrict__ part_arg, int N,
                                           int total, int mode, int nslab, float* __restrict__ out,
                                           int32_t* __restrict__ arg)
{
    const int b = blockIdx.y;
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total) return;
    if (mode == 1) {
        float best = -INFINITY;
        int ba = 0;
        for (int s = 0; s < nslab; ++s) {
            const size_t src = ((size_t)b * nslab + s) * total + o;
            const float v = part[src];
            if (v > best) { best = v; ba = part_arg[src]; }       // earlier slab wins ties: first maximum
        }
        out[(size_t)b * total + o] = best;
        arg[(size_t)b * total + o] = ba;
    } else {
        double s_ = 0;
        for (int s = 0; s < nslab; ++s) s_ += (double)part[((size_t)b * nslab + s) * total + o];
        out[(size_t)b * total + o] = (float)(mode == 0 ? s_ / N : s_);
    }
}

extern "C" long gpe_attn_pool_ws(int B, int N, int P, int C) { return (long)B * gpe_cdiv(N, AP_ROWS) * P * C; }

extern "C" int gpe_attn_pool_fwd(const float* w, int ldw, const float* feat, int ldf, int B, int N, int P, int C,
                                 int mode, float* out, int32_t* arg, float* part, int32_t* part_arg, void* stream)
{
    if (!w || !feat || !out || !part || B <= 0 || N <= 0 || P <= 0 || C <= 0 || ldw < P || ldf < C || mode < 0 ||
        mode > 2 || (long)gpe_cdiv(P, 4) * gpe_cdiv(C, 4) > 256L * AP_TPT)
        return GPE_EINVAL;
    if (mode == 1 && (!arg || !part_arg)) return GPE_EINVAL;
    const int nslab = gpe_cdiv(N, AP_ROWS);
    const size_t lds = (size_t)AP_ROWS * (gpe_round_up(P, 4) + gpe_round_up(C, 4)) * sizeof(float);
    if (lds > 150 * 1024) return GPE_EINVAL;
    GPE_ENSURE_MAX_LDS_N((gpe_attn_pool_part_kernel), 150 * 1024);
    hipLaunchKernelGGL(gpe_attn_pool_part_kernel, dim3(nslab, B), dim3(256), lds, (hipStream_t)stream, w, ldw, feat, ldf, N,
                       P, C, mode, nslab, part, part_arg);
    hipLaunchKernelGGL(gpe_attn_pool_final_kernel, dim3(gpe_cdiv(P * C, 256), B), dim3(256), 0, (hipStream_t)stream, part,
                       part_arg, N, P * C, mode, nslab, out, arg);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// backward, mean / add: gw[n][p] = sc * sum_c feat[n][c] g[b][p][c] ; gf[n][c] = sc * sum_p w[n][p] g[b][p][c]
// (sc = 1/N or 1).  Workgroup = 64 points of a cloud; the cloud's (scaled) g matrix [P][C] and the 64 w / feat rows sit in
// LDS, outputs are register tiles (gf: 4 points x 4 channels, p ascending; gw: 2 points x 4 heads, c ascending — the same
// summation order as a plain per-output loop, so the result does not depend on the tiling).
#define APB_ROWS 64
__global__ __launch_bounds__(256) void gpe_attn_pool_bwd_kernel(const float* __restrict__ w, int ldw,
                                                                const float* __restrict__ feat, int ldf,
                                                                const float* __restrict__ g, int N, int P, int C,
                                                                float sc, float* __restrict__ gw, int ldgw,
                                                                float* __restrict__ gf, int ldgf)
{
    extern __shared__ __align__(16) float sm[];
    const int Pp = (P + 3) & ~3, Cp = (C + 3) & ~3;
    float* gs = sm;                          // [Pp][Cp]   (pad rows / columns zero)
    float* ws = gs + Pp * Cp;                // [APB_ROWS][Pp]
    float* fs = ws + APB_ROWS * Pp;          // [APB_ROWS][Cp]
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * APB_ROWS;
    const int nr = (N - n0 < APB_ROWS) ? (N - n0) : APB_ROWS;
    for (int p_ = wave; p_ < Pp; p_ += 4)
        for (int c = lane; c < Cp; c += 64) gs[p_ * Cp + c] = (p_ < P && c < C) ? g[((size_t)b * P + p_) * C + c] * sc : 0.f;
    for (int r = wave; r < APB_ROWS; r += 4) {
        const long row = (long)b * N + n0 + (r < nr ? r : nr - 1);
        for (int c = lane; c < Pp; c += 64) ws[r * Pp + c] = (c < P && r < nr) ? w[row * ldw + c] : 0.f;
        for (int c = lane; c < Cp; c += 64) fs[r * Cp + c] = (c < C && r < nr) ? feat[row * ldf + c] : 0.f;
    }
    __syncthreads();
    // gf: tiles of 4 points x 4 channels
    const int cqn = Cp >> 2;
    for (int t = tid; t < (APB_ROWS / 4) * cqn; t += 256) {
        const int ng = t / cqn, cq = t - ng * cqn;
        float acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[i][k] = 0.f;
        for (int p_ = 0; p_ < P; ++p_) {
            const float4 gv = *reinterpret_cast<const float4*>(&gs[p_ * Cp + 4 * cq]);
            const float ga[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float wv = ws[(4 * ng + i) * Pp + p_];
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[i][k] = __builtin_fmaf(wv, ga[k], acc[i][k]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * ng + i;
            if (r < nr) {
                float* dst = gf + ((long)b * N + n0 + r) * ldgf + 4 * cq;
#pragma unroll
                for (int k = 0; k < 4; ++k) if (4 * cq + k < C) dst[k] = acc[i][k];
            }
        }
    }
    // gw: tiles of 2 points x 4 heads, channels four at a time in ascending order
    const int pq = Pp >> 2;
    for (int t = tid; t < (APB_ROWS / 2) * pq; t += 256) {
        const int ng = t / pq, pg = t - ng * pq;
        float acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[i][k] = 0.f;
        for (int c4 = 0; c4 < Cp; c4 += 4) {
            float4 fv[2], gv[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) fv[i] = *reinterpret_cast<const float4*>(&fs[(2 * ng + i) * Cp + c4]);
#pragma unroll
            for (int k = 0; k < 4; ++k) gv[k] = *reinterpret_cast<const float4*>(&gs[(4 * pg + k) * Cp + c4]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float s_ = acc[i][k];
                    s_ = __builtin_fmaf(fv[i].x, gv[k].x, s_);
                    s_ = __builtin_fmaf(fv[i].y, gv[k].y, s_);
                    s_ = __builtin_fmaf(fv[i].z, gv[k].z, s_);
                    s_ = __builtin_fmaf(fv[i].w, gv[k].w, s_);
                    acc[i][k] = s_;
                }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 2 * ng + i;
            if (r < nr) {
                float* dst = gw + ((long)b * N + n0 + r) * ldgw + 4 * pg;
#pragma unroll
                for (int k = 0; k < 4; ++k) if (4 * pg + k < P) dst[k] = acc[i][k];
            }
        }
    }
}

// backward, max: only the argmax point of each (b, p, c) receives gradient.  gw / gf must be ZERO on entry.
//   pass 0: thread (b, p) walks c sequentially:  gw[arg][p] += g * feat[arg][c]   (distinct p per thread: no collisions)
//   pass 1: thread (b, c) walks p sequentially:  gf[arg][c] += g * w[arg][p]      (distinct c per thread)
__global__ void gpe_attn_pool_bwd_max_kernel(const float* __restrict__ w, int ldw, const float* __restrict__ feat,
                                             int ldf, const float* __restrict__ g, const int32_t* __restrict__ arg,
                                             int N, int P, int C, float* __restrict__ gw, int ldgw,
                                             float* __restrict__ gf, int ldgf)
{
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const float* gb = g + (size_t)b * P * C;
    const int32_t* ab = arg + (size_t)b * P * C;
    if (t < P) {
        for (int c = 0; c < C; ++c) {
            const long row = (long)b * N + ab[t * C + c];
            gw[row * ldgw + t] += gb[t * C + c] * feat[row * ldf + c];
        }
    } else if (t < P + C) {
        const int c = t - P;
        for (int p_ = 0; p_ < P; ++p_) {
            const long row = (long)b * N + ab[p_ * C + c];
            gf[row * ldgf + c] += gb[p_ * C + c] * w[row * ldw + p_];
        }
    }
}

extern "C" int gpe_attn_pool_bwd(const float* w, int ldw, const float* feat, int ldf, const float* g,
                                 const int32_t* arg, int B, int N, int P, int C, int mode, float* gw, int ldgw,
                                 float* gf, int ldgf, void* stream)
{
    if (!w || !feat || !g || !gw || !gf || B <= 0 || N <= 0 || P <= 0 || C <= 0 || mode < 0 || mode > 2 || ldgw < P ||
        ldgf < C)
        return GPE_EINVAL;
    if (mode == 1) {
        if (!arg) return GPE_EINVAL;
        if (hipMemsetAsync(gw, 0, (size_t)B * N * ldgw * sizeof(float), (hipStream_t)stream) != hipSuccess ||
            hipMemsetAsync(gf, 0, (size_t)B * N * ldgf * sizeof(float), (hipStream_t)stream) != hipSuccess)
            return GPE_ELAUNCH;
        hipLaunchKernelGGL(gpe_attn_pool_bwd_max_kernel, dim3(gpe_cdiv(P + C, 64), B), dim3(64), 0, (hipStream_t)stream, w,
                           ldw, feat, ldf, g, arg, N, P, C, gw, ldgw, gf, ldgf);
    } else {
        const int Pp = gpe_round_up(P, 4), Cp = gpe_round_up(C, 4);
        const size_t lds = ((size_t)Pp * Cp + (size_t)APB_ROWS * (Pp + Cp)) * sizeof(float);
        if (lds > 150 * 1024) return GPE_EINVAL;
        GPE_ENSURE_MAX_LDS_N((gpe_attn_pool_bwd_kernel), 150 * 1024);
        hipLaunchKernelGGL(gpe_attn_pool_bwd_kernel, dim3(gpe_cdiv(N, APB_ROWS), B), dim3(256), lds, (hipStream_t)stream, w, ldw,
                           feat, ldf, g, N, P, C, mode == 0 ? 1.f / N : 1.f, gw, ldgw, gf, ldgf);
    }
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// =====================================================================================================================
// global max / add pooling over equal-sized clouds (mean lives in gpe_pointwise.hip)
// =====================================================================================================================
__global__ __launch_bounds__(1024) void gpe_segment_pool_fwd_kernel(const float* __restrict__ x, int ldx, int N, int C,
                                                                    int mode, float* __restrict__ y, int ldy,
                                                                    int32_t* __restrict__ arg)
{
    __shared__ float rv[16][64];
    __shared__ int ri[16][64];
    __shared__ double rs[16][64];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    float best = -INFINITY;
    int ba = 0;
    double s = 0;
    if (c < C) {
        for (int n = wave; n < N; n += 16) {
            const float v = x[((long)b * N + n) * ldx + c];
            if (mode == 1) { if (v > best) { best = v; ba = n; } }
            else s += (double)v;
        }
    }
    rv[wave][lane] = best; ri[wave][lane] = ba; rs[wave][lane] = s;
    __syncthreads();
    if (wave != 0 || c >= C) return;
    if (mode == 1) {
        for (int w_ = 1; w_ < 16; ++w_) {
            const float v = rv[w_][lane];
            const int a = ri[w_][lane];
            if (v > best || (v == best && a < ba)) { best = v; ba = a; }      // first maximum
        }
        y[(long)b * ldy + c] = best;
        arg[(long)b * C + c] = ba;
    } else {
        double t = 0;
        for (int w_ = 0; w_ < 16; ++w_) t += rs[w_][lane];
        y[(long)b * ldy + c] = (float)t;
    }
}

__global__ void gpe_segment_pool_bwd_kernel(const float* __restrict__ gy, int ldgy, const int32_t* __restrict__ arg, int N,
                                            int C, int mode, long rows, float* __restrict__ gx, int ldgx)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * C) return;
    const long r = e / C;
    const int c = (int)(e - r * C);
    const long b = r / N;
    const int n = (int)(r - b * N);
    const float g = gy[b * ldgy + c];
    gx[r * ldgx + c] = (mode == 1) ? ((arg[b * C + c] == n) ? g : 0.f) : g;
}

extern "C" int gpe_segment_pool_fwd(const float* x, int ldx, int B, int N, int C, int mode, float* y, int ldy,
                                    int32_t* arg, void* stream)
{
    if (!x || !y || B <= 0 || N <= 0 || C <= 0 || ldx < C || ldy < C || (mode != 1 && mode != 2) || (mode == 1 && !arg))
        return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_segment_pool_fwd_kernel, dim3(B, gpe_cdiv(C, 64)), dim3(1024), 0, (hipStream_t)stream, x, ldx, N,
                       C, mode, y, ldy, arg);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

extern "C" int gpe_segment_pool_bwd(const float* gy, int ldgy, const int32_t* arg, int B, int N, int C, int mode,
                                    float* gx, int ldgx, void* stream)
{
    if (!gy || !gx || B <= 0 || N <= 0 || C <= 0 || (mode != 1 && mode != 2) || (mode == 1 && !arg)) return GPE_EINVAL;
    const long rows = (long)B * N;
    hipLaunchKernelGGL(gpe_segment_pool_bwd_kernel, dim3(gpe_cdiv(rows * C, 256)), dim3(256), 0, (hipStream_t)stream, gy, ldgy,
                       arg, N, C, mode, rows, gx, ldgx);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// sum over the k messages of every point: out[i][c] = sum_s a[(i*k+s)][c]   (EdgeConv aggr 'add' / 'mean')
__global__ __launch_bounds__(256) void gpe_edge_sum_k_kernel(const float* __restrict__ a, int lda, long npts, int k, int F,
                                                             float* __restrict__ out, int ldo)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = blockIdx.y * 64 + lane;
    if (c0 >= F) return;
    const long nw = (long)gridDim.x * 4;
    for (long i = (long)blockIdx.x * 4 + wave; i < npts; i += nw) {
        float s = 0.f;
        for (int s_ = 0; s_ < k; ++s_) s += a[(i * k + s_) * lda + c0];
        out[i * ldo + c0] = s;
    }
}

extern "C" int gpe_edge_sum_k(const float* a, int lda, long npts, int k, int F, float* out, int ldo, void* stream)
{
    if (!a || !out || npts <= 0 || k <= 0 || F <= 0 || lda < F || ldo < F) return GPE_EINVAL;
    const int bx = (int)((npts + 3) / 4 < 2048 ? (npts + 3) / 4 : 2048);
    hipLaunchKernelGGL(gpe_edge_sum_k_kernel, dim3(bx, gpe_cdiv(F, 64)), dim3(256), 0, (hipStream_t)stream, a, lda, npts, k,
                       F, out, ldo);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// =====================================================================================================================
// Adam over a flat arena (torch.optim.Adam semantics, amsgrad off, maximize off):
//   g' = g*gscale + wd*p ; m = b1 m + (1-b1) g' ; v = b2 v + (1-b2) g'^2 ;
//   p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps)            bc1 = 1-b1^t, bc2 = 1-b2^t
// zero_grad != 0 also clears g (the next backward writes or accumulates into it).
// =====================================================================================================================
__global__ __launch_bounds__(256) void gpe_adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long n, float lr_over_bc1, float b1,
                                                       float b2, float eps, float wd, float rsqrt_bc2, float gscale,
                                                       int zero_grad, const float* __restrict__ hyper)
{
    const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    // gpe_adam_step_dev: the two step-dependent scalars come from device memory (a captured launch is replayed with new values)
    if (hyper) { lr_over_bc1 = hyper[0]; rsqrt_bc2 = hyper[1]; }
    if (i4 + 3 < n) {
        float4 pv = *reinterpret_cast<float4*>(p + i4), gv = *reinterpret_cast<float4*>(g + i4);
        float4 mv = *reinterpret_cast<float4*>(m + i4), vv = *reinterpret_cast<float4*>(v + i4);
        float* pp = &pv.x; float* gg = &gv.x; float* mm = &mv.x; float* vq = &vv.x;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float gr = gg[t] * gscale + wd * pp[t];
            mm[t] = b1 * mm[t] + (1.f - b1) * gr;
            vq[t] = b2 * vq[t] + (1.f - b2) * gr * gr;
            pp[t] -= lr_over_bc1 * mm[t] / (sqrtf(vq[t]) * rsqrt_bc2 + eps);
        }
        *reinterpret_cast<float4*>(p + i4) = pv;
        *reinterpret_cast<float4*>(m + i4) = mv;
        *reinterpret_cast<float4*>(v + i4) = vv;
        if (zero_grad) *reinterpret_cast<float4*>(g + i4) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        for (long i = i4; i < n; ++i) {
            const float gr = g[i] * gscale + wd * p[i];
            m[i] = b1 * m[i] + (1.f - b1) * gr;
            v[i] = b2 * v[i] + (1.f - b2) * gr * gr;
            p[i] -= lr_over_bc1 * m[i] / (sqrtf(v[i]) * rsqrt_bc2 + eps);
            if (zero_grad) g[i] = 0.f;
        }
    }
}

extern "C" int gpe_adam_step(float* p, float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, long step, float gscale, int zero_grad, void* stream)
{
    if (!p || !g || !m || !v || n <= 0 || step <= 0 || (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15))
        return GPE_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float lr_over_bc1 = (float)((double)lr / bc1);
    const float rsqrt_bc2 = (float)(1.0 / sqrt(bc2));
    hipLaunchKernelGGL(gpe_adam_kernel, dim3(gpe_cdiv(gpe_cdiv(n, 4), 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                       lr_over_bc1, beta1, beta2, eps, weight_decay, rsqrt_bc2, gscale, zero_grad, (const float*)nullptr);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// the same step with its two step-dependent scalars read from DEVICE memory: hyper[0] = lr / (1 - beta1^step),
// hyper[1] = 1 / sqrt(1 - beta2^step) (gpe_adam_hyper computes them on the host).  For steps replayed from a captured hipGraph,
// where kernel arguments are frozen (gpe_amd/graph.py).
extern "C" int gpe_adam_step_dev(float* p, float* g, float* m, float* v, long n, const float* hyper, float beta1, float beta2,
                                 float eps, float weight_decay, float gscale, int zero_grad, void* stream)
{
    if (!p || !g || !m || !v || !hyper || n <= 0 || (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) || (((uintptr_t)hyper) & 3))
        return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_adam_kernel, dim3(gpe_cdiv(gpe_cdiv(n, 4), 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                       0.f, beta1, beta2, eps, weight_decay, 1.f, gscale, zero_grad, hyper);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

extern "C" int gpe_adam_hyper(float lr, float beta1, float beta2, long step, float* out_host)
{
    if (!out_host || step <= 0) return GPE_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    out_host[0] = (float)((double)lr / bc1);
    out_host[1] = (float)(1.0 / sqrt(bc2));
    return GPE_OK;
}

// =====================================================================================================================
// input standardisation: out[r][c] = (x[r][c] - shift[c]) / scale[c]    (nn/data/transforms.py:35-50), C <= 8
// =====================================================================================================================
__global__ void gpe_standardize_kernel(const float* __restrict__ x, long n, int C, float s0, float s1, float s2, float s3,
                                       float s4, float s5, float s6, float s7, float d0, float d1, float d2, float d3,
                                       float d4, float d5, float d6, float d7, float* __restrict__ out)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float sh[8] = {s0, s1, s2, s3, s4, s5, s6, s7}, sc[8] = {d0, d1, d2, d3, d4, d5, d6, d7};
    const int c = (int)(e % C);
    out[e] = (x[e] - sh[c]) / sc[c];
}

extern "C" int gpe_standardize(const float* x, long rows, int C, const float* shift_host, const float* scale_host,
                               float* out, void* stream)
{
    if (!x || !out || !shift_host || !scale_host || rows <= 0 || C <= 0 || C > 8) return GPE_EINVAL;
    float s[8] = {0}, d[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    for (int c = 0; c < C; ++c) { s[c] = shift_host[c]; d[c] = scale_host[c]; }
    const long n = rows * C;
    hipLaunchKernelGGL(gpe_standardize_kernel, dim3(gpe_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, n, C, s[0], s[1],
                       s[2], s[3], s[4], s[5], s[6], s[7], d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], out);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}
