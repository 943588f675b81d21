// The stitch terms of ComposedPatternLoss, active from `epoch_with_stitches` on, as gfx950 kernels:
//   * PatternStitchLoss — similarity + negative (all-pairs triplet or HardNet) term     /root/reference/nn/metrics/losses.py:54-180
//   * supervised stitch tags (MSE) and free-edge classification (BCE with logits)       nn/metrics/composed_loss.py:336-362
//   * the ground-truth pre-processing they need: stitched-edge re-numbering after the panel-order permutation and the
//     panel-origin shift, and the per-panel shift of per-edge ground truth            nn/metrics/composed_loss.py:592-620,705-755
// One workgroup per pattern (<= 2 x 64 stitch tags, 23 x 14 edges): latency-bound small work, value and gradient in one
// launch each, fp64 reductions in a fixed order, no float atomics (duplicate edge references are accumulated serially).
#include "gpe_common.h"
#include <math.h>

#define ST_TPB 256
#define ST_MAXS 64          // stitches per pattern (the shipped data: max_num_stitches = 24)
#define ST_MAXD 8           // stitch tag dimension (shipped: 3)

// flags: 1 stitch (similarity + negative) | 2 HardNet negative | 4 free-edge BCE | 8 supervised tags
struct StitchParams {
    const float* tags; long t_sb, t_sp, t_sl; int D;            // predicted tags (b,p,l,d) at tags + b*t_sb + p*t_sp + l*t_sl + d
    const float* logit; long m_sb, m_sp, m_sl;                  // predicted free-edge logits (b,p,l)
    const int64_t* stitches; const int64_t* nums; int S;        // [B][2][S] pattern-level edge ids, [B]
    const float* gt_mask; const float* gt_tags;                 // dense [B,P,L], [B,P,L,D]
    int B, P, L, flags;
    float margin, sup_w;
};

__device__ __forceinline__ int st_count(const StitchParams& p, int b)
{
    long n = p.nums[b];
    return (int)(n < 0 ? 0 : (n > p.S ? p.S : n));
}

// stage the 2n tags of pattern b: tag t < n = left side of stitch t, tag n + t = its right side
__device__ __forceinline__ void st_load_tags(const StitchParams& p, int b, int n, float* T, int* edge_of)
{
    const int PL = p.P * p.L;
    for (int idx = threadIdx.x; idx < 2 * n; idx += ST_TPB) {
        const int side = idx >= n, i = idx - side * n;
        long e = p.stitches[((size_t)b * 2 + side) * p.S + i];
        e = e < 0 ? 0 : (e >= PL ? PL - 1 : e);                  // the reference would raise an IndexError; stay in bounds
        edge_of[idx] = (int)e;
        const int pp = (int)e / p.L, l = (int)e - pp * p.L;
        for (int d = 0; d < p.D; ++d) T[idx * ST_MAXD + d] = p.tags[b * p.t_sb + pp * p.t_sp + l * p.t_sl + d];
    }
}

__device__ __forceinline__ float st_dist(const float* T, int i, int j, int D)
{
    float s = 0.f;
    for (int d = 0; d < D; ++d) { const float v = T[i * ST_MAXD + d] - T[j * ST_MAXD + d]; s += v * v; }
    return s;
}

// part[b][6] = {similarity_b (already / n_b), sum over the pattern's tags of the per-tag negative term, tag count 2 n_b,
//               BCE sum, supervised squared-error sum, 0}
__global__ __launch_bounds__(ST_TPB) void gpe_stitch_fwd_kernel(StitchParams p, double* __restrict__ part)
{
    __shared__ float T[2 * ST_MAXS * ST_MAXD];
    __shared__ int edge_of[2 * ST_MAXS];
    __shared__ double red[4][ST_TPB];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = st_count(p, b);
    double s_sim = 0, s_neg = 0, s_bce = 0, s_sup = 0;
    if (p.flags & 1) {
        st_load_tags(p, b, n, T, edge_of);
        __syncthreads();
        if (tid < n) s_sim = (double)st_dist(T, tid, tid + n, p.D);
        if (tid < 2 * n) {
            const int bro = tid < n ? tid + n : tid - n;
            if (p.flags & 2) {                                   // HardNet: only the closest other tag counts
                float m = INFINITY;
                for (int j = 0; j < 2 * n; ++j)
                    if (j != tid && j != bro) { const float d = st_dist(T, tid, j, p.D); m = d < m ? d : m; }
                const float gap = p.margin - m;
                s_neg = gap > 0.f ? (double)gap : 0.0;
            } else {                                             // every other tag closer than the margin counts, / (2n)
                double a = 0;
                for (int j = 0; j < 2 * n; ++j)
                    if (j != tid && j != bro) { const float gap = p.margin - st_dist(T, tid, j, p.D); if (gap > 0.f) a += gap; }
                s_neg = a / (2.0 * n);
            }
        }
    }
    const int PL = p.P * p.L;
    if (p.flags & 4) {
        for (int e = tid; e < PL; e += ST_TPB) {
            const int pp = e / p.L, l = e - pp * p.L;
            const double x = p.logit[b * p.m_sb + pp * p.m_sp + l * p.m_sl], y = p.gt_mask[(size_t)b * PL + e];
            s_bce += (x > 0 ? x : 0) - x * y + log1p(exp(-fabs(x)));
        }
    }
    if (p.flags & 8) {
        for (int e = tid; e < PL * p.D; e += ST_TPB) {
            const int d = e % p.D, q = e / p.D, pp = q / p.L, l = q - pp * p.L;
            const float v = p.tags[b * p.t_sb + pp * p.t_sp + l * p.t_sl + d] - p.gt_tags[(size_t)b * PL * p.D + e];
            s_sup += (double)v * v;
        }
    }
    red[0][tid] = s_sim; red[1][tid] = s_neg; red[2][tid] = s_bce; red[3][tid] = s_sup;
    __syncthreads();
    for (int st = ST_TPB / 2; st > 0; st >>= 1) {
        if (tid < st)
#pragma unroll
            for (int q = 0; q < 4; ++q) red[q][tid] += red[q][tid + st];
        __syncthreads();
    }
    if (tid == 0) {
        double* o = part + (size_t)b * 6;
        o[0] = red[0][0] / (double)n;                            // n = 0: 0/0 = NaN, like the reference's sum-of-nothing / 0
        o[1] = red[1][0];
        o[2] = 2.0 * n;
        o[3] = red[2][0];
        o[4] = red[3][0];
        o[5] = 0;
    }
}

// out[0] = (similarity + negative) + sup_w * supervised + free  (the reference's association), out[1..4] = the four terms
__global__ void gpe_stitch_final_kernel(const double* __restrict__ part, StitchParams p, float* __restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < p.B; ++b)
#pragma unroll
        for (int q = 0; q < 5; ++q) s[q] += part[(size_t)b * 6 + q];
    const double npl = (double)p.B * p.P * p.L;
    const double sim = s[0] / p.B, neg = s[1] / s[2], bce = s[3] / npl, sup = s[4] / (npl * p.D);
    double total = 0;
    if (p.flags & 1) total += sim + neg;
    if (p.flags & 8) total += (double)p.sup_w * sup;
    if (p.flags & 4) total += bce;
    out[0] = (float)total;
    out[1] = (p.flags & 1) ? (float)sim : 0.f;
    out[2] = (p.flags & 1) ? (float)neg : 0.f;
    out[3] = (p.flags & 8) ? (float)sup : 0.f;
    out[4] = (p.flags & 4) ? (float)bce : 0.f;
}

// gradient of out[0] w.r.t. the predicted tags (dense g_tags [B,P,L,D]) and free-edge logits (dense g_mask [B,P,L]),
// times the device scalar *gscale
__global__ __launch_bounds__(ST_TPB) void gpe_stitch_bwd_kernel(StitchParams p, const double* __restrict__ part,
                                                                const float* __restrict__ gscale,
                                                                float* __restrict__ g_tags, float* __restrict__ g_mask)
{
    __shared__ float T[2 * ST_MAXS * ST_MAXD];
    __shared__ float G[2 * ST_MAXS * ST_MAXD];
    __shared__ int edge_of[2 * ST_MAXS];
    __shared__ int jstar[2 * ST_MAXS];          // HardNet: closest other tag of an ACTIVE tag, else -1
    __shared__ float wstar[2 * ST_MAXS];        // ... and the share of the gradient each tied minimum receives
    const int b = blockIdx.x, tid = threadIdx.x;
    const float gs = gscale ? gscale[0] : 1.f;
    const int n = st_count(p, b), PL = p.P * p.L;
    const double npl = (double)p.B * PL;
    // dense outputs of this pattern: the supervised term touches every element, everything else starts from zero
    if (g_tags) {
        for (int e = tid; e < PL * p.D; e += ST_TPB) {
            float v = 0.f;
            if (p.flags & 8) {
                const int d = e % p.D, q = e / p.D, pp = q / p.L, l = q - pp * p.L;
                const float df = p.tags[b * p.t_sb + pp * p.t_sp + l * p.t_sl + d] - p.gt_tags[(size_t)b * PL * p.D + e];
                v = (float)(2.0 * df * p.sup_w / (npl * p.D)) * gs;
            }
            g_tags[(size_t)b * PL * p.D + e] = v;
        }
    }
    if (g_mask) {
        for (int e = tid; e < PL; e += ST_TPB) {
            float v = 0.f;
            if (p.flags & 4) {
                const int pp = e / p.L, l = e - pp * p.L;
                const double x = p.logit[b * p.m_sb + pp * p.m_sp + l * p.m_sl], y = p.gt_mask[(size_t)b * PL + e];
                v = (float)((1.0 / (1.0 + exp(-x)) - y) / npl) * gs;
            }
            g_mask[(size_t)b * PL + e] = v;
        }
    }
    if (!(p.flags & 1) || !g_tags || n == 0) return;
    double ntot = 0;
    for (int q = 0; q < p.B; ++q) ntot += part[(size_t)q * 6 + 2];
    st_load_tags(p, b, n, T, edge_of);
    __syncthreads();
    if (tid < 2 * n) {
        const int bro = tid < n ? tid + n : tid - n;
        float g[ST_MAXD];
        // similarity: d/dt_left = 2 (t_left - t_right) / (n B), the right side with the opposite sign
        const float csim = (float)(2.0 / ((double)n * p.B));
        for (int d = 0; d < p.D; ++d) g[d] = csim * (T[tid * ST_MAXD + d] - T[bro * ST_MAXD + d]);
        jstar[tid] = -1;
        wstar[tid] = 0.f;
        if (p.flags & 2) {
            float m = INFINITY;
            int ties = 0;
            for (int j = 0; j < 2 * n; ++j)
                if (j != tid && j != bro) {
                    const float d = st_dist(T, tid, j, p.D);
                    if (d < m) { m = d; ties = 1; jstar[tid] = j; } else if (d == m) ++ties;
                }
            if (!(p.margin - m > 0.f)) jstar[tid] = -1;          // inactive (or no other tag at all)
            else wstar[tid] = 1.f / (float)ties;                 // torch's min() backward shares the gradient among ties
            if (jstar[tid] >= 0) {
                const float c = (float)(-2.0 / ntot) * wstar[tid];
                for (int j = 0; j < 2 * n; ++j)
                    if (j != tid && j != bro && st_dist(T, tid, j, p.D) == m)
                        for (int d = 0; d < p.D; ++d) g[d] += c * (T[tid * ST_MAXD + d] - T[j * ST_MAXD + d]);
                jstar[tid] = __float_as_int(m);                  // keep the minimum itself for the gather pass below
            }
        } else {
            // pair (i,j) appears in tag i's sum and in tag j's: -4 (t_i - t_j) / (2n ntot) for every active pair
            const float c = (float)(-4.0 / (2.0 * n * ntot));
            for (int j = 0; j < 2 * n; ++j)
                if (j != tid && j != bro && p.margin - st_dist(T, tid, j, p.D) > 0.f)
                    for (int d = 0; d < p.D; ++d) g[d] += c * (T[tid * ST_MAXD + d] - T[j * ST_MAXD + d]);
        }
        for (int d = 0; d < p.D; ++d) G[tid * ST_MAXD + d] = g[d];
    }
    __syncthreads();
    if ((p.flags & 2) && tid < 2 * n) {
        // HardNet, the other end: tag `tid` is (one of) the closest tag(s) of every active i with d(i, tid) == min_i
        const int bro = tid < n ? tid + n : tid - n;
        float g[ST_MAXD];
        for (int d = 0; d < p.D; ++d) g[d] = 0.f;
        for (int i = 0; i < 2 * n; ++i) {
            if (i == tid || wstar[i] == 0.f) continue;
            const int bi = i < n ? i + n : i - n;
            if (tid == bi) continue;
            if (st_dist(T, i, tid, p.D) == __int_as_float(jstar[i])) {
                const float c = (float)(2.0 / ntot) * wstar[i];
                for (int d = 0; d < p.D; ++d) g[d] += c * (T[i * ST_MAXD + d] - T[tid * ST_MAXD + d]);
            }
        }
        (void)bro;
        for (int d = 0; d < p.D; ++d) G[tid * ST_MAXD + d] += g[d];
    }
    __syncthreads();                            // also orders the dense initialisation above against the scatter below
    if (tid == 0) {
        // one thread, tag order: an edge referenced by several stitches accumulates deterministically
        for (int t = 0; t < 2 * n; ++t)
            for (int d = 0; d < p.D; ++d)
                g_tags[((size_t)b * PL + edge_of[t]) * p.D + d] += G[t * ST_MAXD + d] * gs;
    }
}

static int st_check(const StitchParams& p, const double* part)
{
    if (p.B <= 0 || p.P <= 0 || p.L <= 0 || !part) return GPE_EINVAL;
    if ((p.flags & 1) && (!p.tags || !p.stitches || !p.nums || p.S <= 0 || p.S > ST_MAXS || p.D <= 0 || p.D > ST_MAXD))
        return GPE_EINVAL;
    if ((p.flags & 4) && (!p.logit || !p.gt_mask)) return GPE_EINVAL;
    if ((p.flags & 8) && (!p.tags || !p.gt_tags || p.D <= 0 || p.D > ST_MAXD)) return GPE_EINVAL;
    return GPE_OK;
}

extern "C" int gpe_stitch_loss_fwd(const float* tags, long t_sb, long t_sp, long t_sl, int D, const float* logit, long m_sb,
                                   long m_sp, long m_sl, const int64_t* stitches, const int64_t* nums, int S,
                                   const float* gt_mask, const float* gt_tags, int B, int P, int L, int flags, float margin,
                                   float sup_w, double* part, float* out5, void* stream)
{
    StitchParams p{tags, t_sb, t_sp, t_sl, D, logit, m_sb, m_sp, m_sl, stitches, nums, S, gt_mask, gt_tags, B, P, L, flags,
                   margin, sup_w};
    if (st_check(p, part) != GPE_OK || !out5) return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_stitch_fwd_kernel, dim3(B), dim3(ST_TPB), 0, (hipStream_t)stream, p, part);
    GPE_CHECK_LAUNCH();
    hipLaunchKernelGGL(gpe_stitch_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, part, p, out5);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

extern "C" int gpe_stitch_loss_bwd(const float* tags, long t_sb, long t_sp, long t_sl, int D, const float* logit, long m_sb,
                                   long m_sp, long m_sl, const int64_t* stitches, const int64_t* nums, int S,
                                   const float* gt_mask, const float* gt_tags, int B, int P, int L, int flags, float margin,
                                   float sup_w, const double* part, const float* gscale, float* g_tags, float* g_mask,
                                   void* stream)
{
    StitchParams p{tags, t_sb, t_sp, t_sl, D, logit, m_sb, m_sp, m_sl, stitches, nums, S, gt_mask, gt_tags, B, P, L, flags,
                   margin, sup_w};
    if (st_check(p, part) != GPE_OK) return GPE_EINVAL;
    if (((flags & 9) && !g_tags) || ((flags & 4) && !g_mask)) return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_stitch_bwd_kernel, dim3(B), dim3(ST_TPB), 0, (hipStream_t)stream, p, part, gscale, g_tags, g_mask);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// ground-truth pre-processing of the stitch terms
// ---------------------------------------------------------------------------------------------------------------------
// out[b][side][i] (i < nums[b]) = the pattern-level edge id after (1) the panel-order permutation: panel q moves to the LAST
// slot r with perm[b][r] == q (composed_loss.py:592-620; no such slot: -1, as the reference's list initialiser), then (2) the
// origin shift of that panel: in-panel id e -> e - lead if e >= lead else num_edges - (lead - e) (:727-755; lead / num_edges
// are indexed by the panel's slot AFTER step 1).  Entries i >= nums[b] are copied.  perm == NULL / lead == NULL skip a step.
__global__ void gpe_stitch_renumber_kernel(const int64_t* __restrict__ st, const int64_t* __restrict__ nums, int S, int P,
                                           int L, const int64_t* __restrict__ perm, const int32_t* __restrict__ lead,
                                           const int32_t* __restrict__ num_edges, int64_t* __restrict__ out)
{
    const int b = blockIdx.x;
    long n = nums[b];
    n = n < 0 ? 0 : (n > S ? S : n);
    for (int idx = threadIdx.x; idx < 2 * S; idx += blockDim.x) {
        const int i = idx % S;
        long e = st[(size_t)b * 2 * S + idx];
        if (i < n) {
            if (perm) {
                const long panel = e / L, inner = e - panel * L;
                long slot = -1;
                for (int r = 0; r < P; ++r) if (perm[(size_t)b * P + r] == panel) slot = r;
                e = slot * L + inner;
            }
            if (lead) {
                // Python floor division for the (pathological) negative id of an unmatched panel
                long panel = e / L;
                if (e < 0 && panel * L != e) --panel;
                const long inner = e - panel * L;
                long g = (long)b * P + panel;
                g = g < 0 ? 0 : g;                               // the reference would wrap around; stay in bounds
                const long ld = lead[g], ne = num_edges[g];
                e = panel * L + (inner >= ld ? inner - ld : ne - (ld - inner));
            }
        }
        out[(size_t)b * 2 * S + idx] = e;
    }
}

extern "C" int gpe_stitch_renumber(const int64_t* stitches, const int64_t* nums, int B, int S, int P, int L,
                                   const int64_t* perm, const int32_t* lead, const int32_t* num_edges, int64_t* out,
                                   void* stream)
{
    if (!stitches || !nums || !out || B <= 0 || S <= 0 || P <= 0 || L <= 0 || (lead && !num_edges)) return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_stitch_renumber_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, stitches, nums, S, P, L, perm,
                       lead, num_edges, out);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// per-edge ground truth [npanels][L][D] follows its panel's new loop origin (composed_loss.py:705-725): rows l < n of a panel
// with n >= 3 edges and lead != 0 become feat[(l + lead) mod n]; padding rows and every other panel are copied.
__global__ void gpe_panel_shift_kernel(const float* __restrict__ feat, int D, const int32_t* __restrict__ lead,
                                       const int32_t* __restrict__ num_edges, long npanels, int L, float* __restrict__ out)
{
    const long total = npanels * L * D;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int d = (int)(e % D);
        const long q = e / D;
        const int l = (int)(q % L);
        const long pn = q / L;
        const int ld = lead[pn];
        int n = num_edges[pn];
        n = n > L ? L : n;
        int src = l;
        if (n >= 3 && ld != 0 && l < n) { src = l + ld; if (src >= n) src -= n; }
        out[e] = feat[(pn * L + src) * D + d];
    }
}

extern "C" int gpe_panel_shift(const float* feat, int D, const int32_t* lead, const int32_t* num_edges, long npanels, int L,
                               float* out, void* stream)
{
    if (!feat || !lead || !num_edges || !out || D <= 0 || npanels <= 0 || L <= 0) return GPE_EINVAL;
    const long total = npanels * L * D;
    hipLaunchKernelGGL(gpe_panel_shift_kernel, dim3((unsigned)gpe_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, feat, D,
                       lead, num_edges, npanels, L, out);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}
