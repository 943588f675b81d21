// "f16x3": the two-term fp16 policy of the split-precision single-role edge kernel (gpe_edgegemm_split_kernel.h) — three
// v_mfma_f32_16x16x32_f16 per fp32 product, operands normalised per TENSOR by a power of two so that fp16's 5-bit exponent
// is never the limit.  Two planes of a 200 x 200 weight take 184 of a wave's 256 accumulation registers, so — unlike the
// three-plane bf16 policy — every shape of the shipped edge MLPs (10 / 13 output tiles x 10 / 13 K chunks) fits.
//
// The largest magnitudes behind the scales are measured ON THE DEVICE, on the launch stream, and live in CALLER-OWNED words
// (include/gpe_hip.h "amax words"): the library keeps no record of tensors.
//   * packed weight: one-workgroup pass over its 16 KCH x Npad floats (a few microseconds) -> workspace word 1;
//   * dense A operand: the caller's `amax_a` word — filled by the entry point that wrote the tensor (the split kernels track the
//     largest magnitude they store, gpe_edge_dz3 likewise) — or, when the caller passes none, a streaming |x| max -> word 0;
//   * gathered A operand relu(P_i + Q_j): the caller's `amax_a` (gpe_edge_pq_amax) or the bound max(P) + max(Q) from one pass
//     over the per-point [P|Q] table -> word 0.
#include "gpe_edgegemm_split_kernel.h"

#define H3_PQ_BLOCKS GPE_WS_H3_PARTS

// largest |x| over rows x cols (row pitch ld), as the bit pattern of a non-negative float (orders like the float; NaN > inf)
__global__ __launch_bounds__(256) void gpe_h3_absmax_kernel(const float* __restrict__ x, long rows, int cols4, long ld,
                                                            unsigned* __restrict__ out)
{
    const long total = rows * cols4;
    unsigned m = 0u;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long r = t / cols4;
        const int q = (int)(t - r * cols4);
        const float4 v = ld4(x + r * ld + 4 * q);
        const unsigned a = __float_as_uint(v.x) & 0x7fffffffu, b = __float_as_uint(v.y) & 0x7fffffffu;
        const unsigned c = __float_as_uint(v.z) & 0x7fffffffu, d = __float_as_uint(v.w) & 0x7fffffffu;
        const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
        const unsigned e = ab > cd ? ab : cd;
        m = m > e ? m : e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)m, o);
        m = m > t ? m : t;
    }
    // one atomic per workgroup: thousands of same-address atomics issued at once serialise in the L2 (measured: 16 k of them
    // took longer than the pass itself)
    __shared__ unsigned red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned a = red[0] > red[1] ? red[0] : red[1], b = red[2] > red[3] ? red[2] : red[3];
        const unsigned r = a > b ? a : b;
        if (r) atomicMax(out, r);
    }
}

// max(P) and max(Q) over the [rows][>= 2H] table (signed maxima; max(P) + max(Q) bounds relu(P_i + Q_j)).  One wave per row
// at a time, lane = column quad, two rows in flight; every workgroup leaves ONE pair of partial maxima in part[2 b], part[2 b + 1]
// (no atomics: see gpe_h3_absmax_kernel), gpe_h3_pqfinish_kernel combines them.
__global__ __launch_bounds__(256) void gpe_h3_pqmax_kernel(const float* __restrict__ pq, long rows, int H, long ld,
                                                           float* __restrict__ part /* [2 * gridDim.x] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h4 = H >> 2;
    float mp = -INFINITY, mq = -INFINITY;
    const long stride = (long)gridDim.x * 4;
    for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += 2 * stride) {
        const float* row0 = pq + r * ld;
        const bool two = r + stride < rows;
        const float* row1 = pq + (two ? r + stride : r) * ld;
        for (int q = lane; q < 2 * h4; q += 64) {
            const float4 v0 = ld4(row0 + 4 * q), v1 = ld4(row1 + 4 * q);
            const float m = fmaxf(fmaxf(fmaxf(v0.x, v0.y), fmaxf(v0.z, v0.w)), fmaxf(fmaxf(v1.x, v1.y), fmaxf(v1.z, v1.w)));
            if (q < h4) mp = fmaxf(mp, m); else mq = fmaxf(mq, m);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mp = fmaxf(mp, __shfl_xor(mp, o)); mq = fmaxf(mq, __shfl_xor(mq, o)); }
    __shared__ float red[8];
    if (lane == 0) { red[2 * wave] = mp; red[2 * wave + 1] = mq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = fmaxf(fmaxf(red[0], red[2]), fmaxf(red[4], red[6]));
        part[2 * blockIdx.x + 1] = fmaxf(fmaxf(red[1], red[3]), fmaxf(red[5], red[7]));
    }
}

// one workgroup: the packed weight's largest magnitude -> slots[1]; clears slots[0] (the A-operand word the multi-workgroup
// pass accumulates into) and the caller's output word the coming launch will atomicMax into
__global__ __launch_bounds__(1024) void gpe_h3_wmax_kernel(const float* __restrict__ w, long n, unsigned* __restrict__ slots,
                                                           unsigned* __restrict__ clear_word)
{
    __shared__ unsigned red[16];
    unsigned m = 0u;
    for (long t = threadIdx.x; t < n; t += 1024) {
        const unsigned a = __float_as_uint(w[t]) & 0x7fffffffu;
        m = m > a ? m : a;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)m, o);
        m = m > t ? m : t;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) m = m > red[i] ? m : red[i];
        slots[1] = m;
        slots[0] = 0u;
        if (clear_word) clear_word[0] = 0u;
    }
}

// gather bound: slots[0] = bits of max(0, maxP + maxQ) (rounded up by a factor 1 + 1e-6: a bound, not a value) from the
// nblk partial pairs of gpe_h3_pqmax_kernel
__global__ __launch_bounds__(256) void gpe_h3_pqfinish_kernel(const float* __restrict__ part, int nblk, unsigned* __restrict__ slots)
{
    float mp = -INFINITY, mq = -INFINITY;
    for (int b = threadIdx.x; b < nblk; b += 256) { mp = fmaxf(mp, part[2 * b]); mq = fmaxf(mq, part[2 * b + 1]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mp = fmaxf(mp, __shfl_xor(mp, o)); mq = fmaxf(mq, __shfl_xor(mq, o)); }
    __shared__ float red[8];
    if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = mp; red[2 * (threadIdx.x >> 6) + 1] = mq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mp = fmaxf(fmaxf(red[0], red[2]), fmaxf(red[4], red[6]));
        mq = fmaxf(fmaxf(red[1], red[3]), fmaxf(red[5], red[7]));
        float b = (mp + mq) * 1.000001f;
        if (!(b > 0.f)) b = (b != b) ? b : 0.f;           // NaN stays NaN (sorts above everything), negative -> all-zero operand
        slots[0] = __float_as_uint(b) & 0x7fffffffu;
    }
}

// the two passes of the gather bound, on stream s -> out[0]
int gpe_h3_pq_passes(unsigned* out, float* part, const float* pq, long rows, int H, long ld, hipStream_t s)
{
    if (!out || !part || (H & 3)) return GPE_EINVAL;
    int gx = (int)((rows + 3) / 4);
    if (gx > H3_PQ_BLOCKS) gx = H3_PQ_BLOCKS;
    hipLaunchKernelGGL(gpe_h3_pqmax_kernel, dim3(gx), dim3(256), 0, s, pq, rows, H, ld, part);
    GPE_CHECK_LAUNCH();
    hipLaunchKernelGGL(gpe_h3_pqfinish_kernel, dim3(1), dim3(256), 0, s, part, gx, out);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}
int gpe_h3_absmax(unsigned* out, const float* x, long rows, int cols, long ld, hipStream_t s)
{
    if (!out || !x) return GPE_EINVAL;
    if (hipMemsetAsync(out, 0, sizeof(unsigned), s) != hipSuccess) return GPE_ELAUNCH;
    const int cols4 = (cols + 3) >> 2;
    const long total = rows * cols4;
    if (total <= 0) return GPE_OK;
    int gx = (int)((total + 255) / 256);
    const int cap = gpe_num_cus() * 8;
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(gpe_h3_absmax_kernel, dim3(gx), dim3(256), 0, s, x, rows, cols4, ld, out);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

int gpe_edgegemm_w8_dispatch(int amode, int emode, int NT, int KCH, const RgParams& p, int stats_nblk, hipStream_t s);   // gpe_edgegemm_w8.hip

template <int AMODE, int EMODE>
static int h3_dispatch(int NT, int KCH, const RgParams& p, int stats_nblk, hipStream_t s)
{
    if (NT == 13 && KCH == 13) return x6_launch<SplitF16x2, 3, 1, 13, AMODE, EMODE>(p, stats_nblk, s);
    if (NT == 13 && KCH == 10) return x6_launch<SplitF16x2, 3, 1, 10, AMODE, EMODE>(p, stats_nblk, s);
    if (NT == 10 && KCH == 13) return x6_launch<SplitF16x2, 2, 2, 13, AMODE, EMODE>(p, stats_nblk, s);
    if (NT == 10 && KCH == 10) return x6_launch<SplitF16x2, 2, 2, 10, AMODE, EMODE>(p, stats_nblk, s);
    return GPE_ENOTSUP_SHAPE;
}

// Returns 1 and launches when the shape is on this kernel's menu, 0 when the caller should try the next kernel,
// < 0 on a launch error.  Needs the f16x3 words of the caller's workspace; without them (or below gpe_h3_min_rows() rows, where the
// scale passes cost more than the kernel saves) the exact kernels run.
int gpe_edgegemm_h3_try(const RgParams& p_in, int amode, int emode, int stats_nblk, hipStream_t s)
{
    unsigned* slots = p_in.ws.h3;
    if (!slots || p_in.M < gpe_h3_min_rows()) return 0;
    RgParams p;
    GpeFold fold;
    if (!x6_prepare(p_in, amode, emode, stats_nblk, p, fold)) return 0;
    const long npts = p_in.k > 0 ? p_in.M / p_in.k : 0;   // rows of the per-point table (p.k may now count pseudo-point rows)
    const int NT = (p.N <= 160) ? 10 : 13;
    const int KCH = (p.K <= 160) ? 10 : 13;
    if (amode == A_GATHER && !p.user_amax_a && (p.H & 3)) return 0;

    // the largest magnitude this launch writes to `out` (forward activations, in-place dz: tensors the next edge GEMM reads as
    // its A operand) goes to the caller's word; the gathered backward writes dz of block 0, which no GEMM of this library reads
    const bool tracks = emode == E_EDGE_FWD || emode == E_BWD_INPLACE;
    const bool reuse = (p.dbg & 128) != 0;                // profiling only: keep the scales of the previous launch (no passes)
    const long wn = (long)16 * KCH * p.Npad;              // the packed weight: 4 KCH k-quads x Npad columns x 4 (gpe_packed_size)
    // (w_ready: the caller's gpe_pack_fold launch left the weight's amax in slots[1], cleared slots[0] and the output word; only
    // honoured when no pass of this call accumulates into slots[0] afterwards — i.e. the A operand's word is the caller's)
    const bool w_ready = p_in.w_ready && p.user_amax_a;
    if (!reuse && !w_ready) {
        hipLaunchKernelGGL(gpe_h3_wmax_kernel, dim3(1), dim3(1024), 0, s, p.wp, wn, slots, tracks ? p.user_amax_out : nullptr);
        GPE_CHECK_LAUNCH();
    }
    if (reuse) {
        p.h3_amax_a = slots;
    } else if (p.user_amax_a) {
        p.h3_amax_a = p.user_amax_a;
    } else if (amode == A_DENSE) {
        const int cols4 = (p.K + 3) >> 2;
        const long total = p.M * cols4;
        int gx = (int)((total + 255) / 256);
        const int cap = gpe_num_cus() * 8;
        if (gx > cap) gx = cap;
        hipLaunchKernelGGL(gpe_h3_absmax_kernel, dim3(gx), dim3(256), 0, s, p.a.base, p.M, cols4, (long)p.a.stride_outer, slots);
        GPE_CHECK_LAUNCH();
        p.h3_amax_a = slots;
    } else {
        // the per-point table behind the gathered operand
        const int rc_pq = gpe_h3_pq_passes(slots, reinterpret_cast<float*>(slots + 2), p.pq, npts, p.H, (long)p.ldpq, s);
        if (rc_pq != GPE_OK) return rc_pq;
        p.h3_amax_a = slots;
    }
    p.h3_amax_w = slots + 1;
    p.amax_out = tracks ? p.user_amax_out : nullptr;

    // k = 16 at the shipped widths: two waves per SIMD (gpe_edgegemm_w8.hip); everything else the single-role kernel
    int rc = gpe_edgegemm_w8_dispatch(amode, emode, NT, KCH, p, stats_nblk, s);
    if (rc != GPE_ENOTSUP_SHAPE) { /* launched (or refused with an error) */ }
    else if (amode == A_GATHER && emode == E_EDGE_FWD) rc = h3_dispatch<A_GATHER, E_EDGE_FWD>(NT, KCH, p, stats_nblk, s);
    else if (amode == A_DENSE && emode == E_EDGE_FWD) rc = h3_dispatch<A_DENSE, E_EDGE_FWD>(NT, KCH, p, stats_nblk, s);
    else if (amode == A_DENSE && emode == E_BWD_INPLACE) rc = h3_dispatch<A_DENSE, E_BWD_INPLACE>(NT, KCH, p, stats_nblk, s);
    else if (amode == A_DENSE && emode == E_BWD_GATHER) rc = h3_dispatch<A_DENSE, E_BWD_GATHER>(NT, KCH, p, stats_nblk, s);
    if (rc == GPE_ENOTSUP_SHAPE) return 0;
    if (rc == GPE_OK) rc = gpe_edge_pseudo_fold(p, fold, s);
    if (rc != GPE_OK) return rc;
    if (tracks && p_in.user_amax_out && p_in.tracked) *p_in.tracked = 1;
    return 1;
}
