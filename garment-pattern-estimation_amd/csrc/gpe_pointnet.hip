// PointNet++ set abstraction for gfx950 (PointNetPlusPlus, /root/reference/nn/net_blocks.py:10-88: torch_geometric
// fps + radius + PointConv, then a global PointNet layer).  Third-party arithmetic, absent from /root/reference; restated
// from the published operators WITH PyG's conventions (oracle/ref_path.py: fps / fps_start / radius / pointconv_edges):
//   farthest point sampling  starts at a caller-given point per cloud (PyG: random_start=True — the host draws it from torch's
//                            generator, ops.fps); distances = fp32 fma chain over x,y,z of (a-b)^2, argmax ties -> lower index
//                            (the two choices upstream leaves to the implementation);
//   ball query               neighbours of a centroid = the first `maxn` points of its cloud in ascending index order with
//                            squared distance <= r^2;
//   PointConv edge list      PyG's PointNetConv default add_self_loops=True: remove_self_loops drops the edge whose source index
//                            (flat point number) EQUALS its target index (flat centroid number) — different index spaces in this
//                            bipartite call, PyG compares them anyway — and add_self_loops appends i -> i for i < B*M: centroid
//                            i also hears from flat point i, whichever cloud that point is in.
// All of it is small integer / latency-bound work next to the two dense MLPs (ops.DenseMLPFn).
#include "gpe_common.h"
#include <math.h>

__device__ __forceinline__ float pn_sqdist(const float* a, const float* b, int C)
{
    float acc = 0.f;
    for (int c = 0; c < C; ++c) { const float d = a[c] - b[c]; acc = __builtin_fmaf(d, d, acc); }
    return acc;
}

// ---------------------------------------------------------------------------------------------------------------------
// farthest point sampling: one workgroup per cloud, the running min-distance of every point in registers
// ---------------------------------------------------------------------------------------------------------------------
#define FPS_T 1024
#define FPS_PER 16                         // points per thread: N <= 16384
__global__ __launch_bounds__(FPS_T) void gpe_fps_kernel(const float* __restrict__ pos, int ldp, int N, int C, int M,
                                                        const int32_t* __restrict__ start, int32_t* __restrict__ out)
{
    __shared__ float rv[FPS_T / 64];
    __shared__ int ri[FPS_T / 64];
    __shared__ float cur[8];
    __shared__ int cur_i;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* P = pos + (size_t)b * N * ldp;
    float mind[FPS_PER];
#pragma unroll
    for (int q = 0; q < FPS_PER; ++q) mind[q] = INFINITY;
    int s0 = start ? start[b] : 0;
    s0 = s0 < 0 ? 0 : (s0 >= N ? N - 1 : s0);
    if (tid == 0) { cur_i = s0; out[(size_t)b * M] = s0; }
    if (tid < C) cur[tid] = P[(size_t)s0 * ldp + tid];
    __syncthreads();
    for (int m = 1; m < M; ++m) {
        float c_[8];
        for (int c = 0; c < C; ++c) c_[c] = cur[c];
        float best = -1.f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < FPS_PER; ++q) {
            const int i = tid + FPS_T * q;
            if (i < N) {
                const float d = pn_sqdist(P + (size_t)i * ldp, c_, C);
                mind[q] = fminf(mind[q], d);
                if (mind[q] > best) { best = mind[q]; bi = i; }        // ascending i inside a thread: first maximum
            }
        }
        // wave argmax (value desc, index asc), then across the 16 waves
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(best, off);
            const int oi = __shfl_xor(bi, off);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        __syncthreads();                   // everyone has read cur[]
        if (lane == 0) { rv[wave] = best; ri[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float bv = rv[0];
            int bb = ri[0];
            for (int w = 1; w < FPS_T / 64; ++w)
                if (rv[w] > bv || (rv[w] == bv && ri[w] < bb)) { bv = rv[w]; bb = ri[w]; }
            cur_i = bb;
            out[(size_t)b * M + m] = bb;
        }
        __syncthreads();
        if (tid < C) cur[tid] = P[(size_t)cur_i * ldp + tid];
        __syncthreads();
    }
}

extern "C" int gpe_fps(const float* pos, int ldp, int B, int N, int C, int M, const int32_t* start, int32_t* idx, void* stream)
{
    if (!pos || !idx || B <= 0 || N <= 0 || C <= 0 || C > 8 || ldp < C || M <= 0 || M > N || N > FPS_T * FPS_PER)
        return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_fps_kernel, dim3(B), dim3(FPS_T), 0, (hipStream_t)stream, pos, ldp, N, C, M, start, idx);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// ball query: one wave per centroid scans its cloud 64 points at a time; ballot + prefix popcount keep index order
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gpe_radius_kernel(const float* __restrict__ pos, int ldp, const int32_t* __restrict__ cidx,
                                                         int N, int C, int M, long total, float r2, int maxn,
                                                         int32_t* __restrict__ nbr, int32_t* __restrict__ cnt)
{
    const int lane = threadIdx.x & 63;
    const long s = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= total) return;
    const long b = s / M;
    const float* P = pos + (size_t)b * N * ldp;
    const int ci = cidx[s];
    float c_[8];
    for (int c = 0; c < C; ++c) c_[c] = P[(size_t)ci * ldp + c];
    int found = 0;
    for (int i0 = 0; i0 < N && found < maxn; i0 += 64) {
        const int i = i0 + lane;
        bool in = false;
        if (i < N) in = pn_sqdist(P + (size_t)i * ldp, c_, C) <= r2;
        const unsigned long long m = __ballot(in);
        if (in) {
            const int slot = found + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (slot < maxn) nbr[s * maxn + slot] = i;
        }
        found += __builtin_popcountll(m);
    }
    if (found > maxn) found = maxn;
    if (lane == 0) cnt[s] = found;
}

extern "C" int gpe_radius(const float* pos, int ldp, const int32_t* cidx, int B, int N, int C, int M, float r, int maxn,
                          int32_t* nbr, int32_t* cnt, void* stream)
{
    if (!pos || !cidx || !nbr || !cnt || B <= 0 || N <= 0 || M <= 0 || C <= 0 || C > 8 || ldp < C || maxn <= 0 || !(r >= 0))
        return GPE_EINVAL;
    const long total = (long)B * M;
    hipLaunchKernelGGL(gpe_radius_kernel, dim3(gpe_cdiv(total, 4)), dim3(256), 0, (hipStream_t)stream, pos, ldp, cidx, N, C, M,
                       total, r * r, maxn, nbr, cnt);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// PyG PointNetConv's add_self_loops=True on the ball-query edge list (see the file header): per centroid s (flat number), the
// slot of the neighbour whose flat POINT number equals s (drop[s], -1 if none: that edge is removed) and the edge count after
// removal + the appended loop (cnt_out[s] = cnt[s] - (drop >= 0) + 1).
__global__ void gpe_pointconv_loops_kernel(const int32_t* __restrict__ nbr, const int32_t* __restrict__ cnt, int N, int M,
                                           long total, int maxn, int32_t* __restrict__ cnt_out, int32_t* __restrict__ drop)
{
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= total) return;
    const long b = s / M;
    const int n = cnt[s];
    int d = -1;
    for (int q = 0; q < n; ++q)
        if (b * N + nbr[s * maxn + q] == s) d = q;          // ascending, duplicate-free list: at most one hit
    drop[s] = d;
    cnt_out[s] = n - (d >= 0 ? 1 : 0) + 1;
}

extern "C" int gpe_pointconv_self_loops(const int32_t* nbr, const int32_t* cnt, int B, int N, int M, int maxn, int32_t* cnt_out,
                                        int32_t* drop, void* stream)
{
    if (!nbr || !cnt || !cnt_out || !drop || B <= 0 || N <= 0 || M <= 0 || M > N || maxn <= 0) return GPE_EINVAL;
    const long total = (long)B * M;
    hipLaunchKernelGGL(gpe_pointconv_loops_kernel, dim3(gpe_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, nbr, cnt, N, M,
                       total, maxn, cnt_out, drop);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// compact edge list: for centroid s with edges [off[s], off[s+1]):  msg[e][0:Cx] = x[j][:] (optional), then
// pos[j] - pos[centroid]  (PointConv message input, nn/net_blocks.py:17,24 -> PyG PointNetConv.message), j = the flat source
// point.  drop == NULL: the edges are the ball-query neighbours.  drop != NULL (PyG's self-loop re-indexing): neighbour slot
// drop[s] is skipped and the LAST edge of the centroid is the appended loop, source = flat point s.
__global__ void gpe_ball_messages_kernel(const float* __restrict__ pos, int ldp, const float* __restrict__ x, int ldx, int Cx,
                                         const int32_t* __restrict__ cidx, const int32_t* __restrict__ nbr,
                                         const int64_t* __restrict__ off, const int32_t* __restrict__ drop, int N, int C, int M,
                                         long total, int slots, int maxn, float* __restrict__ msg, int ldm,
                                         int32_t* __restrict__ seg_of_row)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total * slots) return;
    const long s = t / slots;
    const int q = (int)(t - s * slots);
    const long e0 = off[s], e1 = off[s + 1];
    if (q >= e1 - e0) return;
    const long b = s / M;
    long j;
    if (drop && q == e1 - e0 - 1) j = s;                                   // the appended loop: flat point s -> centroid s
    else {
        const int d = drop ? drop[s] : -1;
        j = b * N + nbr[s * maxn + q + ((d >= 0 && q >= d) ? 1 : 0)];
    }
    const long c = b * N + cidx[s];
    float* o = msg + (e0 + q) * ldm;
    for (int k = 0; k < Cx; ++k) o[k] = x[j * ldx + k];
    for (int k = 0; k < C; ++k) o[Cx + k] = pos[j * ldp + k] - pos[c * ldp + k];
    if (seg_of_row) seg_of_row[e0 + q] = (int32_t)s;
}

extern "C" int gpe_ball_messages(const float* pos, int ldp, const float* x, int ldx, int Cx, const int32_t* cidx,
                                 const int32_t* nbr, const int64_t* off, const int32_t* drop, int B, int N, int C, int M, int maxn,
                                 float* msg, int ldm, int32_t* seg_of_row, void* stream)
{
    if (!pos || !cidx || !nbr || !off || !msg || B <= 0 || N <= 0 || M <= 0 || C <= 0 || maxn <= 0 || Cx < 0 ||
        (Cx > 0 && !x) || ldm < Cx + C)
        return GPE_EINVAL;
    const long total = (long)B * M;
    const int slots = maxn + (drop ? 1 : 0);
    hipLaunchKernelGGL(gpe_ball_messages_kernel, dim3(gpe_cdiv(total * slots, 256)), dim3(256), 0, (hipStream_t)stream, pos, ldp,
                       x, ldx, Cx, cidx, nbr, off, drop, N, C, M, total, slots, maxn, msg, ldm, seg_of_row);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// ragged segment max (PointConv aggr = 'max' over each centroid's edges): y[s][c] = max over rows off[s]..off[s+1]-1,
// arg = the winning row (first maximum); empty segments give 0 / -1 (PyG fills missing targets with 0)
__global__ void gpe_ragged_max_fwd_kernel(const float* __restrict__ x, int ldx, const int64_t* __restrict__ off, long S, int C,
                                          float* __restrict__ y, int ldy, int64_t* __restrict__ arg)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S * C) return;
    const long s = t / C;
    const int c = (int)(t - s * C);
    float best = -INFINITY;
    long ba = -1;
    for (long e = off[s]; e < off[s + 1]; ++e) {
        const float v = x[e * ldx + c];
        if (v > best) { best = v; ba = e; }
    }
    y[s * ldy + c] = (ba >= 0) ? best : 0.f;
    arg[s * C + c] = ba;
}

__global__ void gpe_ragged_max_bwd_kernel(const float* __restrict__ gy, int ldgy, const int64_t* __restrict__ off,
                                          const int64_t* __restrict__ arg, const int32_t* __restrict__ seg_of_row, long E, int C,
                                          float* __restrict__ gx, int ldgx)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * C) return;
    const long e = t / C;
    const int c = (int)(t - e * C);
    const long s = seg_of_row[e];
    gx[e * ldgx + c] = (arg[s * C + c] == e) ? gy[s * ldgy + c] : 0.f;
}

extern "C" int gpe_ragged_max_fwd(const float* x, int ldx, const int64_t* off, long S, int C, float* y, int ldy, int64_t* arg,
                                  void* stream)
{
    if (!x || !off || !y || !arg || S <= 0 || C <= 0 || ldx < C || ldy < C) return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_ragged_max_fwd_kernel, dim3(gpe_cdiv(S * C, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, off, S, C,
                       y, ldy, arg);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

extern "C" int gpe_ragged_max_bwd(const float* gy, int ldgy, const int64_t* off, const int64_t* arg,
                                  const int32_t* seg_of_row, long E, int C, float* gx, int ldgx, void* stream)
{
    if (!gy || !off || !arg || !seg_of_row || !gx || E < 0 || C <= 0 || ldgx < C) return GPE_EINVAL;
    if (E == 0) return GPE_OK;
    hipLaunchKernelGGL(gpe_ragged_max_bwd_kernel, dim3(gpe_cdiv(E * C, 256)), dim3(256), 0, (hipStream_t)stream, gy, ldgy, off, arg,
                       seg_of_row, E, C, gx, ldgx);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}
