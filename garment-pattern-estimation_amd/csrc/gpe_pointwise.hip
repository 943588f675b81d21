// Bandwidth-bound kernels of the path for gfx950: weight packing / BatchNorm folding, the EdgeConv neighbourhood
// gather with fp64 BN statistics, BN finalisation (forward + backward coefficient algebra), max-aggregation
// finish, kNN graph transposition + deterministic pull-style scatter, segment mean pool, LSTM cell pointwise.
// Reference lines each one replaces are listed in include/gpe_hip.h.
#include "gpe_common.h"
#include <math.h>

extern "C" int gpe_abi_version(void) { return 7; }

// compute units the persistent kernels may fill: the device's count minus the caller's reservation (gpe_reserve_cus_set)
static int g_reserved_cus = 0;
extern "C" int gpe_reserve_cus_set(int n)
{
    if (n < 0 || n > 192) return GPE_EINVAL;
    const int prev = g_reserved_cus;
    g_reserved_cus = n;
    return prev;
}
int gpe_num_cus()
{
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256 - g_reserved_cus;
    int& c = cus[dev & 63];
    if (!c) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) c = prop.multiProcessorCount;
        if (c <= 0) c = 256;
    }
    const int left = c - g_reserved_cus;
    return left > 8 ? left : 8;
}

__device__ __forceinline__ float4 pw_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void pw_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ---------------------------------------------------------------------------------------------------------
// weight packing: packed[(kc*4 + plane)*Npad*4 + n*4 + t] = w(n, 16*kc + 4*plane + t) * col_scale
// ---------------------------------------------------------------------------------------------------------
// K extent of a packed weight: whole 16-k chunks, and a K in (96, 208] is filled up (with zeros) to the 10 or 13 chunks the
// register-stationary edge kernels keep resident — they load their full chunk count whatever the real K is.
static int gpe_pack_kpad(int K)
{
    const int k16 = gpe_round_up(K, 16);
    return (k16 > 96 && k16 < 160) ? 160 : (k16 > 160 && k16 < 208) ? 208 : k16;
}
extern "C" long gpe_packed_size(int N, int K) { return (long)gpe_round_up(N, 16) * gpe_pack_kpad(K); }

__global__ void gpe_pack_kernel(const float* __restrict__ w, int ldw, int N, int K, int transpose,
                                const float* __restrict__ col_scale, float* __restrict__ wp, int Npad, long total)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int t = (int)(e & 3);
    const long r = e >> 2;
    const int n = (int)(r % Npad);
    const long kq = r / Npad;                 // = kc*4 + plane
    const int k = (int)(kq * 4 + t);
    float v = 0.f;
    if (n < N && k < K) {
        v = transpose ? w[(size_t)k * ldw + n] : w[(size_t)n * ldw + k];
        if (col_scale) v *= col_scale[k];
    }
    wp[e] = v;
}

extern "C" int gpe_pack_weight(const float* w, int ldw, int N, int K, int transpose, const float* col_scale,
                               float* wp, void* stream)
{
    if (!w || !wp || N <= 0 || K <= 0 || ldw < (transpose ? N : K)) return GPE_EINVAL;
    const int Npad = gpe_round_up(N, 16);
    const long total = (long)Npad * gpe_pack_kpad(K);
    hipLaunchKernelGGL(gpe_pack_kernel, dim3(gpe_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, ldw, N, K,
                       transpose, col_scale, wp, Npad, total);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// multi-pack: every weight-derived operand of a model (plain / transposed / gate-interleaved packs, the P|Q split of
// the first edge-MLP Linear, b_ih + b_hh) refreshed by ONE launch after an optimizer step, instead of ~45 launches
// per training step.  Job table in device memory, 64 bytes per job (mirrored by ops.PackPlan):
// ---------------------------------------------------------------------------------------------------------
struct GpePackJob {
    const float* w;         // source matrix / vector
    const float* w2;        // second source (kind 5)
    float* out;
    long total;             // output elements
    long first_block;       // first 256-thread block of this job
    int ldw, N, K, kind, Npad, aux;
};
static_assert(sizeof(GpePackJob) == 64, "job layout is part of the ABI (ops.PackPlan builds it with numpy)");

// one thread = one output quad (4 consecutive elements: the 4 k values of one packed float4); a block covers 1024 elements
__device__ __forceinline__ float gpe_pack_elem(const GpePackJob& jb, int n, int k)
{
    const float* w = jb.w;
    const int ldw = jb.ldw;
    switch (jb.kind) {
        case 0: return (n < jb.N && k < jb.K) ? w[(size_t)n * ldw + k] : 0.f;
        case 1: return (n < jb.N && k < jb.K) ? w[(size_t)k * ldw + n] : 0.f;
        case 2: {                                                   // gate-interleaved (LSTM G = 4, GRU G = 3), aux = H
            const int H = jb.aux, G = jb.N / H;
            const int b = n / (16 * G), gate = (n / 16) % G, u = (b << 4) + (n & 15);
            return (u < H && k < jb.K) ? w[(size_t)(gate * H + u) * ldw + k] : 0.f;
        }
        case 3: {                                                   // [W1a - W1b ; W1b] (N = 2H rows, K = C), aux = H
            const int H = jb.aux, C = jb.K;
            if (!(n < jb.N && k < C)) return 0.f;
            return (n < H) ? w[(size_t)n * ldw + k] - w[(size_t)n * ldw + C + k] : w[(size_t)(n - H) * ldw + C + k];
        }
        case 4: {                                                   // transpose of the above (N = C, K = 2H), aux = H
            const int H = jb.aux, C = jb.N;
            if (!(n < C && k < jb.K)) return 0.f;
            return (k < H) ? w[(size_t)k * ldw + n] - w[(size_t)k * ldw + C + n] : w[(size_t)(k - H) * ldw + C + n];
        }
    }
    return 0.f;
}

typedef _Float16 pk_f16x8 __attribute__((ext_vector_type(8)));

// blocks a job occupies in a gpe_pack_multi launch (ops.PackPlan builds first_block from it).  Row-major sources (kinds 0, 2, 8:
// the reduction index k is the fast index of the weight) go through 64 x 64 tiles: rows are read in whole 256-byte pieces and turned
// in LDS — read column by column (round 2 - 5) every load instruction touched 64 cache lines for 64 floats.  Kind 9: 4096 elements
// per block (a quarter of the same-word atomics).  Everything else: 1024 outputs per block.
#define GPE_PACK_TILE 64
static long gpe_pack_blocks(int kind, long total, int Npad, int K)
{
    if (kind == 0 || kind == 2) return (long)gpe_cdiv(Npad, GPE_PACK_TILE) * gpe_cdiv(total / Npad, GPE_PACK_TILE);
    if (kind == 8) return (long)gpe_cdiv(Npad, GPE_PACK_TILE) * gpe_cdiv(gpe_round_up(K, 32), GPE_PACK_TILE);
    if (kind == 9) return gpe_cdiv(total, 4096);
    return gpe_cdiv(total, 1024);
}
extern "C" long gpe_pack_job_blocks(int kind, long total, int Npad, int K)
{
    if (kind < 0 || kind > 10 || total <= 0 || ((kind == 0 || kind == 2 || kind == 8) && Npad <= 0)) return GPE_EINVAL;
    return gpe_pack_blocks(kind, total, Npad, K);
}


// kinds 0 / 2 / 8: one 64-column x 64-k tile of the output per block
__device__ __forceinline__ void gpe_pack_tile(const GpePackJob& jb, long tb)
{
    __shared__ float tile[GPE_PACK_TILE][GPE_PACK_TILE + 2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ntn = (jb.Npad + GPE_PACK_TILE - 1) / GPE_PACK_TILE;
    const int tk = (int)(tb / ntn), tn = (int)(tb - (long)tk * ntn);
    const int n0 = tn * GPE_PACK_TILE, k0 = tk * GPE_PACK_TILE;
    const bool pair = ((jb.ldw & 1) == 0) && ((((uintptr_t)jb.w) & 7) == 0);       // rows start on 8-byte boundaries
    const int H = jb.aux, G = (jb.kind == 0) ? 1 : jb.N / H;
    // phase 1: a wave reads its 16 columns' source rows two at a time, 256 bytes of each
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int c = wave * 16 + s * 2 + (lane >> 5), kk = 2 * (lane & 31);
        const int n = n0 + c, k = k0 + kk;
        long row = -1;
        if (jb.kind == 0) row = (n < jb.N) ? n : -1;
        else {
            const int b = n / (16 * G), gate = (n >> 4) % G, u = (b << 4) + (n & 15);
            row = (n < jb.Npad && u < H) ? (long)gate * H + u : -1;
        }
        float2 v = make_float2(0.f, 0.f);
        if (row >= 0) {
            const float* src = jb.w + row * jb.ldw + k;
            if (pair && k + 1 < jb.K) v = *reinterpret_cast<const float2*>(src);
            else {
                if (k < jb.K) v.x = src[0];
                if (k + 1 < jb.K) v.y = src[1];
            }
        }
        *reinterpret_cast<float2*>(&tile[c][kk]) = v;
    }
    __syncthreads();
    // phase 2: outputs in their own order (consecutive threads = consecutive columns)
    if (jb.kind != 8) {
        const int Kpad = (int)(jb.total / jb.Npad);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i, kq = q >> 6, c = q & 63;
            if (n0 + c < jb.Npad && k0 + 4 * kq < Kpad)
                *reinterpret_cast<float4*>(jb.out + ((long)((k0 >> 2) + kq) * jb.Npad + n0 + c) * 4) =
                    make_float4(tile[c][4 * kq], tile[c][4 * kq + 1], tile[c][4 * kq + 2], tile[c][4 * kq + 3]);
        }
        return;
    }
    const int KP = (jb.K + 31) & ~31, KG = KP >> 3;
    float sc, inv;
    gpe_h3_scale_of(*reinterpret_cast<const unsigned*>(jb.w2), sc, inv);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + 256 * i, plane = q >> 9, r = q & 511, kg = r >> 6, c = r & 63;
        if (n0 + c < jb.Npad && k0 + 8 * kg < KP) {
            pk_f16x8 o;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float xs = tile[c][8 * kg + t] * sc;
                const _Float16 h = (_Float16)xs;
                o[t] = plane ? (_Float16)(xs - (float)h) : h;
            }
            *reinterpret_cast<pk_f16x8*>(reinterpret_cast<char*>(jb.out) + ((long)(plane * KG + (k0 >> 3) + kg) * jb.Npad + n0 + c) * 16) = o;
        }
    }
}

__global__ __launch_bounds__(256) void gpe_pack_multi_kernel(const GpePackJob* __restrict__ jobs, int njobs, int tiled)
{
    // wave-uniform job lookup: jobs are sorted by first_block
    int ji = 0;
    for (int q = 1; q < njobs; ++q) ji = ((long)blockIdx.x >= jobs[q].first_block) ? q : ji;
    const GpePackJob jb = jobs[ji];
    if (tiled && (jb.kind == 0 || jb.kind == 2 || jb.kind == 8)) {
        gpe_pack_tile(jb, (long)blockIdx.x - jb.first_block);
        return;
    }
    if (jb.kind == 9) {
        // largest |w| of a [N][K] matrix (row pitch ldw) -> atomicMax into the uint32 word at jb.out (zeroed by the caller):
        // phase 1 of the fp16-plane packs below, which normalise a weight by a power of two taken from it.  4096 elements per block
        __shared__ unsigned red9[4];
        const long e9 = ((long)blockIdx.x - jb.first_block) * 4096 + threadIdx.x;
        unsigned m = 0u;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const long i = e9 + 256 * t;
            if (i < jb.total) {
                const long n = i / jb.K;
                const unsigned a = __float_as_uint(jb.w[n * jb.ldw + (i - n * jb.K)]) & 0x7fffffffu;
                m = m > a ? m : a;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned t = (unsigned)__shfl_xor((int)m, o);
            m = m > t ? m : t;
        }
        if ((threadIdx.x & 63) == 0) red9[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned a = red9[0] > red9[1] ? red9[0] : red9[1], b = red9[2] > red9[3] ? red9[2] : red9[3];
            const unsigned r = a > b ? a : b;
            if (r) atomicMax(reinterpret_cast<unsigned*>(jb.out), r);
        }
        return;
    }
    const long e = (((long)blockIdx.x - jb.first_block) * 256 + threadIdx.x) * 4;
    if (e >= jb.total) return;
    if (jb.kind == 8 || jb.kind == 10) {
        // two-term fp16 planes of a weight in the B-fragment order of v_mfma_f32_16x16x32_f16: out = [plane h | plane l], a plane =
        // [KP / 8 k-groups][Npad columns][8 halves] (KP = K rounded up to 32, zero fill): lane (j, g) of the MFMA reads ONE 16-byte
        // piece = k 8 kg .. 8 kg + 7 of column n.  Values: w * 2^sh = h + l with 2^sh from the tensor's largest magnitude (the word at
        // jb.w2, written by a kind-9 job of an earlier launch).  kind 8: gate-interleaved columns (element map of kind 2);
        // kind 10: the plain transpose (element map of kind 1).  A thread writes one piece (16 bytes = "4 floats" of `total`).
        const unsigned u = (unsigned)(e >> 2);
        const unsigned KP = ((unsigned)jb.K + 31u) & ~31u, KG = KP >> 3;
        const unsigned per_plane = KG * (unsigned)jb.Npad;
        const unsigned plane = u / per_plane, r2 = u - plane * per_plane;
        const unsigned kg = r2 / (unsigned)jb.Npad;
        const int n = (int)(r2 - kg * (unsigned)jb.Npad);
        float sc, inv;
        gpe_h3_scale_of(*reinterpret_cast<const unsigned*>(jb.w2), sc, inv);
        GpePackJob je = jb;
        je.kind = (jb.kind == 8) ? 2 : 1;
        pk_f16x8 o;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float xs = gpe_pack_elem(je, n, (int)(8 * kg) + t) * sc;
            const _Float16 h = (_Float16)xs;
            o[t] = plane ? (_Float16)(xs - (float)h) : h;
        }
        *reinterpret_cast<pk_f16x8*>(reinterpret_cast<char*>(jb.out) + (size_t)u * 16) = o;
        return;
    }
    if (jb.kind >= 5) {                                              // vector jobs, element-wise
        for (int t = 0; t < 4 && e + t < jb.total; ++t) {
            const long i = e + t;
            float v;
            if (jb.kind == 5) v = jb.w[i] + jb.w2[i];                                   // b_ih + b_hh
            else if (jb.kind == 6) v = (i < jb.aux) ? jb.w[i] : 0.f;                    // [b1 | 0]
            else v = jb.w[i] + ((i < jb.aux) ? jb.w2[i] : 0.f);                         // GRU: b_ih + [b_hr | b_hz | 0]
            jb.out[i] = v;
        }
        return;
    }
    // matrix jobs: total = Npad * Kpad is a multiple of 4 and e = 4 * (kq * Npad + n)
    const unsigned r = (unsigned)(e >> 2);                           // a job's output is far below 2^31 elements
    const unsigned kq = r / (unsigned)jb.Npad;
    const int n = (int)(r - kq * (unsigned)jb.Npad);
    const int k0 = (int)(kq * 4);
    float4 v;
    v.x = gpe_pack_elem(jb, n, k0); v.y = gpe_pack_elem(jb, n, k0 + 1);
    v.z = gpe_pack_elem(jb, n, k0 + 2); v.w = gpe_pack_elem(jb, n, k0 + 3);
    *reinterpret_cast<float4*>(jb.out + e) = v;
}

extern "C" long gpe_packed_planes_size(int Npad, int K)
{
    if (Npad <= 0 || K <= 0) return GPE_EINVAL;
    return 2L * (((K + 31) & ~31) >> 3) * Npad * 4;
}

extern "C" int gpe_pack_multi(const void* jobs_dev, int njobs, long total_blocks, void* stream)
{
    if (!jobs_dev || njobs <= 0 || total_blocks <= 0 || total_blocks >= (1L << 31)) return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_pack_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const GpePackJob*>(jobs_dev), njobs, 1);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// gate-interleaved packing of an LSTM weight [4H][ldw] (rows i|f|g|o x H, torch.nn.LSTM layout): packed row
// b*64 + gate*16 + u'  <->  original row gate*H + 16*b + u', so one 64-column block holds the four gates of 16 units.
__global__ void gpe_pack_gates_kernel(const float* __restrict__ w, int ldw, int H, int K, float* __restrict__ wp,
                                      int Npad, long total)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int t = (int)(e & 3);
    const long r = e >> 2;
    const int n = (int)(r % Npad);
    const long kq = r / Npad;
    const int k = (int)(kq * 4 + t);
    const int b = n >> 6, gate = (n >> 4) & 3, u = (b << 4) + (n & 15);
    float v = 0.f;
    if (u < H && k < K) v = w[(size_t)(gate * H + u) * ldw + k];
    wp[e] = v;
}

extern "C" long gpe_packed_gates_size(int H, int K) { return 64L * gpe_cdiv(H, 16) * gpe_round_up(K, 16); }

// the same for G gates (G = 3: nn.GRU rows r|z|n): packed row b*16G + gate*16 + u'  <->  original row gate*H + 16b + u'
__global__ void gpe_pack_ngates_kernel(const float* __restrict__ w, int ldw, int H, int G, int K, float* __restrict__ wp,
                                       int Npad, long total)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int t = (int)(e & 3);
    const long r = e >> 2;
    const int n = (int)(r % Npad);
    const long kq = r / Npad;
    const int k = (int)(kq * 4 + t);
    const int b = n / (16 * G), gate = (n / 16) % G, u = (b << 4) + (n & 15);
    float v = 0.f;
    if (u < H && k < K) v = w[(size_t)(gate * H + u) * ldw + k];
    wp[e] = v;
}

extern "C" long gpe_packed_ngates_size(int H, int G, int K) { return 16L * G * gpe_cdiv(H, 16) * gpe_round_up(K, 16); }

extern "C" int gpe_pack_weight_ngates(const float* w, int ldw, int H, int G, int K, float* wp, void* stream)
{
    if (!w || !wp || H <= 0 || K <= 0 || G <= 0 || G > 8 || ldw < K) return GPE_EINVAL;
    const int Npad = 16 * G * gpe_cdiv(H, 16);
    const long total = (long)Npad * gpe_round_up(K, 16);
    hipLaunchKernelGGL(gpe_pack_ngates_kernel, dim3(gpe_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, ldw, H,
                       G, K, wp, Npad, total);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

extern "C" int gpe_pack_weight_gates(const float* w, int ldw, int H, int K, float* wp, void* stream)
{
    if (!w || !wp || H <= 0 || K <= 0 || ldw < K) return GPE_EINVAL;
    const int Npad = 64 * gpe_cdiv(H, 16);
    const long total = (long)Npad * gpe_round_up(K, 16);
    hipLaunchKernelGGL(gpe_pack_gates_kernel, dim3(gpe_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, ldw, H,
                       K, wp, Npad, total);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// What a forward block of the edge MLP needs from the BatchNorm in front of it, in ONE launch (round 6; three before: gpe_pack_weight
// with col_scale = s -> gpe_fold_bias(t) -> the packed weight's amax pass of the f16x3 edge launch): PF_BLOCKS workgroups share
// W' = W diag(s) in the edge kernels' order and b' = b + W t; with the caller's edge workspace every workgroup leaves the largest
// magnitude of its share in a partial slot, takes a ticket behind a release fence, and the LAST one folds the partials into the
// f16x3 weight slot and clears the A-operand slot and the caller's output word (what the edge launch's own pass would have done).
// No workgroup waits for another.  Same arithmetic, same summation orders as the separate kernels: bit-identical results.
// ---------------------------------------------------------------------------------------------------------
#define PF_BLOCKS 64
#define PF_WAVES 16
__global__ __launch_bounds__(64 * PF_WAVES) void gpe_pack_fold_kernel(const float* __restrict__ w, int ldw, int N, int K,
                                                                      const float* __restrict__ col_scale, const float* __restrict__ t,
                                                                      const float* __restrict__ bias, float* __restrict__ wp, int Npad,
                                                                      long total, float* __restrict__ bias_out, unsigned* slots,
                                                                      unsigned* clear_word, unsigned* ticket)
{
    __shared__ double fred[PF_WAVES][64];
    __shared__ unsigned mred[PF_WAVES];
    __shared__ unsigned last_sh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ---- W' = W diag(s), the element order of gpe_pack_kernel; its largest magnitude on the way ----
    unsigned m = 0u;
    for (long e = (long)blockIdx.x * (64 * PF_WAVES) + threadIdx.x; e < total; e += (long)gridDim.x * (64 * PF_WAVES)) {
        const int tq = (int)(e & 3);
        const long r = e >> 2;
        const int n = (int)(r % Npad);
        const long kq = r / Npad;
        const int k = (int)(kq * 4 + tq);
        float v = 0.f;
        if (n < N && k < K) v = w[(size_t)n * ldw + k] * col_scale[k];
        wp[e] = v;
        const unsigned a = __float_as_uint(v) & 0x7fffffffu;
        m = m > a ? m : a;
    }
    // ---- b' = b + W t: one wave per row, lanes stride over k, fixed-order sum of the 64 partials (gpe_fold_bias_kernel) ----
    for (int n0 = blockIdx.x * PF_WAVES; n0 < N; n0 += gridDim.x * PF_WAVES) {
        const int n = n0 + wave;
        double fs = 0.0;
        if (n < N)
            for (int k = lane; k < K; k += 64) fs += (double)w[(size_t)n * ldw + k] * (double)t[k];
        fred[wave][lane] = fs;
        __syncthreads();
        if (lane == 0 && n < N) {
            double acc = bias ? (double)bias[n] : 0.0;
            for (int l = 0; l < 64; ++l) acc += fred[wave][l];
            bias_out[n] = (float)acc;
        }
        __syncthreads();
    }
    if (!slots) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned u = (unsigned)__shfl_xor((int)m, o);
        m = m > u ? m : u;
    }
    if (lane == 0) mred[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < PF_WAVES; ++i) m = m > mred[i] ? m : mred[i];
        __hip_atomic_store(slots + 2 + blockIdx.x, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();                                     // release: the partial before the ticket
        last_sh = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last_sh) return;
    __threadfence();                                         // acquire: every workgroup's partial
    unsigned mm = 0u;
    if (threadIdx.x < gridDim.x) mm = __hip_atomic_load(slots + 2 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 0) {                                         // (gridDim.x <= 64)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned u = (unsigned)__shfl_xor((int)mm, o);
            mm = mm > u ? mm : u;
        }
        if (lane == 0) {
            slots[1] = mm;
            slots[0] = 0u;
            if (clear_word) clear_word[0] = 0u;
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

extern "C" int gpe_pack_fold(const float* w, int ldw, int N, int K, const float* col_scale, const float* t, const float* bias,
                             float* wp, float* bias_out, void* ws, long ws_bytes, uint32_t* clear_word, uint32_t* ticket, void* stream)
{
    if (!w || !col_scale || !t || !wp || !bias_out || N <= 0 || K <= 0 || ldw < K) return GPE_EINVAL;
    unsigned* slots = nullptr;
    if (ws) {
        const GpeEdgeWs e = gpe_edge_ws(ws, ws_bytes);
        if (!e.h3 || !ticket) return GPE_EINVAL;
        slots = e.h3;
    }
    const int Npad = gpe_round_up(N, 16);
    const long total = (long)Npad * gpe_pack_kpad(K);
    hipLaunchKernelGGL(gpe_pack_fold_kernel, dim3(PF_BLOCKS), dim3(64 * PF_WAVES), 0, (hipStream_t)stream, w, ldw, N, K, col_scale, t, bias,
                       wp, Npad, total, bias_out, slots, clear_word, ticket);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// one wave per output row: lanes stride over k (coalesced), butterfly-free fixed-order reduction through LDS
__global__ __launch_bounds__(256) void gpe_fold_bias_kernel(const float* __restrict__ w, int ldw, int N, int K,
                                                            const float* __restrict__ bias, const float* __restrict__ t,
                                                            float* out)
{
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    double s = 0.0;
    if (n < N)
        for (int k = lane; k < K; k += 64) s += (double)w[(size_t)n * ldw + k] * (double)t[k];
    red[wave][lane] = s;
    __syncthreads();
    if (lane == 0 && n < N) {
        double acc = bias ? (double)bias[n] : 0.0;
        for (int l = 0; l < 64; ++l) acc += red[wave][l];
        out[n] = (float)acc;
    }
}

extern "C" int gpe_fold_bias(const float* w, int ldw, int N, int K, const float* bias, const float* t, float* out,
                             void* stream)
{
    if (!w || !t || !out || N <= 0 || K <= 0 || ldw < K) return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_fold_bias_kernel, dim3(gpe_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, w, ldw, N, K,
                       bias, t, out);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// EdgeConv neighbourhood gather + fp64 statistics of a1 = relu(P_i + Q_j)
//   one wave per point at a time: lanes span the H channels as float4, the k neighbour rows are fetched with
//   wave-uniform (shuffle-broadcast) row bases and all k loads in flight before the first use.
// ---------------------------------------------------------------------------------------------------------
#define GS_BLOCKS 512
#ifndef GS_KU
#define GS_KU 8               // neighbour rows in flight per wave
#endif
#ifndef GS_WAVES
#define GS_WAVES 8            // waves per workgroup (GS_BLOCKS workgroups: 4 waves per SIMD; round 6: 4 before — the pass is a chain of
                              // L2 round trips per wave, more waves in flight: 85 -> 61 us per launch at cfg 2; 16: 63)
#endif
template <int KU>
__global__ __launch_bounds__(64 * GS_WAVES) void gpe_gather_stats_kernel(const float* __restrict__ pq, int ldpq, int H,
                                                               const int32_t* __restrict__ jg, int k,
                                                               int B, int N, int pin, double* __restrict__ part)
{
    __shared__ double red[GS_WAVES][2][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = 4 * lane;
    const bool active = c < H;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    // pin: the waves of XCD x (blocks x, x+8, ...) walk the points of clouds x, x+8, ... only, one cloud at a time, so a
    // cloud's Q table (N x H floats, gathered k-fold) is served by that XCD's L2 alone (gpe_common.h)
    GpePointWalk wk = gpe_point_walk(B, N, pin);
    if (GS_WAVES != 4) {                                   // (gpe_point_walk counts four waves per workgroup)
        wk.first = (long)(pin ? (blockIdx.x >> 3) : blockIdx.x) * GS_WAVES + wave;
        wk.stride = (long)(pin ? (gridDim.x >> 3) : gridDim.x) * GS_WAVES;
    }
    for (long u = wk.first; u < wk.count; u += wk.stride) {
        const long i = gpe_walk_point(wk, u, N);
        const int myidx = (lane < k) ? jg[i * k + lane] : 0;
        float4 p4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (active) p4 = pw_ld4(pq + i * ldpq + c);
        // a point's k messages are summed in fp32 (k <= 64 terms: 1e-7 relative, the same partial sums the fused edge kernels keep per
        // tile), the points in fp64: the fp64 conversions and adds per ELEMENT (round 1 - 4) were this pass's whole run time
        float s32[4] = {0.f, 0.f, 0.f, 0.f}, q32[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < k; s0 += KU) {
            float4 nb[KU];
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                nb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (s0 + u < k) {
                    const long jrow = __shfl(myidx, s0 + u);
                    if (active) nb[u] = pw_ld4(pq + jrow * ldpq + H + c);
                }
            }
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                if (s0 + u < k) {
                    const float a0 = fmaxf(p4.x + nb[u].x, 0.f), a1 = fmaxf(p4.y + nb[u].y, 0.f);
                    const float a2 = fmaxf(p4.z + nb[u].z, 0.f), a3 = fmaxf(p4.w + nb[u].w, 0.f);
                    s32[0] += a0; q32[0] = __builtin_fmaf(a0, a0, q32[0]);
                    s32[1] += a1; q32[1] = __builtin_fmaf(a1, a1, q32[1]);
                    s32[2] += a2; q32[2] = __builtin_fmaf(a2, a2, q32[2]);
                    s32[3] += a3; q32[3] = __builtin_fmaf(a3, a3, q32[3]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) { s[t] += (double)s32[t]; q[t] += (double)q32[t]; }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) { red[wave][0][c + t] = s[t]; red[wave][1][c + t] = q[t]; }
    __syncthreads();
    if (tid < H) {
        double a = 0, b = 0;
        for (int w = 0; w < GS_WAVES; ++w) { a += red[w][0][tid]; b += red[w][1][tid]; }
        part[(size_t)blockIdx.x * 2 * H + tid] = a;
        part[(size_t)blockIdx.x * 2 * H + H + tid] = b;
    }
}

extern "C" int gpe_edge_gather_stats(const float* pq, int ldpq, int H, const int32_t* jg, int B, int N, int k,
                                     double* part, void* stream)
{
    if (!pq || !jg || !part || B <= 0 || N <= 0 || k <= 0 || k > 64 || H <= 0 || H > 256 || (H & 3) || (ldpq & 3) ||
        ldpq < 2 * H)
        return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_gather_stats_kernel<GS_KU>, dim3(GS_BLOCKS), dim3(64 * GS_WAVES), 0, (hipStream_t)stream, pq, ldpq, H,
                       jg, k, B, N, gpe_pin_clouds(B) ? 1 : 0, part);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// BatchNorm finalisation
// ---------------------------------------------------------------------------------------------------------
#define BNF_WAVES 16
#define BNF_CPW 16                    // channels per workgroup: a quarter wave per partial row, 4 rows per wave and load
#define BNF_SUB (64 / BNF_CPW)
__global__ __launch_bounds__(64 * BNF_WAVES) void gpe_bn_finalize_kernel(const double* __restrict__ part, int nblk, int C,
                                                              double count, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps,
                                                              float momentum, float* running_mean,
                                                              float* running_var, int64_t* num_batches,
                                                              float* __restrict__ stats)
{
    // one workgroup per 16 channels (round 6: 64 before — four workgroups read the 1.6 MB of partials of a 200-channel block, 13 do
    // now: 9.2 -> 6.5 us per launch; more load chains per wave did not help: scripts/bn_finalize_bench.py); a wave takes four partial
    // rows per load (lane = (row slot, channel)), its 16 waves x 4 slots split the blocks; combined in a fixed order: chains, then
    // row slots, then waves
    __shared__ double red[BNF_WAVES][2][BNF_CPW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ch = lane & (BNF_CPW - 1), sub = lane / BNF_CPW;
    const int c = blockIdx.x * BNF_CPW + ch;
    if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches) *num_batches += 1;
    double s = 0, q = 0;
    if (c < C) {
        // four independent chains: the loads of a chain's next block do not wait for its add
        constexpr int RS = BNF_WAVES * BNF_SUB;              // row slots of the workgroup
        double s4[4] = {0, 0, 0, 0}, q4[4] = {0, 0, 0, 0};
        int b = wave * BNF_SUB + sub;
        for (; b + 3 * RS < nblk; b += 4 * RS) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s4[u] += part[(size_t)(b + u * RS) * 2 * C + c];
                q4[u] += part[(size_t)(b + u * RS) * 2 * C + C + c];
            }
        }
        for (; b < nblk; b += RS) {
            s4[0] += part[(size_t)b * 2 * C + c];
            q4[0] += part[(size_t)b * 2 * C + C + c];
        }
        s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        q = (q4[0] + q4[1]) + (q4[2] + q4[3]);
    }
    // the row slots of a wave (lanes ch, ch + 16, ch + 32, ch + 48), in a fixed order
#pragma unroll
    for (int o = BNF_CPW; o < 64; o <<= 1) {
        s += __shfl_xor(s, o);
        q += __shfl_xor(q, o);
    }
    if (sub == 0) { red[wave][0][ch] = s; red[wave][1][ch] = q; }
    __syncthreads();
    if (wave != 0 || sub != 0 || c >= C) return;
    s = 0; q = 0;
    for (int w_ = 0; w_ < BNF_WAVES; ++w_) { s += red[w_][0][ch]; q += red[w_][1][ch]; }
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0) var = 0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const float sc = (float)((double)gamma[c] * rstd);
    stats[c] = (float)mean;
    stats[C + c] = (float)rstd;
    stats[2 * C + c] = sc;
    stats[3 * C + c] = (float)((double)beta[c] - mean * (double)sc);
    if (running_mean) {
        const double unbiased = (count > 1) ? var * count / (count - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
}

extern "C" int gpe_bn_finalize(const double* part, int nblk, int C, double count, const float* gamma,
                               const float* beta, float eps, float momentum, float* running_mean,
                               float* running_var, int64_t* num_batches, float* stats_out, void* stream)
{
    if (!part || !gamma || !beta || !stats_out || nblk <= 0 || C <= 0 || count <= 0) return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_bn_finalize_kernel, dim3(gpe_cdiv(C, BNF_CPW)), dim3(64 * BNF_WAVES), 0, (hipStream_t)stream, part, nblk,
                       C, count, gamma, beta, eps, momentum, running_mean, running_var, num_batches, stats_out);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

__global__ void gpe_bn_from_running_kernel(const float* rm, const float* rv, int C, const float* gamma,
                                           const float* beta, float eps, float* stats)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double rstd = 1.0 / sqrt((double)rv[c] + (double)eps);
    const float sc = (float)((double)gamma[c] * rstd);
    stats[c] = rm[c];
    stats[C + c] = (float)rstd;
    stats[2 * C + c] = sc;
    stats[3 * C + c] = (float)((double)beta[c] - (double)rm[c] * (double)sc);
}

extern "C" int gpe_bn_from_running(const float* running_mean, const float* running_var, int C, const float* gamma,
                                   const float* beta, float eps, float* stats_out, void* stream)
{
    if (!running_mean || !running_var || !gamma || !beta || !stats_out || C <= 0) return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_bn_from_running_kernel, dim3(gpe_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream,
                       running_mean, running_var, C, gamma, beta, eps, stats_out);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// max aggregation finish: y = s*(s>=0 ? max : min) + t
// ---------------------------------------------------------------------------------------------------------
__global__ void gpe_edge_finish_kernel(const float* __restrict__ mx, const float* __restrict__ mn, int ldagg,
                                       const float* __restrict__ stats, long rows, int C, float* __restrict__ y,
                                       int ldy)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * C) return;
    const long r = e / C;
    const int c = (int)(e - r * C);
    const float s = stats[2 * C + c], t = stats[3 * C + c];
    const float v = (s >= 0.f) ? mx[r * ldagg + c] : mn[r * ldagg + c];
    y[r * ldy + c] = s * v + t;
}

extern "C" int gpe_edge_finish(const float* mx, const float* mn, int ldagg, const float* stats, long rows, int C,
                               float* y, int ldy, void* stream)
{
    if (!mx || !mn || !stats || !y || rows < 0 || C <= 0 || ldagg < C || ldy < C) return GPE_EINVAL;
    if (rows == 0) return GPE_OK;
    hipLaunchKernelGGL(gpe_edge_finish_kernel, dim3(gpe_cdiv(rows * C, 256)), dim3(256), 0, (hipStream_t)stream, mx,
                       mn, ldagg, stats, rows, C, y, ldy);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// EdgeConv backward: per-point sums for the last BN (partials [PS_BLOCKS][2][C] fp64)
// ---------------------------------------------------------------------------------------------------------
#define PS_BLOCKS 512          // (128 left half the CUs without a workgroup: 47 us for 118 MB)
__global__ __launch_bounds__(256) void gpe_point_sums_kernel(const float* __restrict__ g, int ldg,
                                                             const float* __restrict__ mx,
                                                             const float* __restrict__ mn, int ldagg,
                                                             const float* __restrict__ stats, long rows, int C,
                                                             double* __restrict__ part, unsigned* __restrict__ amax_sg)
{
    // amax_sg (may be NULL): receives max |s_c g_ic| — the data term of the bound of a lazily formed dz3 (gpe_edge_dz3_bound)
    const int c = blockIdx.y * 256 + threadIdx.x;
    double s1 = 0, s2 = 0;
    float gm = 0.f, sabs = 0.f;
    if (c < C) {
        const float mean = stats[c], rstd = stats[C + c], s = stats[2 * C + c];
        // 8 rows in flight per thread (a load per dependent fp64 add was a 512-deep latency chain: 0.4 TB/s)
        const float* sp = (s >= 0.f) ? mx : mn;
        const long st = gridDim.x;
        long r = blockIdx.x;
        for (; r + 7 * st < rows; r += 8 * st) {
            float gv[8], sv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { gv[u] = g[(r + u * st) * ldg + c]; sv[u] = sp[(r + u * st) * ldagg + c]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s1 += (double)gv[u];
                s2 += (double)gv[u] * (double)((sv[u] - mean) * rstd);
                gm = fmaxf(gm, fabsf(gv[u]));
            }
        }
        for (; r < rows; r += st) {
            const float gv = g[r * ldg + c];
            const float sel = sp[r * ldagg + c];
            s1 += (double)gv;
            s2 += (double)gv * (double)((sel - mean) * rstd);
            gm = fmaxf(gm, fabsf(gv));
        }
        part[(size_t)blockIdx.x * 2 * C + c] = s1;
        part[(size_t)blockIdx.x * 2 * C + C + c] = s2;
        sabs = fabsf(s);
    }
    if (amax_sg) {                                   // uniform branch; every wave of the block arrives here
        float m = gm * sabs;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0) atomicMax(amax_sg, __float_as_uint(m * 1.0000002f));   // |s| * max|g| rounds once more than |s g|
    }
}

extern "C" int gpe_point_sums_blocks(void) { return PS_BLOCKS; }

extern "C" int gpe_edge_bwd_point_sums(const float* g, int ldg, const float* mx, const float* mn, int ldagg,
                                       const float* stats, long rows, int C, double* part, uint32_t* amax_sg, void* stream)
{
    if (!g || !mx || !mn || !stats || !part || rows <= 0 || C <= 0) return GPE_EINVAL;
    if (amax_sg && hipMemsetAsync(amax_sg, 0, sizeof(uint32_t), (hipStream_t)stream) != hipSuccess) return GPE_ELAUNCH;
    hipLaunchKernelGGL(gpe_point_sums_kernel, dim3(PS_BLOCKS, gpe_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, g, ldg, mx, mn,
                       ldagg, stats, rows, C, part, amax_sg);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// (one thread per channel walking all partial blocks was a 128-deep dependent load chain: 38 us; now 16 waves split the
// blocks and meet in LDS, combined in a fixed order)
#define BNC_WAVES 16
__global__ __launch_bounds__(64 * BNC_WAVES) void gpe_bn_bwd_coef_kernel(const double* __restrict__ part, int nblk,
                                                                        const float* __restrict__ stats, int C, double count,
                                                                        float* __restrict__ coef, float* dgamma, float* dbeta)
{
    __shared__ double red[BNC_WAVES][2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    double s1 = 0, s2 = 0;
    if (c < C) {
        for (int b = wave; b < nblk; b += BNC_WAVES) {
            s1 += part[(size_t)b * 2 * C + c];
            s2 += part[(size_t)b * 2 * C + C + c];
        }
    }
    red[wave][0][lane] = s1;
    red[wave][1][lane] = s2;
    __syncthreads();
    if (wave != 0 || c >= C) return;
    s1 = 0; s2 = 0;
    for (int w_ = 0; w_ < BNC_WAVES; ++w_) { s1 += red[w_][0][lane]; s2 += red[w_][1][lane]; }
    const double mean = stats[c], rstd = stats[C + c], s = stats[2 * C + c];
    const double m1 = s1 / count, m2 = s2 / count;
    const double k2 = s * rstd * m2;
    coef[c] = (float)s;
    coef[C + c] = (float)(s * m1);
    coef[2 * C + c] = (float)k2;
    coef[3 * C + c] = (float)mean;
    if (dgamma) dgamma[c] = (float)s2;
    if (dbeta) dbeta[c] = (float)s1;
}

extern "C" int gpe_bn_bwd_coef(const double* part, int nblk, const float* stats, int C, double count, float* coef,
                               float* dgamma, float* dbeta, void* stream)
{
    if (!part || !stats || !coef || nblk <= 0 || C <= 0 || count <= 0) return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_bn_bwd_coef_kernel, dim3(gpe_cdiv(C, 64)), dim3(64 * BNC_WAVES), 0, (hipStream_t)stream, part, nblk,
                       stats, C, count, coef, dgamma, dbeta);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// dz3 = (a3>0) ? [slot==argsel]*s*g - c1 - (a3-mean)*k2 : 0, written IN PLACE over the stored activation
// a3 [E][lda3].  One wave per POINT (lanes = column quads): the point's selectors and upstream gradient are loaded once,
// its k message rows are streamed DZ3_RB at a time so that a wave always has DZ3_RB independent 16-B loads in flight
// (one row per wave-iteration was latency-bound at 3.5 TB/s of read+write traffic).
#define DZ3_RB 8
__global__ __launch_bounds__(256) void gpe_dz3_kernel(float* __restrict__ a3, int lda3, const float* __restrict__ g,
                                                      int ldg, const uint8_t* __restrict__ amx,
                                                      const uint8_t* __restrict__ amn, int ldagg,
                                                      const float* __restrict__ coef, long npts, int k, int F,
                                                      float gscale, unsigned* __restrict__ amax_out)
{
    // amax_out (f16x3 mode, else NULL): receives the largest |dz3| written, as the bit pattern of a non-negative float
    __shared__ unsigned amax_red[4];
    if (threadIdx.x < 4) amax_red[threadIdx.x] = 0u;
    __syncthreads();
    float amax_run = 0.f;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = (blockIdx.y * 64 + lane) << 2;
    if (c >= F) return;
    const bool all_slots = amx == nullptr;             // aggr 'add' / 'mean': every message carries the point's gradient
    float cs_[4], c1_[4], k2_[4], mu_[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int cc = (c + t < F) ? c + t : c;
        cs_[t] = coef[cc]; c1_[t] = coef[F + cc]; k2_[t] = coef[2 * F + cc]; mu_[t] = coef[3 * F + cc];
    }
    const long nw = (long)gridDim.x * 4;
    for (long i = (long)blockIdx.x * 4 + wave; i < npts; i += nw) {
        uint8_t mxv[4] = {0, 0, 0, 0}, mnv[4] = {0, 0, 0, 0};
        if (!all_slots) {
            const uchar4 smx = *reinterpret_cast<const uchar4*>(amx + i * ldagg + c);
            const uchar4 smn = *reinterpret_cast<const uchar4*>(amn + i * ldagg + c);
            mxv[0] = smx.x; mxv[1] = smx.y; mxv[2] = smx.z; mxv[3] = smx.w;
            mnv[0] = smn.x; mnv[1] = smn.y; mnv[2] = smn.z; mnv[3] = smn.w;
        }
        int sel[4];
        float sg[4];                                   // s * g of this point, per column of the quad
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int cc = (c + t < F) ? c + t : c;
            sel[t] = (cs_[t] >= 0.f) ? mxv[t] : mnv[t];
            sg[t] = cs_[t] * g[i * ldg + cc] * gscale;
        }
        float* base = a3 + i * k * lda3 + c;
        for (int s0 = 0; s0 < k; s0 += DZ3_RB) {
            float4 a[DZ3_RB];
#pragma unroll
            for (int u = 0; u < DZ3_RB; ++u) {
                const int s_ = (s0 + u < k) ? s0 + u : k - 1;          // clamped: unconditional loads
                a[u] = pw_ld4(base + (long)s_ * lda3);
            }
#pragma unroll
            for (int u = 0; u < DZ3_RB; ++u) {
                if (s0 + u < k) {
                    const float av[4] = {a[u].x, a[u].y, a[u].z, a[u].w};
                    float o[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float hit = (all_slots || sel[t] == s0 + u) ? sg[t] : 0.f;
                        o[t] = (c + t < F && av[t] > 0.f) ? hit - c1_[t] - (av[t] - mu_[t]) * k2_[t] : 0.f;
                    }
                    pw_st4(base + (long)(s0 + u) * lda3, make_float4(o[0], o[1], o[2], o[3]));
                    amax_run = fmaxf(fmaxf(amax_run, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
                }
            }
        }
    }
    if (amax_out) {                                    // uniform; lanes past F have left, the barrier counts live waves only
        atomicMax(&amax_red[wave], __float_as_uint(amax_run));
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned a = amax_red[0] > amax_red[1] ? amax_red[0] : amax_red[1];
            const unsigned b = amax_red[2] > amax_red[3] ? amax_red[2] : amax_red[3];
            atomicMax(amax_out, a > b ? a : b);
        }
    }
}

static int dz3_launch(float* a3, int lda3, const float* g, int ldg, const uint8_t* amx, const uint8_t* amn, int ldagg,
                      const float* coef, int B, int N, int k, int F, float gscale, unsigned* amax_out, hipStream_t stream)
{
    if (!a3 || !g || !coef || B <= 0 || N <= 0 || k <= 0 || F <= 0 || (lda3 & 3) || lda3 < F) return GPE_EINVAL;
    if (amx && (!amn || (ldagg & 3) || ldagg < F)) return GPE_EINVAL;
    const long E = (long)B * N * k;
    if (E >= (1L << 31)) return GPE_EINVAL;
    const long npts = (long)B * N;
    const int blocks = (int)((npts + 3) / 4 < 4096 ? (npts + 3) / 4 : 4096);
    // amax_out (may be NULL): the caller's word for the largest |dz3| written — the f16x3 scale of the edge GEMMs that read dz3
    if (amax_out && hipMemsetAsync(amax_out, 0, sizeof(unsigned), stream) != hipSuccess) return GPE_ELAUNCH;
    hipLaunchKernelGGL(gpe_dz3_kernel, dim3(blocks, gpe_cdiv(F, 256)), dim3(256), 0, stream, a3, lda3, g, ldg, amx, amn,
                       ldagg, coef, npts, k, F, gscale, amax_out);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

extern "C" int gpe_edge_dz3(float* a3, int lda3, const float* g, int ldg, const uint8_t* amx, const uint8_t* amn,
                            int ldagg, const float* coef, int B, int N, int k, int F, uint32_t* amax_out, void* stream)
{
    if (!amx || !amn) return GPE_EINVAL;
    return dz3_launch(a3, lda3, g, ldg, amx, amn, ldagg, coef, B, N, k, F, 1.f, amax_out, (hipStream_t)stream);
}

extern "C" int gpe_edge_dz3_all(float* a3, int lda3, const float* g, int ldg, float gscale, const float* coef, int B,
                                int N, int k, int F, uint32_t* amax_out, void* stream)
{
    return dz3_launch(a3, lda3, g, ldg, nullptr, nullptr, 0, coef, B, N, k, F, gscale, amax_out, (hipStream_t)stream);
}

// sums for an inner BN + true weight gradient of the next Linear, from the CENTRED product
// Gc = dz_next^T (a - mean)  (accumulated without cancellation by the reduce-GEMM) and db = colsum(dz_next):
//   sum dy     = sum_f w[f][c] * db[f]
//   sum dy*xhat = rstd_c * sum_f w[f][c] * Gc[f][c]
//   dW_next[f][c] = Gc[f][c]*s_c + db[f]*beta_c          (since mean*s + t = beta)
#define BNG_FQ 16           // f-lanes per column: 64 columns x 16 f-lanes = one 1024-thread workgroup (Cn = 200: 13 rows each)
__global__ __launch_bounds__(64 * BNG_FQ) void gpe_bn_bwd_from_G_kernel(
    const float* __restrict__ G, int ldG, const float* __restrict__ db, const float* __restrict__ w, int ldw, int Cn,
    int C, const float* __restrict__ stats, double* __restrict__ sums, float* __restrict__ dw, int lddw)
{
    // (one thread per column looping over all Cn rows was a 200-deep dependent load chain: 77 us for 120 KB of data)
    __shared__ double red[BNG_FQ][2][64];
    const int lane = threadIdx.x & 63, fq = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    double a = 0, b = 0;
    if (c < C) {
        const double s = stats[2 * C + c], t = stats[3 * C + c], mean = stats[c];
        const double beta = t + mean * s;
        double a2[2] = {0, 0}, b2[2] = {0, 0};
        int f = fq;
        for (; f + BNG_FQ < Cn; f += 2 * BNG_FQ) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ff = f + BNG_FQ * u;
                const double wf = w[(size_t)ff * ldw + c], gf = G[(size_t)ff * ldG + c], dbf = db[ff];
                a2[u] += wf * dbf;
                b2[u] += wf * gf;
                if (dw) dw[(size_t)ff * lddw + c] = (float)(gf * s + dbf * beta);
            }
        }
        if (f < Cn) {
            const double wf = w[(size_t)f * ldw + c], gf = G[(size_t)f * ldG + c], dbf = db[f];
            a2[0] += wf * dbf;
            b2[0] += wf * gf;
            if (dw) dw[(size_t)f * lddw + c] = (float)(gf * s + dbf * beta);
        }
        a = a2[0] + a2[1];
        b = b2[0] + b2[1];
    }
    red[fq][0][lane] = a;
    red[fq][1][lane] = b;
    __syncthreads();
    if (fq != 0 || c >= C) return;
    a = 0; b = 0;
    for (int q = 0; q < BNG_FQ; ++q) { a += red[q][0][lane]; b += red[q][1][lane]; }
    sums[c] = a;
    sums[C + c] = b * (double)stats[C + c];
}

extern "C" int gpe_bn_bwd_from_G(const float* G, int ldG, const float* db, const float* w_next, int ldw, int Cn,
                                 int C, const float* stats, double* sums, float* dw, int lddw, void* stream)
{
    if (!G || !db || !w_next || !stats || !sums || Cn <= 0 || C <= 0 || ldG < C || ldw < C) return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_bn_bwd_from_G_kernel, dim3(gpe_cdiv(C, 64)), dim3(64 * BNG_FQ), 0, (hipStream_t)stream, G, ldG, db,
                       w_next, ldw, Cn, C, stats, sums, dw, lddw);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// kNN graph transposition (one workgroup per cloud) and the pull-style scatter it enables
// ---------------------------------------------------------------------------------------------------------
// LDS_EDGES: the cloud's edge list is scattered and sorted in LDS and written out once, coalesced (it fits up to
// N*k = 32 k edges, the benchmark shape); otherwise the buckets are sorted in place in global memory.
template <bool LDS_EDGES>
__global__ __launch_bounds__(1024) void gpe_knn_reverse_kernel(const int32_t* __restrict__ idx, int N, int k,
                                                               int32_t* __restrict__ rev_off,
                                                               int32_t* __restrict__ rev_edge)
{
    extern __shared__ int sm[];
    int* cnt = sm;                 // [N]
    int* off = sm + N;             // [N+1]
    __shared__ int wsum[16];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int32_t* id = idx + (size_t)b * N * k;
    int32_t* ro = rev_off + (size_t)b * (N + 1);
    int32_t* const re_g = rev_edge + (size_t)b * N * k;
    int32_t* const re = LDS_EDGES ? reinterpret_cast<int32_t*>(sm + 2 * N + 1) : re_g;
    const int E = N * k;
    for (int i = tid; i < N; i += 1024) cnt[i] = 0;
    __syncthreads();
    for (int e = tid; e < E; e += 1024) atomicAdd(&cnt[id[e]], 1);
    __syncthreads();
    // exclusive scan of cnt: each thread owns a contiguous run of `per` entries
    const int per = (N + 1023) / 1024;
    const int beg = tid * per;
    int local = 0;
    for (int i = 0; i < per; ++i) if (beg + i < N) local += cnt[beg + i];
    // block scan of `local` (wave scan + cross-wave)
    int x = local;
    const int lane = tid & 63, wv = tid >> 6;
    for (int d = 1; d < 64; d <<= 1) { int y = __shfl_up(x, d); if (lane >= d) x += y; }
    if (lane == 63) wsum[wv] = x;
    __syncthreads();
    if (tid == 0) { int a = 0; for (int w = 0; w < 16; ++w) { int t = wsum[w]; wsum[w] = a; a += t; } }
    __syncthreads();
    int run = x - local + wsum[wv];
    for (int i = 0; i < per; ++i) if (beg + i < N) { off[beg + i] = run; run += cnt[beg + i]; }
    if (tid == 1023) off[N] = E;
    __syncthreads();
    for (int i = tid; i <= N; i += 1024) ro[i] = off[i];
    for (int i = tid; i < N; i += 1024) cnt[i] = 0;
    __syncthreads();
    for (int e = tid; e < E; e += 1024) {
        const int jn = id[e];
        const int pos = atomicAdd(&cnt[jn], 1);
        re[off[jn] + pos] = e;
    }
    __threadfence_block();
    __syncthreads();
    // deterministic order inside each bucket: ascending edge id
    for (int jn = tid; jn < N; jn += 1024) {
        const int a = off[jn], z = off[jn + 1];
        for (int i = a + 1; i < z; ++i) {
            const int v = re[i];
            int h = i - 1;
            while (h >= a && re[h] > v) { re[h + 1] = re[h]; --h; }
            re[h + 1] = v;
        }
    }
    if (LDS_EDGES) {
        __syncthreads();
        for (int e = tid; e < E; e += 1024) re_g[e] = re[e];
    }
}

extern "C" int gpe_knn_reverse(const int32_t* idx, int B, int N, int k, int32_t* rev_off, int32_t* rev_edge,
                               void* stream)
{
    if (!idx || !rev_off || !rev_edge || B <= 0 || N <= 0 || k <= 0) return GPE_EINVAL;
    const size_t lds = (size_t)(2 * N + 1) * sizeof(int);
    if (lds > 150 * 1024) return GPE_EINVAL;
    const size_t lds_e = lds + (size_t)N * k * sizeof(int);
    if (lds_e <= 150 * 1024) {
        GPE_ENSURE_MAX_LDS_N((gpe_knn_reverse_kernel<true>), 150 * 1024);
        hipLaunchKernelGGL(gpe_knn_reverse_kernel<true>, dim3(B), dim3(1024), lds_e, (hipStream_t)stream, idx, N, k, rev_off,
                           rev_edge);
    } else {
        GPE_ENSURE_MAX_LDS_N((gpe_knn_reverse_kernel<false>), 150 * 1024);
        hipLaunchKernelGGL(gpe_knn_reverse_kernel<false>, dim3(B), dim3(1024), lds, (hipStream_t)stream, idx, N, k, rev_off,
                           rev_edge);
    }
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

__global__ __launch_bounds__(256) void gpe_pull_dq_kernel(const float* __restrict__ dz, int lddz,
                                                          const int32_t* __restrict__ rev_off,
                                                          const int32_t* __restrict__ rev_edge, int B, int N, int k, int H,
                                                          int pin, float* __restrict__ dQ, int lddq)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = 4 * lane;
    const bool active = c < H;
    // pin: a cloud's dz rows (N*k x H, each read once here, right after the kernel that wrote them) stay on one XCD
    const GpePointWalk wk = gpe_point_walk(B, N, pin);
    for (long u = wk.first; u < wk.count; u += wk.stride) {
        int bi, jn;
        gpe_walk_split(wk, u, N, bi, jn);
        const long b = bi;
        const long pt = b * N + jn;
        const int32_t* ro = rev_off + b * (N + 1);
        const int32_t* re = rev_edge + b * (long)N * k;
        const long e0 = b * (long)N * k;
        const int a = ro[jn], z = ro[jn + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = a; i < z; i += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i + u < z && active) v[u] = pw_ld4(dz + (e0 + re[i + u]) * lddz + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        if (active) pw_st4(dQ + pt * lddq + c, acc);
    }
}

extern "C" int gpe_edge_pull_dq(const float* dz, int lddz, const int32_t* rev_off, const int32_t* rev_edge, int B,
                                int N, int k, int H, float* dQ, int lddq, void* stream)
{
    if (!dz || !rev_off || !rev_edge || !dQ || B <= 0 || N <= 0 || k <= 0 || H <= 0 || H > 256 || (H & 3) ||
        (lddz & 3) || (lddq & 3))
        return GPE_EINVAL;
    const long pts = (long)B * N;
    const int pin = gpe_pin_clouds(B) ? 1 : 0;
    int blocks = (int)((pts + 3) / 4 < 2048 ? (pts + 3) / 4 : 2048);
    if (pin) blocks = gpe_round_up(blocks, GPE_NXCD);
    hipLaunchKernelGGL(gpe_pull_dq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dz, lddz, rev_off,
                       rev_edge, B, N, k, H, pin, dQ, lddq);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// segment mean pool (equal-sized segments: every cloud has N points)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void gpe_segment_mean_fwd_kernel(const float* __restrict__ x, int ldx, int N,
                                                                    int C, float* __restrict__ y, int ldy)
{
    __shared__ double red[4][256];
    const int c = threadIdx.x & 255, rg = threadIdx.x >> 8;
    const int b = blockIdx.x;
    for (int c0 = 0; c0 < C; c0 += 256) {
        double s = 0;
        if (c0 + c < C) {
            // 8 rows in flight per thread (one load per dependent fp64 add was a 512-deep latency chain: 150 us)
            const float* px = x + ((size_t)b * N) * ldx + c0 + c;
            double sa[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int n = rg;
            for (; n + 28 < N; n += 32) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = px[(size_t)(n + 4 * u) * ldx];
#pragma unroll
                for (int u = 0; u < 8; ++u) sa[u] += (double)v[u];
            }
            for (; n < N; n += 4) sa[0] += (double)px[(size_t)n * ldx];
            s = ((sa[0] + sa[1]) + (sa[2] + sa[3])) + ((sa[4] + sa[5]) + (sa[6] + sa[7]));
        }
        red[rg][c] = s;
        __syncthreads();
        if (rg == 0 && c0 + c < C)
            y[(size_t)b * ldy + c0 + c] = (float)((red[0][c] + red[1][c] + red[2][c] + red[3][c]) / (double)N);
        __syncthreads();
    }
}

extern "C" int gpe_segment_mean_fwd(const float* x, int ldx, int B, int N, int C, float* y, int ldy, void* stream)
{
    if (!x || !y || B <= 0 || N <= 0 || C <= 0 || ldx < C || ldy < C) return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_segment_mean_fwd_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, x, ldx, N, C, y,
                       ldy);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

__global__ void gpe_segment_mean_bwd_kernel(const float* __restrict__ gy, int ldgy, int N, int C, long rows,
                                            float* __restrict__ gx, int ldgx, int accumulate)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * C) return;
    const long r = e / C;
    const int c = (int)(e - r * C);
    const float v = gy[(r / N) * ldgy + c] / (float)N;
    float* d = gx + r * ldgx + c;
    *d = accumulate ? (*d + v) : v;
}

extern "C" int gpe_segment_mean_bwd(const float* gy, int ldgy, int B, int N, int C, float* gx, int ldgx,
                                    int accumulate, void* stream)
{
    if (!gy || !gx || B <= 0 || N <= 0 || C <= 0) return GPE_EINVAL;
    const long rows = (long)B * N;
    hipLaunchKernelGGL(gpe_segment_mean_bwd_kernel, dim3(gpe_cdiv(rows * C, 256)), dim3(256), 0,
                       (hipStream_t)stream, gy, ldgy, N, C, rows, gx, ldgx, accumulate);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// LSTM cell pointwise (gate order i, f, g, o as in torch.nn.LSTM)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gpe_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void gpe_lstm_cell_fwd_kernel(float* __restrict__ gates, const float* __restrict__ c_prev, long ldc_prev,
                                         float* __restrict__ c, float* __restrict__ h, long h_stride, int Bn, int H)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)Bn * H) return;
    const long b = e / H;
    const int u = (int)(e - b * H);
    float* gr = gates + b * 4 * H;
    const float ig = gpe_sigmoid(gr[u]);
    const float fg = gpe_sigmoid(gr[H + u]);
    const float gg = tanhf(gr[2 * H + u]);
    const float og = gpe_sigmoid(gr[3 * H + u]);
    const float cn = fg * c_prev[b * ldc_prev + u] + ig * gg;
    gr[u] = ig; gr[H + u] = fg; gr[2 * H + u] = gg; gr[3 * H + u] = og;
    c[b * H + u] = cn;
    h[b * h_stride + u] = og * tanhf(cn);
}

extern "C" int gpe_lstm_cell_fwd(float* gates, const float* c_prev, long ldc_prev, float* c, float* h,
                                 long h_stride, int Bn, int H, void* stream)
{
    if (!gates || !c_prev || !c || !h || Bn <= 0 || H <= 0) return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_lstm_cell_fwd_kernel, dim3(gpe_cdiv((long)Bn * H, 256)), dim3(256), 0,
                       (hipStream_t)stream, gates, c_prev, ldc_prev, c, h, h_stride, Bn, H);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

__global__ void gpe_lstm_cell_bwd_kernel(const float* __restrict__ dh_out, long dho_stride,
                                         const float* __restrict__ dh_rec, int n_rec,
                                         const float* __restrict__ dc_next,
                                         const float* __restrict__ gates, const float* __restrict__ c,
                                         const float* __restrict__ c_prev, long ldc_prev,
                                         float* __restrict__ dgates, long dg_stride, float* __restrict__ dc_prev,
                                         int Bn, int H)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)Bn * H) return;
    const long b = e / H;
    const int u = (int)(e - b * H);
    const float* gr = gates + b * 4 * H;
    const float ig = gr[u], fg = gr[H + u], gg = gr[2 * H + u], og = gr[3 * H + u];
    float dh = dh_out ? dh_out[b * dho_stride + u] : 0.f;
    if (dh_rec)
        for (int z = 0; z < n_rec; ++z) dh += dh_rec[(long)z * Bn * H + b * H + u];   // split-K partials
    const float tc = tanhf(c[b * H + u]);
    float dc = dh * og * (1.f - tc * tc);
    if (dc_next) dc += dc_next[b * H + u];
    float* dg = dgates + b * dg_stride;
    dg[u] = dc * gg * ig * (1.f - ig);
    dg[H + u] = dc * c_prev[b * ldc_prev + u] * fg * (1.f - fg);
    dg[2 * H + u] = dc * ig * (1.f - gg * gg);
    dg[3 * H + u] = dh * tc * og * (1.f - og);
    dc_prev[b * H + u] = dc * fg;
}

extern "C" int gpe_lstm_cell_bwd(const float* dh_out, long dho_stride, const float* dh_rec, int n_rec,
                                 const float* dc_next,
                                 const float* gates, const float* c, const float* c_prev, long ldc_prev,
                                 float* dgates, long dg_stride, float* dc_prev, int Bn, int H, void* stream)
{
    if (!gates || !c || !c_prev || !dgates || !dc_prev || Bn <= 0 || H <= 0) return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_lstm_cell_bwd_kernel, dim3(gpe_cdiv((long)Bn * H, 256)), dim3(256), 0,
                       (hipStream_t)stream, dh_out, dho_stride, dh_rec, n_rec, dc_next, gates, c, c_prev, ldc_prev, dgates,
                       dg_stride, dc_prev, Bn, H);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// GRU cell backward (pointwise).  dh = dh_out + dh_dir_next (the z-gated direct path of step t+1) + sum of the n_rec
// partials of dGh_{t+1}.W_hh;  saved = {r, z, n, hn} of this step (gpe_gru_step_fwd), h_prev rows.
//   dGx [.][3H] = {dr_pre, dz_pre, dn_pre}   (gradient w.r.t. the INPUT-side pre-activations: W_ih, b_ih, x)
//   dGh [.][3H] = {dr_pre, dz_pre, dn_pre*r} (w.r.t. the RECURRENT-side pre-activations: W_hh, b_hh, h_prev)
//   dh_dir_prev = dh * z
__global__ void gpe_gru_cell_bwd_kernel(const float* __restrict__ dh_out, long dho_stride,
                                        const float* __restrict__ dh_rec, int n_rec, const float* __restrict__ dh_dir_next,
                                        const float* __restrict__ saved, const float* __restrict__ h_prev, long hp_stride,
                                        float* __restrict__ dgx, float* __restrict__ dgh, long dg_stride,
                                        float* __restrict__ dh_dir_prev, int Bn, int H)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)Bn * H) return;
    const long b = e / H;
    const int u = (int)(e - b * H);
    const float* sv = saved + b * 4 * H;
    const float rg = sv[u], zg = sv[H + u], ng = sv[2 * H + u], hn = sv[3 * H + u];
    float dh = dh_out ? dh_out[b * dho_stride + u] : 0.f;
    if (dh_dir_next) dh += dh_dir_next[b * H + u];
    if (dh_rec)
        for (int z = 0; z < n_rec; ++z) dh += dh_rec[(long)z * Bn * H + b * H + u];
    const float hp = h_prev[b * hp_stride + u];
    const float dn_pre = dh * (1.f - zg) * (1.f - ng * ng);
    const float dz_pre = dh * (hp - ng) * zg * (1.f - zg);
    const float dr_pre = dn_pre * hn * rg * (1.f - rg);
    float* gx = dgx + b * dg_stride;
    float* gh = dgh + b * dg_stride;
    gx[u] = dr_pre; gx[H + u] = dz_pre; gx[2 * H + u] = dn_pre;
    gh[u] = dr_pre; gh[H + u] = dz_pre; gh[2 * H + u] = dn_pre * rg;
    dh_dir_prev[b * H + u] = dh * zg;
}

extern "C" int gpe_gru_cell_bwd(const float* dh_out, long dho_stride, const float* dh_rec, int n_rec,
                                const float* dh_dir_next, const float* saved, const float* h_prev, long hp_stride,
                                float* dgx, float* dgh, long dg_stride, float* dh_dir_prev, int Bn, int H, void* stream)
{
    if (!saved || !h_prev || !dgx || !dgh || !dh_dir_prev || Bn <= 0 || H <= 0) return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_gru_cell_bwd_kernel, dim3(gpe_cdiv((long)Bn * H, 256)), dim3(256), 0, (hipStream_t)stream,
                       dh_out, dho_stride, dh_rec, n_rec, dh_dir_next, saved, h_prev, hp_stride, dgx, dgh, dg_stride,
                       dh_dir_prev, Bn, H);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// sparsemax over rows of width W <= 32 (sparsemax.Sparsemax(dim=1), /root/reference/nn/nets.py:225; Martins &
// Astudillo 2016): out = max(z - tau, 0), tau from the sorted-cumsum support rule; backward nz*(g - mean_nz(g)).
// One row per lane, the row sorted in registers by a fully unrolled odd-even transposition network.
// ---------------------------------------------------------------------------------------------------------
#define SPX_W 32
__global__ __launch_bounds__(256) void gpe_sparsemax_fwd_kernel(const float* __restrict__ z, int ldz, long rows, int W,
                                                                float* __restrict__ out, int ldo)
{
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float v[SPX_W];
#pragma unroll
    for (int i = 0; i < SPX_W; ++i) v[i] = (i < W) ? z[r * ldz + i] : -INFINITY;
#pragma unroll
    for (int pass = 0; pass < SPX_W; ++pass) {
#pragma unroll
        for (int i = (pass & 1); i + 1 < SPX_W; i += 2) {
            const float a = v[i], b = v[i + 1];
            v[i] = fmaxf(a, b); v[i + 1] = fminf(a, b);          // descending
        }
    }
    float cs = 0.f, tau = 0.f;
#pragma unroll
    for (int i = 0; i < SPX_W; ++i) {
        if (i < W) {
            cs += v[i];
            if (1.f + (float)(i + 1) * v[i] > cs) tau = (cs - 1.f) / (float)(i + 1);   // support grows monotonically
        }
    }
    for (int i = 0; i < W; ++i) out[r * ldo + i] = fmaxf(z[r * ldz + i] - tau, 0.f);
}

__global__ __launch_bounds__(256) void gpe_sparsemax_bwd_kernel(const float* __restrict__ out, int ldo,
                                                                const float* __restrict__ g, int ldg, long rows, int W,
                                                                float* __restrict__ gz, int ldgz)
{
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float s = 0.f, n = 0.f;
    for (int i = 0; i < W; ++i)
        if (out[r * ldo + i] > 0.f) { s += g[r * ldg + i]; n += 1.f; }
    const float m = (n > 0.f) ? s / n : 0.f;
    for (int i = 0; i < W; ++i) gz[r * ldgz + i] = (out[r * ldo + i] > 0.f) ? g[r * ldg + i] - m : 0.f;
}

extern "C" int gpe_sparsemax_fwd(const float* z, int ldz, long rows, int W, float* out, int ldo, void* stream)
{
    if (!z || !out || rows < 0 || W <= 0 || W > SPX_W || ldz < W || ldo < W) return GPE_EINVAL;
    if (rows == 0) return GPE_OK;
    hipLaunchKernelGGL(gpe_sparsemax_fwd_kernel, dim3(gpe_cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, z, ldz,
                       rows, W, out, ldo);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

extern "C" int gpe_sparsemax_bwd(const float* out, int ldo, const float* g, int ldg, long rows, int W, float* gz,
                                 int ldgz, void* stream)
{
    if (!out || !g || !gz || rows < 0 || W <= 0 || ldo < W || ldg < W || ldgz < W) return GPE_EINVAL;
    if (rows == 0) return GPE_OK;
    hipLaunchKernelGGL(gpe_sparsemax_bwd_kernel, dim3(gpe_cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, out,
                       ldo, g, ldg, rows, W, gz, ldgz);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// entmax.SparsemaxLoss() as the reference calls it (nn/metrics/composed_loss.py:4,196,323-332; entmax is third-party and
// un-vendored: the published Fenchel-Young sparsemax loss, Martins & Astudillo 2016 / Blondel et al. 2019, restated):
//   p = sparsemax(x);  L_r = (1 - |p|^2)/2 + <p - e_t, x>;  loss = mean_r L_r (reduction 'elementwise_mean', nothing ignored);
//   dL_r/dx = p - e_t.   One row per lane (same in-register sort as the forward above); gx receives (p - e_t)/rows,
// part[blockIdx] the block's fp64 sum of L_r; gpe_sparsemax_loss_finish adds the partials in index order.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gpe_sparsemax_loss_kernel(const float* __restrict__ x, int ldx,
                                                                 const int32_t* __restrict__ target, long rows, int W,
                                                                 float* __restrict__ gx, int ldg, double* __restrict__ part,
                                                                 int* __restrict__ bad)
{
    __shared__ double red[256];
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double L = 0.0;
    if (r < rows) {
        float v[SPX_W];
#pragma unroll
        for (int i = 0; i < SPX_W; ++i) v[i] = (i < W) ? x[r * ldx + i] : -INFINITY;
#pragma unroll
        for (int pass = 0; pass < SPX_W; ++pass) {
#pragma unroll
            for (int i = (pass & 1); i + 1 < SPX_W; i += 2) {
                const float a = v[i], b = v[i + 1];
                v[i] = fmaxf(a, b); v[i + 1] = fminf(a, b);
            }
        }
        float cs = 0.f, tau = 0.f;
#pragma unroll
        for (int i = 0; i < SPX_W; ++i) {
            if (i < W) {
                cs += v[i];
                if (1.f + (float)(i + 1) * v[i] > cs) tau = (cs - 1.f) / (float)(i + 1);
            }
        }
        int t = target[r];
        if (t < 0 || t >= W) { atomicOr(bad, 1); t = 0; }
        const float inv = 1.f / (float)rows;
        float pp = 0.f, dot = 0.f;
        for (int i = 0; i < W; ++i) {
            const float xi = x[r * ldx + i];
            const float p = fmaxf(xi - tau, 0.f);
            const float d = p - (i == t ? 1.f : 0.f);
            pp += p * p;
            dot += d * xi;
            gx[r * ldg + i] = d * inv;
        }
        L = (double)((1.f - pp) * 0.5f + dot);
    }
    red[threadIdx.x] = L;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(64) void gpe_sparsemax_loss_finish_kernel(const double* __restrict__ part, int nblk, long rows,
                                                                       float* __restrict__ loss)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += part[i];
    loss[0] = (float)(s / (double)rows);
}

// part: >= ceil(rows/256) doubles; bad: one int, set to 1 when a target is outside [0, W) (the caller raises)
extern "C" int gpe_sparsemax_loss(const float* x, int ldx, const int32_t* target, long rows, int W, float* gx, int ldg,
                                  double* part, float* loss, int* bad, void* stream)
{
    if (!x || !target || !gx || !part || !loss || !bad || rows <= 0 || W <= 0 || W > SPX_W || ldx < W || ldg < W) return GPE_EINVAL;
    const int nblk = (int)gpe_cdiv(rows, 256);
    hipLaunchKernelGGL(gpe_sparsemax_loss_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, ldx, target, rows, W, gx,
                       ldg, part, bad);
    GPE_CHECK_LAUNCH();
    hipLaunchKernelGGL(gpe_sparsemax_loss_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, part, nblk, rows, loss);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// out = alpha[0] * x with alpha on the device (the upstream gradient of a scalar loss term: no host read-back)
__global__ void gpe_scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ alpha, float* __restrict__ out, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = alpha[0] * x[i];
}

extern "C" int gpe_scale_dev(const float* x, const float* alpha, float* out, long n, void* stream)
{
    if (!x || !alpha || !out || n < 0) return GPE_EINVAL;
    if (n == 0) return GPE_OK;
    hipLaunchKernelGGL(gpe_scale_dev_kernel, dim3((unsigned)gpe_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, alpha, out, n);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// y[r][c] = s[c]*a[r][c] + t[c]   (BatchNorm applied to a stored post-ReLU activation; dense-MLP last layer)
__global__ void gpe_bn_apply_kernel(const float* __restrict__ a, int lda, const float* __restrict__ stats, long rows,
                                    int C, float a_scale, float t_scale, float* __restrict__ y, int ldy)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * C) return;
    const long r = e / C;
    const int c = (int)(e - r * C);
    y[r * ldy + c] = stats[2 * C + c] * (a[r * lda + c] * a_scale) + stats[3 * C + c] * t_scale;
}

extern "C" int gpe_bn_apply_scaled(const float* a, int lda, const float* stats, long rows, int C, float a_scale,
                                   float t_scale, float* y, int ldy, void* stream)
{
    if (!a || !stats || !y || rows < 0 || C <= 0 || lda < C || ldy < C) return GPE_EINVAL;
    if (rows == 0) return GPE_OK;
    hipLaunchKernelGGL(gpe_bn_apply_kernel, dim3(gpe_cdiv(rows * C, 256)), dim3(256), 0, (hipStream_t)stream, a, lda,
                       stats, rows, C, a_scale, t_scale, y, ldy);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

extern "C" int gpe_bn_apply(const float* a, int lda, const float* stats, long rows, int C, float* y, int ldy,
                            void* stream)
{
    return gpe_bn_apply_scaled(a, lda, stats, rows, C, 1.f, 1.f, y, ldy, stream);
}

// ---------------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------------
__global__ void gpe_reduce_inner_kernel(const float* __restrict__ x, long x_so, long x_si, int T, int R, int C,
                                        float* __restrict__ y, int ldy, int accumulate)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)R * C) return;
    const long r = e / C;
    const int c = (int)(e - r * C);
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += x[r * x_so + t * x_si + c];
    float* d = y + r * ldy + c;
    *d = accumulate ? (*d + s) : s;
}

extern "C" int gpe_reduce_inner(const float* x, long x_so, long x_si, int T, int R, int C, float* y, int ldy,
                                int accumulate, void* stream)
{
    if (!x || !y || T <= 0 || R <= 0 || C <= 0) return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_reduce_inner_kernel, dim3(gpe_cdiv((long)R * C, 256)), dim3(256), 0, (hipStream_t)stream,
                       x, x_so, x_si, T, R, C, y, ldy, accumulate);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

__global__ void gpe_w1_split_kernel(const float* __restrict__ w1, int ldw1, const float* __restrict__ b1, int H,
                                    int C, float* __restrict__ wpq, int ldwpq, float* __restrict__ bias_pq)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < H * C) {
        const int h = e / C, c = e - h * C;
        const float wa = w1[(size_t)h * ldw1 + c], wb = w1[(size_t)h * ldw1 + C + c];
        wpq[(size_t)h * ldwpq + c] = wa - wb;
        wpq[(size_t)(H + h) * ldwpq + c] = wb;
    }
    if (e < 2 * H) bias_pq[e] = (e < H) ? b1[e] : 0.f;
}

extern "C" int gpe_w1_split(const float* w1, int ldw1, const float* b1, int H, int C, float* wpq, int ldwpq,
                            float* bias_pq, void* stream)
{
    if (!w1 || !b1 || !wpq || !bias_pq || H <= 0 || C <= 0 || ldw1 < 2 * C || ldwpq < C) return GPE_EINVAL;
    const int total = (H * C > 2 * H) ? H * C : 2 * H;
    hipLaunchKernelGGL(gpe_w1_split_kernel, dim3(gpe_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w1, ldw1,
                       b1, H, C, wpq, ldwpq, bias_pq);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

__global__ void gpe_w1_grad_kernel(const float* __restrict__ dwpq, int ld, int H, int C, float* __restrict__ dw1,
                                   int lddw1)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= H * C) return;
    const int h = e / C, c = e - h * C;
    const float dp = dwpq[(size_t)h * ld + c], dq = dwpq[(size_t)(H + h) * ld + c];
    dw1[(size_t)h * lddw1 + c] = dp;
    dw1[(size_t)h * lddw1 + C + c] = dq - dp;
}

extern "C" int gpe_w1_grad_from_pq(const float* dwpq, int ld, int H, int C, float* dw1, int lddw1, void* stream)
{
    if (!dwpq || !dw1 || H <= 0 || C <= 0 || ld < C || lddw1 < 2 * C) return GPE_EINVAL;
    hipLaunchKernelGGL(gpe_w1_grad_kernel, dim3(gpe_cdiv(H * C, 256)), dim3(256), 0, (hipStream_t)stream, dwpq, ld,
                       H, C, dw1, lddw1);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

__global__ void gpe_add_kernel(const float* a, const float* b, float* out, long n)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) out[e] = a[e] + b[e];
}

__global__ void gpe_scale_kernel(const float* x, float alpha, float* out, long n)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) out[e] = alpha * x[e];
}

extern "C" int gpe_scale(const float* x, float alpha, float* out, long n, void* stream)
{
    if (!x || !out || n < 0) return GPE_EINVAL;
    if (n == 0) return GPE_OK;
    hipLaunchKernelGGL(gpe_scale_kernel, dim3(gpe_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, alpha, out, n);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

extern "C" int gpe_add(const float* a, const float* b, float* out, long n, void* stream)
{
    if (!a || !b || !out || n < 0) return GPE_EINVAL;
    if (n == 0) return GPE_OK;
    hipLaunchKernelGGL(gpe_add_kernel, dim3(gpe_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// out[(b*T + t)*H + h] = x[b*x_sb + t*x_st + h] * mask[(b*T + t)*H + h] — the inter-layer dropout of nn.LSTM / nn.GRU
// (/root/reference/nn/net_blocks.py:346,374,418-420,469 pass `dropout` to torch's recurrent modules, which multiply the
// output sequence of every layer but the last by a Bernoulli(1-p)/(1-p) mask).  x is a strided [Bn, T, H] view (the h history
// of a recurrent stack, or a dense gradient); mask and out are dense.
__global__ void gpe_mul_rows_kernel(const float* __restrict__ x, long x_sb, long x_st, const float* __restrict__ mask, int T,
                                    int H, long n, float* __restrict__ out)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int h = (int)(e % H);
    const long q = e / H;
    const int t = (int)(q % T);
    const long b = q / T;
    out[e] = x[b * x_sb + t * x_st + h] * mask[e];
}

extern "C" int gpe_mul_rows(const float* x, long x_sb, long x_st, const float* mask, long Bn, int T, int H, float* out,
                            void* stream)
{
    if (!x || !mask || !out || Bn < 0 || T <= 0 || H <= 0) return GPE_EINVAL;
    const long n = Bn * T * H;
    if (n == 0) return GPE_OK;
    hipLaunchKernelGGL(gpe_mul_rows_kernel, dim3(gpe_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, x_sb, x_st, mask, T, H,
                       n, out);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// explicit EdgeConv message inputs — the general formulation behind DynamicEdgeConv (nn/net_blocks.py:124-135):
//   out[e][0:C] = x_i, out[e][C:2C] = x_j - x_i   for edge e = i*k + s, j = jg[e]  (row pitch ldo >= 2C, pad columns zeroed)
// Used only for first-block widths the fused P|Q path does not take (EConv_hidden not a multiple of 4, or > 256): the edge MLP
// then runs as a dense MLP over the E message rows.  Backward: gx_i = sum_s (g1 - g2)[i*k+s] + sum_{e : jg[e] = i} g2[e],
// the second sum pulled through the transposed graph in ascending edge order (deterministic, no atomics).
// ---------------------------------------------------------------------------------------------------------
__global__ void gpe_edge_inputs_fwd_kernel(const float* __restrict__ x, int ldx, int C, const int32_t* __restrict__ jg, int k,
                                           long E, float* __restrict__ out, int ldo)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * ldo) return;
    const long e = t / ldo;
    const int c = (int)(t - e * ldo);
    float v = 0.f;
    if (c < 2 * C) {
        const long i = e / k;
        const float xi = x[i * ldx + (c < C ? c : c - C)];
        v = (c < C) ? xi : x[(long)jg[e] * ldx + (c - C)] - xi;
    }
    out[t] = v;
}

extern "C" int gpe_edge_inputs_fwd(const float* x, int ldx, int C, const int32_t* jg, long npts, int k, float* out, int ldo,
                                   void* stream)
{
    if (!x || !jg || !out || C <= 0 || ldx < C || npts <= 0 || k <= 0 || ldo < 2 * C) return GPE_EINVAL;
    const long total = npts * k * ldo;
    hipLaunchKernelGGL(gpe_edge_inputs_fwd_kernel, dim3((unsigned)gpe_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       C, jg, k, npts * k, out, ldo);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

// rev_off [B][N+1], rev_edge [B][N*k]: the transposed graph of gpe_knn_reverse (edge numbers LOCAL to the cloud)
__global__ void gpe_edge_inputs_bwd_kernel(const float* __restrict__ g, int ldg, int C, const int32_t* __restrict__ rev_off,
                                           const int32_t* __restrict__ rev_edge, int N, int k, long npts,
                                           float* __restrict__ gx, int ldgx)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npts * C) return;
    const long i = t / C;
    const int c = (int)(t - i * C);
    const long b = i / N;
    const int il = (int)(i - b * N);
    float s = 0.f;
    for (int sl = 0; sl < k; ++sl) {
        const float* row = g + (i * k + sl) * ldg;
        s += row[c] - row[C + c];
    }
    const int32_t* off = rev_off + b * (N + 1);
    const int32_t* ed = rev_edge + b * (long)N * k;
    const long ebase = b * (long)N * k;
    for (int q = off[il]; q < off[il + 1]; ++q) s += g[(ebase + ed[q]) * ldg + C + c];
    gx[i * ldgx + c] = s;
}

extern "C" int gpe_edge_inputs_bwd(const float* g, int ldg, int C, const int32_t* rev_off, const int32_t* rev_edge, int B,
                                   int N, int k, float* gx, int ldgx, void* stream)
{
    if (!g || !rev_off || !rev_edge || !gx || C <= 0 || ldg < 2 * C || B <= 0 || N <= 0 || k <= 0 || ldgx < C) return GPE_EINVAL;
    const long total = (long)B * N * C;
    hipLaunchKernelGGL(gpe_edge_inputs_bwd_kernel, dim3((unsigned)gpe_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, g, ldg,
                       C, rev_off, rev_edge, N, k, (long)B * N, gx, ldgx);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}
