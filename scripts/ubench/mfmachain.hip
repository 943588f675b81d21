// Micro-benchmark (gfx950): issue interval vs dependent latency of v_mfma_f32_16x16x32_bf16 and v_mfma_f32_16x16x4_f32:
// one wave per SIMD issues 24 MFMAs per iteration round-robin over C independent accumulator chains, C = 1..12.
//   build: hipcc --offload-arch=gfx950 -O3 -o mfmachain mfmachain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int C, int KIND>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters)
{
    const int lane = threadIdx.x & 63;
    f32x4 acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    u32x4 a = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3f803f80u, 0x3f803f81u, 0x3f803f80u, 0x3f803f80u};
    float fa = lane * 0.001f, fb = 1.0f + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            if (KIND == 0)
                acc[u % C] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[u % C], 0, 0, 0);
            else acc[u % C] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[u % C], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int C, int KIND>
static void run(float* out, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<C, KIND>), dim3(256), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<C, KIND>), dim3(256), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s chains=%2d : %.1f cycles per MFMA (2.4 GHz)\n", KIND ? "fp32 16x16x4 " : "bf16 16x16x32", C, ms * 1e6 / iters / 24 * 2.4);
}

int main()
{
    float* out; hipMalloc(&out, 4096);
    const int iters = 20000;
    run<1, 0>(out, iters); run<2, 0>(out, iters); run<3, 0>(out, iters); run<4, 0>(out, iters); run<6, 0>(out, iters);
    run<8, 0>(out, iters); run<12, 0>(out, iters);
    run<1, 1>(out, iters); run<2, 1>(out, iters); run<4, 1>(out, iters); run<12, 1>(out, iters);
    return 0;
}
