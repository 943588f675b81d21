// Micro-benchmark (gfx950), to run first thing next round: what does ONE wave per SIMD pay for scalar and vector
// instructions inside a stream of fp32 MFMAs (the single-role edge kernels' layout)?  DESIGN.md 9 prices every non-MFMA
// instruction of those kernels at 7-10 cycles from the SQ counters, scalar address arithmetic included; this measures it
// directly.  Per loop iteration (wall time x clock, one 256-thread workgroup per CU):
//   M   : 12 x v_mfma_f32_16x16x4_f32 (4 accumulator chains)
//   MS  : the same + 24 s_mul_i32 (2 behind every MFMA; a dependent chain of 4 scalars, like a row address)
//   MV  : the same + 24 v_fma_f32
//   S,V : the 24 scalar / vector instructions alone
// build: hipcc --offload-arch=gfx950 -O3 -o saluissue saluissue.hip      run: ./saluissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, int seed)
{
    const int lane = threadIdx.x & 63;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = 1.f + lane * 1e-3f, b = 0.5f;
    float r[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    const float y = 1.0001f, z = 0.5f;
    int s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;     // wave-uniform: live in SGPRs
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            if (KIND == 0 || KIND == 1 || KIND == 2)
                acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u & 3], 0, 0, 0);
            if (KIND == 1 || KIND == 3) {
                asm volatile("s_mul_i32 %0, %0, %1" : "+s"(s0) : "s"(s1));
                asm volatile("s_mul_i32 %0, %0, %1" : "+s"(s2) : "s"(s3));
            }
            if (KIND == 2 || KIND == 4) {
                r[(2 * u) & 7] = __builtin_fmaf(r[(2 * u) & 7], y, z);
                r[(2 * u + 1) & 7] = __builtin_fmaf(r[(2 * u + 1) & 7], y, z);
            }
        }
    }
    float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + (float)(s0 ^ s2);
    for (int i = 0; i < 8; ++i) s += r[i];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int KIND>
static void run(const char* name, int cus, int iters)
{
    float* out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(cus), dim3(256), 0, 0, out, iters, 3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(cus), dim3(256), 0, 0, out, iters, 3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-4s %8.3f ms   %7.1f cycles per iteration at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / iters);
    hipFree(out);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 200000;
    run<0>("M", cus, iters);
    run<1>("MS", cus, iters);
    run<2>("MV", cus, iters);
    run<3>("S", cus, iters);
    run<4>("V", cus, iters);
    return 0;
}
