// Micro-benchmark (gfx950): does ONE wave overlap its own bf16 MFMAs with its own VALU work, and what does a
// v_mfma_f32_16x16x32_bf16 cost?  One 256-thread workgroup per CU = one wave per SIMD (the single-role edge kernels' layout).
// Variants, cycles per loop iteration (s_memtime):
//   M    : 12 independent MFMAs (4 accumulator chains x 3)
//   V    : 24 independent VALU ops (v_pk_add / cvt / and)
//   MV   : both, interleaved 1 MFMA : 2 VALU
//   M4   : the same 12 MFMAs as fp32 16x16x4 (reference)
//   build: hipcc --offload-arch=gfx950 -O3 -o selfissue selfissue.hip      run: ./selfissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters)
{
    const int lane = threadIdx.x & 63;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    u32x4 a = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3f803f80u, 0x3f803f81u, 0x3f803f80u, 0x3f803f80u};
    float r[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    const float y = 1.0001f + lane * 1e-6f, z = 0.5f;
    float fa = lane * 0.001f, fb = 1.0f + lane;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            if (KIND == 0 || KIND == 2)
                acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[u & 3], 0, 0, 0);
            if (KIND == 3) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[u & 3], 0, 0, 0);
            if (KIND == 1 || KIND == 2) {
                r[(2 * u) & 7] = __builtin_fmaf(r[(2 * u) & 7], y, z);
                r[(2 * u + 1) & 7] = __builtin_fmaf(r[(2 * u + 1) & 7], y, z);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += r[i];
    if (s == 123.456f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    float* out; long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    const int iters = 20000;
    const char* names[4] = {"M  (12 bf16 16x16x32 MFMA)", "V  (24 VALU fma)", "MV (12 MFMA + 24 VALU)", "M4 (12 fp32 16x16x4 MFMA)"};
    for (int kind = 0; kind < 4; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            if (kind == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            hipDeviceSynchronize();
        }
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-30s %8.1f counter ticks per iteration\n", names[kind], (double)c / iters);
    }
    // wall-clock version (the cycle counter may tick at a fixed 100 MHz): time the kernels
    for (int kind = 0; kind < 4; ++kind) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
        if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
        if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
        if (kind == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-30s %8.3f ms  -> %.1f ns per iteration (x2.4 = cycles at 2.4 GHz: %.0f)\n", names[kind], ms, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
    }
    return 0;
}
