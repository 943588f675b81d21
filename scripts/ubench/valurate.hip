// VALU issue rates on gfx950 at the kNN kernel's occupancy (4 waves per SIMD): scalar vs packed fp32, and the exact
// sub+fma mix of gpe_knn_kernel's inner loop with its operands already in registers.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float y, float z)
{
    const int tid = threadIdx.x;
    float r[16];
    f2 p[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = tid + i;
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = (f2){(float)(tid + i), (float)(tid - i)};
    const f2 yy = {y, y + 1.f}, zz = {z, z + 1.f};
    float q0 = y, q1 = y * 2, q2 = y * 3, q3 = y * 4;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) r[i] = __builtin_fmaf(r[i], y, z);
        } else if (KIND == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], yy, zz);
        } else if (KIND == 2) {                       // kNN mix, packed: 8 pk_sub + 8 pk_fma = 16 pair-dims
            const float qa[4] = {q0, q1, q2, q3};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f2 qq = (f2){qa[a], qa[a]};
                const f2 d0 = qq - yy, d1 = qq - zz;
                p[2 * a] = __builtin_elementwise_fma(d0, d0, p[2 * a]);
                p[2 * a + 1] = __builtin_elementwise_fma(d1, d1, p[2 * a + 1]);
            }
            q0 += 1.f;                                // keep the subtraction loop-variant
        } else if (KIND == 3) {                       // kNN mix, scalar: 16 v_sub + 16 v_fma
            const float qa[4] = {q0, q1, q2, q3};
            const float pa[4] = {yy.x, yy.y, zz.x, zz.y};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float d = qa[a] - pa[b];
                    r[4 * a + b] = __builtin_fmaf(d, d, r[4 * a + b]);
                }
            q0 += 1.f;
        }
        if (KIND >= 2) { q1 += q0; q2 += q0; q3 += q0; }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    if (s == 123.456f) out[tid] = s;
}
template <int KIND>
static void run(const char* name, int wgs, int iters, double lane_ops_per_iter, int instr_per_iter)
{
    float* out;
    hipMalloc(&out, 4096);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double waves_per_simd = wgs * 4.0 / 1024.0;
    const double cyc = ms * 1e-3 * 2.4e9 / (iters * waves_per_simd);      // SIMD cycles per wave-iteration at 2.4 GHz
    printf("%-34s wgs=%4d  %.3f ms  %.1f cyc/iter/wave  %.2f cyc/instr  %.1f lanes-ops/clk/SIMD\n", name, wgs, ms, cyc,
           cyc / instr_per_iter, lane_ops_per_iter * 64 / cyc);
    hipFree(out);
}
int main()
{
    for (int wgs : {256, 1024}) {
        run<0>("16 x v_fma_f32", wgs, 20000, 16, 16);
        run<1>("8 x v_pk_fma_f32", wgs, 20000, 16, 8);
        run<2>("kNN mix packed (8 pk_add+8 pk_fma)", wgs, 20000, 32, 16);
        run<3>("kNN mix scalar (16 sub+16 fma)", wgs, 20000, 32, 32);
    }
    return 0;
}
