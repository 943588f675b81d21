// Micro-benchmark: what overlaps with an fp32-MFMA-issuing wave on the SAME SIMD of a gfx950 CU?
// One 512-thread workgroup per CU (2 waves per SIMD).  Waves 0-3 ("A") issue a stream of independent MFMAs, waves 4-7
// ("B", the SIMD partners) run one of several instruction mixes.  Timed: A alone, B alone, A and B together.
//   build: hipcc --offload-arch=gfx950 -O3 -o coissue coissue.hip     run: ./coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

enum { B_NONE = 0, B_VALU_DEP, B_VALU_IND, B_LDS, B_VMEM, B_SALU, B_VALU_SPARSE, B_PKFMA };

template <int AKIND, int BKIND>
__global__ __launch_bounds__(512, 2) void k(float* out, const float4* big, long nbig, int iters, int runA, int runB, int place, int prioB)
{
    extern __shared__ float4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 4096; e += 512) lds[e] = make_float4(e, 1, 2, 3);
    __syncthreads();
    // place 0: A = waves 0-3, B = waves 4-7 (SIMD partners); 1: A = SIMD 0-2 (waves 0,1,2,4,5,6), B = SIMD 3 (waves 3,7);
    // 2: A = waves 4-7 (younger), B = waves 0-3
    const bool isA = place == 0 ? wave < 4 : place == 1 ? (wave & 3) != 3 : wave >= 4;
    if (isA) {
        if (!runA) return;
        f32x4 acc[12];
        for (int i = 0; i < 12; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        float a = lane * 0.001f, b = 1.0f + lane;
        s16x4 ab = {(short)lane, 1, 2, 3}, bb = {3, 2, 1, (short)lane};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    if (AKIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                    else acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab, bb, acc[i], 0, 0, 0);
                }
        }
        float s = 0;
        for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (s == 123.456f) out[tid] = s;
    } else {
        if (!runB) return;
        if (prioB) __builtin_amdgcn_s_setprio(3);
        float x = lane, y = 1.0001f, z = 0.5f;
        float r[8] = {1, 2, 3, 4, 5, 6, 7, 8};
        const int n = iters;
        if (BKIND == B_VALU_DEP) {
            for (int it = 0; it < n; ++it)
#pragma unroll
                for (int u = 0; u < 48; ++u) x = __builtin_fmaf(x, y, z);
        } else if (BKIND == B_VALU_IND) {
            for (int it = 0; it < n; ++it)
#pragma unroll
                for (int u = 0; u < 6; ++u)
#pragma unroll
                    for (int q = 0; q < 8; ++q) r[q] = __builtin_fmaf(r[q], y, z);
        } else if (BKIND == B_PKFMA) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p[4] = {{1, 2}, {3, 4}, {5, 6}, {7, 8}}, yy = {y, y}, zz = {z, z};
            for (int it = 0; it < n; ++it)
#pragma unroll
                for (int u = 0; u < 12; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) p[q] = __builtin_elementwise_fma(p[q], yy, zz);
            x = p[0][0] + p[1][1] + p[2][0] + p[3][1];
        } else if (BKIND == B_VALU_SPARSE) {   // integer VALU (no fp32 FMA datapath)
            int xi = lane, yi = 12345;
            for (int it = 0; it < n; ++it)
#pragma unroll
                for (int u = 0; u < 48; ++u) xi = (xi ^ yi) + (xi >> 3);
            x = xi;
        } else if (BKIND == B_LDS) {
            float4 s4 = make_float4(0, 0, 0, 0);
            int idx = lane;
            for (int it = 0; it < n; ++it)
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    const float4 v = lds[(idx + 64 * u) & 4095];
                    s4.x += v.x; idx += 7;
                }
            x = s4.x;
        } else if (BKIND == B_VMEM) {
            float4 s4 = make_float4(0, 0, 0, 0);
            long base = ((long)blockIdx.x * 4 + (wave & 3)) * 64 + lane;
            const long stride = (long)gridDim.x * 256;
            for (int it = 0; it < n; ++it) {
                float4 v[12];
#pragma unroll
                for (int u = 0; u < 12; ++u) { v[u] = big[base % nbig]; base += stride; }
#pragma unroll
                for (int u = 0; u < 12; ++u) s4.x += v[u].x;
            }
            x = s4.x;
        } else if (BKIND == B_SALU) {
            int sv = __builtin_amdgcn_readfirstlane(wave);
            for (int it = 0; it < n; ++it)
#pragma unroll
                for (int u = 0; u < 48; ++u) sv = __builtin_amdgcn_readfirstlane(sv * 3 + 1);
            x = sv;
        }
        for (int q = 0; q < 8; ++q) x += r[q];
        if (x == 123.456f) out[tid] = x;
    }
}

template <int AKIND, int BKIND>
static void run(const char* name, float* out, const float4* big, long nbig, int iters, int place = 0, int prioB = 0)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[3];
    const int cfg[3][2] = {{1, 0}, {0, 1}, {1, 1}};
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<AKIND, BKIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int c = 0; c < 3; ++c) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL((k<AKIND, BKIND>), dim3(256), dim3(512), 100 * 1024, 0, out, big, nbig, iters, cfg[c][0], cfg[c][1], place, prioB);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[c], e0, e1);
        }
    }
    printf("%-28s A=%7.3f ms  B=%7.3f ms  A+B together=%7.3f ms   (sum %.3f, max %.3f)\n", name, ms[0], ms[1], ms[2],
           ms[0] + ms[1], ms[0] > ms[1] ? ms[0] : ms[1]);
}

int main()
{
    float* out; float4* big;
    const long nbig = 64L << 20;   // 1 GiB of float4
    hipMalloc(&out, 4096); hipMalloc(&big, nbig * sizeof(float4));
    hipMemset(big, 0, nbig * sizeof(float4));
    const int iters = 2000;        // A: 2000*48 MFMAs per wave
    printf("A = fp32 MFMA 16x16x4 stream (waves 0-3), B = partner waves 4-7\n");
    run<0, B_VALU_DEP>("B: dependent v_fma chain", out, big, nbig, iters);
    run<0, B_VALU_IND>("B: 8 independent v_fma", out, big, nbig, iters);
    run<0, B_PKFMA>("B: v_pk_fma_f32", out, big, nbig, iters);
    run<0, B_VALU_SPARSE>("B: integer VALU chain", out, big, nbig, iters);
    run<0, B_LDS>("B: ds_read_b128", out, big, nbig, iters);
    run<0, B_VMEM>("B: global_load_dwordx4", out, big, nbig, iters);
    run<0, B_SALU>("B: SALU/readfirstlane", out, big, nbig, iters);
    printf("--- B at s_setprio(3)\n");
    run<0, B_VALU_DEP>("B: dependent v_fma chain", out, big, nbig, iters, 0, 1);
    run<0, B_LDS>("B: ds_read_b128", out, big, nbig, iters, 0, 1);
    run<0, B_VMEM>("B: global_load_dwordx4", out, big, nbig, iters, 0, 1);
    printf("--- A = younger waves 4-7, B = waves 0-3\n");
    run<0, B_VALU_DEP>("B: dependent v_fma chain", out, big, nbig, iters, 2, 0);
    run<0, B_LDS>("B: ds_read_b128", out, big, nbig, iters, 2, 0);
    run<0, B_VMEM>("B: global_load_dwordx4", out, big, nbig, iters, 2, 0);
    printf("--- A on SIMD 0-2 (6 waves), B on SIMD 3 (2 waves)\n");
    run<0, B_VALU_DEP>("B: dependent v_fma chain", out, big, nbig, iters, 1, 0);
    run<0, B_LDS>("B: ds_read_b128", out, big, nbig, iters, 1, 0);
    run<0, B_VMEM>("B: global_load_dwordx4", out, big, nbig, iters, 1, 0);
    printf("A = bf16 MFMA 16x16x16 stream\n");
    run<1, B_VALU_DEP>("B: dependent v_fma chain", out, big, nbig, iters);
    run<1, B_VALU_IND>("B: 8 independent v_fma", out, big, nbig, iters);
    run<1, B_LDS>("B: ds_read_b128", out, big, nbig, iters);
    run<1, B_VMEM>("B: global_load_dwordx4", out, big, nbig, iters);
    return 0;
}
