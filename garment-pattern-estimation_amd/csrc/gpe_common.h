// Internal helpers shared by the gfx950 kernels of libgpe_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GPE_OK 0
#define GPE_EINVAL (-22)
#define GPE_ELAUNCH (-5)

#define GPE_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e_ = hipGetLastError();                   \
        if (e_ != hipSuccess) return GPE_ELAUNCH;            \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int gpe_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline int gpe_round_up(int a, int b) { return (a + b - 1) / b * b; }

// 2-level row addressing used by every dense row loader: logical row r of a [rows, cols] operand lives at
//   base + (r / inner) * stride_outer + (r % inner) * stride_inner          (element units)
// inner <= 0 means "single level": base + r * stride_outer.
struct GpeRows {
    const float* base;
    long stride_outer;
    long stride_inner;
    int inner;
};

__device__ __forceinline__ const float* gpe_row_ptr(const GpeRows& a, long r)
{
    if (a.inner <= 0) return a.base + r * a.stride_outer;
    long o = r / a.inner;
    long i = r - o * a.inner;
    return a.base + o * a.stride_outer + i * a.stride_inner;
}

__device__ __forceinline__ bool gpe_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// exact n / d for 32-bit unsigned n via fp64 reciprocal + one correction step (~8 instructions instead of the
// ~100+ of a 64-bit integer division; matters in the 1-wave/SIMD MFMA kernels where VALU work is not hidden)
__device__ __forceinline__ unsigned gpe_udiv(unsigned n, unsigned d, double rcp)
{
    // branch-free on purpose: a branch between two global loads makes the compiler wait for the first load at the join
    unsigned q = (unsigned)__double2uint_rz((double)n * rcp);
    const unsigned up = ((unsigned long long)(q + 1) * d <= n) ? 1u : 0u;
    const unsigned dn = ((unsigned long long)q * d > n) ? 1u : 0u;
    return q + up - dn;
}
