// Internal helpers shared by the gfx950 kernels of libgpe_hip.so (not part of the C ABI).
#pragma once
#include <cstdlib>
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

// do the byte ranges [p, p + n) and [q, q + m) meet?  (NULL never overlaps.)  The edge entry points refuse an output that overlaps
// one of their inputs: several kernels write rows through a buffer descriptor and read through plain pointers, which the compiler
// treats as disjoint memory — aliased, the stores lose their order against the loads (gpe_edgegemm_w8_kernel.h, W8_BUFSTORE)
static inline bool gpe_overlap(const void* p, size_t n, const void* q, size_t m)
{
    if (!p || !q || !n || !m) return false;
    const uintptr_t a = (uintptr_t)p, b = (uintptr_t)q;
    return a < b + m && b < a + n;
}

// Measurement switches of the library (A/B runs inside one GPU session: scripts/gpu_session.sh ab): environment variables that are
// ONLY consulted when GPE_DEBUG=1 is set — a production process cannot be steered off the product path by a stray variable.
//   GPE_W8 GPE_REV GPE_LAZY_DZ3 GPE_H3_LEFT          edge kernels        GPE_RD_DEEP GPE_RD_NOPC        reduce-GEMM paths
//   GPE_KNN_PROBE _PIN _VEC _SPLIT _EXACT _F32FILTER GPE_KNN_SORTED      kNN      GPE_WV_KS GPE_WV_BJ GPE_RNN_F32   recurrences
static inline const char* gpe_dbg_env_str(const char* name)
{
    static const int on = getenv("GPE_DEBUG") ? atoi(getenv("GPE_DEBUG")) : 0;
    return on ? getenv(name) : nullptr;
}
static inline int gpe_dbg_env(const char* name, int dflt)
{
    const char* e = gpe_dbg_env_str(name);
    return e ? atoi(e) : dflt;
}

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

// ---- per-device launch state ------------------------------------------------------------------------------------------
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE property of a kernel: a process that drives several
// GPUs (nn.DataParallel-style) must set it once on each.  One bit per device ordinal per call site.
#define GPE_ENSURE_MAX_LDS_N(fn, bytes_)                                                                         \
    do {                                                                                                         \
        static unsigned long long done_ = 0;                                                                     \
        int dev_ = 0;                                                                                            \
        if (hipGetDevice(&dev_) != hipSuccess) return GPE_ELAUNCH;                                               \
        const unsigned long long bit_ = 1ull << (dev_ & 63);                                                     \
        if (!(done_ & bit_)) {                                                                                   \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (bytes_)) != hipSuccess)                                                     \
                return GPE_ELAUNCH;                                                                              \
            done_ |= bit_;                                                                                       \
        }                                                                                                        \
    } while (0)

// kernels with static __shared__ arrays must leave room for them below the 160 KB of a CU
#define GPE_ENSURE_MAX_LDS(fn) GPE_ENSURE_MAX_LDS_N(fn, 160 * 1024)

// compute units of the CURRENT device (cached per device ordinal); defined in gpe_pointwise.hip
int gpe_num_cus();
// ---- caller-owned workspace of the edge entry points (gpe_edge_mlp_fwd / _bwd / gpe_edge_redgemm / gpe_edge_pq_amax) -----------
// The library allocates nothing: a caller passes `ws` of gpe_edge_ws_bytes(...) bytes (16-B aligned) and the entry point carves
//   [0, GPE_WS_DUMMY_BYTES)   the dummy store image of the straight-line edge kernels (64 rows x 512 floats; written, never read)
//   [.., + GPE_WS_H3_BYTES)   f16x3: word 0 = A-operand amax measured in-call, word 1 = packed-weight amax, then the partial
//                             maxima of the gather-bound pass
//   the rest                  per-pseudo-point rows of a k > 16 launch (folded after the kernel)
// Two calls may run concurrently on different streams as long as they do not share a workspace.  NULL / too small: the entry
// point runs the variants that need none (slower), never an error.
#define GPE_WS_DUMMY_BYTES (64 * 512 * 4)
#define GPE_WS_H3_PARTS 1024                         // workgroups (= partial maxima pairs) of the gather-bound pass
#define GPE_WS_H3_BYTES (((2 + 2 * GPE_WS_H3_PARTS) * 4 + 255) & ~255)
struct GpeEdgeWs { float* dummy; unsigned* h3; char* pseudo; size_t pseudo_bytes; };
static inline GpeEdgeWs gpe_edge_ws(void* ws, long bytes)
{
    GpeEdgeWs o = {nullptr, nullptr, nullptr, 0};
    if (!ws || (((uintptr_t)ws) & 15) || bytes < (long)(GPE_WS_DUMMY_BYTES + GPE_WS_H3_BYTES)) return o;
    char* b = static_cast<char*>(ws);
    o.dummy = reinterpret_cast<float*>(b);
    o.h3 = reinterpret_cast<unsigned*>(b + GPE_WS_DUMMY_BYTES);
    o.pseudo = b + GPE_WS_DUMMY_BYTES + GPE_WS_H3_BYTES;
    o.pseudo_bytes = (size_t)bytes - (GPE_WS_DUMMY_BYTES + GPE_WS_H3_BYTES);
    return o;
}
// f16x3 mode (gpe_edgegemm_h3.hip).  Operand scales come from "amax words": device words holding the bit pattern of a
// non-negative float >= the largest magnitude of a tensor.  They are CALLER-OWNED: an entry point that writes an activation /
// dz tensor fills the word the caller passes as `amax_out`, the entry point that consumes the tensor takes it as `amax_a` /
// `amax_u` / `amax_v`.  The library keeps no record of tensors.
// bound of relu(P_i + Q_j) over a [rows][>= 2H] table -> out[0]; part = 2 * GPE_WS_H3_PARTS floats of workspace
int gpe_h3_pq_passes(unsigned* out, float* part, const float* pq, long rows, int H, long ld, hipStream_t s);
// largest |x| over [rows][cols] (pitch ld) -> out[0] (cleared first, on the stream)
int gpe_h3_absmax(unsigned* out, const float* x, long rows, int cols, long ld, hipStream_t s);
// rows below which the f16x3 kernels are not used (their scale passes cost more than they save at small E); part of the
// arithmetic mode: gpe_f16x3_min_rows_set (gpe_rowgemm.hip)
#define GPE_H3_MIN_ROWS_DEFAULT 32768
long gpe_h3_min_rows();
// power of two that brings a tensor whose largest magnitude has the bit pattern `amax` into [2^14, 2^15), and its inverse.
// The exponent is clamped to +-100 (tensors below 2^-86 lose relative precision gracefully), zero / non-finite -> 1.
__device__ __forceinline__ void gpe_h3_scale_of(unsigned amax, float& s, float& inv)
{
    const int e = (int)((amax >> 23) & 0xff);
    int sh = (amax == 0u || e == 255) ? 0 : 141 - e;                              // 14 - (e - 127)
    sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
    s = __uint_as_float((unsigned)(127 + sh) << 23);
    inv = __uint_as_float((unsigned)(127 - sh) << 23);
}

// ---- cloud -> XCD pinning ---------------------------------------------------------------------------------------------
// Workgroup b is dispatched to XCD b % 8 (observed placement; a wrong guess costs speed, never correctness), and every
// XCD has its own 4 MiB L2.  Kernels that re-read a cloud's per-point table (kNN candidates, the Q rows of the EdgeConv
// gather: 1.2-3.3 MB per cloud) therefore keep all work of cloud c on XCD c % 8, so the table is fetched from HBM by ONE
// L2 instead of eight.  The map below turns (xcd = id % 8, slot = id / 8) into a position in that XCD's own work list:
//   cloud = xcd + 8 * (slot / units_per_cloud),  unit = slot % units_per_cloud.
// Used when B >= 8 (otherwise it would idle whole XCDs); clouds past B are skipped by the caller (b >= B).
#define GPE_NXCD 8
static inline bool gpe_pin_clouds(int B) { return B >= GPE_NXCD; }

// wave-granular walk over the B*N points of a batch, one point per wave per step (4 waves per 256-thread block):
//   unpinned: wave w of block g visits points g*4+w, +4*gridDim.x, ...
//   pinned (gridDim.x % 8 == 0): the blocks of XCD x visit the points of clouds x, x+8, ... in order.
struct GpePointWalk { long first, count, stride; int xcd, pin; };
__device__ __forceinline__ GpePointWalk gpe_point_walk(int B, int N, int pin)
{
    GpePointWalk w;
    const int wave = threadIdx.x >> 6;
    w.pin = pin;
    if (pin) {
        w.xcd = blockIdx.x & (GPE_NXCD - 1);
        const int nbx = (B - w.xcd + GPE_NXCD - 1) / GPE_NXCD;          // clouds b = xcd (mod 8), b < B
        w.first = (long)(blockIdx.x >> 3) * 4 + wave;
        w.count = (long)nbx * N;
        w.stride = (long)(gridDim.x >> 3) * 4;
    } else {
        w.xcd = 0;
        w.first = (long)blockIdx.x * 4 + wave;
        w.count = (long)B * N;
        w.stride = (long)gridDim.x * 4;
    }
    return w;
}
// step u -> (cloud b, point i inside the cloud); B*N < 2^31 is a precondition of every caller
__device__ __forceinline__ void gpe_walk_split(const GpePointWalk& w, long u, int N, int& b, int& i)
{
    const unsigned jc = (unsigned)u / (unsigned)N;
    i = (int)((unsigned)u - jc * (unsigned)N);
    b = w.pin ? w.xcd + GPE_NXCD * (int)jc : (int)jc;
}
__device__ __forceinline__ long gpe_walk_point(const GpePointWalk& w, long u, int N)
{
    if (!w.pin) return u;
    int b, i;
    gpe_walk_split(w, u, N, b, i);
    return (long)b * N + i;
}

// tile sequence of a persistent workgroup over B equal clouds of `tpc` tiles each:
//   unpinned (tpc == 0): blockIdx.x, +gridDim.x, ...
//   pinned: XCD x = blockIdx.x % 8 takes clouds x, x+8, ...; its gridDim.x/8 workgroups stride through each cloud's tiles.
// Requires (host-checked) gridDim.x % 8 == 0, B % 8 == 0, gridDim.x / 8 <= tpc.  Positions past the end give tile numbers
// >= num_tiles in both modes, so `tile < num_tiles` stays the loop condition.
struct GpeTileSeq { int t, c, step, tpc, rev, nt; };
// Direction policy of the persistent edge kernels (host side): the kernels of an EdgeConv layer hand 0.6 - 0.8 GB tensors to each
// other, and the last ~ 200 MB a kernel touched are the part of them the memory-side cache (256 MB) can still hold.  Fixed per
// entry point, so that every producer / consumer pair walks in opposite directions — forward: F2 (gather) up, F3 (dense) down;
// backward: dense reduce-GEMM up, B3 (in place) down, gathered reduce-GEMM up, B2 (gathered) down, pull_dq up.
// GPE_REV=0 keeps every kernel walking up (A/B measurements).
static inline int gpe_walk_rev(int down)
{
    static const int off = gpe_dbg_env("GPE_REV", 1) == 0;
    return off ? 0 : down;
}
// rev != 0 walks the same sequence from the far end: pinned — XCD x takes its clouds in the order x + 8 (ncl - 1), ..., x + 8, x
// (a cloud stays on its XCD); unpinned — mirrored tile numbers nt - 1 - t.  Consecutive kernels that hand a tensor on walk in
// opposite directions, so that the consumer starts on what the producer touched last (gpe_walk_direction below).
__device__ __forceinline__ GpeTileSeq gpe_tile_seq(int tpc, int rev = 0, int clouds = 0, int num_tiles = 0)
{
    GpeTileSeq s;
    s.tpc = tpc;
    s.rev = rev;
    s.nt = num_tiles;
    s.step = tpc ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    s.t = tpc ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    s.c = tpc ? (int)(blockIdx.x & (GPE_NXCD - 1)) : 0;
    if (tpc && rev) s.c += GPE_NXCD * (clouds / GPE_NXCD - 1);
    return s;
}
__device__ __forceinline__ int gpe_seq_tile(const GpeTileSeq& s)
{
    if (s.tpc) return (s.c < 0) ? 0x7fffffff : s.c * s.tpc + s.t;
    return (s.rev && s.t < s.nt) ? s.nt - 1 - s.t : s.t;
}
__device__ __forceinline__ void gpe_seq_advance(GpeTileSeq& s)
{
    s.t += s.step;
    if (s.tpc && s.t >= s.tpc) { s.t -= s.tpc; s.c += s.rev ? -GPE_NXCD : GPE_NXCD; }
}

