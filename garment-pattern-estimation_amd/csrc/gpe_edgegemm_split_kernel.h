// Split-precision variants of the single-role fused row GEMM of gpe_edgegemm_sr.hip: the per-edge MLP of DynamicEdgeConv
// (/root/reference/nn/net_blocks.py:43-47,124-135 forward; its input-gradient half in backward) with fp32-GRADE products on the
// 16-bit matrix pipe.  One kernel body, two split policies (template parameter SP):
//
//   SplitBf16x3  ("bf16x6", gpe_edgegemm_x6.hip): x = h + m + l in three bf16 terms (8+8+8 = 24 mantissa bits, fp32's exponent
//                range: no scaling needed), six products  ah.bh + (ah.bm + am.bh) + (ah.bl + am.bm + al.bh)  [+ O(2^-24)];
//   SplitF16x2   ("f16x3",  gpe_edgegemm_h3.hip): x.s = h + l in two fp16 terms (11+11 = 22 bits + the sign of l = 23), three
//                products  ah.bh + (ah.bl + al.bh)  [+ O(2^-23)].  fp16 has a 5-bit exponent, so each operand is first multiplied
//                by a power of two s that brings the TENSOR's largest magnitude into [2^14, 2^15) — exact, undone by two exact
//                multiplications in the epilogue.  An element of magnitude >= 2^-17 of the tensor's largest keeps full
//                relative precision (l is a normal fp16); smaller ones keep an absolute error < 2^-40 of the largest.  The
//                largest magnitudes come in as bit patterns in device memory (RgParams::h3_amax_a / h3_amax_w), measured on the device by
//                gpe_edgegemm_h3.hip — no host read-back.
//
// Why: v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate and a wave streaming it owns its SIMD — the epilogue's VALU
// work adds to the MFMA time.  v_mfma_f32_16x16x32_{bf16,f16} is 16x faster per FLOP and co-issues with
// VALU / LDS / VMEM.  A two-term bf16 split (bf16x3, gpe_edgegemm.hip) keeps 16 mantissa bits: approximate.
//
// Structure = the single-role kernel (one persistent 256-thread workgroup per CU, one wave per SIMD, 512 VGPRs):
//   * the wave's slice of the weight is split once in the prologue and stays resident as 16-bit B fragments (AGPRs);
//   * the A tile stays fp32 in LDS (same 159 KB layout: A[2] + C); a wave splits each A fragment on the fly right after the
//     ds_read (the split's VALU ops co-issue with the MFMAs);
//   * K runs in slabs of 32 (one MFMA k-extent); slot q = 4*slab + mtile carries the memory pipeline: slot 0 issues every
//     global load of the iteration, the next slots commit the staged rows to the other A buffer, the last 16 slots run the
//     epilogue of the previous tile one row each;
//   * N-tiles: wave w owns tiles [AQ*w, AQ*w+AQ) for all 64 rows; the BQ left-over tiles are split over the waves BY K SLAB
//     (wave w takes slabs w, w+4): holding a left-over tile's whole K in every wave would not fit the register file.  The
//     four partial products meet in LDS — in the rows of the just-consumed A buffer that the SAME wave re-stages next, so no
//     extra barrier — and are summed in a fixed order (bit-reproducible).
#pragma once
#include "gpe_rowgemm.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define X6_PB 16          // rows a wave stages / finishes per tile
// scheduling fences around a slot's memory slice: always for the fp32-tile policy.  The two-plane policy runs faster WITHOUT
// them (its slots hold 6 - 12 short MFMAs: letting the scheduler weave the slice into them, under max-ilp, measured 13.78 ->
// 13.46 ms per step in one session, scripts/ab_bench.sh); the knob stays for A/B builds.
#ifndef X6_PLANES_FENCE
#define X6_PLANES_FENCE 0
#endif
#define X6_FENCE(planes) (!(planes) || X6_PLANES_FENCE)
#define X6_NPW 4          // max points per wave per tile (gather / aggregation paths)
// default left-over scheme per kernel kind (see the LEFT template parameter), from the A/B of profiles/r04_c_left_schemes.md
// (us per launch at cfg 2, schemes 0 / 1 / 2): F2 480 / 462 / 448, F3 480 / 516 / 495, B3 518 / 529 / 521, B2 528 / 526 / 517
#ifndef X6_LEFT_F2
#define X6_LEFT_F2 2
#define X6_LEFT_F3 0
#define X6_LEFT_B3 0
#define X6_LEFT_B2 2
#endif
#define GPE_ENOTSUP_SHAPE 12345

// A wave has 256 architectural VGPRs + 256 accumulation VGPRs; MFMA takes its B operand from either file.  The resident
// weights (208 registers) are pinned in AGPRs by hand: left to itself the allocator keeps them architectural and, in the
// gather variants, spills them to scratch memory — reloaded every chunk behind an s_waitcnt vmcnt(0).
__device__ __forceinline__ float x6_pin_agpr(float x)
{
    float a;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(x));
    return a;
}
__device__ __forceinline__ float4 x6_UNUSED_pin4(const float4 v)
{
    return make_float4(x6_pin_agpr(v.x), x6_pin_agpr(v.y), x6_pin_agpr(v.z), x6_pin_agpr(v.w));
}

// P row of (pseudo-)point x: x itself, or x / f when a k > 16 point runs as f pseudo-points (RgParams::pmagic, gpe_edgegemm_sr.hip)
template <bool PSEUDO>
__device__ __forceinline__ long x6_prow(int x, unsigned pmagic) { return PSEUDO ? (long)__umulhi((unsigned)x, pmagic) : (long)x; }

// Wave-uniform choice among the (<= X6_NPW) P rows of a wave's points.  Arguments BY VALUE and selects on values: written
// as `if (idx == q) dst = arr_q` the compiler turns the phi of loads into a load through a phi of pointers into the lambda
// closure, which pins the closure AND every captured local (v[], act[], ...) in scratch memory — each access then drags
// an s_waitcnt vmcnt(0) through the load pipeline.
__device__ __forceinline__ float4 x6_sel4(const float4 a0, const float4 a1, const float4 a2, const float4 a3, int idx)
{
    float4 r = a0;
    r.x = (idx == 1) ? a1.x : r.x; r.y = (idx == 1) ? a1.y : r.y; r.z = (idx == 1) ? a1.z : r.z; r.w = (idx == 1) ? a1.w : r.w;
    r.x = (idx == 2) ? a2.x : r.x; r.y = (idx == 2) ? a2.y : r.y; r.z = (idx == 2) ? a2.z : r.z; r.w = (idx == 2) ? a2.w : r.w;
    r.x = (idx == 3) ? a3.x : r.x; r.y = (idx == 3) ? a3.y : r.y; r.z = (idx == 3) ? a3.z : r.z; r.w = (idx == 3) ? a3.w : r.w;
    return r;
}

typedef __bf16 x6_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 x6_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 x6_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 x6_f16x2 __attribute__((ext_vector_type(2)));
typedef float x6_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned x6_u32x4 __attribute__((ext_vector_type(4)));

// ---- split policies --------------------------------------------------------------------------------------------------------
// P planes per operand; NPROD plane products per fp32 product, listed small terms first: product t multiplies plane pa(t) of
// the A fragment with plane pw(t) of the resident weight.
struct SplitBf16x3 {
    static constexpr int P = 3, NPROD = 6;
    static constexpr bool SCALED = false;
    __device__ static constexpr int pa(int t) { return t == 0 ? 2 : t == 1 ? 0 : t == 2 ? 1 : t == 3 ? 1 : 0; }   // l h m m h h
    __device__ static constexpr int pw(int t) { return t == 0 ? 0 : t == 1 ? 2 : t == 2 ? 1 : t == 3 ? 0 : t == 4 ? 1 : 0; }   // H L M H M H
    __device__ static __forceinline__ unsigned cvt2(float a, float b)              // {bf16(a) | bf16(b) << 16}, RNE
    {
        const x6_f32x2 v = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, x6_bf16x2));
    }
    __device__ static __forceinline__ void split2(float a, float b, unsigned (&o)[3])
    {
        o[0] = cvt2(a, b);
        const float ra = a - __uint_as_float(o[0] << 16), rb = b - __uint_as_float(o[0] & 0xffff0000u);
        o[1] = cvt2(ra, rb);
        const float sa = ra - __uint_as_float(o[1] << 16), sb = rb - __uint_as_float(o[1] & 0xffff0000u);
        o[2] = cvt2(sa, sb);
    }
    __device__ static __forceinline__ f32x4 mfma(const x6_u32x4 a, const x6_u32x4 b, const f32x4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(x6_bf16x8, a), __builtin_bit_cast(x6_bf16x8, b), c, 0, 0, 0);
    }
};

struct SplitF16x2 {
    static constexpr int P = 2, NPROD = 3;
    static constexpr bool SCALED = true;
    __device__ static constexpr int pa(int t) { return t == 0 ? 1 : 0; }    // l h h
    __device__ static constexpr int pw(int t) { return t == 1 ? 1 : 0; }    // H L H
    __device__ static __forceinline__ void split2(float a, float b, unsigned (&o)[2])
    {
        const x6_f32x2 v = {a, b};
        const x6_f16x2 h = __builtin_convertvector(v, x6_f16x2);                   // v_cvt_pk_f16_f32, RNE
        const x6_f32x2 r = v - __builtin_convertvector(h, x6_f32x2);               // exact in fp32
        o[0] = __builtin_bit_cast(unsigned, h);
        o[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, x6_f16x2));
    }
    __device__ static __forceinline__ f32x4 mfma(const x6_u32x4 a, const x6_u32x4 b, const f32x4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(x6_f16x8, a), __builtin_bit_cast(x6_f16x8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ void scale_of(unsigned amax, float& s, float& inv) { gpe_h3_scale_of(amax, s, inv); }
};

// P-plane split of 8 consecutive k-values of a lane's MFMA operand: f[0..7] -> planes (8 16-bit values each, k order kept)
template <class SP> struct X6Frag { x6_u32x4 pl[SP::P]; };

template <class SP>
__device__ __forceinline__ X6Frag<SP> x6_split8(const float4 lo, const float4 hi)
{
    unsigned q0[SP::P], q1[SP::P], q2[SP::P], q3[SP::P];
    SP::split2(lo.x, lo.y, q0);
    SP::split2(lo.z, lo.w, q1);
    SP::split2(hi.x, hi.y, q2);
    SP::split2(hi.z, hi.w, q3);
    X6Frag<SP> f;
#pragma unroll
    for (int t = 0; t < SP::P; ++t) f.pl[t] = (x6_u32x4){q0[t], q1[t], q2[t], q3[t]};
    return f;
}
__device__ __forceinline__ float4 x6_scale4(const float4 v, float s) { return make_float4(v.x * s, v.y * s, v.z * s, v.w * s); }
// keep a resident B fragment in accumulation registers (MFMA reads B from either file)
__device__ __forceinline__ void x6_pin(x6_u32x4& v) { asm volatile("" : "+a"(v)); }

// C-tile pitch in floats
template <class SP> __host__ __device__ constexpr int x6_ldc(int NT, int KCH) { return 16 * NT + 4; }
// 16-byte chunks per plane row of the two-plane A tile: 2 KCH of data, padded to a count = 2 (mod 4)
__host__ __device__ constexpr int x6_pchunks(int KCH) { return ((2 * KCH) & 3) == 2 ? 2 * KCH : 2 * KCH + 2; }
// float index, inside the consumed A buffer, of the 16 K-partials that wave `w` leaves for left-over tile `b` of row `row`.
// fp32 tile: columns [16 (w BQ + b), +16) of the row.  Planes: the first 256 bytes of the row in plane b (BQ <= 2 = planes) —
// either way inside memory that only the row's owner re-stages, which is why no barrier separates the two.
template <bool PLANES, int BQ, int LDA, int PPITCH, int PLANE>
__device__ __forceinline__ int x6_scr(int row, int w, int b)
{
    if constexpr (PLANES) return (b * PLANE + row * PPITCH) / 4 + 16 * w;
    else return row * LDA + 16 * (w * BQ + b);
}

// K16: k == 16 (the benchmark configuration): every wave owns exactly ONE point per tile (see gpe_edgegemm_sr.hip).
// KCH = K extent in 16-wide chunks (the packed-weight granularity); KS = ceil(KCH / 2) slabs of 32.
// PSEUDO: a k > 16 point runs as pseudo-points (RgParams::pmagic != 0) — its own instances, so that the others do not carry the
// division (the non-k16 gather-backward variants sit at the register limit)
// AGGT (forward only): 1 / 0 = the per-point max / min / argmax / argmin tracking of the aggregated last block is compiled in /
// out; -1 = decided at run time by p.agg (the tracking then always runs: 24 VALU per row, only the stores are skipped).  The
// k = 16 instances of the benchmark configuration are instantiated with 0 and 1, and with 2 = tracked AND the activation rows stored
// as _Float16 (RgParams::out_half: the aggregated block whose backward forms dz3 lazily, DESIGN.md 8 row g).
// LEFT (two-plane policy only): where a wave multiplies its K slabs (wave, wave + 4) of the BQ left-over tiles.
//   0  inside the slot loop under the wave-dependent `(sl & 3) == wave` — every slot becomes a conditional block that modifies accL,
//      which the compiler merges with up to 48 register copies per join and runs through accumulation registers with `s_nop 7` +
//      read-back (13 x 13 gather forward: 504 v_mov_b64 + 340 v_accvgpr moves per tile and wave, scripts/isa_mix.py);
//   1  in a branch-free loop of their own behind the slot loop (the A fragments are read a second time; a slab past the end
//      multiplies zero weights against a clamped address).  -45 % static instructions, but the loop's LDS reads and MFMAs sit
//      exposed in front of the tile's barrier: F2 480 -> 462 us, but F3 480 -> 516, B3 518 -> 529 (profiles/r04_c_left_schemes.md);
//   2  ROTATED slabs: wave w walks the K slabs in the order w, w + 1, .. (mod KS), so that ITS left-over slabs are iterations 0
//      and 4 of every wave — compile-time positions, no branch, no second read.  The resident weights are loaded in that order;
//      a wave's accumulation order over K differs from its neighbours' (fixed per wave: still bit-reproducible).
// LAZY (k = 16 in-place backward only): the dense A operand is the stored activation of the aggregated block and dz3 is formed
// from it in commit_row (RgParams::lz_*): 28 VALU per row quad instead of a separate 1.3 GB pass.
template <class SP, int AQ, int BQ, int KCH, int AMODE, int EMODE, bool K16, bool PSEUDO = false, int AGGT = -1, int LEFT = 0,
          bool LAZY = false>
__global__ __launch_bounds__(256, 1) void gpe_edgegemm_split_kernel(RgParams p, int stats_nblk)
{
    static_assert(!LAZY || (K16 && AMODE == A_DENSE && EMODE == E_BWD_INPLACE && !PSEUDO), "lazy dz3: k = 16 in-place backward");
    constexpr bool TRACK = (EMODE == E_EDGE_FWD) && AGGT != 0;
    constexpr bool OUTH = AGGT == 2;                     // 2 = tracked AND the activation rows stored in fp16 (RgParams::out_half)
    constexpr int NT = 4 * AQ + BQ;
    constexpr int KS = (KCH + 1) / 2;
    constexpr bool KTAIL = (KCH & 1) != 0;               // last slab holds only 16 k: lane groups g >= 2 contribute zeros
    constexpr int LSL = (KS + 3) / 4;                    // slabs of a left-over tile per wave (K-split over the 4 waves)
    // LDS layout.  fp32 policy: A tile fp32 [64][LDA] x 2 buffers + C [64][LDC].  PLANES (SplitF16x2): the A tile is kept as the
    // policy's two 16-bit planes, split ONCE when a staged row is committed — h plane [64][PPITCH bytes] then l plane, x 2 buffers;
    // a plane row is x6_pchunks(KCH) sixteen-byte chunks: a count = 2 (mod 4) keeps every 16-lane group of a ds_read_b128 that
    // walks down a column (groups {0-3,12-15,20-27}, ... of MI355X_MICROARCH.md "LDS") on 16 distinct bank quads.
    constexpr bool PLANES = SP::SCALED;
    constexpr bool ROT = PLANES && BQ > 0 && LEFT == 2;          // rotated slab order (see LEFT above)
    constexpr bool TAILLOOP = PLANES && BQ > 0 && LEFT == 1;
    constexpr int LDA = 16 * KCH + 4;
    constexpr int LDC = x6_ldc<SP>(NT, KCH);
    constexpr int PPITCH = 16 * x6_pchunks(KCH);         // bytes per plane row
    constexpr int PLANE = RG_BM * PPITCH;                // bytes per plane
    constexpr int AWORDS = PLANES ? (2 * PLANE) / 4 : RG_BM * LDA;   // one A buffer, in floats
    constexpr int NSLOT = 4 * KS;                        // memory-pipeline slots per tile: slot q = 4*slab + mtile
    // slot 0 issues every global load; the staged rows are committed in the CM_SLOTS slots right before the epilogue (late
    // enough for the loads to have landed, and BEFORE the epilogue's conditional stores: see gpe_edgegemm_sr.hip); the last
    // slots finish EPR rows of the previous tile each
    constexpr int EPR = (NSLOT >= 28) ? 1 : 2;
    constexpr int EP_START = NSLOT - X6_PB / EPR;
    constexpr int CM_SLOTS = 4, CMR = X6_PB / CM_SLOTS;
    constexpr int CM_START = EP_START - CM_SLOTS;
    static_assert(CM_START >= 2, "K too short for the slot schedule");
    // The 13 x 13 backward variants of the two-plane policy stage their 16 rows in TWO batches of 8 through the same registers
    // (batch 0: issued in slot 0, committed in slots CM_START, CM_START + 1; batch 1: issued in slot CM_START + 2, committed in
    // the tile's last two slots): 32 staging registers instead of 64, which is what keeps them free of scratch spills.
    constexpr int NH = (SP::SCALED && NT == 13 && KCH == 13 && EMODE != E_EDGE_FWD) ? 2 : 1;
    constexpr int RBH = X6_PB / NH;
    // ... and load the stored activation of epilogue row u in slot u (the row is finished in slot EP_START + u, 12 slots later)
    // instead of all 16 in slot 0: at most 12 - 13 of them are live at a time.
    // (gathered activations come from the cloud's L2-resident table: 8 slots of lead instead of 12, 9 rows live)
    constexpr bool ACT_LATE = NH == 2;
    constexpr int ACT_SHIFT = (EMODE == E_BWD_GATHER) ? 4 : 0;
    static_assert(!ACT_LATE || (EPR == 1 && EP_START >= 8), "late activation loads assume one epilogue row per slot");
    static_assert(NH == 1 || AMODE == A_DENSE, "the second batch would gather through the NEXT tile's neighbour rows");
    constexpr bool GATHER_ACT = (EMODE == E_BWD_GATHER);

    extern __shared__ __align__(16) float smem[];
    __shared__ unsigned amax_sh[4];                      // SplitF16x2: the waves' largest written magnitudes (kernel tail)
    float* const Abuf0 = smem;
    float* const Abuf1 = smem + AWORDS;
    float* const Cs = smem + 2 * AWORDS;                 // [64][LDC]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int g_tail = (g >= 2) ? (g & 1) : g;           // ROT: chunk read by a lane of the 16-wide tail slab
    const int rows_w = p.R >> 2;                         // rows of a tile this wave stages / finishes (<= X6_PB)
    const int rb = wave * rows_w;
    const int rk16 = (65536 + p.k - 1) / p.k;            // u / k == (u * rk16) >> 16 for u < 64
    const int PT = p.R / p.k, npw = PT >> 2;             // points per tile / per wave: first point of this wave's
                                                         // share of tile t is t*PT + wave*npw — no per-tile division
    const int c = lane << 2;                             // this lane's column quad
    const bool k_on = c < p.K, n_on = c < p.N;
    const int ck = k_on ? c : 0, cn = n_on ? c : 0;      // clamped quads for the unconditional loads

    for (int e = tid; e < 2 * AWORDS; e += 256) smem[e] = 0.f;

    // ---- operand scales (SplitF16x2 only): powers of two from the measured largest magnitudes -------------------------------
    float sA = 1.f, sW = 1.f, invA = 1.f, invW = 1.f;
    if constexpr (SP::SCALED) {
        SP::scale_of(p.h3_amax_a[0], sA, invA);
        SP::scale_of(p.h3_amax_w[0], sW, invW);
    }

    // ---- weights: resident 16-bit B fragments, SP::P planes -----------------------------------------------------------------
    // lane (j, g) of slab sl holds k = 32 sl + 8 g + {0..7} of column 16*tile + j: two float4 of the packed weight
    // (chunk 2 sl + (g >> 1), k-quads 2 (g & 1) and 2 (g & 1) + 1)
    x6_u32x4 wP[SP::P][AQ][KS];
    x6_u32x4 lP[SP::P][BQ > 0 ? BQ : 1][LSL];
    {
        auto load_frag = [&](int col, int sl) -> X6Frag<SP> {
            const int cc = (col < p.Npad) ? col : p.Npad - 1;
            const int kc = 2 * sl + (g >> 1);
            const bool on = col < p.Npad && kc < KCH;
            const int kcc = kc < KCH ? kc : KCH - 1;
            float4 f0 = ld4(p.wp + (((long)(kcc * 4 + 2 * (g & 1))) * p.Npad + cc) * 4);
            float4 f1 = ld4(p.wp + (((long)(kcc * 4 + 2 * (g & 1) + 1)) * p.Npad + cc) * 4);
            if (SP::SCALED) { f0 = x6_scale4(f0, sW); f1 = x6_scale4(f1, sW); }
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            return x6_split8<SP>(on ? f0 : z, on ? f1 : z);
        };
#pragma unroll
        for (int i = 0; i < AQ; ++i)
#pragma unroll
            for (int sl = 0; sl < KS; ++sl) {
                // ROT: register set `sl` holds the slab this wave multiplies in ITERATION sl
                const int slw = ROT ? ((sl + wave >= KS) ? sl + wave - KS : sl + wave) : sl;
                const X6Frag<SP> f = load_frag(16 * (AQ * wave + i) + j, slw);
#pragma unroll
                for (int t = 0; t < SP::P; ++t) { wP[t][i][sl] = f.pl[t]; x6_pin(wP[t][i][sl]); }
            }
#pragma unroll
        for (int b = 0; b < BQ; ++b)
#pragma unroll
            for (int q = 0; q < LSL; ++q) {
                const int sl = wave + 4 * q;               // this wave's K slabs of the left-over tiles
                const X6Frag<SP> f = load_frag(sl < KS ? 16 * (4 * AQ + b) + j : p.Npad, sl < KS ? sl : 0);
#pragma unroll
                for (int t = 0; t < SP::P; ++t) lP[t][b][q] = f.pl[t];
            }
    }

    // ---- epilogue constants + running state ---------------------------------------------------------------------------
    double stS[4] = {0, 0, 0, 0}, stQ[4] = {0, 0, 0, 0};
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 cs4 = bias4, c14 = bias4, k24 = bias4, mu4 = bias4;
    if (n_on) {
        if (EMODE == E_EDGE_FWD) {
            if (p.bias) {
                bias4.x = p.bias[c];
                if (c + 1 < p.N) bias4.y = p.bias[c + 1];
                if (c + 2 < p.N) bias4.z = p.bias[c + 2];
                if (c + 3 < p.N) bias4.w = p.bias[c + 3];
            }
        } else {                                         // N % 4 == 0 guaranteed by the dispatcher
            cs4 = ld4(p.coef_out + c); c14 = ld4(p.coef_out + p.N + c);
            // both operand scales (powers of two) are undone in the BatchNorm-backward factor of z: the epilogue multiplies the RAW
            // accumulator
            if (SP::SCALED) cs4 = x6_scale4(cs4, invW * invA);
            k24 = ld4(p.coef_out + 2 * p.N + c); mu4 = ld4(p.coef_out + 3 * p.N + c);
            // dz = (a>0) ? fma(z, s', fma(-k2, a, mean k2 - c1)) : 0 — two fmas per element (c14 := mean k2 - c1, k24 := -k2)
            c14 = make_float4(__builtin_fmaf(mu4.x, k24.x, -c14.x), __builtin_fmaf(mu4.y, k24.y, -c14.y),
                              __builtin_fmaf(mu4.z, k24.z, -c14.z), __builtin_fmaf(mu4.w, k24.w, -c14.w));
            k24 = make_float4(-k24.x, -k24.y, -k24.z, -k24.w);
        }
    }
    const float invAW = invA * invW;
    float amax_run = 0.f;                                // SplitF16x2: largest magnitude written to p.out (-> RgParams::amax_out)
    float s32[4], q32[4], vmx[4], vmn[4];
    int imx[4], imn[4];
    float4 dp;
    int es = 0, ept = 0;                                 // row inside the current point, point inside this wave's share
    long e_row0 = 0, e_pt0 = 0; int e_rv = 0;            // tile being finished
    // gather forward: the activation rows leave through a buffer descriptor of the tile's `out` rows (scalar row offset, constant
    // lane offset) with the sc1 flavour as the instruction's aux bit.  Round 3 / 4 wrote this store as `asm volatile
    // ("global_store_dwordx4 … sc1")`: in the register-tight instances the compiler then waits `vmcnt(0)` in front of every row — for
    // the previous row's write acknowledgement (found in round 5 on the two-waves-per-SIMD kernel, profiles/r05_a_w8_schedules.md)
    __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0, 0x00020000);

    float4 v[RBH];                                       // rows staged for the next tile (one batch)
    // LAZY: coefficients of the BatchNorm behind the aggregation for this lane's K columns, and — per staged tile — the point's
    // s * g quad and winning slots
    // dz3 = (a>0) ? [slot wins] s g - c1 - (a - mean) k2 : 0, evaluated as fma(-k2, a, base) with base = s g + (mean k2 - c1) for the
    // winning slot, (mean k2 - c1) for the others: lzs = s, lznc = mean k2 - c1, lznk = -k2 per column; lz_sel / lz_won per staged tile
    float lzs[4] = {0.f, 0.f, 0.f, 0.f}, lznc[4] = {0.f, 0.f, 0.f, 0.f}, lznk[4] = {0.f, 0.f, 0.f, 0.f};
    int lz_sel[4] = {0, 0, 0, 0};
    float lz_won[4] = {0.f, 0.f, 0.f, 0.f};
    float4 lz_gq = make_float4(0.f, 0.f, 0.f, 0.f);
    uchar4 lz_sx = make_uchar4(0, 0, 0, 0), lz_sn = make_uchar4(0, 0, 0, 0);
    if constexpr (LAZY) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (c + t < p.K) {
                // (all three carry the operand scale sA — a power of two, exact — so that commit_row's split needs no multiply)
                const float k2 = p.lz_coef[2 * p.K + c + t];
                lzs[t] = p.lz_coef[c + t] * sA;
                lznc[t] = __builtin_fmaf(p.lz_coef[3 * p.K + c + t], k2, -p.lz_coef[p.K + c + t]) * sA;
                lznk[t] = -k2 * sA;
            }
        }
    }
    float4 pvs0, pvs1, pvs2, pvs3;                       // P rows of the points being staged (gather)
    pvs0 = pvs1 = pvs2 = pvs3 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 act[(EMODE != E_EDGE_FWD) ? X6_PB : 1];       // stored activations of the tile being finished (backward)
    float4 pve0, pve1, pve2, pve3;                       // P rows of the points being finished (E_BWD_GATHER)
    pve0 = pve1 = pve2 = pve3 = make_float4(0.f, 0.f, 0.f, 0.f);
    int s_rv = 0;                                        // valid rows of the tile being staged

    // Row addressing of the gathers stays in VGPRs: 16 rows x 64-bit scalar addresses (plus their clamps) do not fit the
    // SGPR file next to this kernel's ~60 live scalars, and SGPR spills go to scratch memory (every reload is a
    // scratch_load + s_waitcnt vmcnt(0) in the middle of the load pipeline).  So a tile's neighbour rows are loaded
    // lane-distributed ONE ITERATION AHEAD (lane L <-> row rb + min(L, rows_w-1)), a row's value is broadcast with
    // ds_bpermute, and `vz` (an opaque zero) keeps the point indices per-lane as well.
    int vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
    // LICM would hoist ~250 per-row scalars (row numbers, LDS offsets, point indices — all functions of rb, rows_w and
    // rk16) out of the persistent tile loop and the register allocator would then spill them to scratch; re-deriving them
    // from an opaque per-iteration zero keeps them transient.
    int rbl = rb, rwl = rows_w, rkl = rk16;
#define X6_REFRESH_SCALARS()                                   \
    {                                                          \
        int sz_;                                               \
        asm volatile("s_mov_b32 %0, 0" : "=s"(sz_));           \
        rbl = rb + sz_; rwl = rows_w + sz_; rkl = rk16 + sz_;  \
    }
    auto load_jgv = [&](int tile) -> int {
        const long row0 = (long)tile * p.R;
        const int rv = (int)((p.M - row0 < p.R) ? (p.M - row0) : p.R);
        int r = rbl + ((lane < rwl) ? lane : rwl - 1);
        r = (r < rv - 1) ? r : rv - 1;
        return p.jg[row0 + r];
    };
    int jgv_s = 0, jgv_e = 0;            // neighbour rows for the NEXT stage (A_GATHER) / the NEXT epilogue (E_BWD_GATHER)
    int jgv_cur = 0;                     // ACT_LATE: the neighbour rows of the tile being finished (jgv_e moves on in slot 0)

    // ---- VMEM issue: everything this iteration will need --------------------------------------------------------------
    // stored activation of epilogue row u of the tile being finished (e_row0 / e_rv / jgv_cur set by issue_epi_loads)
    auto issue_act_load = [&](int u) {
        const int last = e_rv - 1;
        int r = rbl + ((u < rwl) ? u : rwl - 1);
        r = (r < last) ? r : last;                                      // clamp: unconditional loads
        if (EMODE == E_BWD_INPLACE) act[u] = ld4(p.out + (e_row0 + r) * p.ldo + cn);
        else {
            const int jj = __builtin_amdgcn_ds_bpermute(u << 2, jgv_cur);
            act[u] = ld4(p.pq + (long)jj * p.ldpq + p.H + cn);
        }
    };
    auto issue_epi_loads = [&](int tile) {
        e_row0 = (long)tile * p.R;
        e_pt0 = (long)tile * PT + wave * npw;
        e_rv = (int)((p.M - e_row0 < p.R) ? (p.M - e_row0) : p.R);
        es = 0; ept = 0;
        if (EMODE == E_EDGE_FWD && AMODE == A_GATHER)
            orsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.out) + e_row0 * p.ldo * 4, 0, RG_BM * p.ldo * 4, 0x00020000);
#pragma unroll
        for (int t = 0; t < 4; ++t) { s32[t] = 0.f; q32[t] = 0.f; vmx[t] = -INFINITY; vmn[t] = INFINITY; imx[t] = 0; imn[t] = 0; }
        dp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (EMODE == E_EDGE_FWD) return;
        // NOTE: every load below is unconditional (clamped rows, clamped column quad): a register that is loaded under a
        // branch needs a copy at the join, and that copy waits for the load right there — no pipelining left
        const int last = e_rv - 1;
        jgv_cur = jgv_e;
        if (!ACT_LATE) {
#pragma unroll
            for (int u = 0; u < X6_PB; ++u) issue_act_load(u);
        }
        if (GATHER_ACT) {
            const int pt0 = tile * PT + wave * npw + vz, ptl = tile * PT + ((last * rkl) >> 16);
            pve0 = ld4(p.pq + x6_prow<PSEUDO>((pt0 + 0 < ptl) ? pt0 + 0 : ptl, p.pmagic) * p.ldpq + cn);
            pve1 = ld4(p.pq + x6_prow<PSEUDO>((pt0 + 1 < ptl) ? pt0 + 1 : ptl, p.pmagic) * p.ldpq + cn);
            pve2 = ld4(p.pq + x6_prow<PSEUDO>((pt0 + 2 < ptl) ? pt0 + 2 : ptl, p.pmagic) * p.ldpq + cn);
            pve3 = ld4(p.pq + x6_prow<PSEUDO>((pt0 + 3 < ptl) ? pt0 + 3 : ptl, p.pmagic) * p.ldpq + cn);
        }
    };
    auto issue_stage_loads = [&](int tile, int h) {
        const long row0 = (long)tile * p.R;
        s_rv = (int)((p.M - row0 < p.R) ? (p.M - row0) : p.R);
        const int last = s_rv - 1;
        if constexpr (LAZY) if (h == 0) {                 // once per tile: the second batch keeps the registers
            // (issued AHEAD of the row loads: the memory counter retires in order, so commit_row's wait for its row covers these —
            // behind them, the first commit had to wait for every row of the batch)
            // the point of this wave's 16 rows (k = 16: one point per wave and tile), clamped into the last valid point
            const long pt0 = (long)tile * PT + wave * npw, ptl = (long)tile * PT + ((last * rkl) >> 16);
            const long pt = pt0 < ptl ? pt0 : ptl;
            // (dword loads at clamped columns: the gradient rows are the caller's [B*N, K] tensor as it is — no 16-B alignment, no
            // pad columns; the pad lanes' coefficients are 0)
            const float* gr = p.lz_g + pt * p.lz_ldg + ck;
            const int rem = p.K - 1 - ck;
            lz_gq = make_float4(gr[0], gr[rem < 1 ? rem : 1], gr[rem < 2 ? rem : 2], gr[rem < 3 ? rem : 3]);
            lz_sx = *reinterpret_cast<const uchar4*>(p.lz_amx + pt * p.lz_ldagg + ck);
            lz_sn = *reinterpret_cast<const uchar4*>(p.lz_amn + pt * p.lz_ldagg + ck);
        }
#pragma unroll
        for (int uu = 0; uu < RBH; ++uu) {
            const int u = h * RBH + uu;
            int r = rbl + ((u < rwl) ? u : rwl - 1);
            r = (r < last) ? r : last;
            if (AMODE == A_GATHER) {
                const int jj = __builtin_amdgcn_ds_bpermute(u << 2, jgv_s);
                v[uu] = ld4(p.pq + (long)jj * p.ldpq + p.H + ck);
            } else {
                // rows are 16-B aligned and padded to a multiple of 4 columns (checked by the dispatcher)
                if constexpr (LAZY) {
                    // the stored activation is fp16 (8 bytes per quad; pitch in halves): the raw words travel in v[].x / .y
                    const uint2 hq = *reinterpret_cast<const uint2*>(reinterpret_cast<const _Float16*>(p.a.base) + (row0 + r) * p.a.stride_outer + ck);
                    v[uu].x = __uint_as_float(hq.x); v[uu].y = __uint_as_float(hq.y);
                } else
                    v[uu] = ld4(p.a.base + (row0 + r) * p.a.stride_outer + ck);
            }
        }
        if (AMODE == A_GATHER) {
            const int pt0 = tile * PT + wave * npw + vz, ptl = tile * PT + ((last * rkl) >> 16);
            pvs0 = ld4(p.pq + x6_prow<PSEUDO>((pt0 + 0 < ptl) ? pt0 + 0 : ptl, p.pmagic) * p.ldpq + ck);
            pvs1 = ld4(p.pq + x6_prow<PSEUDO>((pt0 + 1 < ptl) ? pt0 + 1 : ptl, p.pmagic) * p.ldpq + ck);
            pvs2 = ld4(p.pq + x6_prow<PSEUDO>((pt0 + 2 < ptl) ? pt0 + 2 : ptl, p.pmagic) * p.ldpq + ck);
            pvs3 = ld4(p.pq + x6_prow<PSEUDO>((pt0 + 3 < ptl) ? pt0 + 3 : ptl, p.pmagic) * p.ldpq + ck);
        }
    };
    // ---- LDS commit of staged row u (compile-time u) ------------------------------------------------------------------
    auto commit_row = [&](float* An, int u) {
        if ((!K16 && u >= rwl) || !k_on) return;
        const int r = rbl + u;
        float4 o = v[u % RBH];
        if (AMODE == A_GATHER) {
            const float4 pv = K16 ? pvs0 : x6_sel4(pvs0, pvs1, pvs2, pvs3, (u * rkl) >> 16);
            o.x = fmaxf(o.x + pv.x, 0.f); o.y = fmaxf(o.y + pv.y, 0.f);
            o.z = fmaxf(o.z + pv.z, 0.f); o.w = fmaxf(o.w + pv.w, 0.f);
        }
        if constexpr (LAZY) {
            // dz3 of slot u of the wave's point (gpe_dz3_kernel's arithmetic): the message that won the aggregation carries s * g
            const x6_f32x2 a01 = __builtin_convertvector(__builtin_bit_cast(x6_f16x2, __float_as_uint(o.x)), x6_f32x2);
            const x6_f32x2 a23 = __builtin_convertvector(__builtin_bit_cast(x6_f16x2, __float_as_uint(o.y)), x6_f32x2);
            const float av[4] = {a01[0], a01[1], a23[0], a23[1]};
            if (u == 0) {                                    // (u is a compile-time constant at every call site) once per tile
                const float gq[4] = {lz_gq.x, lz_gq.y, lz_gq.z, lz_gq.w};
                const int sx[4] = {lz_sx.x, lz_sx.y, lz_sx.z, lz_sx.w}, sn[4] = {lz_sn.x, lz_sn.y, lz_sn.z, lz_sn.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    lz_sel[t] = (lzs[t] >= 0.f) ? sx[t] : sn[t];
                    lz_won[t] = __builtin_fmaf(lzs[t], gq[t], lznc[t]);
                }
            }
            float dz[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float base = (lz_sel[t] == u) ? lz_won[t] : lznc[t];
                dz[t] = (av[t] > 0.f) ? __builtin_fmaf(lznk[t], av[t], base) : 0.f;
            }
            o = make_float4(dz[0], dz[1], dz[2], dz[3]);
        }
        if (r >= s_rv) o = make_float4(0.f, 0.f, 0.f, 0.f);      // rows past the end of a partial last tile
        if constexpr (PLANES) {
            // the row's quad, normalised and split once: 4 values -> 8 bytes in each plane
            unsigned q0[SP::P], q1[SP::P];
            const float sc = LAZY ? 1.f : sA;                    // LAZY: dz3 was formed from pre-scaled coefficients
            SP::split2(o.x * sc, o.y * sc, q0);
            SP::split2(o.z * sc, o.w * sc, q1);
            char* row = reinterpret_cast<char*>(An) + r * PPITCH + 2 * c;
            *reinterpret_cast<uint2*>(row) = make_uint2(q0[0], q1[0]);
            *reinterpret_cast<uint2*>(row + PLANE) = make_uint2(q0[1], q1[1]);
        } else
            st4(&An[r * LDA + c], o);
    };
    // ---- epilogue of row u (compile-time u) of the tile being finished ------------------------------------------------
    auto epi_row = [&](int u, const float4 zraw) {
        if (!K16 && u >= rwl) return;
        // the operand scales (powers of two) are undone inside the epilogue's fma: forward through invAW, backward through cs4
        const float4 z = zraw;
        const int r = rbl + u;
        if (r >= e_rv) return;               // K16: all 16 rows of the wave's point are valid or none is (uniform)
        const int slot = K16 ? u : es;
        if (n_on) {
            if (EMODE == E_EDGE_FWD) {
                const float vv[4] = {fmaxf(SP::SCALED ? __builtin_fmaf(z.x, invAW, bias4.x) : z.x + bias4.x, 0.f),
                                     fmaxf(SP::SCALED ? __builtin_fmaf(z.y, invAW, bias4.y) : z.y + bias4.y, 0.f),
                                     fmaxf(SP::SCALED ? __builtin_fmaf(z.z, invAW, bias4.z) : z.z + bias4.z, 0.f),
                                     fmaxf(SP::SCALED ? __builtin_fmaf(z.w, invAW, bias4.w) : z.w + bias4.w, 0.f)};
                // largest magnitude written: per row here — or, where the per-point maxima are tracked anyway, once per point below
                if (SP::SCALED && !TRACK) amax_run = fmaxf(fmaxf(amax_run, fmaxf(vv[0], vv[1])), fmaxf(vv[2], vv[3]));
                // gather variant: the activation rows stream out past L2 so that they do not evict the cloud's Q table
                // (counter fetch of this kernel 199 -> <145 MB against 109 MB compulsory, same run time: profiles/r02_b)
                if (AMODE == A_GATHER) {
                    const x6_u32x4 oq = {__float_as_uint(vv[0]), __float_as_uint(vv[1]), __float_as_uint(vv[2]), __float_as_uint(vv[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(oq, orsrc, c * 4, r * p.ldo * 4, 16);
                } else if constexpr (OUTH) {
                    // fp16 rows (RNE), 8 bytes per quad: the backward only forms dz3 from this tensor (mask + a tiny-coefficient term)
                    // (clamped to the largest finite fp16: an activation beyond 65504 must not become inf in the stored copy — the
                    // backward would turn it into NaN gradients; mx / mn, the statistics and the amax word keep the fp32 value)
                    const x6_f32x2 v01 = {fminf(vv[0], 65504.f), fminf(vv[1], 65504.f)}, v23 = {fminf(vv[2], 65504.f), fminf(vv[3], 65504.f)};
                    *reinterpret_cast<uint2*>(reinterpret_cast<_Float16*>(p.out) + (e_row0 + r) * p.ldo + c) =
                        make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(v01, x6_f16x2)),
                                   __builtin_bit_cast(unsigned, __builtin_convertvector(v23, x6_f16x2)));
                } else st4(p.out + (e_row0 + r) * p.ldo + c, make_float4(vv[0], vv[1], vv[2], vv[3]));
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    s32[t] += vv[t];
                    q32[t] = __builtin_fmaf(vv[t], vv[t], q32[t]);
                    if constexpr (TRACK) {
                        if (vv[t] > vmx[t]) { vmx[t] = vv[t]; imx[t] = slot; }
                        if (vv[t] < vmn[t]) { vmn[t] = vv[t]; imn[t] = slot; }
                    }
                }
            } else {
                float4 av = act[u];
                if (GATHER_ACT) {
                    const float4 pv = K16 ? pve0 : x6_sel4(pve0, pve1, pve2, pve3, (u * rkl) >> 16);
                    av.x = fmaxf(av.x + pv.x, 0.f); av.y = fmaxf(av.y + pv.y, 0.f);
                    av.z = fmaxf(av.z + pv.z, 0.f); av.w = fmaxf(av.w + pv.w, 0.f);
                }
                float4 o;
                o.x = (av.x > 0.f) ? __builtin_fmaf(z.x, cs4.x, __builtin_fmaf(k24.x, av.x, c14.x)) : 0.f;
                o.y = (av.y > 0.f) ? __builtin_fmaf(z.y, cs4.y, __builtin_fmaf(k24.y, av.y, c14.y)) : 0.f;
                o.z = (av.z > 0.f) ? __builtin_fmaf(z.z, cs4.z, __builtin_fmaf(k24.z, av.z, c14.z)) : 0.f;
                o.w = (av.w > 0.f) ? __builtin_fmaf(z.w, cs4.w, __builtin_fmaf(k24.w, av.w, c14.w)) : 0.f;
                st4(p.out + (e_row0 + r) * p.ldo + c, o);
                if (SP::SCALED && EMODE == E_BWD_INPLACE)
                    amax_run = fmaxf(fmaxf(amax_run, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
                dp.x += o.x; dp.y += o.y; dp.z += o.z; dp.w += o.w;
            }
        }
        if (K16 ? (u == X6_PB - 1) : (++es == p.k)) {        // a point is complete (K16: compile-time)
            if (n_on) {
                const long gpt = e_pt0 + (K16 ? 0 : ept);
                if constexpr (TRACK && SP::SCALED) amax_run = fmaxf(fmaxf(amax_run, fmaxf(vmx[0], vmx[1])), fmaxf(vmx[2], vmx[3]));
                if (TRACK && (AGGT >= 1 || p.agg)) {
                    const long o = gpt * p.oldagg + c;
                    st4(p.mx + o, make_float4(vmx[0], vmx[1], vmx[2], vmx[3]));
                    st4(p.mn + o, make_float4(vmn[0], vmn[1], vmn[2], vmn[3]));
                    *reinterpret_cast<uchar4*>(p.oamx + o) = make_uchar4(imx[0], imx[1], imx[2], imx[3]);
                    *reinterpret_cast<uchar4*>(p.oamn + o) = make_uchar4(imn[0], imn[1], imn[2], imn[3]);
                }
                if (EMODE == E_BWD_GATHER) st4(p.dP + gpt * p.lddp + c, dp);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) { vmx[t] = -INFINITY; vmn[t] = INFINITY; imx[t] = 0; imn[t] = 0; }
            dp = make_float4(0.f, 0.f, 0.f, 0.f);
            es = 0; ++ept;
        }
    };
    auto epi_flush_stats = [&]() {
        if (EMODE == E_EDGE_FWD) {
#pragma unroll
            for (int t = 0; t < 4; ++t) { stS[t] += (double)s32[t]; stQ[t] += (double)q32[t]; }
        }
    };

    // ---- tile sequence of this workgroup --------------------------------------------------------------------------------
    // unpinned: blockIdx.x, +gridDim.x, ...   pinned (p.pin_tpc > 0; gridDim.x % 8 == 0, B % 8 == 0, tiles never straddle
    // clouds): this workgroup sits on XCD x = blockIdx.x % 8 and takes every (gridDim.x/8)-th tile of clouds x, x+8, ... —
    // the gathered Q table of a cloud (3.3 MB at the shipped sizes) is then read through ONE 4 MiB L2 instead of eight.
    // Sequence positions past the end map to tile numbers >= num_tiles in both modes.
    // p.rev: the same sequence from the far end (pinned: the XCD's clouds last to first; unpinned: mirrored tile numbers) — the
    // kernel that consumes a tensor starts on what its producer touched last (gpe_common.h)
    GpeTileSeq sq = gpe_tile_seq(p.pin_tpc, p.rev, p.pin_clouds, p.num_tiles);
    auto seq_tile = [&]() -> int { return gpe_seq_tile(sq); };
    auto seq_advance = [&]() { gpe_seq_advance(sq); };                        // host: gridDim.x / 8 <= pin_tpc

    // ---- prologue: stage tile 0 ---------------------------------------------------------------------------------------
    __syncthreads();                                     // A buffers zeroed
    int tile = seq_tile();
    seq_advance();
    int next = seq_tile();
    seq_advance();
    int next2 = seq_tile();
    if (tile < p.num_tiles) {
        if (AMODE == A_GATHER) jgv_s = load_jgv(tile);
        if (GATHER_ACT) jgv_e = load_jgv(tile);
    }
    if (tile < p.num_tiles && !(p.dbg & 1)) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            issue_stage_loads(tile, h);
#pragma unroll
            for (int uu = 0; uu < RBH; ++uu) commit_row(Abuf0, h * RBH + uu);
        }
    }
    if (AMODE == A_GATHER && tile < p.num_tiles) jgv_s = load_jgv(next < p.num_tiles ? next : tile);
    __syncthreads();

    int buf = 0, prev = -1;
    for (; tile < p.num_tiles; tile = next, next = next2, seq_advance(), next2 = seq_tile()) {
        X6_REFRESH_SCALARS()
        const float* As = buf ? Abuf1 : Abuf0;
        float* An = buf ? Abuf0 : Abuf1;
        const bool do_epi = prev >= 0 && !(p.dbg & 2);
        const bool do_stage = next < p.num_tiles && !(p.dbg & 1);

        f32x4 acc[4][AQ], accL[BQ > 0 ? BQ : 1][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int i = 0; i < AQ; ++i) acc[mt][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < (BQ > 0 ? BQ : 1); ++b) accL[b][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

        // A fragment of slot (sl, mt): 8 consecutive k of row 16 mt + j.  fp32 tile: read raw, split right before use; PLANES: the
        // fragment's planes are read as they are.  In the 16-wide tail slab the lane groups g >= 2 lie past the row: they read a
        // valid address and contribute zeros (their weights are zero as well).
        float4 raw0, raw1;
        X6Frag<SP> nf;
        auto read_raw = [&](int sl, int mt) {
            if constexpr (ROT) {
                // iteration sl of this wave = slab (sl + wave) mod KS (wave-uniform).  In the 16-wide tail slab the lane groups
                // g >= 2 lie past the row: they read a valid chunk (finite data) against the zero weights load_frag gave them
                const int slr = (sl + wave >= KS) ? sl + wave - KS : sl + wave;
                const int ge = (KTAIL && slr == KS - 1) ? g_tail : g;
                const char* src = reinterpret_cast<const char*>(As) + (16 * mt + j) * PPITCH + 16 * (4 * slr + ge);
#pragma unroll
                for (int t = 0; t < SP::P; ++t) nf.pl[t] = *reinterpret_cast<const x6_u32x4*>(src + t * PLANE);
                return;
            }
            const bool dead = KTAIL && sl == KS - 1 && g >= 2;
            if constexpr (PLANES) {
                const char* src = reinterpret_cast<const char*>(As) + (16 * mt + j) * PPITCH + 16 * (4 * sl + (dead ? (g & 1) : g));
#pragma unroll
                for (int t = 0; t < SP::P; ++t) {
                    nf.pl[t] = *reinterpret_cast<const x6_u32x4*>(src + t * PLANE);
                    if (dead) nf.pl[t] = (x6_u32x4){0u, 0u, 0u, 0u};
                }
            } else {
                const float* src = &As[(16 * mt + j) * LDA + 32 * sl + 8 * (dead ? (g & 1) : g)];
                raw0 = ld4(src); raw1 = ld4(src + 4);
                if (dead) { raw0 = make_float4(0.f, 0.f, 0.f, 0.f); raw1 = raw0; }
            }
        };
        float4 zq[EPR];
#pragma unroll
        for (int q = 0; q < EPR; ++q) zq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        read_raw(0, 0);

#pragma unroll
        for (int sl = 0; sl < KS; ++sl) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int q = 4 * sl + mt;
                X6Frag<SP> af;
                if constexpr (PLANES) af = nf; else af = x6_split8<SP>(raw0, raw1);
                if (q + 1 < NSLOT) read_raw((q + 1) >> 2, (q + 1) & 3);
                // ---- this slot's slice of the memory pipeline ----
                if (q == 0) {
                    issue_epi_loads(prev >= 0 ? prev : tile);            // clamped: results unused when !do_epi
                    issue_stage_loads(next < p.num_tiles ? next : tile, 0);  // clamped: results unused when !do_stage
                    if (GATHER_ACT) jgv_e = load_jgv(tile);              // this tile is finished in the next iteration
                    if (AMODE == A_GATHER) jgv_s = load_jgv(next2 < p.num_tiles ? next2 : tile);
                }
                if (ACT_LATE && EMODE != E_EDGE_FWD && q >= ACT_SHIFT && q < X6_PB + ACT_SHIFT) issue_act_load(q - ACT_SHIFT);
                if (NH == 1 && q >= CM_START && q < EP_START) {
                    if (do_stage) {
#pragma unroll
                        for (int c4 = 0; c4 < CMR; ++c4) {
                            const int u = (q - CM_START) * CMR + c4;
                            if (u < X6_PB) commit_row(An, u);
                        }
                    }
                }
                if (NH == 2) {
                    if ((q == CM_START || q == CM_START + 1) && do_stage) {
#pragma unroll
                        for (int c4 = 0; c4 < RBH / 2; ++c4) commit_row(An, (q - CM_START) * (RBH / 2) + c4);
                    }
                    if (q == CM_START + 2) issue_stage_loads(next < p.num_tiles ? next : tile, 1);
                    if (q >= NSLOT - 2 && do_stage) {
#pragma unroll
                        for (int c4 = 0; c4 < RBH / 2; ++c4) commit_row(An, RBH + (q - (NSLOT - 2)) * (RBH / 2) + c4);
                    }
                }
                if (q >= EP_START) {
                    if (do_epi) {
#pragma unroll
                        for (int e2 = 0; e2 < EPR; ++e2) {
                            const int u = (q - EP_START) * EPR + e2;
                            if (u < X6_PB) epi_row(u, zq[e2]);
                        }
                    }
                }
                if (q + 1 >= EP_START && q + 1 < NSLOT) {                // C rows of the NEXT slot's epilogue (LDS prefetch)
#pragma unroll
                    for (int e2 = 0; e2 < EPR; ++e2) {
                        const int u = (q + 1 - EP_START) * EPR + e2;
                        const int rr = rbl + ((u < X6_PB) ? u : X6_PB - 1);
                        zq[e2] = ld4(&Cs[((rr < RG_BM) ? rr : RG_BM - 1) * LDC + cn]);
                    }
                }
                if (X6_FENCE(PLANES)) __builtin_amdgcn_sched_barrier(0);   // memory slice stays in front of this slot's MFMAs
                // ---- SP::NPROD plane products per (tile, slab): small terms first; tiles innermost = independent accumulators ----
#pragma unroll
                for (int t = 0; t < SP::NPROD; ++t)
#pragma unroll
                    for (int i = 0; i < AQ; ++i) acc[mt][i] = SP::mfma(af.pl[SP::pa(t)], wP[SP::pw(t)][i][sl], acc[mt][i]);
                if constexpr (ROT) {
                    if ((sl & 3) == 0) {                     // compile time: iterations 0 and 4 carry this wave's slabs wave, wave + 4
#pragma unroll
                        for (int t = 0; t < SP::NPROD; ++t)
#pragma unroll
                            for (int b = 0; b < BQ; ++b) accL[b][mt] = SP::mfma(af.pl[SP::pa(t)], lP[SP::pw(t)][b][sl >> 2], accL[b][mt]);
                    }
                } else if constexpr (!TAILLOOP) {
                    if (BQ > 0 && (sl & 3) == wave) {        // this wave's K slab of the left-over tiles
#pragma unroll
                        for (int t = 0; t < SP::NPROD; ++t)
#pragma unroll
                            for (int b = 0; b < BQ; ++b) accL[b][mt] = SP::mfma(af.pl[SP::pa(t)], lP[SP::pw(t)][b][sl >> 2], accL[b][mt]);
                    }
                }
                if (X6_FENCE(PLANES)) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (TAILLOOP) {
            // LEFT == 1: the left-over tiles' K slabs of this wave (wave, wave + 4) in a loop of their own, branch-free: a slab past
            // the end multiplies the zero weights it was given in the prologue against a valid (clamped) address, and the A
            // fragments are read a second time (2 LSL x 4 ds_read_b128 per plane pair — the tile is in LDS anyway).
#pragma unroll
            for (int q = 0; q < LSL; ++q) {
                const int slr = wave + 4 * q;                              // wave-uniform
                const int slc = slr < KS ? slr : KS - 1;
                const bool dead = KTAIL && slc == KS - 1 && g >= 2;        // 16-wide tail slab: lane groups past the row read a
                                                                           // valid chunk (finite data) against zero weights
                const char* base = reinterpret_cast<const char*>(As) + j * PPITCH + 16 * (4 * slc + (dead ? (g & 1) : g));
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    X6Frag<SP> lf;
#pragma unroll
                    for (int t = 0; t < SP::P; ++t) lf.pl[t] = *reinterpret_cast<const x6_u32x4*>(base + 16 * mt * PPITCH + t * PLANE);
#pragma unroll
                    for (int t = 0; t < SP::NPROD; ++t)
#pragma unroll
                        for (int b = 0; b < BQ; ++b) accL[b][mt] = SP::mfma(lf.pl[SP::pa(t)], lP[SP::pw(t)][b][q], accL[b][mt]);
                }
            }
        }
        if (do_epi) epi_flush_stats();
        __syncthreads();                                 // (1) every wave is done with C (epilogue of the previous tile) and with As
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < AQ; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Cs[(16 * mt + 4 * g + r) * LDC + 16 * (AQ * wave + i) + j] = acc[mt][i][r];
        if (BQ > 0) {
            // left-over tiles: this wave's K-partial of row r goes to the scratch slot of that row INSIDE the consumed A buffer
            // (columns [16 (wave BQ + b), +16)); the row's owner sums the four partials after the barrier
            float* Sc = const_cast<float*>(As);
#pragma unroll
            for (int b = 0; b < BQ; ++b)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Sc[x6_scr<PLANES, BQ, LDA, PPITCH, PLANE>(16 * mt + 4 * g + r, wave, b) + j] = accL[b][mt][r];
        }
        __syncthreads();                                 // (2) C complete, partials complete, next A tile complete
        if (BQ > 0) {
            // rows rb .. rb + rows_w - 1 are mine: I finish them, and I am the only wave that re-stages them in this buffer
            const int ur = lane >> 2, cq = (lane & 3) << 2;
            if (ur < rwl) {
#pragma unroll
                for (int b = 0; b < BQ; ++b) {
                    const float4 p0 = ld4(&As[x6_scr<PLANES, BQ, LDA, PPITCH, PLANE>(rbl + ur, 0, b) + cq]);
                    const float4 p1 = ld4(&As[x6_scr<PLANES, BQ, LDA, PPITCH, PLANE>(rbl + ur, 1, b) + cq]);
                    const float4 p2 = ld4(&As[x6_scr<PLANES, BQ, LDA, PPITCH, PLANE>(rbl + ur, 2, b) + cq]);
                    const float4 p3 = ld4(&As[x6_scr<PLANES, BQ, LDA, PPITCH, PLANE>(rbl + ur, 3, b) + cq]);
                    float4 o;
                    o.x = (p0.x + p1.x) + (p2.x + p3.x); o.y = (p0.y + p1.y) + (p2.y + p3.y);
                    o.z = (p0.z + p1.z) + (p2.z + p3.z); o.w = (p0.w + p1.w) + (p2.w + p3.w);
                    st4(&Cs[(rbl + ur) * LDC + 16 * (4 * AQ + b) + cq], o);
                }
            }
        }
        prev = tile;
        buf ^= 1;
    }
    // ---- tail: epilogue of the last tile -------------------------------------------------------------------------------
    if (prev >= 0 && !(p.dbg & 2)) {
        issue_epi_loads(prev);
        if (ACT_LATE) {
#pragma unroll
            for (int u = 0; u < X6_PB; ++u) issue_act_load(u);
        }
#pragma unroll
        for (int u = 0; u < X6_PB; ++u) {
            const int rr = rbl + u;
            epi_row(u, ld4(&Cs[((rr < RG_BM) ? rr : RG_BM - 1) * LDC + cn]));
        }
        epi_flush_stats();
    }
    if (SP::SCALED && p.amax_out && EMODE != E_BWD_GATHER) {
        // largest magnitude this workgroup wrote (non-negative floats order like their bit patterns; a NaN sorts above inf)
        float m = amax_run;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) amax_sh[wave] = __float_as_uint(m);
    }
    __syncthreads();
    if (SP::SCALED && p.amax_out && EMODE != E_BWD_GATHER && tid == 0) {
        // one atomic per workgroup (same-address atomics issued by every wave at once serialise in the L2)
        const unsigned a = amax_sh[0] > amax_sh[1] ? amax_sh[0] : amax_sh[1], b = amax_sh[2] > amax_sh[3] ? amax_sh[2] : amax_sh[3];
        atomicMax(p.amax_out, a > b ? a : b);
    }
    if (EMODE == E_EDGE_FWD && p.stats_part) {
        double* red = reinterpret_cast<double*>(smem);          // [4 waves][2][16*NT]
        if (n_on) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                red[(wave * 2 + 0) * (16 * NT) + c + t] = stS[t];
                red[(wave * 2 + 1) * (16 * NT) + c + t] = stQ[t];
            }
        }
        __syncthreads();
        if (tid < p.N) {
            constexpr int NC = 16 * NT;
            const double ss = (red[0 * NC + tid] + red[2 * NC + tid]) + (red[4 * NC + tid] + red[6 * NC + tid]);
            const double qq = (red[1 * NC + tid] + red[3 * NC + tid]) + (red[5 * NC + tid] + red[7 * NC + tid]);
            for (int b = blockIdx.x; b < stats_nblk; b += gridDim.x) {
                double* dst = p.stats_part + (size_t)b * 2 * p.N;
                dst[tid] = (b == (int)blockIdx.x) ? ss : 0.0;
                dst[p.N + tid] = (b == (int)blockIdx.x) ? qq : 0.0;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Left-over scheme (template parameter LEFT) per kernel kind of the k = 16 two-plane instances.  GPE_H3_LEFT = four digits
// "F2 F3 B3 B2" (gather forward, dense forward, in-place backward, gathered backward), each 0 or 2, overrides the table
// (scheme 1 was measured — profiles/r04_c_left_schemes.md — never the fastest, and is no longer instantiated).
static int x6_left_scheme(int amode, int emode)
{
    static int tab[4] = {-1, 0, 0, 0};
    if (tab[0] < 0) {
        const int def[4] = {X6_LEFT_F2, X6_LEFT_F3, X6_LEFT_B3, X6_LEFT_B2};
        const char* e = gpe_dbg_env_str("GPE_H3_LEFT");
        for (int i = 3; i >= 0; --i) {
            int v = def[i];
            if (e && strlen(e) == 4 && (e[i] == '0' || e[i] == '2')) v = e[i] - '0';
            tab[i] = v;
        }
    }
    const int kind = (emode == E_EDGE_FWD) ? (amode == A_GATHER ? 0 : 1) : (emode == E_BWD_INPLACE ? 2 : 3);
    return tab[kind];
}

template <class SP, int AQ, int BQ, int KCH, int AMODE, int EMODE, bool K16, bool PSEUDO = false, int AGGT = -1, int LEFT = 0,
          bool LAZY = false>
static int x6_launch_k(const RgParams& p, int stats_nblk, hipStream_t s)
{
    constexpr int NT = 4 * AQ + BQ;
    constexpr int LDA = 16 * KCH + 4, LDC = x6_ldc<SP>(NT, KCH);
    constexpr int AWORDS = SP::SCALED ? (2 * RG_BM * 16 * x6_pchunks(KCH)) / 4 : RG_BM * LDA;
    const size_t lds = (size_t)(2 * AWORDS + RG_BM * LDC) * sizeof(float);
    // 16 bytes of static __shared__ (amax_sh) sit beside the dynamic image
    GPE_ENSURE_MAX_LDS_N((gpe_edgegemm_split_kernel<SP, AQ, BQ, KCH, AMODE, EMODE, K16, PSEUDO, AGGT, LEFT, LAZY>), 160 * 1024 - 64);
    int gx = gpe_num_cus();
    if (gx > p.num_tiles) gx = p.num_tiles;
    if (stats_nblk > 0 && gx > stats_nblk) gx = stats_nblk;
    hipLaunchKernelGGL((gpe_edgegemm_split_kernel<SP, AQ, BQ, KCH, AMODE, EMODE, K16, PSEUDO, AGGT, LEFT, LAZY>), dim3(gx), dim3(256), lds, s, p, stats_nblk);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

template <class SP, int AQ, int BQ, int KCH, int AMODE, int EMODE>
static int x6_launch(const RgParams& p, int stats_nblk, hipStream_t s)
{
    // only the k = 16 aggregated dense forward of the scaled policy stores fp16 rows / only its in-place backward forms dz3 lazily
    if (p.out_half && !(SP::SCALED && EMODE == E_EDGE_FWD && AMODE == A_DENSE && p.k == 16 && p.agg && !p.pmagic)) return GPE_EINVAL;
    if (p.lz_g && !(SP::SCALED && EMODE == E_BWD_INPLACE && AMODE == A_DENSE && p.k == 16)) return GPE_EINVAL;
    if constexpr (EMODE != E_BWD_INPLACE) {          // the in-place backward needs nothing per point: never pseudo-points
        if (p.pmagic)
            return p.k == 16 ? x6_launch_k<SP, AQ, BQ, KCH, AMODE, EMODE, true, true>(p, stats_nblk, s)
                             : x6_launch_k<SP, AQ, BQ, KCH, AMODE, EMODE, false, true>(p, stats_nblk, s);
    }
    if constexpr (SP::SCALED) {
        // the benchmark configuration (k = 16, whole points): forward with the aggregation tracking compiled in / out, and the
        // left-over scheme of the kernel kind (x6_left_scheme: GPE_H3_LEFT overrides for A/B measurements)
        if (p.k == 16) {
            const int left = x6_left_scheme(AMODE, EMODE);
            if constexpr (EMODE == E_EDGE_FWD) {
                if (p.agg) {
                    if constexpr (AMODE == A_DENSE) {
                        // fp16 activation rows (row g): the aggregated dense forward only
                        if (p.out_half) {
                            if (left == 2) return x6_launch_k<SP, AQ, BQ, KCH, AMODE, EMODE, true, false, 2, 2>(p, stats_nblk, s);
                            return x6_launch_k<SP, AQ, BQ, KCH, AMODE, EMODE, true, false, 2, 0>(p, stats_nblk, s);
                        }
                    }
                    if (left == 2) return x6_launch_k<SP, AQ, BQ, KCH, AMODE, EMODE, true, false, 1, 2>(p, stats_nblk, s);
                    return x6_launch_k<SP, AQ, BQ, KCH, AMODE, EMODE, true, false, 1, 0>(p, stats_nblk, s);
                }
                if (left == 2) return x6_launch_k<SP, AQ, BQ, KCH, AMODE, EMODE, true, false, 0, 2>(p, stats_nblk, s);
                return x6_launch_k<SP, AQ, BQ, KCH, AMODE, EMODE, true, false, 0, 0>(p, stats_nblk, s);
            } else {
                if constexpr (EMODE == E_BWD_INPLACE && AMODE == A_DENSE) {
                    // lazy dz3: always the rotated-slab instance (the in-loop scheme spills 33 registers to scratch with the extra
                    // per-lane coefficient quads; the rotated one does not)
                    if (p.lz_g) return x6_launch_k<SP, AQ, BQ, KCH, AMODE, EMODE, true, false, -1, 2, true>(p, stats_nblk, s);
                }
                if (left == 2) return x6_launch_k<SP, AQ, BQ, KCH, AMODE, EMODE, true, false, -1, 2>(p, stats_nblk, s);
                return x6_launch_k<SP, AQ, BQ, KCH, AMODE, EMODE, true, false, -1, 0>(p, stats_nblk, s);
            }
        }
    }
    return p.k == 16 ? x6_launch_k<SP, AQ, BQ, KCH, AMODE, EMODE, true>(p, stats_nblk, s)
                     : x6_launch_k<SP, AQ, BQ, KCH, AMODE, EMODE, false>(p, stats_nblk, s);
}

// Shape checks + re-tiling shared by both policies.  Returns 1 when the shape is on the single-role split kernels' menu (then
// `p` is the re-tiled copy: every wave owns whole points, R = 4 * npw * k with npw * k <= 16; `fold` says what to fold after the
// launch when a k > 16 point ran as pseudo-points), 0 when it is not.
static int x6_prepare(const RgParams& p_in, int amode, int emode, int stats_nblk, RgParams& p, GpeFold& fold)
{
    p = p_in;
    fold = GpeFold{};
    fold.f = 1;
    if (p.N <= 96 || p.N > 208 || p.K <= 96 || p.K > 208) return 0;
    if (emode != E_EDGE_FWD && (p.N & 3)) return 0;      // the backward epilogues use aligned 16-B coefficient loads
    if (amode == A_GATHER && (p.K & 3)) return 0;
    if (amode == A_DENSE && (p.a.inner > 0 || (p.a.stride_outer & 3) || p.a.stride_outer < ((p.K + 3) & ~3) ||
                             (((uintptr_t)p.a.base) & 15)))
        return 0;                                        // dense rows must be aligned + padded for plain 16-B loads
    if (p.k < 1) return 0;
    const bool per_point = amode == A_GATHER || emode == E_BWD_GATHER || (emode == E_EDGE_FWD && p.agg);
    // k > 16: rows that need nothing per point are tiled 4 rows per "point"; the per-point variants run a point as f pseudo-points
    // of <= 16 rows whose results are folded afterwards (gpe_edge_pseudo_setup / _fold, gpe_edgegemm_sr.hip)
    if (!gpe_edge_pseudo_setup(p, per_point, emode, fold)) return 0;
    const int npw = X6_PB / p.k;                         // points per wave per tile
    if (per_point && npw > X6_NPW) return 0;
    p.R = 4 * npw * p.k;
    p.num_tiles = gpe_cdiv(p.M, p.R);
    p.pin_tpc = 0;
    if ((amode == A_GATHER || emode == E_BWD_GATHER) && p.pin_clouds > 0 && gpe_pin_clouds(p.pin_clouds) &&
        p.pin_clouds % GPE_NXCD == 0) {
        // gather variants only (dense streaming tiles have nothing to keep in L2): tiles must not straddle clouds and
        // the launcher must keep gridDim.x a multiple of 8 with gridDim.x / 8 <= tiles per cloud
        const long rows_per_cloud = p.M / p.pin_clouds;
        const int gx = gpe_num_cus();
        if (rows_per_cloud % p.R == 0 && gx % GPE_NXCD == 0 && gx <= p.num_tiles &&
            (stats_nblk <= 0 || gx <= stats_nblk) && rows_per_cloud / p.R >= gx / GPE_NXCD)
            p.pin_tpc = (int)(rows_per_cloud / p.R);
    }
    return 1;
}
