// Reduce-GEMM family for gfx950:  G[Mg, Ng] = sum over rows r of U[r, :]^T V[r, :]   (+ colsum of U)
//
// This is the weight-gradient half of every Linear on the path (nn.Linear backward under
// /root/reference/nn/trainer.py:97 `loss.backward()`).  V rows are either dense or the EdgeConv gather
// relu(P_i + Q_j) rebuilt on the fly.  Because BatchNorm's backward reductions are linear in the same products,
// G and colsum(U) are also all that the BN backward of the previous block needs (gpe_bn_bwd_from_G) — no extra
// pass over the E edges.
//
// Structure: ONE persistent 256-thread workgroup per CU (grid = #CUs), 512-VGPR budget per wave.
//   * 32-row operand tiles are fetched global -> registers one tile AHEAD (the dependent idx -> Q-row gather
//     included), written to a double-buffered LDS image after the current tile's MFMAs: one barrier per tile, HBM/L2
//     latency fully under the matrix pipe;
//   * the 4 waves tile G 2x2: wave (wm, wn) keeps the accumulators of M-tiles [wm*MH, ..) x N-tiles [wn*NH, ..)
//     in registers across ALL its row tiles (v_mfma_f32_16x16x4_f32, reduction dim = rows); 13 x 13 tiles split
//     7/6 x 7/6 instead of 4/3/3/3 rows of 13 (86 % vs 81 % balance, half the operand reads);
//   * one partial per workgroup at the end; a second kernel sums the partials in a fixed order (fp64), so results
//     are run-to-run deterministic.
#include "gpe_common.h"
#include <stdlib.h>

long gpe_gemm_x6_red_ws(int Mg, int Ng);                                                 // gpe_gemm_x6.hip
int gpe_gemm_x6_redgemm(const GpeRows& u, const GpeRows& v, const float* v_shift, long rows, int Mg, int Ng, float* part, bool want_cs,
                        int* nsplit, int* MgPad, int* NgPad, double** part_cs, hipStream_t s);
extern "C" int gpe_debug_get(void);
static int g_rd_math = 0;            // 0: exact fp32 MFMA, 1: bf16x3, 2: f16x3 where the operand scales are known (gpe_math_set)
void gpe_redgemm_set_math(int m) { g_rd_math = m; }

#define RD_RT 32

enum { V_GATHER = 0, V_DENSE = 1 };

struct RdParams {
    long rows;
    int Mg, Ng, MgPad, NgPad;
    int num_tiles;
    GpeRows u;
    GpeRows v;                                   // V_DENSE
    const float* pq; int ldpq; int H; const int32_t* jg; int k; double rcp_k;   // V_GATHER (global neighbour rows)
    unsigned kmagic;                             // ceil(2^32 / k): row / k == umulhi(row, kmagic) while row * k < 2^32 (pc kernel)
    unsigned umagic, vmagic;                     // the same for u.inner / v.inner (2-level rows of the deep kernel)
    int pin_clouds;                              // B when the rows are B equal clouds (gpe_edge_redgemm), else 0
    const float* v_shift;                        // optional [Ng]: V := V - shift on valid rows (BN centring)
    int vec;                                     // rows aligned to 16 B and padded to 4 columns: plain 16-B loads
    int pin_tpc;                                 // gather variants of the pc/b3 kernels: tiles per cloud when pinned (gpe_common.h)
    int rev;                                     // walk the tile sequence from the far end (gpe_common.h GpeTileSeq)
    float* part;                                 // [gridDim.x][MgPad][NgPad]
    double* part_cs;                             // [gridDim.x][MgPad]
    // f16x3 variant of the b3 kernel: bit patterns of the largest magnitudes of U and of V - shift (device memory)
    const unsigned* amax_u;
    const unsigned* amax_v;
    // LAZY dz3 (f16x3, k = 16, dense V): U is the stored activation a3 of the aggregated block and the producers form dz3 from it
    // (gpe_edge_dz3's arithmetic; RgParams::lz_* of the edge kernels has the same fields).  NULL = off.
    const float* lz_g; int lz_ldg;
    const uint8_t* lz_amx; const uint8_t* lz_amn; int lz_ldagg;
    const float* lz_coef;                        // [4][Mg] = {s, c1, k2, mean}
};
// b3 kernel, compile-time switches (A/B builds through scripts/ab_build.sh; run-time flags cost this kernel registers it does not
// have: a run-time `pipe` flag spilled 120 of them in the gathered 13 x 13 instance):
//   RD_B3_PIPE  producers' order: 0 (default) = fetch tile t + 1 ... left-over MFMAs ... commit it (round 4); 1 = the two operands
//               alternate commit / fetch one tile further ahead, so that every load has most of an iteration to land (round 5).
//               MEASURED SLOWER: gathered 13 x 13 383 / 391 -> 425 / 416 us, dense 10 x 13 376 / 379 -> 382 / 378 (one session, twice
//               each, profiles/r05_b_redgemm_probe.md) — the producers are not waiting for memory: per tile and SIMD the kernel
//               issues 129 MFMAs (2.1 k cycles) next to ~510 producer VALU + 60 LDS / VMEM instructions, and those add up
//   RD_B3_DBG   timing-only ablation (WRONG RESULTS): 1 = no commit (split + LDS writes), 2 = no row loads, 4 = no consumer MFMAs,
//               8 = no left-over MFMAs
#ifndef RD_B3_PIPE
#define RD_B3_PIPE 0
#endif
#ifndef RD_B3_DBG
#define RD_B3_DBG 0
#endif
//   RD_B3_PRIO       s_setprio of the producer waves (the fp32 producer/consumer kernel runs them at 3)
//   RD_B3_LEFT_CONS  1 = the left-over tiles (odd row / column of the 13 x 13 / 10 x 13 tile grid) are multiplied by the CONSUMER waves
//                    (parked three quarters of the time) instead of the producers (the critical path of an iteration)
// Measured in one session (us per launch at cfg 2 inside the training step, gathered 13 x 13 / dense 10 x 13; profiles/r05_b_redgemm_probe.md):
//   round 4's kernel 381 / 369 (378 / 368) | producers at priority 3: 363 / 361 | at priority 1: 359 / 359 | left-overs on the consumers:
//   444 / 403 (13 x 13 spills 15 registers; the 10 x 13 instance, which does not, is slower as well) | both: 478 / 406.
// -> producers at priority 1; the left-overs stay with the producers: whatever either wave of a SIMD issues adds to the tile time.
#ifndef RD_B3_PRIO
#define RD_B3_PRIO 1
#endif
#ifndef RD_B3_LEFT_CONS
#define RD_B3_LEFT_CONS 0
#endif

__device__ __forceinline__ float4 rd_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

__device__ __forceinline__ float4 rd_ld4_guard(const float* p, int nvalid, bool vec)
{
    if (nvalid >= 4 && vec) return rd_ld4(p);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nvalid > 0) v.x = p[0];
    if (nvalid > 1) v.y = p[1];
    if (nvalid > 2) v.z = p[2];
    if (nvalid > 3) v.w = p[3];
    return v;
}

// MFMA body over one 32-row tile for a compile-time (MC x NC) block of 16x16 tiles — no branches
template <int MH, int NH, int MC, int NC>
__device__ __forceinline__ void rd_mma(const float* ub, const float* vb, int LDU, int LDV, int mt0, int nt0, int j,
                                       int g, f32x4 (&acc)[MH][NH])
{
#pragma unroll
    for (int r0 = 0; r0 < RD_RT; r0 += 4) {
        float a[MC], b[NC];
#pragma unroll
        for (int q = 0; q < MC; ++q) a[q] = ub[(r0 + g) * LDU + 16 * (mt0 + q) + j];
#pragma unroll
        for (int n = 0; n < NC; ++n) b[n] = vb[(r0 + g) * LDV + 16 * (nt0 + n) + j];
#pragma unroll
        for (int q = 0; q < MC; ++q)
#pragma unroll
            for (int n = 0; n < NC; ++n)
                acc[q][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[n], acc[q][n], 0, 0, 0);
    }
}

template <int MH, int NH, int VMODE>
__global__ __launch_bounds__(256, 1) void gpe_redgemm_kernel(RdParams p)
{
    constexpr int UC = 32 * MH;                    // U columns of this block (2*MH tiles of 16)
    constexpr int LDU = UC + 16;                   // stride == 16 (mod 32): conflict-free b32 operand reads
    constexpr int VC = 32 * NH;
    constexpr int LDV = VC + 16;
    extern __shared__ __align__(16) float smem[];
    float* Us = smem;                              // [2][RD_RT * LDU]
    float* Vs = smem + 2 * RD_RT * LDU;            // [2][RD_RT * LDV]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;
    const int m0 = blockIdx.y * UC;
    const int ucols = (p.Mg - m0 < UC) ? (p.Mg - m0) : UC;
    const int mt_blk = (ucols + 15) >> 4, nt_all = (p.Ng + 15) >> 4;
    const int mt0 = wm * MH, nt0 = wn * NH;
    const int mc = (mt_blk - mt0 < MH) ? (mt_blk - mt0) : MH;     // may be <= 0
    const int nc = (nt_all - nt0 < NH) ? (nt_all - nt0) : NH;

    f32x4 acc[MH][NH];
#pragma unroll
    for (int q = 0; q < MH; ++q)
#pragma unroll
        for (int n = 0; n < NH; ++n) acc[q][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    double cs = 0.0;

    // operand staging: lane = column quad, rows = wave + 4*q (wave-uniform), fetched global -> registers one tile
    // ahead of the MFMAs that consume them.  With p.vec (aligned, padded rows: every internal edge buffer) each row is
    // ONE plain 16-B load per lane, unconditional with a clamped row, so all RQ loads stay in flight; the guarded
    // scalar-tail loader (whose two paths force a wait between loads) is only for ragged external tensors.
    constexpr int RQ = RD_RT / 4;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int cq = lane << 2;
    const bool u_on = cq < UC && cq < ucols;
    const bool v_on = cq < VC && cq < p.Ng;
    float4 ur[RQ], vr[RQ], vr2[VMODE == V_GATHER ? RQ : 1];
    unsigned rmask = 0;
    float sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.v_shift && v_on) {
#pragma unroll
        for (int t = 0; t < 4; ++t) if (cq + t < p.Ng) sh[t] = p.v_shift[cq + t];
    }

    auto fetch = [&](int tile) {
        const long row0 = (long)tile * RD_RT;
        const int rv = (int)((p.rows - row0 < RD_RT) ? (p.rows - row0) : RD_RT);
        rmask = 0;
        if (p.vec) {
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                const int r = uwave + 4 * q;
                if (r < rv) rmask |= 1u << q;
                const long gr = row0 + ((r < rv) ? r : rv - 1);          // clamped: the load is unconditional
                if (u_on) ur[q] = rd_ld4(p.u.base + gr * p.u.stride_outer + m0 + cq);
                if (v_on) {
                    if (VMODE == V_DENSE) vr[q] = rd_ld4(p.v.base + gr * p.v.stride_outer + cq);
                    else {
                        const long i = (long)gpe_udiv((unsigned)gr, (unsigned)p.k, p.rcp_k);
                        const long jj = p.jg[gr];
                        vr[q] = rd_ld4(p.pq + i * p.ldpq + cq);
                        vr2[q] = rd_ld4(p.pq + jj * p.ldpq + p.H + cq);
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
            const int r = uwave + 4 * q;
            ur[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            vr[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (VMODE == V_GATHER) vr2[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < rv) {
                rmask |= 1u << q;
                const long gr = row0 + r;
                if (u_on) {
                    const float* src = gpe_row_ptr(p.u, gr) + m0 + cq;
                    ur[q] = rd_ld4_guard(src, ucols - cq, gpe_aligned16(src));
                }
                if (v_on) {
                    if (VMODE == V_DENSE) {
                        const float* src = gpe_row_ptr(p.v, gr) + cq;
                        vr[q] = rd_ld4_guard(src, p.Ng - cq, gpe_aligned16(src));
                    } else {
                        const long i = (long)gpe_udiv((unsigned)gr, (unsigned)p.k, p.rcp_k);
                        const long jj = p.jg[gr];
                        vr[q] = rd_ld4(p.pq + i * p.ldpq + cq);               // H % 4 == 0, ldpq % 4 == 0
                        vr2[q] = rd_ld4(p.pq + jj * p.ldpq + p.H + cq);
                    }
                }
            }
        }
    };
    auto commit = [&](int buf) {
        float* ub = Us + buf * RD_RT * LDU;
        float* vb = Vs + buf * RD_RT * LDV;
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
            const int r = uwave + 4 * q;
            const bool ok = (rmask >> q) & 1u;
            if (cq < UC) {
                float4 u = ur[q];
                if (!(ok && u_on)) u = make_float4(0.f, 0.f, 0.f, 0.f);
                else {                                              // ragged last quad (padded columns are not data)
                    if (cq + 1 >= ucols) u.y = 0.f;
                    if (cq + 2 >= ucols) u.z = 0.f;
                    if (cq + 3 >= ucols) u.w = 0.f;
                }
                *reinterpret_cast<float4*>(&ub[r * LDU + cq]) = u;
            }
            if (cq < VC) {
                float4 v = vr[q];
                if (VMODE == V_GATHER) {
                    v.x = fmaxf(v.x + vr2[q].x, 0.f); v.y = fmaxf(v.y + vr2[q].y, 0.f);
                    v.z = fmaxf(v.z + vr2[q].z, 0.f); v.w = fmaxf(v.w + vr2[q].w, 0.f);
                }
                v.x -= sh[0]; v.y -= sh[1]; v.z -= sh[2]; v.w -= sh[3];
                if (!(ok && v_on)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                else {
                    if (cq + 1 >= p.Ng) v.y = 0.f;
                    if (cq + 2 >= p.Ng) v.z = 0.f;
                    if (cq + 3 >= p.Ng) v.w = 0.f;
                }
                *reinterpret_cast<float4*>(&vb[r * LDV + cq]) = v;
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < p.num_tiles) fetch(tile);
    int buf = 0;
    for (; tile < p.num_tiles; tile += gridDim.x) {
        commit(buf);
        __syncthreads();            // tile visible; every wave is past the MFMAs that read buffer buf^1
        // in flight under the MFMAs below.  Unconditional (the last iteration re-fetches its own tile and drops it):
        // registers loaded under a branch are copied at the join, which waits for the loads before the first MFMA
        fetch(tile + (int)gridDim.x < p.num_tiles ? tile + (int)gridDim.x : tile);
        const float* ub = Us + buf * RD_RT * LDU;
        const float* vb = Vs + buf * RD_RT * LDV;
        if (tid < ucols) {
            float c32[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < RD_RT; r += 4) {
                c32[0] += ub[r * LDU + tid]; c32[1] += ub[(r + 1) * LDU + tid];
                c32[2] += ub[(r + 2) * LDU + tid]; c32[3] += ub[(r + 3) * LDU + tid];
            }
            cs += ((double)c32[0] + (double)c32[1]) + ((double)c32[2] + (double)c32[3]);
        }
        // the wave's block is (mc x nc) tiles with mc in {MH, MH-1}, nc in {NH, NH-1} in every shipped shape: four
        // branch-free specialisations; anything else takes the guarded generic body
        if (mc == MH && nc == NH) rd_mma<MH, NH, MH, NH>(ub, vb, LDU, LDV, mt0, nt0, j, g, acc);
        else if (mc == MH && nc == NH - 1) rd_mma<MH, NH, MH, (NH > 1 ? NH - 1 : 1)>(ub, vb, LDU, LDV, mt0, nt0, j, g, acc);
        else if (mc == MH - 1 && nc == NH) rd_mma<MH, NH, (MH > 1 ? MH - 1 : 1), NH>(ub, vb, LDU, LDV, mt0, nt0, j, g, acc);
        else if (mc == MH - 1 && nc == NH - 1)
            rd_mma<MH, NH, (MH > 1 ? MH - 1 : 1), (NH > 1 ? NH - 1 : 1)>(ub, vb, LDU, LDV, mt0, nt0, j, g, acc);
        else if (mc > 0 && nc > 0) {
#pragma unroll
            for (int r0 = 0; r0 < RD_RT; r0 += 4) {
                float a[MH], b[NH];
#pragma unroll
                for (int q = 0; q < MH; ++q) a[q] = ub[(r0 + g) * LDU + 16 * (mt0 + q) + j];
#pragma unroll
                for (int n = 0; n < NH; ++n) b[n] = vb[(r0 + g) * LDV + 16 * (nt0 + n) + j];
#pragma unroll
                for (int q = 0; q < MH; ++q) {
                    if (q < mc) {
#pragma unroll
                        for (int n = 0; n < NH; ++n)
                            if (n < nc)
                                acc[q][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[n], acc[q][n], 0, 0, 0);
                    }
                }
            }
        }
        buf ^= 1;
    }

    // ---- one partial per workgroup ---------------------------------------------------------------------
    float* dst = p.part + (size_t)blockIdx.x * p.MgPad * p.NgPad;
#pragma unroll
    for (int q = 0; q < MH; ++q) {
#pragma unroll
        for (int n = 0; n < NH; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * (mt0 + q) + 4 * g + r, nn = 16 * (nt0 + n) + j;
                if (q < MH && m < p.MgPad && nn < p.NgPad) dst[(size_t)m * p.NgPad + nn] = acc[q][n][r];
            }
    }
    if (tid < ucols) p.part_cs[(size_t)blockIdx.x * p.MgPad + m0 + tid] = cs;
}

// ---------------------------------------------------------------------------------------------------------
// Producer/consumer variant for the two edge shapes (tile grids 13x13 and 10x13): one persistent 512-thread workgroup
// per CU, two waves per SIMD.
//   consumers (waves 0-3): 2x2 blocks of (MT/2) x (NT/2) accumulator tiles, nothing but ds_read_b32 operands + MFMAs;
//   producers (waves 4-7): fetch the next 32-row tile (plain 16-B loads of aligned rows; gather = P_i + Q_j), commit it
//                          to the other LDS buffer, keep the fp64 column sums of U from their staging registers, and
//                          compute the left-over tile row / column (MT or NT odd) so each SIMD carries ~MT*NT/4 tiles.
// One barrier per tile.  Requires p.vec (aligned, 4-padded rows) — true for every internal edge buffer.
// ---------------------------------------------------------------------------------------------------------
template <int MT, int NT, int VMODE>
__global__ __launch_bounds__(512, 2) void gpe_redgemm_pc_kernel(RdParams p)
{
    constexpr int MB = MT / 2, NB = NT / 2;
    constexpr int LEFT = (MT & 1) * NT + (NT & 1) * (MT - (MT & 1));     // left-over tiles
    constexpr int PMAX = (LEFT + 3) / 4;
    constexpr int UC = 16 * MT, VC = 16 * NT;
    constexpr int LDU = (UC % 32 == 16) ? UC : UC + 16;
    constexpr int LDV = (VC % 32 == 16) ? VC : VC + 16;
    constexpr int RQ = RD_RT / 4;
    extern __shared__ __align__(16) float smem[];
    float* Us = smem;                              // [2][RD_RT * LDU]
    float* Vs = smem + 2 * RD_RT * LDU;            // [2][RD_RT * LDV]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w4 = wave & 3;
    const int j = lane & 15, g = lane >> 4;
    float* dst = p.part + (size_t)blockIdx.x * p.MgPad * p.NgPad;

    if (wave < 4) {
        // ================================ consumers ================================
        const int mt0 = (w4 & 1) * MB, nt0 = (w4 >> 1) * NB;
        f32x4 acc[MB][NB];
#pragma unroll
        for (int q = 0; q < MB; ++q)
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[q][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __syncthreads();                           // prologue: tile 0 staged
        int buf = 0;
        GpeTileSeq sq = gpe_tile_seq(p.pin_tpc, p.rev, p.pin_clouds, p.num_tiles);
        for (int tile = gpe_seq_tile(sq); tile < p.num_tiles; gpe_seq_advance(sq), tile = gpe_seq_tile(sq)) {
            const float* ub = Us + buf * RD_RT * LDU;
            const float* vb = Vs + buf * RD_RT * LDV;
#pragma unroll
            for (int r0 = 0; r0 < RD_RT; r0 += 4) {
                float a[MB], b[NB];
#pragma unroll
                for (int q = 0; q < MB; ++q) a[q] = ub[(r0 + g) * LDU + 16 * (mt0 + q) + j];
#pragma unroll
                for (int n = 0; n < NB; ++n) b[n] = vb[(r0 + g) * LDV + 16 * (nt0 + n) + j];
#pragma unroll
                for (int q = 0; q < MB; ++q)
#pragma unroll
                    for (int n = 0; n < NB; ++n)
                        acc[q][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[n], acc[q][n], 0, 0, 0);
            }
            __syncthreads();
            buf ^= 1;
        }
#pragma unroll
        for (int q = 0; q < MB; ++q)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    dst[(size_t)(16 * (mt0 + q) + 4 * g + r) * p.NgPad + 16 * (nt0 + n) + j] = acc[q][n][r];
        __syncthreads();                           // tail: producers' column sums in LDS
    } else {
        // ================================ producers ================================
        __builtin_amdgcn_s_setprio(3);             // memory-issuing waves win arbitration against the MFMA stream
        const int cq = lane << 2;
        const bool u_on = cq < p.Mg, v_on = cq < p.Ng;
        const int cu = u_on ? cq : 0, cv = v_on ? cq : 0;
        float sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.v_shift && v_on) {
#pragma unroll
            for (int t = 0; t < 4; ++t) if (cq + t < p.Ng) sh[t] = p.v_shift[cq + t];
        }
        // my left-over tiles: list index t = w4, w4+4, ...  ->  (m, n)
        int offU[PMAX > 0 ? PMAX : 1], offV[PMAX > 0 ? PMAX : 1];
        int my_count = 0;
#pragma unroll
        for (int s_ = 0; s_ < PMAX; ++s_) {
            const int t = w4 + 4 * s_;
            int m = 0, n = 0;
            if (t < LEFT) {
                ++my_count;
                if ((MT & 1) && t < NT) { m = MT - 1; n = t; }
                else { const int t2 = t - (MT & 1) * NT; m = t2; n = NT - 1; }
            }
            offU[s_] = 16 * m; offV[s_] = 16 * n;
        }
        f32x4 accP[PMAX > 0 ? PMAX : 1];
#pragma unroll
        for (int s_ = 0; s_ < (PMAX > 0 ? PMAX : 1); ++s_) accP[s_] = (f32x4){0.f, 0.f, 0.f, 0.f};
        double csd[4] = {0, 0, 0, 0};
        float4 ur[RQ], vr[RQ], vr2[VMODE == V_GATHER ? RQ : 1];

        int jgv = 0;                                // V_GATHER: lane q <-> neighbour row of this wave's q-th row
        // A partial last tile is fetched as the LAST 32 rows of the operands (rows - 32 ..: in bounds, the launcher guarantees
        // rows >= 32): row order inside a tile is irrelevant to the sums, the rows that belong to the previous tile are
        // zeroed by commit's slow path, and every tile's row addresses are base + q * (4 rows) with no per-row clamp.
        const long us4 = 4L * p.u.stride_outer, vs4 = 4L * p.v.stride_outer;
        auto tile_row0 = [&](int tile) -> long {
            const long row0 = (long)tile * RD_RT;
            return (row0 + RD_RT <= p.rows) ? row0 : p.rows - RD_RT;
        };
        auto load_jgv = [&](int tile) -> int {
            return p.jg[tile_row0(tile) + w4 + 4 * ((lane < RQ) ? lane : RQ - 1)];
        };
        auto fetch = [&](int tile) {
            const long rb = tile_row0(tile) + w4;
            const float* up = p.u.base + rb * p.u.stride_outer + cu;
            const float* vp = p.v.base + rb * p.v.stride_outer + cv;
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                ur[q] = rd_ld4(up + q * us4);
                if (VMODE == V_DENSE) vr[q] = rd_ld4(vp + q * vs4);
                else {
                    // wave-uniform row: one s_mul_hi_u32 (the double-reciprocal gpe_udiv is ~8 VALU instructions per row here)
                    const long i = (long)__umulhi((unsigned)(rb + 4 * q), p.kmagic);
                    // neighbour row: prefetched lane-distributed one tile ahead (load_jgv) — read here as p.jg[gr] it is
                    // a vector load whose result every later load of this fetch has to wait for (in-order vmcnt)
                    const long jj = __builtin_amdgcn_readlane(jgv, q);
                    vr[q] = rd_ld4(p.pq + i * p.ldpq + cv);
                    vr2[q] = rd_ld4(p.pq + jj * p.ldpq + p.H + cv);
                }
            }
        };
        auto commit = [&](int buf, int tile) {
            float* ub = Us + buf * RD_RT * LDU;
            float* vb = Vs + buf * RD_RT * LDV;
            const long row0 = (long)tile * RD_RT;
            const int rv = (int)((p.rows - row0 < RD_RT) ? (p.rows - row0) : RD_RT);
            float c32[4] = {0.f, 0.f, 0.f, 0.f};
            if (rv == RD_RT) {
                // full tile (every tile but possibly the last): no masks at all.  Columns >= Mg / Ng of the LDS image then
                // hold whatever the clamped loads delivered (pad columns of the rows, or a copy of columns 0..3): column m of
                // U only ever reaches row m of the product and entry m of the column sums, column n of V only column n, and
                // gpe_redgemm_finish reads m < Mg, n < Ng alone.  (112 v_cndmask + 16 exec branches per tile otherwise,
                // each ~6 cycles of a SIMD that is otherwise issuing fp32 MFMAs.)
                if (cq < UC) {
#pragma unroll
                    for (int q = 0; q < RQ; ++q) {
                        const float4 u = ur[q];
                        c32[0] += u.x; c32[1] += u.y; c32[2] += u.z; c32[3] += u.w;
                        *reinterpret_cast<float4*>(&ub[(w4 + 4 * q) * LDU + cq]) = u;
                    }
                }
                if (cq < VC) {
#pragma unroll
                    for (int q = 0; q < RQ; ++q) {
                        float4 v = vr[q];
                        if (VMODE == V_GATHER) {
                            v.x = fmaxf(v.x + vr2[q].x, 0.f); v.y = fmaxf(v.y + vr2[q].y, 0.f);
                            v.z = fmaxf(v.z + vr2[q].z, 0.f); v.w = fmaxf(v.w + vr2[q].w, 0.f);
                        }
                        v.x -= sh[0]; v.y -= sh[1]; v.z -= sh[2]; v.w -= sh[3];
                        *reinterpret_cast<float4*>(&vb[(w4 + 4 * q) * LDV + cq]) = v;
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < RQ; ++q) {
                    const int r = w4 + 4 * q;
                    const bool ok = r >= RD_RT - rv;           // (see fetch: a partial tile holds the operands' last 32 rows)
                    if (cq < UC) {
                        float4 u = ur[q];
                        if (!(ok && u_on)) u = make_float4(0.f, 0.f, 0.f, 0.f);
                        else {
                            if (cq + 1 >= p.Mg) u.y = 0.f;
                            if (cq + 2 >= p.Mg) u.z = 0.f;
                            if (cq + 3 >= p.Mg) u.w = 0.f;
                        }
                        c32[0] += u.x; c32[1] += u.y; c32[2] += u.z; c32[3] += u.w;
                        *reinterpret_cast<float4*>(&ub[r * LDU + cq]) = u;
                    }
                    if (cq < VC) {
                        float4 v = vr[q];
                        if (VMODE == V_GATHER) {
                            v.x = fmaxf(v.x + vr2[q].x, 0.f); v.y = fmaxf(v.y + vr2[q].y, 0.f);
                            v.z = fmaxf(v.z + vr2[q].z, 0.f); v.w = fmaxf(v.w + vr2[q].w, 0.f);
                        }
                        v.x -= sh[0]; v.y -= sh[1]; v.z -= sh[2]; v.w -= sh[3];
                        if (!(ok && v_on)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                        else {
                            if (cq + 1 >= p.Ng) v.y = 0.f;
                            if (cq + 2 >= p.Ng) v.z = 0.f;
                            if (cq + 3 >= p.Ng) v.w = 0.f;
                        }
                        *reinterpret_cast<float4*>(&vb[r * LDV + cq]) = v;
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) csd[t] += (double)c32[t];
        };

        GpeTileSeq sq = gpe_tile_seq(p.pin_tpc, p.rev, p.pin_clouds, p.num_tiles);
        int tile = gpe_seq_tile(sq);
        gpe_seq_advance(sq);
        int next = gpe_seq_tile(sq);
        gpe_seq_advance(sq);
        int next2 = gpe_seq_tile(sq);
        if (VMODE == V_GATHER) jgv = load_jgv(tile < p.num_tiles ? tile : 0);
        if (tile < p.num_tiles) { fetch(tile); commit(0, tile); }
        if (VMODE == V_GATHER && tile < p.num_tiles) jgv = load_jgv(next < p.num_tiles ? next : tile);
        __syncthreads();                           // prologue
        int buf = 0;
        for (; tile < p.num_tiles; tile = next, next = next2, gpe_seq_advance(sq), next2 = gpe_seq_tile(sq)) {
            // unconditional (clamped tile): registers loaded under a branch are copied at the join, and that copy waits
            // for the loads right there — the fetch then runs as 8 serial round trips and the whole kernel at its pace
            fetch(next < p.num_tiles ? next : tile);
            if (VMODE == V_GATHER) jgv = load_jgv(next2 < p.num_tiles ? next2 : tile);
            if (PMAX > 0) {
                const float* ub = Us + buf * RD_RT * LDU;
                const float* vb = Vs + buf * RD_RT * LDV;
#pragma unroll
                for (int r0 = 0; r0 < RD_RT; r0 += 4) {
#pragma unroll
                    for (int s_ = 0; s_ < PMAX; ++s_) {
                        if (s_ < my_count) {
                            const float a = ub[(r0 + g) * LDU + offU[s_] + j];
                            const float b = vb[(r0 + g) * LDV + offV[s_] + j];
                            accP[s_] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, accP[s_], 0, 0, 0);
                        }
                    }
                }
            }
            if (next < p.num_tiles) commit(buf ^ 1, next);
            __syncthreads();
            buf ^= 1;
        }
#pragma unroll
        for (int s_ = 0; s_ < PMAX; ++s_) {
            if (s_ < my_count) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    dst[(size_t)(offU[s_] + 4 * g + r) * p.NgPad + offV[s_] + j] = accP[s_][r];
            }
        }
        double* red = reinterpret_cast<double*>(smem);          // [4][UC]
        if (cq < UC) {
#pragma unroll
            for (int t = 0; t < 4; ++t) red[w4 * UC + cq + t] = csd[t];
        }
        __syncthreads();                           // tail
    }
    if (tid < p.Mg) {
        const double* red = reinterpret_cast<const double*>(smem);
        p.part_cs[(size_t)blockIdx.x * p.MgPad + tid] =
            (red[tid] + red[UC + tid]) + (red[2 * UC + tid] + red[3 * UC + tid]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16x3 variant of the producer/consumer kernel (gpe_math_set(1)): same roles, same partial/colsum outputs, but the
// products run on the bf16 matrix pipe — u = uh + ul, v = vh + vl (two bf16 each), u*v ~= ul*vh + uh*vl + uh*vh with
// fp32 accumulation (v_mfma_f32_16x16x32_bf16; the reduction dim of ONE instruction is a whole 32-row tile).  A bf16
// MFMA stream leaves the SIMD's other wave free to issue, so the producers' loads really do hide under it
// (profiles/r01_d_coissue_ubench.md).
//
// The MFMA wants, per lane, 8 consecutive reduction indices (= ROWS) of one operand column, so the LDS image is the
// TRANSPOSE of the row tile: entry (plane, g, col) = 16 bytes = bf16 of rows 8g..8g+7 of column `col`, stored at
//     ((plane*4 + g)*4 + (col & 3)) * S + (col >> 2)            [16-byte units],  S = cols/4 + pad,  S % 8 == 2
// A producer lane owns columns 4l..4l+3 of rows 8*w4..8*w4+7, so its four ds_write_b128 per plane land on consecutive
// 16-byte slots across lanes (conflict-free), and a consumer's 16 lanes (i = 0..15, same g) hit 8 distinct 16-byte bank
// groups per 8 lanes because S % 8 == 2.
typedef __bf16 rd_bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 rd_bf16x8_t __attribute__((ext_vector_type(8)));
typedef float rd_f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void rd_split_pair(float x0, float x1, unsigned& hw, unsigned& lw)
{
    const rd_f32x2_t x = {x0, x1};
    hw = __builtin_bit_cast(unsigned, __builtin_convertvector(x, rd_bf16x2_t));          // v_cvt_pk_bf16_f32 (RNE)
    const rd_f32x2_t r = {x0 - __uint_as_float(hw << 16), x1 - __uint_as_float(hw & 0xffff0000u)};
    lw = __builtin_bit_cast(unsigned, __builtin_convertvector(r, rd_bf16x2_t));
}
__device__ __forceinline__ f32x4 rd_mfma32(const uint4 a, const uint4 b, const f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(rd_bf16x8_t, a), __builtin_bit_cast(rd_bf16x8_t, b),
                                                   c, 0, 0, 0);
}
// f16x3 (gpe_math_set(4)): the same kernel with two-term fp16 splits of operands normalised per tensor by a power of two
// (gpe_edgegemm_split_kernel.h has the arithmetic; the scales come from the notes / bounds of gpe_edgegemm_h3.hip)
typedef _Float16 rd_f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 rd_f16x8_t __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ void rd_split_pair_p(float x0, float x1, unsigned& hw, unsigned& lw)
{
    if constexpr (F16) {
        const rd_f32x2_t x = {x0, x1};
        const rd_f16x2_t h = __builtin_convertvector(x, rd_f16x2_t);
        const rd_f32x2_t r = x - __builtin_convertvector(h, rd_f32x2_t);
        hw = __builtin_bit_cast(unsigned, h);
        lw = __builtin_bit_cast(unsigned, __builtin_convertvector(r, rd_f16x2_t));
    } else
        rd_split_pair(x0, x1, hw, lw);
}
template <bool F16>
__device__ __forceinline__ f32x4 rd_mfma32_p(const uint4 a, const uint4 b, const f32x4 c)
{
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rd_f16x8_t, a), __builtin_bit_cast(rd_f16x8_t, b), c, 0, 0, 0);
    else
        return rd_mfma32(a, b, c);
}
__device__ __forceinline__ float rd_comp(const float4& v, int t) { return t == 0 ? v.x : t == 1 ? v.y : t == 2 ? v.z : v.w; }

template <int CT> struct RdB3Layout {                 // CT = number of 16-column tiles of the operand
    static constexpr int Q = 4 * CT;                  // column quads
    static constexpr int S = Q + ((10 - (Q % 8)) % 8);        // S % 8 == 2
    static constexpr int BYTES = 32 * S * 16;         // 2 planes x 4 row blocks x 4 (col & 3) x S slots x 16 B
    __device__ static __forceinline__ int slot(int plane, int g, int col)
    {
        return (((plane * 4 + g) * 4 + (col & 3)) * S + (col >> 2)) * 16;
    }
};

template <int MT, int NT, int VMODE, bool F16 = false, bool LAZY = false>
__global__ __launch_bounds__(512, 2) void gpe_redgemm_b3_kernel(RdParams p)
{
    static_assert(!LAZY || (F16 && VMODE == V_DENSE), "lazy dz3: f16x3 reduce-GEMM with a dense V");
    float sU = 1.f, sV = 1.f, invU = 1.f, invV = 1.f;
    if constexpr (F16) {
        gpe_h3_scale_of(p.amax_u[0], sU, invU);
        // V enters as V - shift: |V - shift| <= max|V| + max|shift| (uniform loop over <= 208 values, once per workgroup)
        float bound = __uint_as_float(p.amax_v[0]);
        if (p.v_shift) {
            float smx = 0.f;
            for (int c = 0; c < p.Ng; ++c) smx = fmaxf(smx, fabsf(p.v_shift[c]));
            bound += smx;
        }
        gpe_h3_scale_of(__float_as_uint(bound), sV, invV);
    }
    static_assert(RD_RT == 32, "one v_mfma_f32_16x16x32_bf16 reduces exactly one row tile");
    constexpr int MB = MT / 2, NB = NT / 2;
    constexpr int LEFT = (MT & 1) * NT + (NT & 1) * (MT - (MT & 1));     // left-over tiles
    constexpr int PMAX = (LEFT + 3) / 4;
    constexpr int UC = 16 * MT, VC = 16 * NT;
    using LU = RdB3Layout<MT>;
    using LV = RdB3Layout<NT>;
    constexpr int RQ = RD_RT / 4;                     // 8 consecutive rows per producer wave
    extern __shared__ __align__(16) char smem_b3[];
    char* const Ub = smem_b3;                         // [2][LU::BYTES]
    char* const Vb = smem_b3 + 2 * LU::BYTES;         // [2][LV::BYTES]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w4 = wave & 3;
    const int j = lane & 15, g = lane >> 4;
    float* dst = p.part + (size_t)blockIdx.x * p.MgPad * p.NgPad;

    // this wave's share of the left-over tiles (producers and consumers are numbered 0..3 alike): list index t = w4, w4+4, ...  ->  (m, n)
    int offU[PMAX > 0 ? PMAX : 1], offV[PMAX > 0 ? PMAX : 1];
    int my_count = 0;
#pragma unroll
    for (int s_ = 0; s_ < PMAX; ++s_) {
        const int t = w4 + 4 * s_;
        int m = 0, n = 0;
        if (t < LEFT) {
            ++my_count;
            if ((MT & 1) && t < NT) { m = MT - 1; n = t; }
            else { const int t2 = t - (MT & 1) * NT; m = t2; n = NT - 1; }
        }
        offU[s_] = 16 * m; offV[s_] = 16 * n;
    }
    if (wave < 4) {
        // ================================ consumers ================================
        const int mt0 = (w4 & 1) * MB, nt0 = (w4 >> 1) * NB;
        f32x4 acc[MB][NB];
#pragma unroll
        for (int q = 0; q < MB; ++q)
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[q][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 accL[(RD_B3_LEFT_CONS && PMAX > 0) ? PMAX : 1];
#pragma unroll
        for (int s_ = 0; s_ < ((RD_B3_LEFT_CONS && PMAX > 0) ? PMAX : 1); ++s_) accL[s_] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __syncthreads();                           // prologue: tile 0 staged
        int buf = 0;
        GpeTileSeq sq = gpe_tile_seq(p.pin_tpc, p.rev, p.pin_clouds, p.num_tiles);
        for (int tile = gpe_seq_tile(sq); tile < p.num_tiles; gpe_seq_advance(sq), tile = gpe_seq_tile(sq)) {
            const char* ub = Ub + buf * LU::BYTES;
            const char* vb = Vb + buf * LV::BYTES;
            uint4 bh[NB], bl[NB];
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                bh[n] = *reinterpret_cast<const uint4*>(vb + LV::slot(0, g, 16 * (nt0 + n) + j));
                bl[n] = *reinterpret_cast<const uint4*>(vb + LV::slot(1, g, 16 * (nt0 + n) + j));
            }
            if (!(RD_B3_DBG & 4))
#pragma unroll
            for (int q = 0; q < MB; ++q) {
                const uint4 ah = *reinterpret_cast<const uint4*>(ub + LU::slot(0, g, 16 * (mt0 + q) + j));
                const uint4 al = *reinterpret_cast<const uint4*>(ub + LU::slot(1, g, 16 * (mt0 + q) + j));
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[q][n] = rd_mfma32_p<F16>(al, bh[n], acc[q][n]);
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[q][n] = rd_mfma32_p<F16>(ah, bl[n], acc[q][n]);
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[q][n] = rd_mfma32_p<F16>(ah, bh[n], acc[q][n]);
            }
            if constexpr (RD_B3_LEFT_CONS && PMAX > 0) {
#pragma unroll
                for (int s_ = 0; s_ < PMAX; ++s_) {
                    if (s_ < my_count) {
                        const uint4 ah = *reinterpret_cast<const uint4*>(ub + LU::slot(0, g, offU[s_] + j));
                        const uint4 al = *reinterpret_cast<const uint4*>(ub + LU::slot(1, g, offU[s_] + j));
                        const uint4 bh2 = *reinterpret_cast<const uint4*>(vb + LV::slot(0, g, offV[s_] + j));
                        const uint4 bl2 = *reinterpret_cast<const uint4*>(vb + LV::slot(1, g, offV[s_] + j));
                        f32x4 a3 = rd_mfma32_p<F16>(al, bh2, accL[s_]);
                        a3 = rd_mfma32_p<F16>(ah, bl2, a3);
                        accL[s_] = rd_mfma32_p<F16>(ah, bh2, a3);
                    }
                }
            }
            __syncthreads();
            buf ^= 1;
        }
        if constexpr (RD_B3_LEFT_CONS && PMAX > 0) {
#pragma unroll
            for (int s_ = 0; s_ < PMAX; ++s_) {
                if (s_ < my_count) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dst[(size_t)(offU[s_] + 4 * g + r) * p.NgPad + offV[s_] + j] = accL[s_][r] * invU * invV;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < MB; ++q)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    dst[(size_t)(16 * (mt0 + q) + 4 * g + r) * p.NgPad + 16 * (nt0 + n) + j] = acc[q][n][r] * invU * invV;
        __syncthreads();                           // tail: producers' column sums in LDS
    } else {
        // ================================ producers ================================
        if (RD_B3_PRIO) __builtin_amdgcn_s_setprio(RD_B3_PRIO);
        const int cq = lane << 2;
        const bool u_on = cq < p.Mg, v_on = cq < p.Ng;
        const int cu = u_on ? cq : 0, cv = v_on ? cq : 0;
        float sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.v_shift && v_on) {
#pragma unroll
            for (int t = 0; t < 4; ++t) if (cq + t < p.Ng) sh[t] = p.v_shift[cq + t];
        }
        f32x4 accP[PMAX > 0 ? PMAX : 1];
#pragma unroll
        for (int s_ = 0; s_ < (PMAX > 0 ? PMAX : 1); ++s_) accP[s_] = (f32x4){0.f, 0.f, 0.f, 0.f};
        double csd[4] = {0, 0, 0, 0};
        float4 ur[RQ], vr[RQ], vr2[VMODE == V_GATHER ? RQ : 1];
        // LAZY: this lane's coefficient quads, and per fetched tile the s * g quad + winning slots of the wave's point (k = 16 and
        // 32-row tiles aligned to points: a wave's 8 consecutive rows are slots 8 (w4 & 1) .. + 7 of ONE point)
        // (dz3 = fma(-k2, a, base), base = s g + (mean k2 - c1) for the winning slot, mean k2 - c1 for the others: gpe_edgegemm_split_kernel.h)
        float lzs[4] = {0.f, 0.f, 0.f, 0.f}, lznc[4] = {0.f, 0.f, 0.f, 0.f}, lznk[4] = {0.f, 0.f, 0.f, 0.f};
        float4 lz_gq = make_float4(0.f, 0.f, 0.f, 0.f);
        uchar4 lz_sx = make_uchar4(0, 0, 0, 0), lz_sn = make_uchar4(0, 0, 0, 0);
        if constexpr (LAZY) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (cq + t < p.Mg) {
                    const float k2 = p.lz_coef[2 * p.Mg + cq + t];
                    lzs[t] = p.lz_coef[cq + t];
                    lznc[t] = __builtin_fmaf(p.lz_coef[3 * p.Mg + cq + t], k2, -p.lz_coef[p.Mg + cq + t]);
                    lznk[t] = -k2;
                }
            }
        }

        int jgv = 0;                                // V_GATHER: lane q <-> neighbour row of this wave's q-th row
        // The producers' instruction diet of gpe_redgemm_pc_kernel, applied here in round 4: a partial last tile
        // is fetched as the LAST 32 rows of the operands (in bounds: the launcher guarantees rows >= 32; row order inside a
        // tile is irrelevant to the sums; the rows that belong to the previous tile are zeroed by commit's slow path), so every
        // tile's row addresses are linear, full tiles run without a single mask (pad columns only ever reach pad outputs, which
        // gpe_redgemm_finish never reads — NaN / inf there are harmless), and the gathered row's point is one s_mul_hi_u32 on the
        // wave-uniform row instead of the fp64 reciprocal chain.
        auto tile_row0 = [&](int tile) -> long {
            const long row0 = (long)tile * RD_RT;
            return (row0 + RD_RT <= p.rows) ? row0 : p.rows - RD_RT;
        };
        auto load_jgv = [&](int tile) -> int {
            return p.jg[tile_row0(tile) + RQ * w4 + ((lane < RQ) ? lane : RQ - 1)];
        };
        // U rows first, then V rows: the memory counter retires in order, so a commit of U can wait for "all but the V loads"
        auto fetchU = [&](int tile) {
            const long rb = tile_row0(tile) + RQ * w4;                      // 8 CONSECUTIVE rows per wave
            const float* up = p.u.base + rb * p.u.stride_outer + cu;
            if constexpr (LAZY) {
                const long pt = rb >> 4;                                    // k = 16 (host-checked); wave-uniform
                const float* gr = p.lz_g + pt * p.lz_ldg + cu;                  // dword loads at clamped columns: any row pitch
                const int rem = p.Mg - 1 - cu;
                lz_gq = make_float4(gr[0], gr[rem < 1 ? rem : 1], gr[rem < 2 ? rem : 2], gr[rem < 3 ? rem : 3]);
                lz_sx = *reinterpret_cast<const uchar4*>(p.lz_amx + pt * p.lz_ldagg + cu);
                lz_sn = *reinterpret_cast<const uchar4*>(p.lz_amn + pt * p.lz_ldagg + cu);
            }
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                if constexpr (LAZY) {
                    // fp16 rows of the stored activation (pitch in halves): the raw words travel in ur[].x / .y
                    const uint2 hq = *reinterpret_cast<const uint2*>(reinterpret_cast<const _Float16*>(p.u.base) + (rb + q) * p.u.stride_outer + cu);
                    ur[q].x = __uint_as_float(hq.x); ur[q].y = __uint_as_float(hq.y);
                } else
                    ur[q] = rd_ld4(up + q * p.u.stride_outer);
            }
        };
        auto fetchV = [&](int tile) {
            const long rb = tile_row0(tile) + RQ * w4;
            const float* vp = p.v.base + rb * p.v.stride_outer + cv;
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                if (VMODE == V_DENSE) vr[q] = rd_ld4(vp + q * p.v.stride_outer);
                else {
                    const long i = (long)__umulhi((unsigned)(rb + q), p.kmagic);
                    const long jj = __builtin_amdgcn_readlane(jgv, q);       // prefetched one tile ahead (load_jgv)
                    vr[q] = rd_ld4(p.pq + i * p.ldpq + cv);
                    vr2[q] = rd_ld4(p.pq + jj * p.ldpq + p.H + cv);
                }
            }
        };
        auto commitU = [&](int buf, int tile) {
            char* ub = Ub + buf * LU::BYTES;
            const long row0 = (long)tile * RD_RT;
            const int rv = (int)((p.rows - row0 < RD_RT) ? (p.rows - row0) : RD_RT);
            float c32[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (LAZY) {
                // U row = dz3 of slot 8 (w4 & 1) + q of the wave's point, formed from the stored activation (gpe_dz3_kernel's arithmetic)
                const float gq[4] = {lz_gq.x, lz_gq.y, lz_gq.z, lz_gq.w};
                const int sx[4] = {lz_sx.x, lz_sx.y, lz_sx.z, lz_sx.w}, sn[4] = {lz_sn.x, lz_sn.y, lz_sn.z, lz_sn.w};
                float sg[4];
                int sel[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) { sel[t] = (lzs[t] >= 0.f) ? sx[t] : sn[t]; sg[t] = __builtin_fmaf(lzs[t], gq[t], lznc[t]); }
                const int slot0 = (w4 & 1) * RQ;
#pragma unroll
                for (int q = 0; q < RQ; ++q) {
                    typedef _Float16 rd_h2 __attribute__((ext_vector_type(2)));
                    typedef float rd_f2 __attribute__((ext_vector_type(2)));
                    const rd_f2 a01 = __builtin_convertvector(__builtin_bit_cast(rd_h2, __float_as_uint(ur[q].x)), rd_f2);
                    const rd_f2 a23 = __builtin_convertvector(__builtin_bit_cast(rd_h2, __float_as_uint(ur[q].y)), rd_f2);
                    const float av[4] = {a01[0], a01[1], a23[0], a23[1]};
                    float dz[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float base = (sel[t] == slot0 + q) ? sg[t] : lznc[t];
                        dz[t] = (av[t] > 0.f) ? __builtin_fmaf(lznk[t], av[t], base) : 0.f;
                    }
                    ur[q] = make_float4(dz[0], dz[1], dz[2], dz[3]);
                }
            }
            if (rv != RD_RT) {
                // partial last tile (see fetch: it holds the operands' last 32 rows): rows of the previous tile and pad columns -> 0
#pragma unroll
                for (int q = 0; q < RQ; ++q) {
                    const bool ok = RQ * w4 + q >= RD_RT - rv;
                    if (!(ok && u_on)) ur[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    else {
                        if (cq + 1 >= p.Mg) ur[q].y = 0.f;
                        if (cq + 2 >= p.Mg) ur[q].z = 0.f;
                        if (cq + 3 >= p.Mg) ur[q].w = 0.f;
                    }
                    // (V needs no mask: a zero U row contributes 0 * finite to every product and nothing to the column sums)
                }
            }
            if (cq < UC) {
#pragma unroll
                for (int q = 0; q < RQ; ++q) { c32[0] += ur[q].x; c32[1] += ur[q].y; c32[2] += ur[q].z; c32[3] += ur[q].w; }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    unsigned hw[4], lw[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        rd_split_pair_p<F16>(rd_comp(ur[2 * e], t) * sU, rd_comp(ur[2 * e + 1], t) * sU, hw[e], lw[e]);
                    *reinterpret_cast<uint4*>(ub + LU::slot(0, w4, cq + t)) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
                    *reinterpret_cast<uint4*>(ub + LU::slot(1, w4, cq + t)) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) csd[t] += (double)c32[t];
        };
        auto commitV = [&](int buf) {
            char* vb = Vb + buf * LV::BYTES;
            if (cq < VC) {
#pragma unroll
                for (int q = 0; q < RQ; ++q) {
                    float4 v = vr[q];
                    if (VMODE == V_GATHER) {
                        v.x = fmaxf(v.x + vr2[q].x, 0.f); v.y = fmaxf(v.y + vr2[q].y, 0.f);
                        v.z = fmaxf(v.z + vr2[q].z, 0.f); v.w = fmaxf(v.w + vr2[q].w, 0.f);
                    }
                    v.x -= sh[0]; v.y -= sh[1]; v.z -= sh[2]; v.w -= sh[3];
                    vr[q] = v;
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    unsigned hw[4], lw[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        rd_split_pair_p<F16>(rd_comp(vr[2 * e], t) * sV, rd_comp(vr[2 * e + 1], t) * sV, hw[e], lw[e]);
                    *reinterpret_cast<uint4*>(vb + LV::slot(0, w4, cq + t)) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
                    *reinterpret_cast<uint4*>(vb + LV::slot(1, w4, cq + t)) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
                }
            }
        };


        // Order of a producer wave's iteration: see RD_B3_PIPE above.  Phase switches (RD_B3_DBG builds, us per launch at cfg 2,
        // gathered / dense, stand-alone launches on random data): production 527 / 426, no commit 399 / 320, no row loads 378 / 301,
        // neither 253 / 202, no consumer MFMAs 446 / 354, no MFMAs at all 394 / 331, nothing at all 94 / 85 — the phases ADD UP.
        GpeTileSeq sq = gpe_tile_seq(p.pin_tpc, p.rev, p.pin_clouds, p.num_tiles);
        int tile = gpe_seq_tile(sq);
        gpe_seq_advance(sq);
        int next = gpe_seq_tile(sq);
        gpe_seq_advance(sq);
        int next2 = gpe_seq_tile(sq);
        gpe_seq_advance(sq);
        int next3 = gpe_seq_tile(sq);
        constexpr bool pipe = RD_B3_PIPE != 0;
        if (VMODE == V_GATHER) jgv = load_jgv(tile < p.num_tiles ? tile : 0);
        if (tile < p.num_tiles) { fetchU(tile); fetchV(tile); commitU(0, tile); commitV(0); }
        if (VMODE == V_GATHER && tile < p.num_tiles) jgv = load_jgv(next < p.num_tiles ? next : tile);
        if (pipe && tile < p.num_tiles) {
            fetchU(next < p.num_tiles ? next : tile);                       // committed in the first iteration
            fetchV(next < p.num_tiles ? next : tile);
            if (VMODE == V_GATHER) jgv = load_jgv(next2 < p.num_tiles ? next2 : tile);
        }
        __syncthreads();                           // prologue
        int buf = 0;
        for (; tile < p.num_tiles; tile = next, next = next2, next2 = next3, gpe_seq_advance(sq), next3 = gpe_seq_tile(sq)) {
            if constexpr (pipe) {
                // U of tile t + 1 (fetched during iteration t - 1) -> LDS, U of tile t + 2 requested; then the same for V: each set
                // of loads has the other operand's commit, the left-over MFMAs and the barrier to land
                const bool cm = next < p.num_tiles && !(RD_B3_DBG & 1);
                const int t2 = next2 < p.num_tiles ? next2 : tile;           // clamped: unconditional loads
                if (cm) commitU(buf ^ 1, next);
                if (!(RD_B3_DBG & 2)) fetchU(t2);
                if (cm) commitV(buf ^ 1);
                if (!(RD_B3_DBG & 2)) fetchV(t2);
                if (VMODE == V_GATHER) jgv = load_jgv(next3 < p.num_tiles ? next3 : tile);
            } else {
                if (!(RD_B3_DBG & 2)) { fetchU(next < p.num_tiles ? next : tile); fetchV(next < p.num_tiles ? next : tile); }
                if (VMODE == V_GATHER) jgv = load_jgv(next2 < p.num_tiles ? next2 : tile);
            }
            if (PMAX > 0 && !RD_B3_LEFT_CONS && !(RD_B3_DBG & 8)) {
                const char* ub = Ub + buf * LU::BYTES;
                const char* vb = Vb + buf * LV::BYTES;
#pragma unroll
                for (int s_ = 0; s_ < PMAX; ++s_) {
                    if (s_ < my_count) {
                        const uint4 ah = *reinterpret_cast<const uint4*>(ub + LU::slot(0, g, offU[s_] + j));
                        const uint4 al = *reinterpret_cast<const uint4*>(ub + LU::slot(1, g, offU[s_] + j));
                        const uint4 bh = *reinterpret_cast<const uint4*>(vb + LV::slot(0, g, offV[s_] + j));
                        const uint4 bl = *reinterpret_cast<const uint4*>(vb + LV::slot(1, g, offV[s_] + j));
                        f32x4 a3 = rd_mfma32_p<F16>(al, bh, accP[s_]);   // same shape back to back: accumulator forwarding is fine
                        a3 = rd_mfma32_p<F16>(ah, bl, a3);
                        accP[s_] = rd_mfma32_p<F16>(ah, bh, a3);
                    }
                }
            }
            if (!pipe && next < p.num_tiles && !(RD_B3_DBG & 1)) { commitU(buf ^ 1, next); commitV(buf ^ 1); }
            __syncthreads();
            buf ^= 1;
        }
        if constexpr (!RD_B3_LEFT_CONS) {
#pragma unroll
            for (int s_ = 0; s_ < PMAX; ++s_) {
                if (s_ < my_count) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dst[(size_t)(offU[s_] + 4 * g + r) * p.NgPad + offV[s_] + j] = accP[s_][r] * invU * invV;
                }
            }
        }
        double* red = reinterpret_cast<double*>(smem_b3);       // [4][UC]
        if (cq < UC) {
#pragma unroll
            for (int t = 0; t < 4; ++t) red[w4 * UC + cq + t] = csd[t];
        }
        __syncthreads();                           // tail
    }
    if (tid < p.Mg) {
        const double* red = reinterpret_cast<const double*>(smem_b3);
        p.part_cs[(size_t)blockIdx.x * p.MgPad + tid] =
            (red[tid] + red[UC + tid]) + (red[2 * UC + tid] + red[3 * UC + tid]);
    }
}

// Fixed-order (deterministic) reduction of the per-workgroup partials.  256 threads = 32 consecutive output elements x 8
// partial groups: group q sums partials q, q+8, ... in fp64 (2 chains), the 8 group sums are combined through LDS in
// index order.  8x the threads of a one-thread-per-element loop: with 256 partials of a 208 x 208 product (44 MB) the
// serial version was latency-bound at ~0.4 TB/s.
// ---------------------------------------------------------------------------------------------------------
// Deep-reduction variant for ROW-POOR dense products (the LSTM / GRU weight gradients: 10 k rows against a 1000 x 250
// output).  The big-block kernel above gives such a product ~6 row tiles per workgroup, a 58 MB partial image and the
// guarded scalar loader (its rows are 2-level [sequence][step] descriptors): 150-175 us for 5 GFLOP.  Here the OUTPUT is
// cut small instead — 64 x 64 per workgroup, grid (row split, M blocks, N blocks), 3 workgroups per CU — so the row
// split stays <= 32 and a workgroup still runs tens of row tiles.  Rows are addressed through the
// 2-level descriptor with plain 16-B loads (aligned pitches, rows padded to 4 columns: every internal sequence buffer).
// ---------------------------------------------------------------------------------------------------------
#define RDD_B 64
#define RDD_LD 80                     // == 16 (mod 32): conflict-free b32 operand reads
#define RDD_MAX_GX 32

// 2-level row offset with the division as one v_mul_hi_u32 (host: r * inner < 2^32)
__device__ __forceinline__ long rd_row_off_magic(const GpeRows& a, unsigned r, unsigned magic)
{
    if (a.inner <= 1) return (long)r * a.stride_outer;
    const unsigned o = __umulhi(r, magic);
    return (long)o * a.stride_outer + (long)(r - o * (unsigned)a.inner) * a.stride_inner;
}

__device__ __forceinline__ long rd_row_off(const GpeRows& a, unsigned r, double rcp_inner)
{
    if (a.inner <= 0) return (long)r * a.stride_outer;
    const unsigned o = gpe_udiv(r, (unsigned)a.inner, rcp_inner);
    return (long)o * a.stride_outer + (long)(r - o * (unsigned)a.inner) * a.stride_inner;
}

// VVEC: V rows take 16-B loads too; otherwise V (an external tensor such as the raw N x 3 positions) is read with clamped
// scalar loads — U, the gradient operand, is always an internal aligned buffer.
template <bool VVEC>
__global__ __launch_bounds__(256, 3) void gpe_redgemm_deep_kernel(RdParams p)
{
    extern __shared__ __align__(16) float smem[];
    float* Us = smem;                              // [2][RD_RT * RDD_LD]
    float* Vs = smem + 2 * RD_RT * RDD_LD;         // [2][RD_RT * RDD_LD]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;
    const int m0 = blockIdx.y * RDD_B, n0 = blockIdx.z * RDD_B;
    const bool want_cs = p.part_cs != nullptr && blockIdx.z == 0;

    // staging map: thread -> rows sr, sr + 16 of the tile, column quad cq of both operands
    const int sr = tid >> 4, cq = (tid & 15) << 2;
    const bool u_on = m0 + cq < p.Mg, v_on = n0 + cq < p.Ng;
    const int ucol = u_on ? m0 + cq : 0, vcol = v_on ? n0 + cq : 0;      // clamped: the loads are unconditional
    float sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.v_shift && v_on) {
#pragma unroll
        for (int t = 0; t < 4; ++t) if (n0 + cq + t < p.Ng) sh[t] = p.v_shift[n0 + cq + t];
    }
    float4 ur[2], vr[2];
    unsigned rmask = 0;
    bool tfull = false;                            // the staged tile has all 32 rows (uniform)
    auto fetch = [&](int tile) {
        const long row0 = (long)tile * RD_RT;
        const int rv = (int)((p.rows - row0 < RD_RT) ? (p.rows - row0) : RD_RT);
        rmask = 0;
        tfull = rv == RD_RT;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = sr + 16 * h;
            if (r < rv) rmask |= 1u << h;
            const unsigned gr = (unsigned)(row0 + ((r < rv) ? r : rv - 1));
            ur[h] = rd_ld4(p.u.base + rd_row_off_magic(p.u, gr, p.umagic) + ucol);
            const float* vp = p.v.base + rd_row_off_magic(p.v, gr, p.vmagic);
            if constexpr (VVEC) vr[h] = rd_ld4(vp + vcol);
            else {                                                  // clamped columns: masked at commit
                const int last = p.Ng - 1;
                vr[h].x = vp[vcol < last ? vcol : last];
                vr[h].y = vp[vcol + 1 < last ? vcol + 1 : last];
                vr[h].z = vp[vcol + 2 < last ? vcol + 2 : last];
                vr[h].w = vp[vcol + 3 < last ? vcol + 3 : last];
            }
        }
    };
    auto commit = [&](int buf) {
        float* ub = Us + buf * RD_RT * RDD_LD;
        float* vb = Vs + buf * RD_RT * RDD_LD;
        if (tfull) {
            // full tile: no masks.  Columns past Mg / Ng of the 64-wide block carry whatever the clamped loads delivered:
            // column m of U only reaches row m of the product and entry m of the column sums, column n of V only column
            // n, and gpe_redgemm_finish reads m < Mg, n < Ng alone (same argument as gpe_redgemm_pc_kernel).
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = sr + 16 * h;
                float4 v = vr[h];
                v.x -= sh[0]; v.y -= sh[1]; v.z -= sh[2]; v.w -= sh[3];
                *reinterpret_cast<float4*>(&ub[r * RDD_LD + cq]) = ur[h];
                *reinterpret_cast<float4*>(&vb[r * RDD_LD + cq]) = v;
            }
            return;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = sr + 16 * h;
            const bool ok = (rmask >> h) & 1u;
            float4 u = ur[h], v = vr[h];
            v.x -= sh[0]; v.y -= sh[1]; v.z -= sh[2]; v.w -= sh[3];
            if (!(ok && u_on)) u = make_float4(0.f, 0.f, 0.f, 0.f);
            else {                                                  // ragged last quad (padded columns are not data)
                if (m0 + cq + 1 >= p.Mg) u.y = 0.f;
                if (m0 + cq + 2 >= p.Mg) u.z = 0.f;
                if (m0 + cq + 3 >= p.Mg) u.w = 0.f;
            }
            if (!(ok && v_on)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            else {
                if (n0 + cq + 1 >= p.Ng) v.y = 0.f;
                if (n0 + cq + 2 >= p.Ng) v.z = 0.f;
                if (n0 + cq + 3 >= p.Ng) v.w = 0.f;
            }
            *reinterpret_cast<float4*>(&ub[r * RDD_LD + cq]) = u;
            *reinterpret_cast<float4*>(&vb[r * RDD_LD + cq]) = v;
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[q][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    double cs = 0.0;
    const int cs_col = tid & 63, cs_rg = tid >> 6;     // column sums of U: thread = (column, group of 8 rows)

    int tile = blockIdx.x;
    if (tile < p.num_tiles) fetch(tile);
    int buf = 0;
    for (; tile < p.num_tiles; tile += gridDim.x) {
        commit(buf);
        __syncthreads();            // tile visible; every wave is past the MFMAs that read buffer buf^1
        fetch(tile + (int)gridDim.x < p.num_tiles ? tile + (int)gridDim.x : tile);     // unconditional (see above)
        const float* ub = Us + buf * RD_RT * RDD_LD;
        const float* vb = Vs + buf * RD_RT * RDD_LD;
        if (want_cs) {
            const float* c = ub + (8 * cs_rg) * RDD_LD + cs_col;
            const float s0 = (c[0] + c[RDD_LD]) + (c[2 * RDD_LD] + c[3 * RDD_LD]);
            const float s1 = (c[4 * RDD_LD] + c[5 * RDD_LD]) + (c[6 * RDD_LD] + c[7 * RDD_LD]);
            cs += (double)s0 + (double)s1;
        }
#pragma unroll
        for (int r0 = 0; r0 < RD_RT; r0 += 4) {
            float a[2], b[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) a[q] = ub[(r0 + g) * RDD_LD + 16 * (2 * wm + q) + j];
#pragma unroll
            for (int n = 0; n < 2; ++n) b[n] = vb[(r0 + g) * RDD_LD + 16 * (2 * wn + n) + j];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[q][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[n], acc[q][n], 0, 0, 0);
        }
        buf ^= 1;
    }

    // ---- one partial per workgroup (same image as the big-block kernel: gpe_redgemm_finish sums them) ------------
    float* dst = p.part + (size_t)blockIdx.x * p.MgPad * p.NgPad;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * (2 * wm + q) + 4 * g + r, nn = n0 + 16 * (2 * wn + n) + j;
                dst[(size_t)m * p.NgPad + nn] = acc[q][n][r];
            }
    if (want_cs) {
        __syncthreads();                                           // the operand tiles are dead
        double* red = reinterpret_cast<double*>(smem);             // [4][64]
        red[cs_rg * 64 + cs_col] = cs;
        __syncthreads();
        if (tid < RDD_B)
            p.part_cs[(size_t)blockIdx.x * p.MgPad + m0 + tid] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
    }
}

// rows a 16-B loader can take through the 2-level descriptor: aligned base and pitches, rows padded to 4 columns
static bool rd_rows_vec2(const GpeRows& r, int cols)
{
    const long pitch = r.inner > 0 ? r.stride_inner : r.stride_outer;
    if ((((uintptr_t)r.base) & 15) || (r.stride_outer & 3)) return false;
    if (r.inner > 0 && (r.stride_inner & 3)) return false;
    return pitch >= ((cols + 3) & ~3) || (cols & 3) == 0;
}

// ---------------------------------------------------------------------------------------------------------
// Thin products (Ng <= 4: the weight gradient of a Linear on raw xyz positions, 65 536 x 400 against 65 536 x 3): nothing for
// the matrix pipe, the job is to stream U once.  Workgroup = a contiguous row range, wave w takes its rows w, w+4, ...,
// lane = column quads lane, lane+64, ... of U (QL per lane); RDT_RB rows in flight per wave, fp32 fma chains of a few dozen
// rows per wave, then the four waves are added in fp64 and one partial per workgroup goes to gpe_redgemm_finish.
// HBM-bound (rows * Mg * 4 bytes): 80 -> ~25 us on the shape above, against the 224 x 256 big-block kernel.
// ---------------------------------------------------------------------------------------------------------
#define RDT_GX 512
#define RDT_RB 4
template <int QL>
__global__ __launch_bounds__(256) void gpe_redgemm_thin_kernel(RdParams p)
{
    extern __shared__ __align__(16) float smem[];              // [4 waves][MgPad][5]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long rpb = (p.rows + gridDim.x - 1) / gridDim.x;
    const long r_begin = (long)blockIdx.x * rpb;
    const long r_end = (r_begin + rpb < p.rows) ? r_begin + rpb : p.rows;
    int cq[QL];
#pragma unroll
    for (int q = 0; q < QL; ++q) {
        const int c = 4 * (lane + 64 * q);
        cq[q] = (c < p.Mg) ? c : 0;                            // clamped: unconditional loads
    }
    float sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.v_shift) {
#pragma unroll
        for (int n = 0; n < 4; ++n) if (n < p.Ng) sh[n] = p.v_shift[n];
    }
    float acc[QL][4][4], cs[QL][4];
#pragma unroll
    for (int q = 0; q < QL; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            cs[q][c] = 0.f;
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[q][c][n] = 0.f;
        }
    for (long r = r_begin + wave; r < r_end; r += 4 * RDT_RB) {
        float4 u[RDT_RB][QL];
        float v[RDT_RB][4];
#pragma unroll
        for (int j = 0; j < RDT_RB; ++j) {
            const long rr = (r + 4 * j < r_end) ? r + 4 * j : r;            // clamped to a row of this wave (masked below)
            const float* up = p.u.base + rr * p.u.stride_outer;
            const float* vp = p.v.base + rr * p.v.stride_outer;
#pragma unroll
            for (int q = 0; q < QL; ++q) u[j][q] = rd_ld4(up + cq[q]);
#pragma unroll
            for (int n = 0; n < 4; ++n) v[j][n] = vp[(n < p.Ng) ? n : 0] - sh[n];
        }
#pragma unroll
        for (int j = 0; j < RDT_RB; ++j) {
            if (r + 4 * j < r_end) {                                        // uniform
#pragma unroll
                for (int q = 0; q < QL; ++q) {
                    const float uu[4] = {u[j][q].x, u[j][q].y, u[j][q].z, u[j][q].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        cs[q][c] += uu[c];
#pragma unroll
                        for (int n = 0; n < 4; ++n) acc[q][c][n] = __builtin_fmaf(uu[c], v[j][n], acc[q][c][n]);
                    }
                }
            }
        }
    }
    // ---- four waves -> one partial ------------------------------------------------------------------------------------
    float* mine = smem + (size_t)wave * p.MgPad * 5;
#pragma unroll
    for (int q = 0; q < QL; ++q) {
        const int c0 = 4 * (lane + 64 * q);
        if (c0 < p.MgPad) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int n = 0; n < 4; ++n) mine[(c0 + c) * 5 + n] = acc[q][c][n];
                mine[(c0 + c) * 5 + 4] = cs[q][c];
            }
        }
    }
    __syncthreads();
    float* dst = p.part + (size_t)blockIdx.x * p.MgPad * p.NgPad;            // NgPad == 4
    const size_t wstride = (size_t)p.MgPad * 5;
    for (int m = tid; m < p.MgPad; m += 256) {
        const float* s0 = smem + (size_t)m * 5;
#pragma unroll
        for (int n = 0; n < 4; ++n)
            dst[(size_t)m * 4 + n] = (float)(((double)s0[n] + (double)s0[wstride + n]) +
                                             ((double)s0[2 * wstride + n] + (double)s0[3 * wstride + n]));
        if (p.part_cs)
            p.part_cs[(size_t)blockIdx.x * p.MgPad + m] = ((double)s0[4] + (double)s0[wstride + 4]) +
                                                          ((double)s0[2 * wstride + 4] + (double)s0[3 * wstride + 4]);
    }
}

// row split of the deep kernel: ~3 workgroups per CU, at least 4 row tiles each, <= RDD_MAX_GX partial images
static int rdd_gx(int Mg, int Ng, long num_tiles, int cus)
{
    const long blocks = (long)gpe_cdiv(Mg, RDD_B) * gpe_cdiv(Ng, RDD_B);
    long gx = gpe_cdiv(3L * cus, blocks);
    if (gx > RDD_MAX_GX) gx = RDD_MAX_GX;
    if (num_tiles >= 0 && gx > num_tiles / 4) gx = num_tiles / 4;
    return gx < 1 ? 1 : (int)gx;
}

#define RD_FIN_E 32
#define RD_FIN_Q 8
__global__ __launch_bounds__(RD_FIN_E * RD_FIN_Q) void gpe_redgemm_finish(
    const float* __restrict__ part, const double* __restrict__ part_cs, int nblk, int Mg, int Ng, int MgPad, int NgPad,
    float* G, int ldg, float* colsum, int accumulate)
{
    __shared__ double red[RD_FIN_Q][RD_FIN_E];
    const int el = threadIdx.x & (RD_FIN_E - 1), q = threadIdx.x / RD_FIN_E;
    const long total = (long)Mg * Ng;
    const long e = (long)blockIdx.x * RD_FIN_E + el;
    // blocks [0, ceil(total/32)) reduce G; the blocks after them reduce the column sums
    const long g_blocks = (total + RD_FIN_E - 1) / RD_FIN_E;
    const bool is_cs = (long)blockIdx.x >= g_blocks;
    double s0 = 0, s1 = 0;
    if (!is_cs) {
        if (e < total) {
            const int m = (int)(e / Ng), n = (int)(e - (long)m * Ng);
            const size_t stride = (size_t)MgPad * NgPad;
            const float* src = part + (size_t)m * NgPad + n;
            int b = q;
            for (; b + RD_FIN_Q < nblk; b += 2 * RD_FIN_Q) {
                s0 += (double)src[(size_t)b * stride];
                s1 += (double)src[(size_t)(b + RD_FIN_Q) * stride];
            }
            if (b < nblk) s0 += (double)src[(size_t)b * stride];
        }
    } else {
        const long c = ((long)blockIdx.x - g_blocks) * RD_FIN_E + el;
        if (c < Mg)
            for (int b = q; b < nblk; b += RD_FIN_Q) s0 += part_cs[(size_t)b * MgPad + c];
    }
    red[q][el] = s0 + s1;
    __syncthreads();
    if (q == 0) {
        double s = red[0][el];
#pragma unroll
        for (int i = 1; i < RD_FIN_Q; ++i) s += red[i][el];
        if (!is_cs) {
            if (e < total) {
                const int m = (int)(e / Ng), n = (int)(e - (long)m * Ng);
                float* d = G + (size_t)m * ldg + n;
                *d = accumulate ? (*d + (float)s) : (float)s;
            }
        } else {
            const long c = ((long)blockIdx.x - g_blocks) * RD_FIN_E + el;
            if (c < Mg) colsum[c] = accumulate ? (colsum[c] + (float)s) : (float)s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
static int rd_pick(int need, const int* opts, int n)
{
    for (int i = 0; i < n; ++i) if (opts[i] >= need) return opts[i];
    return -1;
}
static const int RD_MH_OPTS[3] = {2, 5, 7};
static const int RD_NH_OPTS[4] = {1, 5, 7, 8};

static int rd_num_cus() { return gpe_num_cus(); }

// geometry shared by the workspace query and the launcher (no device query here: the ws size must be computable on a
// CPU-only box, so it is sized for the largest grid we ever launch)
#define RD_MAX_GX 256
// rows < 0: workspace query -> the largest M block (an upper bound of every geometry below: gx * MgPad grows with MH).
static void rd_geometry(int Mg, int Ng, long rows, int* MH, int* NH, int* gy, int* MgPad, int* NgPad)
{
    // (64-row output blocks for the row-poor LSTM weight gradients were tried and measured slower: 2.43 vs 1.95 ms per
    // step — only 16 of 64 staging lanes carry U columns)
    const int mt = gpe_cdiv(Mg, 16), nt = gpe_cdiv(Ng, 16);
    const int mtb = mt < 14 ? mt : 14;
    int mh = rd_pick(gpe_cdiv(mtb, 2), RD_MH_OPTS, 3);
    // (r02: for the row-poor LSTM weight gradients — 10 k rows against a 1000 x 250 output — a narrower M block with fewer
    // row splits was tried: partials 52 -> 16 MB, but the reduce-GEMM time went UP, 1.85 -> 2.56 ms per step: with 2 M-tiles
    // per wave the V operand is re-read 3.5x as often from LDS and the MFMA stream is too short to hide it.  `rows` stays in
    // the signature for the next attempt.)
    (void)rows;
    *MH = mh;
    *NH = rd_pick(gpe_cdiv(nt, 2), RD_NH_OPTS, 4);
    *gy = gpe_cdiv(mt, 2 * (*MH));
    *MgPad = (*gy) * 32 * (*MH);
    *NgPad = gpe_round_up(Ng, 16);
}

extern "C" long gpe_redgemm_ws(int Mg, int Ng)
{
    int MH, NH, gy, MgPad, NgPad;
    rd_geometry(Mg, Ng, -1, &MH, &NH, &gy, &MgPad, &NgPad);
    if (NH < 0) return -1;
    const long gx = RD_MAX_GX / gy > 0 ? RD_MAX_GX / gy : 1;
    const long big = gx * MgPad * NgPad + 2L * gx * MgPad + 8;
    const long dM = gpe_round_up(Mg, RDD_B), dN = gpe_round_up(Ng, RDD_B);
    const long deep = RDD_MAX_GX * dM * dN + 2L * RDD_MAX_GX * dM + 8;
    const long mr = gpe_round_up(Mg, 4);
    const long thin = Ng <= 4 ? (long)RDT_GX * mr * 4 + 2L * RDT_GX * mr + 8 : 0;      // gpe_redgemm_thin_kernel
    const long m2 = big > deep ? big : deep;
    const long m3 = m2 > thin ? m2 : thin;
    const long x6 = gpe_gemm_x6_red_ws(Mg, Ng);
    return m3 > x6 ? m3 : x6;
}

template <int MH, int NH, int VMODE>
static int rd_launch(const RdParams& p, dim3 grid, hipStream_t s)
{
    const size_t lds = (size_t)2 * RD_RT * ((32 * MH + 16) + (32 * NH + 16)) * sizeof(float);
    GPE_ENSURE_MAX_LDS((gpe_redgemm_kernel<MH, NH, VMODE>));
    hipLaunchKernelGGL((gpe_redgemm_kernel<MH, NH, VMODE>), grid, dim3(256), lds, s, p);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

template <int MT, int NT, int VMODE>
static int rd_pc_launch(const RdParams& p, int gx, hipStream_t s)
{
    constexpr int UC = 16 * MT, VC = 16 * NT;
    constexpr int LDU = (UC % 32 == 16) ? UC : UC + 16;
    constexpr int LDV = (VC % 32 == 16) ? VC : VC + 16;
    const size_t lds = (size_t)2 * RD_RT * (LDU + LDV) * sizeof(float);
    GPE_ENSURE_MAX_LDS((gpe_redgemm_pc_kernel<MT, NT, VMODE>));
    hipLaunchKernelGGL((gpe_redgemm_pc_kernel<MT, NT, VMODE>), dim3(gx), dim3(512), lds, s, p);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

template <int MT, int NT, int VMODE, bool F16 = false, bool LAZY = false>
static int rd_b3_launch(const RdParams& p, int gx, hipStream_t s)
{
    const size_t lds = (size_t)2 * (RdB3Layout<MT>::BYTES + RdB3Layout<NT>::BYTES);
    GPE_ENSURE_MAX_LDS((gpe_redgemm_b3_kernel<MT, NT, VMODE, F16, LAZY>));
    hipLaunchKernelGGL((gpe_redgemm_b3_kernel<MT, NT, VMODE, F16, LAZY>), dim3(gx), dim3(512), lds, s, p);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

template <int VMODE>
static int rd_dispatch(int MH, int NH, const RdParams& p, dim3 grid, hipStream_t s)
{
#define RD_CASE(M_, N_) if (MH == M_ && NH == N_) return rd_launch<M_, N_, VMODE>(p, grid, s)
    RD_CASE(2, 1); RD_CASE(2, 5); RD_CASE(2, 7); RD_CASE(2, 8);
    RD_CASE(5, 1); RD_CASE(5, 5); RD_CASE(5, 7); RD_CASE(5, 8);
    RD_CASE(7, 1); RD_CASE(7, 5); RD_CASE(7, 7); RD_CASE(7, 8);
#undef RD_CASE
    return GPE_EINVAL;
}

static bool rd_rows_vec(const GpeRows& r, int cols)
{
    return r.inner <= 0 && !(r.stride_outer & 3) && r.stride_outer >= ((cols + 3) & ~3) && !(((uintptr_t)r.base) & 15);
}

static int rd_run(RdParams& p, int vmode, float* G, int ldG, float* colsum, float* part, int accumulate,
                  hipStream_t s)
{
    // aligned, 4-padded rows take the plain unconditional 16-B loader (all RQ loads in flight); anything else the
    // guarded scalar-tail loader
    p.vec = rd_rows_vec(p.u, p.Mg) && (vmode == V_GATHER || rd_rows_vec(p.v, p.Ng));
    int MH, NH, gy, MgPad, NgPad;
    rd_geometry(p.Mg, p.Ng, p.rows, &MH, &NH, &gy, &MgPad, &NgPad);
    if (MH < 0 || NH < 0) return GPE_EINVAL;
    p.MgPad = MgPad; p.NgPad = NgPad;
    p.num_tiles = gpe_cdiv(p.rows, RD_RT);
    int cus = rd_num_cus();
    if (cus > RD_MAX_GX) cus = RD_MAX_GX;
    int gx = cus / gy;
    if (gx < 1) gx = 1;
    if (gx > p.num_tiles) gx = p.num_tiles > 0 ? p.num_tiles : 1;
    p.part = part;
    // row-poor dense products (fewer than 64 row tiles per workgroup of the big-block grid) with a 16-B loadable U
    // thin products: stream U once (Ng <= 4, plain 16-B loadable U rows, single-level rows on both sides)
    if (vmode == V_DENSE && !p.lz_g && p.Ng <= 4 && p.Mg <= 1024 && p.rows >= 4096 && p.u.inner <= 0 && p.v.inner <= 0 &&
        rd_rows_vec(p.u, p.Mg)) {
        const int gxt = (int)(p.rows / 64 < RDT_GX ? p.rows / 64 : RDT_GX);
        p.MgPad = gpe_round_up(p.Mg, 4); p.NgPad = 4;
        size_t toff = (size_t)gxt * p.MgPad * 4;
        toff = (toff + 1) & ~(size_t)1;
        p.part_cs = colsum ? reinterpret_cast<double*>(part + toff) : nullptr;
        const size_t lds = (size_t)4 * p.MgPad * 5 * sizeof(float);
        const int ql = gpe_cdiv(p.MgPad, 256);
        if (ql <= 1) { GPE_ENSURE_MAX_LDS((gpe_redgemm_thin_kernel<1>)); hipLaunchKernelGGL(gpe_redgemm_thin_kernel<1>, dim3(gxt), dim3(256), lds, s, p); }
        else if (ql == 2) { GPE_ENSURE_MAX_LDS((gpe_redgemm_thin_kernel<2>)); hipLaunchKernelGGL(gpe_redgemm_thin_kernel<2>, dim3(gxt), dim3(256), lds, s, p); }
        else { GPE_ENSURE_MAX_LDS((gpe_redgemm_thin_kernel<4>)); hipLaunchKernelGGL(gpe_redgemm_thin_kernel<4>, dim3(gxt), dim3(256), lds, s, p); }
        GPE_CHECK_LAUNCH();
        const long fin_t = gpe_cdiv((long)p.Mg * p.Ng, RD_FIN_E) + (colsum ? gpe_cdiv(p.Mg, RD_FIN_E) : 0);
        hipLaunchKernelGGL(gpe_redgemm_finish, dim3(fin_t), dim3(RD_FIN_E * RD_FIN_Q), 0, s, p.part, p.part_cs, gxt, p.Mg,
                           p.Ng, p.MgPad, p.NgPad, G, ldG, colsum, accumulate);
        GPE_CHECK_LAUNCH();
        return GPE_OK;
    }
    // f16x3 mode: row-rich dense products off the edge kernels' menu (the decoders' weight gradients, 10304 rows x 1000 x 250; the
    // [P|Q] projection's, 65536 x 400 x 150) on the bf16 pipe, three-term splits (gpe_gemm_x6.hip): same partial image, same finish
    if (vmode == V_DENSE && g_rd_math == 2 && !p.lz_g && !(gpe_debug_get() & 16384) &&
        !(gpe_cdiv(p.Ng, 16) == 13 && (gpe_cdiv(p.Mg, 16) == 13 || gpe_cdiv(p.Mg, 16) == 10) && gy == 1 && p.amax_u && p.amax_v)) {
        int ns = 0, mp = 0, np = 0;
        double* pcs = nullptr;
        const int rc = gpe_gemm_x6_redgemm(p.u, p.v, p.v_shift, p.rows, p.Mg, p.Ng, part, colsum != nullptr, &ns, &mp, &np, &pcs, s);
        if (rc < 0) return rc;
        if (rc == 1) {
            const long fin_x = gpe_cdiv((long)p.Mg * p.Ng, RD_FIN_E) + (colsum ? gpe_cdiv(p.Mg, RD_FIN_E) : 0);
            hipLaunchKernelGGL(gpe_redgemm_finish, dim3(fin_x), dim3(RD_FIN_E * RD_FIN_Q), 0, s, part, pcs, ns, p.Mg, p.Ng, mp, np, G, ldG,
                               colsum, accumulate);
            GPE_CHECK_LAUNCH();
            return GPE_OK;
        }
    }
    const long in_max = (p.u.inner > p.v.inner ? p.u.inner : p.v.inner) > 1 ? (p.u.inner > p.v.inner ? p.u.inner : p.v.inner) : 1;
    static const int dbg_deep = gpe_dbg_env("GPE_RD_DEEP", -1);   // measurement override: 0 = never the deep kernel
    // the edge weight-gradient shapes (<= 208 x 208 outputs, plain 16-B rows, >= 4 row tiles per workgroup) stay on the
    // producer/consumer kernels below at every size: measured (profiles/r04_h_rd_paths.md, dense V, 150 x 200) 48 / 64 / 106 / 206 us
    // against the deep kernel's 70 / 107 / 204 / 403 us at E = 41 k / 66 k / 131 k / 262 k rows (f16x3: 46 / 53 / 70 / 127 us); the deep
    // kernel keeps the row-poor decoder products it was built for (10 k rows x 1000 x 250: 79 against 154 us)
    const bool edge_shape = gpe_cdiv(p.Ng, 16) == 13 && (gpe_cdiv(p.Mg, 16) == 13 || gpe_cdiv(p.Mg, 16) == 10) && gy == 1 &&
                            rd_rows_vec(p.u, p.Mg) && rd_rows_vec(p.v, p.Ng) && p.num_tiles >= 4L * gx;
    if (vmode == V_DENSE && g_rd_math != 1 && !p.lz_g && !edge_shape && dbg_deep != 0 && p.num_tiles > 0 && p.num_tiles < 64L * gx && p.rows * in_max < (1L << 32) &&
        p.rows < (1L << 31) && rd_rows_vec2(p.u, p.Mg)) {
        p.umagic = p.u.inner > 1 ? (unsigned)(((1ull << 32) + p.u.inner - 1) / p.u.inner) : 0;
        p.vmagic = p.v.inner > 1 ? (unsigned)(((1ull << 32) + p.v.inner - 1) / p.v.inner) : 0;
        const int dgx = rdd_gx(p.Mg, p.Ng, p.num_tiles, cus);
        const int dgy = gpe_cdiv(p.Mg, RDD_B), dgz = gpe_cdiv(p.Ng, RDD_B);
        p.MgPad = dgy * RDD_B; p.NgPad = dgz * RDD_B;
        size_t doff = (size_t)dgx * p.MgPad * p.NgPad;
        doff = (doff + 1) & ~(size_t)1;
        p.part_cs = colsum ? reinterpret_cast<double*>(part + doff) : nullptr;
        const size_t lds = (size_t)4 * RD_RT * RDD_LD * sizeof(float);
        if (rd_rows_vec2(p.v, p.Ng)) hipLaunchKernelGGL(gpe_redgemm_deep_kernel<true>, dim3(dgx, dgy, dgz), dim3(256), lds, s, p);
        else hipLaunchKernelGGL(gpe_redgemm_deep_kernel<false>, dim3(dgx, dgy, dgz), dim3(256), lds, s, p);
        GPE_CHECK_LAUNCH();
        const long total_d = (long)p.Mg * p.Ng;
        const long fin_d = gpe_cdiv(total_d, RD_FIN_E) + (colsum ? gpe_cdiv(p.Mg, RD_FIN_E) : 0);
        hipLaunchKernelGGL(gpe_redgemm_finish, dim3(fin_d), dim3(RD_FIN_E * RD_FIN_Q), 0, s, p.part, p.part_cs, dgx, p.Mg,
                           p.Ng, p.MgPad, p.NgPad, G, ldG, colsum, accumulate);
        GPE_CHECK_LAUNCH();
        return GPE_OK;
    }
    size_t off = (size_t)gx * MgPad * NgPad;
    off = (off + 1) & ~(size_t)1;                                  // 8-B align the fp64 section
    p.part_cs = reinterpret_cast<double*>(part + off);
    dim3 grid(gx, gy);
    int rc = GPE_EINVAL;
    const int mt_all = gpe_cdiv(p.Mg, 16), nt_all = gpe_cdiv(p.Ng, 16);
    static const int dbg_nopc = gpe_dbg_env("GPE_RD_NOPC", 0);     // measurement override
    const bool pc_ok = !dbg_nopc && gy == 1 && nt_all == 13 && (mt_all == 13 || mt_all == 10) && rd_rows_vec(p.u, p.Mg) &&
                       (vmode == V_GATHER || rd_rows_vec(p.v, p.Ng)) && p.num_tiles >= 4 * gx &&
                       (vmode != V_GATHER || (p.k > 1 && p.rows * p.k < (1L << 32)));   // umulhi row / k (kmagic)
    p.pin_tpc = 0;
    if (pc_ok && vmode == V_GATHER && p.pin_clouds > 0 && gpe_pin_clouds(p.pin_clouds) && p.pin_clouds % GPE_NXCD == 0) {
        const long rows_per_cloud = p.rows / p.pin_clouds;
        if (rows_per_cloud % RD_RT == 0 && gx % GPE_NXCD == 0 && rows_per_cloud / RD_RT >= gx / GPE_NXCD)
            p.pin_tpc = (int)(rows_per_cloud / RD_RT);
    }
    if (p.lz_g) {
        // lazy dz3: only the f16x3 dense-V kernel forms U on the fly; the caller asked gpe_edge_lazy_dz3_ok first
        if (!(pc_ok && g_rd_math == 2 && p.amax_u && p.amax_v && vmode == V_DENSE && p.k == 16 && (p.rows & 15) == 0)) return GPE_EINVAL;
        rc = (mt_all == 13) ? rd_b3_launch<13, 13, V_DENSE, true, true>(p, gx, s) : rd_b3_launch<10, 13, V_DENSE, true, true>(p, gx, s);
    } else if (pc_ok && g_rd_math == 2 && p.amax_u && p.amax_v) {
        if (mt_all == 13 && vmode == V_DENSE) rc = rd_b3_launch<13, 13, V_DENSE, true>(p, gx, s);
        else if (mt_all == 13) rc = rd_b3_launch<13, 13, V_GATHER, true>(p, gx, s);
        else if (vmode == V_DENSE) rc = rd_b3_launch<10, 13, V_DENSE, true>(p, gx, s);
        else rc = rd_b3_launch<10, 13, V_GATHER, true>(p, gx, s);
    } else if (pc_ok && g_rd_math == 1) {
        if (mt_all == 13 && vmode == V_DENSE) rc = rd_b3_launch<13, 13, V_DENSE>(p, gx, s);
        else if (mt_all == 13) rc = rd_b3_launch<13, 13, V_GATHER>(p, gx, s);
        else if (vmode == V_DENSE) rc = rd_b3_launch<10, 13, V_DENSE>(p, gx, s);
        else rc = rd_b3_launch<10, 13, V_GATHER>(p, gx, s);
    } else if (pc_ok) {
        if (mt_all == 13 && vmode == V_DENSE) rc = rd_pc_launch<13, 13, V_DENSE>(p, gx, s);
        else if (mt_all == 13) rc = rd_pc_launch<13, 13, V_GATHER>(p, gx, s);
        else if (vmode == V_DENSE) rc = rd_pc_launch<10, 13, V_DENSE>(p, gx, s);
        else rc = rd_pc_launch<10, 13, V_GATHER>(p, gx, s);
    } else
        rc = (vmode == V_DENSE) ? rd_dispatch<V_DENSE>(MH, NH, p, grid, s) : rd_dispatch<V_GATHER>(MH, NH, p, grid, s);
    if (rc != GPE_OK) return rc;
    const long total = (long)p.Mg * p.Ng;
    const long fin_blocks = gpe_cdiv(total, RD_FIN_E) + (colsum ? gpe_cdiv(p.Mg, RD_FIN_E) : 0);
    hipLaunchKernelGGL(gpe_redgemm_finish, dim3(fin_blocks), dim3(RD_FIN_E * RD_FIN_Q), 0, s, p.part, p.part_cs, gx,
                       p.Mg, p.Ng, MgPad, NgPad, G, ldG, colsum, accumulate);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

extern "C" int gpe_redgemm(const float* u, long u_so, long u_si, int u_inner, const float* v, long v_so,
                           long v_si, int v_inner, const float* v_shift, long rows, int Mg, int Ng, float* G,
                           int ldg, float* colsum, float* part, int accumulate, void* stream)
{
    if (!u || !v || !G || !part || rows < 0 || Mg <= 0 || Ng <= 0 || Ng > 256 || ldg < Ng) return GPE_EINVAL;
    RdParams p = {};
    p.rows = rows; p.Mg = Mg; p.Ng = Ng;
    p.u = GpeRows{u, u_so, u_si, u_inner};
    p.v = GpeRows{v, v_so, v_si, v_inner};
    p.v_shift = v_shift;
    return rd_run(p, V_DENSE, G, ldg, colsum, part, accumulate, (hipStream_t)stream);
}

extern "C" int gpe_edge_redgemm(const float* u, int ldu, int v_mode, const float* v, int ldv, const float* pq,
                                int ldpq, const int32_t* jg, const float* v_shift, int B, int N, int k, int Mg,
                                int Ng, float* G, int ldG, float* colsum, float* part, const uint32_t* amax_u,
                                const uint32_t* amax_v, void* ws, long ws_bytes, const float* lz_g, int lz_ldg,
                                const uint8_t* lz_amx, const uint8_t* lz_amn, int lz_ldagg, const float* lz_coef, void* stream)
{
    if (!u || !G || !part || B <= 0 || N <= 0 || k <= 0 || Mg <= 0 || Ng <= 0 || Ng > 256 || ldG < Ng || ldu < Mg)
        return GPE_EINVAL;
    if (v_mode == 0 && (!pq || !jg || (Ng & 3) || (ldpq & 3))) return GPE_EINVAL;
    if ((long)B * N * k >= (1L << 31)) return GPE_EINVAL;
    if (v_mode == 1 && (!v || ldv < Ng)) return GPE_EINVAL;
    RdParams p = {};
    p.rows = (long)B * N * k; p.Mg = Mg; p.Ng = Ng;
    p.u = GpeRows{u, ldu, 0, 0};
    p.v = GpeRows{v, ldv, 0, 0};
    p.pq = pq; p.ldpq = ldpq; p.H = Ng; p.jg = jg; p.k = k; p.rcp_k = 1.0 / k; p.kmagic = (unsigned)(((1ull << 32) + k - 1) / k); p.v_shift = v_shift;
    p.pin_clouds = B;
    if (lz_g) {
        if (!lz_amx || !lz_amn || !lz_coef || lz_ldg < Mg || (lz_ldagg & 3) || lz_ldagg < ((Mg + 3) & ~3) || k != 16 || v_mode != 1 ||
            !amax_u || !amax_v)
            return GPE_EINVAL;
        p.lz_g = lz_g; p.lz_ldg = lz_ldg; p.lz_amx = lz_amx; p.lz_amn = lz_amn; p.lz_ldagg = lz_ldagg; p.lz_coef = lz_coef;
    }
    if (g_rd_math == 2 && amax_u && p.rows >= gpe_h3_min_rows()) {
        // f16x3: the operand scales must be known without a pass over an E-row tensor — U (a dz tensor) through the caller's amax
        // word of it, V through the caller's word (dense: the amax the forward kernel measured; gathered: a bound of
        // relu(P_i + Q_j), gpe_edge_pq_amax) or, for the gathered operand only, the bound passes run here; else exact fp32
        p.amax_u = amax_u;
        p.amax_v = amax_v;
        if (!p.amax_v && v_mode == 0) {
            const GpeEdgeWs w = gpe_edge_ws(ws, ws_bytes);
            if (w.h3 && gpe_h3_pq_passes(w.h3, reinterpret_cast<float*>(w.h3 + 2), pq, (long)B * N, Ng, ldpq, (hipStream_t)stream) == GPE_OK)
                p.amax_v = w.h3;
        }
        if (!p.amax_v) p.amax_u = nullptr;
    }
    return rd_run(p, v_mode == 0 ? V_GATHER : V_DENSE, G, ldG, colsum, part, 0, (hipStream_t)stream);
}
