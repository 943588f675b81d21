// Two waves per SIMD for the f16x3 edge kernels of the benchmark configuration (k = 16, widths 200 / 200 / 150): the per-edge MLP of
// DynamicEdgeConv (/root/reference/nn/net_blocks.py:43-47,124-135 forward; its input-gradient half in backward), same arithmetic,
// same LDS image and same epilogues as gpe_edgegemm_split_kernel.h (SplitF16x2: three v_mfma_f32_16x16x32_f16 per fp32 product on
// tensor-normalised two-term fp16 planes), other division of labour.
//
// Why (profiles/r04_m_sq_wave_states.md): the single-role kernel runs ONE wave per SIMD that does everything
// for its share — stage 16 rows of the next tile, multiply, finish 16 rows of the previous tile — and measures matrix pipe 0.17-0.27
// busy, 0.2-0.3 of its cycles parked at waits and 0.2 issue-stalled: neither roof.  A lone wave cannot issue its epilogue VALU while
// it waits for its own LDS fragment, and its 184 resident weight registers leave no room for a second wave.
//
// Here: ONE persistent 512-thread workgroup per CU = two waves per SIMD, 256 registers each.
//   * MFMA work of a 64-row tile is split over the 8 waves by OUTPUT TILE, the left-overs by K:
//       wave w owns output tile w for the whole K (KS slabs of 32);
//       NT = 13: tiles 8..11 are split in two K halves over the wave pairs (w, w + 4), tile 12 in KS single slabs over the waves;
//       NT = 10: tiles 8, 9 are split in four K quarters over the waves of one role each.
//     A wave walks the K slabs in ROTATED order (start `rot`, wave-uniform) chosen so that its slabs of the split tiles are its
//     iterations 0 .. SL2-1 (and 0 for tile 12): every register index is a compile-time constant and every A fragment is read once.
//     Resident weights: (KS + SL2 + 1) slabs x 2 planes x 4 registers = 96 (13 x 13), against 184 in the single-role kernel.
//   * The memory pipeline is split by ROLE: waves 0-3 ("stagers") load and commit the 16 rows of point (wave & 3) of the NEXT tile,
//     waves 4-7 ("finishers") run the epilogue of the 16 rows of point (wave & 3) of the PREVIOUS tile.  A SIMD holds one wave of
//     each role (waves w and w + 4), so one wave's epilogue VALU issues under the other's MFMAs and waits.  k = 16: a point is still
//     finished by ONE wave — max / min tracking and the per-point sums need no exchange.
//   * The K-partials of the split tiles meet in the rows of the just-consumed A buffer (as in the single-role kernel); the
//     finisher of a point folds them into the C tile right after the tile's second barrier, and the stager of the same point —
//     the only wave that re-stages those rows — waits for the finisher's flag before its first commit (one LDS word per pair).
#pragma once
#include "gpe_edgegemm_split_kernel.h"
#include <type_traits>

// scheduling fence at the end of every slot (per translation unit, build.py): see the measurement table in gpe_edgegemm_w8.hip
#ifndef W8_SLOT_FENCE
#define W8_SLOT_FENCE 1
#endif
typedef unsigned w8_u32x2 __attribute__((ext_vector_type(2)));
// keeps a loaded quad (and therefore its load) in front of this point: see the finisher's epilogue
#ifndef W8_B2_SC1
#define W8_B2_SC1 1
#endif
#ifndef W8_BUFSTORE
// bit 0 F2, 1 F3, 2 B3, 3 B2: row stores through the buffer descriptor.  NOT the in-place backward (B3): it loads the rows it later
// overwrites through plain global pointers, the compiler takes buffer accesses and global accesses for disjoint memory, and the
// mixed form measurably loses stores' ordering against those loads (gradients 10 % off and different from run to run:
// test_lazy_dz3_matches_the_in_place_pass, test_two_streams_one_device; bisected per kernel on the GPU)
#define W8_BUFSTORE 11
#endif
__device__ __forceinline__ void w8_pin4(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

// KK = rows per point, compile time: 16 (BASELINE cfg 2 / 3 / 5: one point of 16 rows per wave and tile), 5 (the shipped YAMLs,
// models/att/att.yaml:94: three points of 5 rows per wave, 60-row tiles — rows 60..63 of the LDS tile belong to nobody: they are
// multiplied like the others and never read) or 4 (BASELINE cfg 4, k = 20: a point runs as five PSEUDO-points of four rows whose
// per-point results are folded afterwards — gpe_edge_pseudo_setup / _fold, gpe_edgegemm_sr.hip; RgParams::pmagic != 0 then turns
// a pseudo-point number into its P row; rows that need nothing per point are simply tiled by four).  Every per-row decision (which point, which slot, where a point ends) is a
// compile-time function of the row number u, so the slots stay straight-line code for both.
template <int NT, int KCH, int AMODE, int EMODE, int AGGT, bool LAZY, int KK = 16>
__global__ __launch_bounds__(512) void gpe_edgegemm_w8_kernel(RgParams p, int stats_nblk)
{
    using SP = SplitF16x2;
    static_assert(KK == 16 || KK == 5 || KK == 4, "neighbourhood sizes with a compile-time row schedule");
    static_assert(!LAZY || KK == 16, "lazy dz3 needs one point per wave");
    constexpr int NPW = 16 / KK;                         // points per wave and tile
    constexpr int RW = NPW * KK;                         // rows a wave stages / finishes per tile (16 or 15)
    constexpr int TR = 4 * RW;                           // rows per tile (64 or 60)
    static_assert(NT == 13 || (NT == 10 && KCH == 13), "shapes of the shipped edge MLPs");
    static_assert(KCH == 13 || KCH == 10, "K = 200 / 150");
    static_assert(!LAZY || (AMODE == A_DENSE && EMODE == E_BWD_INPLACE), "lazy dz3: in-place backward");
    static_assert(EMODE != E_BWD_INPLACE || (W8_BUFSTORE & 4) == 0, "the in-place backward loads the rows it overwrites through plain "
                  "pointers: its stores must be plain too (buffer and global accesses are disjoint memory to the compiler)");
    constexpr bool FWD = EMODE == E_EDGE_FWD;
    constexpr bool TRACK = FWD && AGGT != 0;
    constexpr bool OUTH = FWD && AGGT == 2;              // activation rows stored as _Float16 (RgParams::out_half)
    constexpr bool GATHER_ACT = EMODE == E_BWD_GATHER;
    constexpr int KS = (KCH + 1) / 2;
    constexpr bool KTAIL = (KCH & 1) != 0;               // last slab holds only 16 k: lane groups g >= 2 contribute zeros
    constexpr bool PAIRS = NT == 13;
    constexpr int SL2 = PAIRS ? (KS + 1) / 2 : (KS + 3) / 4;     // slabs of a split tile per wave
    constexpr int PPITCH = 16 * x6_pchunks(KCH);         // bytes per plane row
    constexpr int PLANE = RG_BM * PPITCH;                // bytes per plane
    constexpr int AWORDS = (2 * PLANE) / 4;              // one A buffer (h plane, l plane), in floats
    constexpr int LDC = 16 * NT + 4;
    constexpr int NSLOT = 4 * KS;                        // slot q = 4 * iteration + mtile
    // K-partials of the split tiles: 64-byte slots (16 floats) in the plane rows of the consumed A buffer
    // (only inside the bytes every commit rewrites: a partial left in a pad chunk would meet the next tile's MFMAs as fp16 garbage —
    // NaN x zero weight.  The dispatcher checks 2 * roundup(K, 4) >= 64 * SPP.)
    constexpr int SPP = (KCH == 13) ? 6 : 4;             // slots per plane row
    constexpr int NLS = PAIRS ? KS - 1 : 0;              // tile 12: one partial goes to C directly, KS - 1 to slots
    constexpr int NSCR = PAIRS ? 4 + NLS : 6;
    static_assert(NSCR <= 2 * SPP, "partials do not fit the row");
    // stager: two batches of 8 rows through the same registers
    constexpr int C0 = (NSLOT >= 28) ? 8 : 6;            // batch 0: issued in slot 0, committed in slots C0 .. C0 + 3 (2 rows each)
    constexpr int I1 = C0 + 4;                           // batch 1: issued in slot I1, committed in slots C1 .. C1 + 3
    constexpr int C1 = (NSLOT >= 28) ? 22 : 16;
    // finisher: one row per slot in the last 16 slots
    constexpr int EP0 = NSLOT - 16;
    // stored-activation rows in flight per finisher (backward variants): a ring; the gathered backward with several points per wave
    // (three or four P rows, per-point sums) has 8 registers fewer to give
    constexpr int RING = (GATHER_ACT && NPW > 1) ? 6 : 8;

    extern __shared__ __align__(16) float smem[];
    __shared__ unsigned amax_sh[4];
    __shared__ int flag_sh[4];
    float* const Abuf0 = smem;
    float* const Abuf1 = smem + AWORDS;
    float* const Cs = smem + 2 * AWORDS;                 // [64][LDC]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, pt = wave & 3;           // role 0 stages, role 1 finishes; both for point `pt` of a tile
    const int j = lane & 15, g = lane >> 4;
    const int g_tail = (g >= 2) ? (g & 1) : g;
    const int rb = RW * pt;
    const int c = lane << 2;                             // this lane's column quad
    const bool n_on = c < p.N;
    // staging is branch-free: lanes past the last K quad repeat that quad (same address, same data) instead of being masked — a load
    // whose only use sits in a conditional block is sunk into it by the compiler and then waits for its own round trip
    const int ck = (c < p.K) ? c : ((p.K - 1) & ~3);
    // ... and so is the epilogue: lanes past the last N quad repeat it (their statistics are never read)
    const int cn = n_on ? c : ((p.N - 1) & ~3);

    // ---- division of the MFMA work (all wave-uniform) -----------------------------------------------------------------------
    //   rot        first K slab of this wave's rotated walk: iteration i multiplies slab (rot + i) mod KS
    //   sec_tile   the split tile this wave contributes to in iterations 0 .. SL2-1; iteration i is real while sec_real(i)
    //   PAIRS: pair (pt, pt + 4) shares tile 8 + pt — role 0 takes slabs pt .. pt + SL2-1, role 1 the KS - SL2 that follow;
    //          tile 12: slab `rot` (iteration 0) of every wave whose rot is not taken by an earlier wave (role 0: rot = pt = 0..3;
    //          role 1: rot = pt + SL2 mod KS, new exactly when rot >= 4)
    //   else : quad (role) shares tile 8 + role — quarter pt takes slabs 2 pt, 2 pt + 1 (< KS)
    int rot, sec_tile, sec_slot;
    bool sec_direct, last_on = false, last_direct = false;
    int last_slot = 0;
    if constexpr (PAIRS) {
        rot = role ? pt + SL2 : pt;
        if (rot >= KS) rot -= KS;
        sec_tile = 8 + pt; sec_direct = role == 0; sec_slot = pt;
        last_on = role == 0 || rot >= 4;
        last_direct = wave == 0;
        last_slot = 4 + (role == 0 ? pt - 1 : rot - 1);
    } else {
        rot = 2 * pt;
        sec_tile = 8 + role; sec_direct = pt == 0; sec_slot = 3 * role + pt - 1;
    }

    for (int e = tid; e < 2 * AWORDS; e += 512) smem[e] = 0.f;
    if (tid < 4) flag_sh[tid] = 0;

    float sA, sW, invA, invW;
    SP::scale_of(p.h3_amax_a[0], sA, invA);
    SP::scale_of(p.h3_amax_w[0], sW, invW);

    // ---- weights: resident fp16 B fragments, two planes -------------------------------------------------------------------
    // lane (j, g) of slab sl holds k = 32 sl + 8 g + {0..7} of column 16 * tile + j (see gpe_edgegemm_split_kernel.h)
    x6_u32x4 wP[2][KS], sP[2][SL2], tP[2];
    {
        auto load_frag = [&](int col, int sl) -> X6Frag<SP> {
            const int cc = (col < p.Npad) ? col : p.Npad - 1;
            const int kc = 2 * sl + (g >> 1);
            const bool on = col < p.Npad && kc < KCH;
            const int kcc = kc < KCH ? kc : KCH - 1;
            float4 f0 = ld4(p.wp + (((long)(kcc * 4 + 2 * (g & 1))) * p.Npad + cc) * 4);
            float4 f1 = ld4(p.wp + (((long)(kcc * 4 + 2 * (g & 1) + 1)) * p.Npad + cc) * 4);
            f0 = x6_scale4(f0, sW); f1 = x6_scale4(f1, sW);
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            return x6_split8<SP>(on ? f0 : z, on ? f1 : z);
        };
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            const int sl = (rot + i >= KS) ? rot + i - KS : rot + i;
            const X6Frag<SP> f = load_frag(16 * wave + j, sl);
            wP[0][i] = f.pl[0]; wP[1][i] = f.pl[1];
        }
#pragma unroll
        for (int i = 0; i < SL2; ++i) {
            const int sl = (rot + i >= KS) ? rot + i - KS : rot + i;
            const bool real = PAIRS ? (role == 0 || i < KS - SL2) : (rot + i < KS);
            const X6Frag<SP> f = load_frag(real ? 16 * sec_tile + j : p.Npad, sl);
            sP[0][i] = f.pl[0]; sP[1][i] = f.pl[1];
        }
        {
            const X6Frag<SP> f = load_frag((PAIRS && last_on) ? 16 * 12 + j : p.Npad, rot);
            tP[0] = f.pl[0]; tP[1] = f.pl[1];
        }
    }

    // ---- tile sequence of this workgroup (gpe_common.h: cloud -> XCD pinning, walk direction) ---------------------------------
    GpeTileSeq sq = gpe_tile_seq(p.pin_tpc, p.rev, p.pin_clouds, p.num_tiles);

    int vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
    // per-iteration opaque zero: keeps the per-row scalars (row numbers, LDS offsets) transient instead of hoisted + spilled
    int rbl = rb;
#define W8_REFRESH_SCALARS()                                   \
    {                                                          \
        int sz_;                                               \
        asm volatile("s_mov_b32 %0, 0" : "=s"(sz_));           \
        rbl = rb + sz_;                                        \
    }
    // neighbour rows of this wave's points in `tile`, lane-distributed: lane L <-> row rb + min(L, RW - 1) (clamped into the tile)
    auto load_jgv = [&](int tile) -> int {
        const long row0 = (long)tile * TR;
        const int rv = (int)((p.M - row0 < TR) ? (p.M - row0) : TR);
        int r = rbl + ((lane < RW) ? lane : RW - 1);
        r = (r < rv - 1) ? r : rv - 1;
        return p.jg[row0 + r];
    };
    // P row of (pseudo-)point x (wave-uniform): x itself, or x / f when a k > 16 point runs as f pseudo-points
    // (only the four-row instances are ever launched with pseudo-points)
    auto prow = [&](long x) -> long { return (KK == 4 && p.pmagic) ? (long)__umulhi((unsigned)x, p.pmagic) : x; };
    // float index, inside an A buffer, of partial slot s of row `row`
    auto scr = [&](int row, int s) -> int { return ((s >= SPP ? PLANE + (s - SPP) * 64 : s * 64) + row * PPITCH) >> 2; };

    // ---- the persistent tile loop, one instance per role ----------------------------------------------------------------------
    auto body = [&](auto role_tag) {
        constexpr int ROLE = decltype(role_tag)::value;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

        // ---- stager state ----
        float4 v[8];
        // gather: P rows of the (at most two) points a batch of 8 rows touches — batch h covers points (8 h) / KK and the next one
        constexpr int NPB = NPW > 1 ? 2 : 1;
        float4 pvs[NPB];
#pragma unroll
        for (int q = 0; q < NPB; ++q) pvs[q] = zero4;
        int s_rv = 0, jgv_a = 0, jgv_b = 0;              // neighbour rows of the tile being staged / of the one after it
        float lzs[4] = {0.f, 0.f, 0.f, 0.f}, lznc[4] = {0.f, 0.f, 0.f, 0.f}, lznk[4] = {0.f, 0.f, 0.f, 0.f};
        int lz_sel[4] = {0, 0, 0, 0};
        float lz_won[4] = {0.f, 0.f, 0.f, 0.f};
        float4 lz_gq = zero4;
        uchar4 lz_sx = make_uchar4(0, 0, 0, 0), lz_sn = make_uchar4(0, 0, 0, 0);
        if constexpr (ROLE == 0 && LAZY) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (ck + t < p.K) {
                    // (all three carry the operand scale sA — a power of two, exact — so that commit_row's split needs no multiply)
                    const float k2 = p.lz_coef[2 * p.K + ck + t];
                    lzs[t] = p.lz_coef[ck + t] * sA;
                    lznc[t] = __builtin_fmaf(p.lz_coef[3 * p.K + ck + t], k2, -p.lz_coef[p.K + ck + t]) * sA;
                    lznk[t] = -k2 * sA;
                }
            }
        }
        // ---- finisher state ----
        double stS[4] = {0, 0, 0, 0}, stQ[4] = {0, 0, 0, 0};
        float4 bias4 = zero4, cs4 = zero4, c14 = zero4, k24 = zero4;
        if constexpr (ROLE == 1) {
            {
                if constexpr (FWD) {
                    if (p.bias) {
                        bias4.x = p.bias[cn];
                        if (cn + 1 < p.N) bias4.y = p.bias[cn + 1];
                        if (cn + 2 < p.N) bias4.z = p.bias[cn + 2];
                        if (cn + 3 < p.N) bias4.w = p.bias[cn + 3];
                    }
                } else {                                 // N % 4 == 0 guaranteed by the dispatcher
                    cs4 = ld4(p.coef_out + cn); c14 = ld4(p.coef_out + p.N + cn);
                    cs4 = x6_scale4(cs4, invW * invA);   // both operand scales are undone in the BatchNorm-backward factor of z
                    k24 = ld4(p.coef_out + 2 * p.N + cn);
                    const float4 mu4 = ld4(p.coef_out + 3 * p.N + cn);
                    // dz = (a>0) ? fma(z, s', fma(-k2, a, mean k2 - c1)) : 0
                    c14 = make_float4(__builtin_fmaf(mu4.x, k24.x, -c14.x), __builtin_fmaf(mu4.y, k24.y, -c14.y),
                                      __builtin_fmaf(mu4.z, k24.z, -c14.z), __builtin_fmaf(mu4.w, k24.w, -c14.w));
                    k24 = make_float4(-k24.x, -k24.y, -k24.z, -k24.w);
                }
            }
        }
        const float invAW = invA * invW;
        float amax_run = 0.f;
        float s32[4] = {0.f, 0.f, 0.f, 0.f}, q32[4] = {0.f, 0.f, 0.f, 0.f};
        float vmx[4], vmn[4];
        int imx[4], imn[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { vmx[t] = -INFINITY; vmn[t] = INFINITY; imx[t] = 0; imn[t] = 0; }
        float4 dp = zero4;
        float4 act[(ROLE == 1 && !FWD) ? RING : 1];
        // P rows of the points being finished (E_BWD_GATHER): the current point's and the next one's (ring of two)
        constexpr int NPE = NPW > 3 ? 2 : NPW;           // (three rows fit: the ring costs the k = 5 instance more registers than it saves)
        float4 pve[NPE];
        long e_ptl = 0;
#pragma unroll
        for (int q = 0; q < NPE; ++q) pve[q] = zero4;
        long e_row0 = 0, e_pt = 0; int e_rv = 0;         // tile being finished; e_pt = this wave's first point
        // The epilogue of a row sits under ONE wave-uniform branch (nothing to finish in the first iteration and for an absent point of
        // a ragged last tile).  The compiler sinks a load whose only use lies in a conditional block INTO the block, where it then
        // waits for its own round trip — so the row's C quad and stored activation are pinned (w8_pin4) in front of the branch: the
        // loads stay where they are issued, one slot / one ring turn ahead.  (Fully branch-free slots were built first — stores through
        // a zero-sized descriptor — and spilled 50 - 90 registers: the scheduler stretches every live range across the tile.)
        // Row stores go through a buffer descriptor of the tile's `out` rows: scalar row offset, constant lane offset, no 64-bit
        // address arithmetic, and the sc1 (write-through, not kept in L2) flavour without inline assembly — behind an `asm volatile`
        // store the compiler waited vmcnt(0) before every row, i.e. for the previous row's write acknowledgement.
        bool epi_on = false;
        __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0, 0x00020000);
        float amax_t = 0.f;
        int jgv_e = 0, jgv_cur = 0;                      // neighbour rows of the tile being multiplied / being finished

        // ---- stager: VMEM issue + LDS commit --------------------------------------------------------------------------------
        auto issue_stage_loads = [&](int tile_s, int h) {
            const long row0 = (long)tile_s * TR;
            s_rv = (int)((p.M - row0 < TR) ? (p.M - row0) : TR);
            const int last = s_rv - 1;
            // the first point of this wave's rows, clamped into the last valid point of the tile
            const long pt0 = (long)tile_s * (4 * NPW) + pt * NPW, ptl = (long)tile_s * (4 * NPW) + last / KK;
            const long ptc = pt0 < ptl ? pt0 : ptl;
            if (h == 0) {
                if constexpr (LAZY) {
                    // (issued AHEAD of the row loads: the memory counter retires in order; dword loads at clamped columns — the
                    // gradient rows are the caller's [B*N, K] tensor as it is)
                    const float* gr = p.lz_g + ptc * p.lz_ldg + ck;
                    const int rem = p.K - 1 - ck;
                    lz_gq = make_float4(gr[0], gr[rem < 1 ? rem : 1], gr[rem < 2 ? rem : 2], gr[rem < 3 ? rem : 3]);
                    lz_sx = *reinterpret_cast<const uchar4*>(p.lz_amx + ptc * p.lz_ldagg + ck);
                    lz_sn = *reinterpret_cast<const uchar4*>(p.lz_amn + ptc * p.lz_ldagg + ck);
                }
            }
            if constexpr (AMODE == A_GATHER) {
                const int pb = (8 * h) / KK;             // first point of this batch (wave-relative)
#pragma unroll
                for (int q = 0; q < NPB; ++q) {
                    const long pg = pt0 + ((pb + q < NPW) ? pb + q : NPW - 1);
                    pvs[q] = ld4(p.pq + (prow(pg < ptl ? pg : ptl) + vz) * p.ldpq + ck);
                }
            }
#pragma unroll
            for (int uu = 0; uu < 8; ++uu) {
                const int u = (8 * h + uu < RW) ? 8 * h + uu : RW - 1;     // (KK = 5: the 16th row does not exist — loaded twice, committed never)
                int r = rbl + u;
                r = (r < last) ? r : last;
                if constexpr (AMODE == A_GATHER) {
                    const int jj = __builtin_amdgcn_ds_bpermute(u << 2, jgv_a);
                    v[uu] = ld4(p.pq + (long)jj * p.ldpq + p.H + ck);
                } else if constexpr (LAZY) {
                    // the stored activation is fp16 (8 bytes per quad; pitch in halves): the raw words travel in v[].x / .y
                    const uint2 hq = *reinterpret_cast<const uint2*>(reinterpret_cast<const _Float16*>(p.a.base) + (row0 + r) * p.a.stride_outer + ck);
                    v[uu].x = __uint_as_float(hq.x); v[uu].y = __uint_as_float(hq.y);
                } else
                    v[uu] = ld4(p.a.base + (row0 + r) * p.a.stride_outer + ck);
            }
        };
        auto commit_row = [&](float* An, int u) {
            if (u >= RW) return;                         // (compile time)
            const int r = rbl + u;
            float4 o = v[u & 7];
            if constexpr (AMODE == A_GATHER) {
                const float4 pv = pvs[u / KK - (8 * (u / 8)) / KK];
                o.x = fmaxf(o.x + pv.x, 0.f); o.y = fmaxf(o.y + pv.y, 0.f);
                o.z = fmaxf(o.z + pv.z, 0.f); o.w = fmaxf(o.w + pv.w, 0.f);
            }
            if constexpr (LAZY) {
                // dz3 of slot u of the wave's point (gpe_dz3_kernel's arithmetic): the message that won the aggregation carries s * g
                const x6_f32x2 a01 = __builtin_convertvector(__builtin_bit_cast(x6_f16x2, __float_as_uint(o.x)), x6_f32x2);
                const x6_f32x2 a23 = __builtin_convertvector(__builtin_bit_cast(x6_f16x2, __float_as_uint(o.y)), x6_f32x2);
                const float av[4] = {a01[0], a01[1], a23[0], a23[1]};
                if (u == 0) {                            // (u is a compile-time constant at every call site) once per tile
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
            if (r >= s_rv) o = zero4;                    // rows past the end of a partial last tile
            unsigned q0[2], q1[2];
            const float sc = LAZY ? 1.f : sA;            // LAZY: dz3 was formed from pre-scaled coefficients
            SP::split2(o.x * sc, o.y * sc, q0);
            SP::split2(o.z * sc, o.w * sc, q1);
            char* row = reinterpret_cast<char*>(An) + r * PPITCH + 2 * ck;
            *reinterpret_cast<uint2*>(row) = make_uint2(q0[0], q1[0]);
            *reinterpret_cast<uint2*>(row + PLANE) = make_uint2(q0[1], q1[1]);
        };
        // the finisher of this wave's point has folded the partials that lie in the rows about to be re-staged (flag = number of
        // folded tiles).  Bounded spin; a flag that never comes (it cannot, short of a scheduling fault: the finisher sets it right
        // after the tile's second barrier, which this wave has passed too) TRAPS — committing over partials that were not folded
        // would be silent corruption
        auto wait_folded = [&](int want) {
            int spins = 0;
            while (__hip_atomic_load(&flag_sh[pt], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) __builtin_trap();
            }
        };

        // ---- finisher: VMEM issue + epilogue of one row -------------------------------------------------------------------------
        // stored activation of row u of `tile_e` (whose neighbour rows are jv) -> ring entry u % RING
        auto issue_act_load = [&](int u, int tile_e, int jv) {
            if constexpr (ROLE == 1 && !FWD) {
                const long row0 = (long)tile_e * TR;
                const int rv = (int)((p.M - row0 < TR) ? (p.M - row0) : TR);
                if (u >= RW) u = RW - 1;                 // (KK = 5: no 16th row)
                int r = rbl + u;
                r = (r < rv - 1) ? r : rv - 1;           // clamp: unconditional loads
                if constexpr (EMODE == E_BWD_INPLACE) act[u % RING] = ld4(p.out + (row0 + r) * p.ldo + cn);
                else {
                    const int jj = __builtin_amdgcn_ds_bpermute(u << 2, jv);
                    act[u % RING] = ld4(p.pq + (long)jj * p.ldpq + p.H + cn);
                }
            }
        };
        auto begin_epi = [&](int tile_e, bool on) {
            e_row0 = (long)tile_e * TR;
            e_pt = (long)tile_e * (4 * NPW) + pt * NPW;
            e_rv = (int)((p.M - e_row0 < TR) ? (p.M - e_row0) : TR);
            epi_on = on && rbl < e_rv;                   // KK = 16: all rows of the wave's point are valid or none is
            constexpr int ES = OUTH ? 2 : 4;             // bytes per stored element
            orsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.out) + e_row0 * p.ldo * ES, 0, RG_BM * p.ldo * ES, 0x00020000);
#pragma unroll
            for (int t = 0; t < 4; ++t) { s32[t] = 0.f; q32[t] = 0.f; vmx[t] = -INFINITY; vmn[t] = INFINITY; imx[t] = 0; imn[t] = 0; }
            dp = zero4;
            amax_t = 0.f;
            if constexpr (GATHER_ACT) {
                const long ptl = (long)tile_e * (4 * NPW) + (e_rv - 1) / KK;
                if constexpr (NPW > NPE) e_ptl = ptl;
#pragma unroll
                for (int q = 0; q < NPE; ++q) pve[q] = ld4(p.pq + (prow(e_pt + q < ptl ? e_pt + q : ptl) + vz) * p.ldpq + cn);
            }
        };
        auto epi_row = [&](int u, float4 z) {
            if (u >= RW) return;                         // (compile time)
            if constexpr (GATHER_ACT && NPW > NPE) {
                // first row of point q >= 1: its P row was requested one point earlier; request the next point's into the entry the
                // finished point has freed (unconditional, clamped — and pinned where it is first needed: see w8_pin4)
                if (u % KK == 0 && u / KK >= 1) {
                    const int q = u / KK;
                    w8_pin4(pve[q % NPE]);
                    if (q + 1 < NPW) {
                        const long pg = e_pt + q + 1;
                        pve[(q + 1) % NPE] = ld4(p.pq + (prow(pg < e_ptl ? pg : e_ptl) + vz) * p.ldpq + cn);
                    }
                }
            }
            w8_pin4(z);
            if constexpr (!FWD) w8_pin4(act[u % RING]);
            if (!epi_on) return;
            const int r = rbl + u;                       // row inside the tile: wave-uniform -> the descriptor's scalar offset
            if constexpr (KK != 16) {
                if (r >= e_rv) return;                   // a ragged last tile ends between the points of a wave
            }
            if constexpr (FWD) {
                const float vv[4] = {fmaxf(__builtin_fmaf(z.x, invAW, bias4.x), 0.f), fmaxf(__builtin_fmaf(z.y, invAW, bias4.y), 0.f),
                                     fmaxf(__builtin_fmaf(z.z, invAW, bias4.z), 0.f), fmaxf(__builtin_fmaf(z.w, invAW, bias4.w), 0.f)};
                if constexpr (!TRACK) amax_t = fmaxf(fmaxf(amax_t, fmaxf(vv[0], vv[1])), fmaxf(vv[2], vv[3]));
                if constexpr (OUTH) {
                    // fp16 rows (RNE), clamped to the largest finite fp16 (gpe_edgegemm_split_kernel.h)
                    const x6_f32x2 v01 = {fminf(vv[0], 65504.f), fminf(vv[1], 65504.f)}, v23 = {fminf(vv[2], 65504.f), fminf(vv[3], 65504.f)};
                    const w8_u32x2 hq = {__builtin_bit_cast(unsigned, __builtin_convertvector(v01, x6_f16x2)),
                                         __builtin_bit_cast(unsigned, __builtin_convertvector(v23, x6_f16x2))};
                    if constexpr ((W8_BUFSTORE & 2) != 0)
                        __builtin_amdgcn_raw_buffer_store_b64(hq, orsrc, cn * 2, r * p.ldo * 2, 0);
                    else
                        *reinterpret_cast<uint2*>(reinterpret_cast<_Float16*>(p.out) + (e_row0 + r) * p.ldo + cn) = make_uint2(hq[0], hq[1]);
                } else {
                    const x6_u32x4 oq = {__float_as_uint(vv[0]), __float_as_uint(vv[1]), __float_as_uint(vv[2]), __float_as_uint(vv[3])};
                    // gather variant: the activation rows stream out past L2 (sc1) so that they do not evict the cloud's Q table
                    if constexpr ((W8_BUFSTORE & (AMODE == A_GATHER ? 1 : 2)) != 0)
                        __builtin_amdgcn_raw_buffer_store_b128(oq, orsrc, cn * 4, r * p.ldo * 4, AMODE == A_GATHER ? 16 : 0);
                    else
                        st4(p.out + (e_row0 + r) * p.ldo + cn, make_float4(vv[0], vv[1], vv[2], vv[3]));
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    s32[t] += vv[t];
                    q32[t] = __builtin_fmaf(vv[t], vv[t], q32[t]);
                    if constexpr (TRACK) {
                        if (vv[t] > vmx[t]) { vmx[t] = vv[t]; imx[t] = u % KK; }
                        if (vv[t] < vmn[t]) { vmn[t] = vv[t]; imn[t] = u % KK; }
                    }
                }
            } else {
                float4 av = act[u % RING];
                if constexpr (GATHER_ACT) {
                    const float4 pv = pve[(u / KK) % NPE];
                    av.x = fmaxf(av.x + pv.x, 0.f); av.y = fmaxf(av.y + pv.y, 0.f);
                    av.z = fmaxf(av.z + pv.z, 0.f); av.w = fmaxf(av.w + pv.w, 0.f);
                }
                float4 o;
                o.x = (av.x > 0.f) ? __builtin_fmaf(z.x, cs4.x, __builtin_fmaf(k24.x, av.x, c14.x)) : 0.f;
                o.y = (av.y > 0.f) ? __builtin_fmaf(z.y, cs4.y, __builtin_fmaf(k24.y, av.y, c14.y)) : 0.f;
                o.z = (av.z > 0.f) ? __builtin_fmaf(z.z, cs4.z, __builtin_fmaf(k24.z, av.z, c14.z)) : 0.f;
                o.w = (av.w > 0.f) ? __builtin_fmaf(z.w, cs4.w, __builtin_fmaf(k24.w, av.w, c14.w)) : 0.f;
                const x6_u32x4 oq = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
                // (gathered backward: sc1 like the gather forward — its dz rows are read by another kernel much later, and kept in L2 they
                // evict the cloud's Q table this kernel's epilogue keeps re-reading: round 3 measured 1122 -> 974 MB fetched per launch
                // for that change but lost 6 % to the `asm volatile` store it needed then)
                if constexpr ((W8_BUFSTORE & (EMODE == E_BWD_INPLACE ? 4 : 8)) != 0)
                    __builtin_amdgcn_raw_buffer_store_b128(oq, orsrc, cn * 4, r * p.ldo * 4, (GATHER_ACT && W8_B2_SC1) ? 16 : 0);
                else
                    st4(p.out + (e_row0 + r) * p.ldo + cn, o);
                if constexpr (EMODE == E_BWD_INPLACE)
                    amax_t = fmaxf(fmaxf(amax_t, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
                dp.x += o.x; dp.y += o.y; dp.z += o.z; dp.w += o.w;
            }
            if (u % KK == KK - 1) {                      // a point is complete (compile time)
                if constexpr (TRACK) amax_t = fmaxf(amax_t, fmaxf(fmaxf(vmx[0], vmx[1]), fmaxf(vmx[2], vmx[3])));
                {
                    const long gpt = e_pt + u / KK;
                    if constexpr (TRACK) {
                        const long o = gpt * p.oldagg + cn;
                        st4(p.mx + o, make_float4(vmx[0], vmx[1], vmx[2], vmx[3]));
                        st4(p.mn + o, make_float4(vmn[0], vmn[1], vmn[2], vmn[3]));
                        *reinterpret_cast<uchar4*>(p.oamx + o) = make_uchar4(imx[0], imx[1], imx[2], imx[3]);
                        *reinterpret_cast<uchar4*>(p.oamn + o) = make_uchar4(imn[0], imn[1], imn[2], imn[3]);
                    }
                    if constexpr (GATHER_ACT) st4(p.dP + gpt * p.lddp + cn, dp);
                }
                // (per point, not per wave: a ragged last tile can end between the points of a KK = 5 wave)
                amax_run = fmaxf(amax_run, amax_t);
                if constexpr (FWD) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) { stS[t] += (double)s32[t]; stQ[t] += (double)q32[t]; }
                }
                if constexpr (NPW > 1) {                 // the next point of this wave starts clean
#pragma unroll
                    for (int t = 0; t < 4; ++t) { vmx[t] = -INFINITY; vmn[t] = INFINITY; imx[t] = 0; imn[t] = 0; s32[t] = 0.f; q32[t] = 0.f; }
                    dp = zero4;
                }
            }
        };

        // ---- prologue: stage tile 0 ----------------------------------------------------------------------------------------
        int tile = gpe_seq_tile(sq);
        gpe_seq_advance(sq);
        int next = gpe_seq_tile(sq);
        gpe_seq_advance(sq);
        int next2 = gpe_seq_tile(sq);
        if constexpr (ROLE == 0) {
            if (tile < p.num_tiles) {
                if constexpr (AMODE == A_GATHER) jgv_a = load_jgv(tile);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    issue_stage_loads(tile, h);
#pragma unroll
                    for (int uu = 0; uu < 8; ++uu) commit_row(Abuf0, 8 * h + uu);
                }
                // neighbour rows of the tile staged during the first iteration
                if constexpr (AMODE == A_GATHER) jgv_b = load_jgv(next < p.num_tiles ? next : tile);
            }
        }
        __syncthreads();

        int buf = 0, prev = -1, iter = 0;
        for (; tile < p.num_tiles; tile = next, next = next2, gpe_seq_advance(sq), next2 = gpe_seq_tile(sq)) {
            W8_REFRESH_SCALARS()
            const float* As = buf ? Abuf1 : Abuf0;
            float* An = buf ? Abuf0 : Abuf1;
            const bool do_epi = prev >= 0;
            const bool do_stage = next < p.num_tiles;
            const int tile_s = do_stage ? next : tile;   // clamped: results unused when !do_stage
            const int tile_e = do_epi ? prev : tile;     // clamped: results unused when !do_epi

            f32x4 acc[4], accS[4], accT[4];              // own tile; split tile; tile 12
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; accS[mt] = acc[mt]; accT[mt] = acc[mt];
            }
            // A fragment of slot (i, mt): 8 consecutive k of row 16 mt + j in slab (rot + i) mod KS, both planes as they lie in LDS.
            // In the 16-wide tail slab the lane groups g >= 2 lie past the row: they read a valid chunk against zero weights.
            X6Frag<SP> nf;
            auto read_frag = [&](int i, int mt) {
                const int slr = (rot + i >= KS) ? rot + i - KS : rot + i;
                const int ge = (KTAIL && slr == KS - 1) ? g_tail : g;
                const char* src = reinterpret_cast<const char*>(As) + (16 * mt + j) * PPITCH + 16 * (4 * slr + ge);
                nf.pl[0] = *reinterpret_cast<const x6_u32x4*>(src);
                nf.pl[1] = *reinterpret_cast<const x6_u32x4*>(src + PLANE);
            };
            float4 zq = zero4;
            read_frag(0, 0);

#pragma unroll
            for (int i = 0; i < KS; ++i) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int q = 4 * i + mt;
                    const X6Frag<SP> af = nf;
                    if (q + 1 < NSLOT) read_frag((q + 1) >> 2, (q + 1) & 3);
                    // ---- this slot's slice of the memory pipeline ----
                    if constexpr (ROLE == 0) {
                        if (q == 0) {
                            if constexpr (AMODE == A_GATHER) {
                                jgv_a = jgv_b;
                                jgv_b = load_jgv(next2 < p.num_tiles ? next2 : tile);
                            }
                            issue_stage_loads(tile_s, 0);
                        }
                        if (q == C0) wait_folded(iter);
                        // (unconditional: without a next tile the current one is committed once more into the buffer nobody reads)
                        if (q >= C0 && q < C0 + 4) { commit_row(An, 2 * (q - C0)); commit_row(An, 2 * (q - C0) + 1); }
                        if (q == I1) issue_stage_loads(tile_s, 1);
                        if (q >= C1 && q < C1 + 4) { commit_row(An, 8 + 2 * (q - C1)); commit_row(An, 9 + 2 * (q - C1)); }
                    } else {
                        if (q == 0) {
                            begin_epi(tile_e, do_epi);
                            if constexpr (GATHER_ACT) { jgv_cur = jgv_e; jgv_e = load_jgv(tile); }
                        }
                        if (q >= EP0 && q - EP0 < RW) {
                            epi_row(q - EP0, zq);
                            // the ring entry this row has freed takes the next row that maps to it: row u + RING of the tile being
                            // finished — or, after the entry's last row, row u % RING of the tile being MULTIPLIED (finished in the next
                            // iteration: its first RING rows arrive a whole MFMA block ahead)
                            if (q - EP0 + RING < RW) issue_act_load(q - EP0 + RING, tile_e, jgv_cur);
                            else issue_act_load((q - EP0) % RING, tile, jgv_e);
                        }
                        if (q + 1 >= EP0 && q + 1 < NSLOT) {         // C row of the NEXT slot's epilogue (LDS prefetch)
                            const int rr = rbl + ((q + 1 - EP0 < RW) ? q + 1 - EP0 : RW - 1);
                            zq = ld4(&Cs[rr * LDC + cn]);
                        }
                    }
                    // ---- plane products: small terms first; the split tiles' independent accumulators interleaved.  (In the slots
                    // without split-tile work the three MFMAs form one dependent chain: with two waves per SIMD the partner's MFMAs
                    // fill the gaps, and separate hh / correction accumulators cost 16 registers this kernel does not have.) ----
                    acc[mt] = SP::mfma(af.pl[1], wP[0][i], acc[mt]);                         // l . H
                    if (i < SL2) accS[mt] = SP::mfma(af.pl[1], sP[0][i], accS[mt]);
                    if (PAIRS && i == 0) accT[mt] = SP::mfma(af.pl[1], tP[0], accT[mt]);
                    acc[mt] = SP::mfma(af.pl[0], wP[1][i], acc[mt]);                         // h . L
                    if (i < SL2) accS[mt] = SP::mfma(af.pl[0], sP[1][i], accS[mt]);
                    if (PAIRS && i == 0) accT[mt] = SP::mfma(af.pl[0], tP[1], accT[mt]);
                    acc[mt] = SP::mfma(af.pl[0], wP[0][i], acc[mt]);                         // h . H
                    if (i < SL2) accS[mt] = SP::mfma(af.pl[0], sP[0][i], accS[mt]);
                    if (PAIRS && i == 0) accT[mt] = SP::mfma(af.pl[0], tP[0], accT[mt]);
#if W8_SLOT_FENCE
                    // the slots are straight-line code: without a fence the scheduler treats the whole tile as one region and stretches
                    // every live range across it (49 - 75 spilled registers); inside a slot it still weaves the memory slice into the MFMAs
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
            }
            __syncthreads();                             // (1) finishers are done with C, every wave with As
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Cs[(16 * mt + 4 * g + r) * LDC + 16 * wave + j] = acc[mt][r];
            {
                // split tiles: the K-partial goes to C (first contributor) or to this wave's slot in the rows of the consumed A buffer
                float* Sc = const_cast<float*>(As);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * mt + 4 * g + r;
                        float* dst = sec_direct ? &Cs[row * LDC + 16 * sec_tile + j] : &Sc[scr(row, sec_slot) + j];
                        *dst = accS[mt][r];
                    }
                if constexpr (PAIRS) {
                    if (last_on) {
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = 16 * mt + 4 * g + r;
                                float* dst = last_direct ? &Cs[row * LDC + 16 * 12 + j] : &Sc[scr(row, last_slot) + j];
                                *dst = accT[mt][r];
                            }
                    }
                }
            }
            __syncthreads();                             // (2) C complete, partials complete, next A tile complete
            if constexpr (ROLE == 1) {
                // fold the partials of this point's 16 rows into C, in slot order (bit-reproducible), then release the rows
                // (KK = 5: a wave owns 15 rows — the 16th lane group repeats the 15th row's fold: same values, same addresses)
                const int ur = (lane >> 2) < RW ? (lane >> 2) : RW - 1, cq = (lane & 3) << 2;
                const int row = rbl + ur;
                if constexpr (PAIRS) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        float* cd = &Cs[row * LDC + 16 * (8 + b) + cq];
                        const float4 a0 = ld4(cd), a1 = ld4(&As[scr(row, b) + cq]);
                        st4(cd, make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w));
                    }
                    float* cd = &Cs[row * LDC + 16 * 12 + cq];
                    float4 s = ld4(cd);
#pragma unroll
                    for (int l = 0; l < NLS; ++l) {
                        const float4 a1 = ld4(&As[scr(row, 4 + l) + cq]);
                        s.x += a1.x; s.y += a1.y; s.z += a1.z; s.w += a1.w;
                    }
                    st4(cd, s);
                } else {
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        float* cd = &Cs[row * LDC + 16 * (8 + b) + cq];
                        float4 s = ld4(cd);
#pragma unroll
                        for (int l = 0; l < 3; ++l) {
                            const float4 a1 = ld4(&As[scr(row, 3 * b + l) + cq]);
                            s.x += a1.x; s.y += a1.y; s.z += a1.z; s.w += a1.w;
                        }
                        st4(cd, s);
                    }
                }
                // (LDS operations of a wave are performed in order: the flag store follows the partial reads)
                __hip_atomic_store(&flag_sh[pt], iter + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            prev = tile;
            buf ^= 1;
            ++iter;
        }
        // ---- tail: epilogue of the last tile ------------------------------------------------------------------------------
        if constexpr (ROLE == 1) {
            {
                W8_REFRESH_SCALARS()
                begin_epi(prev >= 0 ? prev : 0, prev >= 0);
                jgv_cur = jgv_e;
                // the first RING rows were requested in the last iteration; the others as the ring frees
#pragma unroll
                for (int u = 0; u < RW; ++u) {
                    epi_row(u, ld4(&Cs[(rbl + u) * LDC + cn]));
                    if (u + RING < RW) issue_act_load(u + RING, prev >= 0 ? prev : 0, jgv_cur);
                }
            }
            if (p.amax_out && !GATHER_ACT) {
                // largest magnitude this wave wrote (non-negative floats order like their bit patterns; a NaN sorts above inf)
                float m = amax_run;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
                if (lane == 0) amax_sh[pt] = __float_as_uint(m);
            }
        }
        __syncthreads();                                 // every fold is done: the A buffers may be reused for the statistics
        if constexpr (ROLE == 1 && FWD) {
            if (p.stats_part) {
                double* red = reinterpret_cast<double*>(smem);          // [4 finishers][2][16*NT]
                if (n_on) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        red[(pt * 2 + 0) * (16 * NT) + c + t] = stS[t];
                        red[(pt * 2 + 1) * (16 * NT) + c + t] = stQ[t];
                    }
                }
            }
        }
        __syncthreads();
    };
    __syncthreads();                                     // A buffers zeroed, flags cleared
    // measurement aid (gpe_debug_set 1024 / 2048): static issue priority for the finishers / the stagers (MI355X_MICROARCH.md "two
    // waves per SIMD": the second-dispatched half of a 512-thread workgroup loses the VALU arbitration at equal priority)
    // (measured, profiles/r05_a_w8_schedules.md: finishers at priority 1 — B2 431 -> 417 us, F2 375 -> 410, F3 / B3 +-0; stagers at
    // priority 1 — nothing: only the gathered backward takes the finisher priority by default)
    if (((p.dbg & 1024) || EMODE == E_BWD_GATHER) && role == 1) __builtin_amdgcn_s_setprio(1);
    if ((p.dbg & 2048) && role == 0) __builtin_amdgcn_s_setprio(1);
    if (role == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});

    if (p.amax_out && EMODE != E_BWD_GATHER && tid == 0) {
        // one atomic per workgroup (same-address atomics issued by every wave at once serialise in the L2)
        const unsigned a = amax_sh[0] > amax_sh[1] ? amax_sh[0] : amax_sh[1], b = amax_sh[2] > amax_sh[3] ? amax_sh[2] : amax_sh[3];
        atomicMax(p.amax_out, a > b ? a : b);
    }
    if (FWD && p.stats_part && tid < p.N) {
        const double* red = reinterpret_cast<const double*>(smem);
        constexpr int NC = 16 * NT;
        const double ss = (red[0 * NC + tid] + red[2 * NC + tid]) + (red[4 * NC + tid] + red[6 * NC + tid]);
        const double qq = (red[1 * NC + tid] + red[3 * NC + tid]) + (red[5 * NC + tid] + red[7 * NC + tid]);
        for (int b = blockIdx.x; b < stats_nblk; b += gridDim.x) {
            double* dst = p.stats_part + (size_t)b * 2 * p.N;
            dst[tid] = (b == (int)blockIdx.x) ? ss : 0.0;
            dst[p.N + tid] = (b == (int)blockIdx.x) ? qq : 0.0;
        }
    }
#undef W8_REFRESH_SCALARS
}

template <int NT, int KCH, int AMODE, int EMODE, int AGGT, bool LAZY, int KK = 16>
static int w8_launch(const RgParams& p, int stats_nblk, hipStream_t s)
{
    constexpr int LDC = 16 * NT + 4;
    constexpr int AWORDS = (2 * RG_BM * 16 * x6_pchunks(KCH)) / 4;
    const size_t lds = (size_t)(2 * AWORDS + RG_BM * LDC) * sizeof(float);
    // 32 bytes of static __shared__ (amax_sh, flag_sh) sit beside the dynamic image
    GPE_ENSURE_MAX_LDS_N((gpe_edgegemm_w8_kernel<NT, KCH, AMODE, EMODE, AGGT, LAZY, KK>), 160 * 1024 - 64);
    int gx = gpe_num_cus();
    if (gx > p.num_tiles) gx = p.num_tiles;
    if (stats_nblk > 0 && gx > stats_nblk) gx = stats_nblk;
    hipLaunchKernelGGL((gpe_edgegemm_w8_kernel<NT, KCH, AMODE, EMODE, AGGT, LAZY, KK>), dim3(gx), dim3(512), lds, s, p, stats_nblk);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}
