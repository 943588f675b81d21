// Persistent execution of an LSTM stack whose row tiles outnumber the chip (the shipped PANEL decoder: 736 rows x 3 layers x 14
// steps x 250 units at BASELINE cfg 2) — ONE launch per direction, f16x3 arithmetic.  Reference: nn.LSTM under the decoders,
// /root/reference/nn/net_blocks.py:363-402.  Companion of gpe_rnn_persist.hip (stacks with one row tile per workgroup).
//
// Why a second kernel.  gpe_rnn_persist.hip splits the K extent of ONE 16-row tile over the four waves of a workgroup: the right
// shape for a latency chain (the 32-row pattern decoder), the wrong one for 10 - 12 row tiles per workgroup — every tile-step then
// pays its own poll -> load -> MFMA -> barrier -> store -> drain chain with nothing under it (measured in round 6: 688 us forward
// against 441 us of diagonal launches).  Here a WAVE owns row tiles: workgroup (layer l, row group rg, unit block nb) keeps its weight
// slices (64 gate columns of 16 units: W_hh_l and W_ih_l, 64 KB each as fp16 planes) in LDS for all T steps; wave w of it walks
// the tiles rg + RG (w + 4 q), q = 0, 1, .., of every step t with the whole K extent, no barrier and no LDS exchange:
//   * operands swapped in the MFMA — A = the weight fragment (gate columns become the D rows), B = the state rows — so lane (j, g)
//     ends up with all four gates of units 4 g .. 4 g + 3 of row j: the cell update needs no cross-lane traffic and h leaves as ONE
//     16-byte store per lane;
//   * the exchanged operand is published ALREADY SPLIT into its two fp16 terms (hsplit: [layer][slot][row][k-group][plane][8]):
//     a state row is read by the 16 unit blocks of its own layer and the 16 of the layer above, so the split is done once by the
//     producer instead of 32 times by the consumers, and a consumer's B fragment is two 16-byte loads straight into MFMA registers;
//   * the next item's flags are polled ONCE and, when they stand, its loads are issued before the current item's products:
//     the hand-off latency of one tile hides under the arithmetic of another (an item whose flags are not up yet is waited for
//     after the current one — never before it: a wave that owns a single tile would wait for itself).
// Inter-workgroup visibility: recipe R1 as in gpe_rnn_persist.hip — sc1 (write-through) stores of the payload, the storing wave
// drains vmcnt, one relaxed agent-scope counter increment per (workgroup, tile, step); consumers poll relaxed and read sc1.
// Counter (l, slot, rt) counts the unit blocks that published h_{l, slot - 1} of tile rt; slot 0 = the start state.
#include "gpe_common.h"
#include <math.h>

extern "C" int gpe_debug_get(void);

#define PM_MAXL 4
#define PM_MAXS 8                 // k-steps of 32: K (= units) <= 256
#define PM_SPIN_LIMIT (1u << 23)
#define PM_FS 32                  // words per arrival counter: one 128-byte line each (counters that share a line serialise their
                                  // agent-scope increments and polls: measured 7 us per item with packed counters)
#define PM_SA 4096.f
#define PM_INV_SA (1.f / 4096.f)

typedef unsigned pm_u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 pm_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pm_f16x2 __attribute__((ext_vector_type(2)));
typedef float pm_f32x2 __attribute__((ext_vector_type(2)));

struct PmFwdParams {
    int L, T, Bn, H, NB, RG, NRT, KP;
    const float* xproj0; long xp0_sb, xp0_st;
    const void* w0[PM_MAXL];                 // W_hh_l: fp16 plane pack of the gate-interleaved matrix (gpe_pack_multi kind 8)
    const void* w1[PM_MAXL];                 // W_ih_l, layers > 0
    const unsigned* s0[PM_MAXL]; const unsigned* s1[PM_MAXL];        // amax words of the plane packs
    const float* bias[PM_MAXL];              // b_ih + b_hh of layers > 0
    float* hs; long hs_sl, hs_sb, hs_st;
    float* cs; long cs_sl, cs_st;
    float* saved; long sv_sl, sv_st;
    unsigned* flags;                         // [L][T + 1][NRT] arrival counters
    char* hsplit;                            // [L][T + 1][16 NRT rows][KP / 8][2 planes][8 halves]
    unsigned long long* trace;               // measurement aid (gpe_debug_set 8192): [grid][4 waves][T * PM_MAXQ][8] stamps of lane 0, else NULL
};
#ifndef PM_NW
#define PM_NW 4                   // waves per workgroup (8 = two per SIMD was measured: no faster — what the two waves of a SIMD issue adds up)
#endif
#define PM_MAXQ (16 / PM_NW)      // row tiles a wave may own (PM_NW * PM_MAXQ = 16 tiles per workgroup)
// timing probes of a measurement build (wrong numbers): 1 no payload loads after the first item, 2 no plain stores, 4 no products,
// 8 a cell update without transcendental functions
#ifndef PM_PROBE
#define PM_PROBE 0
#endif
#define PM_STAMP(i)                                                                                                     \
    do {                                                                                                                \
        if (p.trace && lane == 0) p.trace[(((long)blockIdx.x * PM_NW + wave) * (T * PM_MAXQ) + it) * 8 + (i)] = wall_clock64(); \
    } while (0)

__device__ __forceinline__ float pm_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ unsigned pm_poll(const unsigned* flag)
{
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

__device__ __forceinline__ void pm_spin(const unsigned* flag, unsigned need)
{
    unsigned spins = 0;
    while (pm_poll(flag) < need) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > PM_SPIN_LIMIT) __builtin_trap();
    }
}

__device__ __forceinline__ void pm_split2(float a, float b, float s, unsigned& h, unsigned& l)
{
    const pm_f32x2 v = {a * s, b * s};
    const pm_f16x2 hh = __builtin_convertvector(v, pm_f16x2);
    const pm_f32x2 r = v - __builtin_convertvector(hh, pm_f32x2);
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, pm_f16x2));
}

// four consecutive floats v[0 .. nvalid) to dst (dst + 4 may run past the row: only valid elements are written); al = the
// alignment every such quad of the tensor has: 16 / 8 / 4 bytes
__device__ __forceinline__ void pm_store4(float* dst, const float (&v)[4], int nvalid, int al)
{
    if (nvalid >= 4 && al == 16) { *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]); return; }
    if (al >= 8) {
        if (nvalid >= 2) *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
        else if (nvalid == 1) dst[0] = v[0];
        if (nvalid >= 4) *reinterpret_cast<float2*>(dst + 2) = make_float2(v[2], v[3]);
        else if (nvalid == 3) dst[2] = v[2];
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) if (r < nvalid) dst[r] = v[r];
}

// publish the state quad h4 (units 16 nb + 4 g .. + 3 of tile row j; invalid units already zero) as its two fp16 planes:
// lanes g and g ^ 1 (16 lanes apart) hold the two halves of k-group 2 nb + (g >> 1); the even one stores plane 0 (the leading
// terms of all eight units), the odd one plane 1 (the residuals) — one 16-byte sc1 store per lane.
// hsplit is TILE-MAJOR: [tile][plane][k-step][row 16][k-group 4][8 halves] — the B fragment of one k-step and plane is 1 KB of
// consecutive bytes, so a consumer's 16-byte-per-lane load touches sixteen full 64-byte segments instead of 32 half-used ones
// (the vector memory pipe of a CU, shared by its four waves, is what bounds an item: measured)
__device__ __forceinline__ void pm_publish_split(__amdgpu_buffer_rsrc_t rs, int rt, int j, int NS, int nb, int g, const float (&h4)[4])
{
    unsigned hi01, hi23, lo01, lo23;
    pm_split2(h4[0], h4[1], PM_SA, hi01, lo01);
    pm_split2(h4[2], h4[3], PM_SA, hi23, lo23);
    const bool odd = g & 1;
    const unsigned s0 = odd ? hi01 : lo01, s1 = odd ? hi23 : lo23;
    const unsigned r0 = (unsigned)__shfl_xor((int)s0, 16), r1 = (unsigned)__shfl_xor((int)s1, 16);
    pm_u32x4 piece;
    piece[0] = odd ? r0 : hi01; piece[1] = odd ? r1 : hi23;
    piece[2] = odd ? lo01 : r0; piece[3] = odd ? lo23 : r1;
    const int kgroup = 2 * nb + (g >> 1);
    const int off = rt * (NS * 2048) + ((((odd ? NS : 0) + (kgroup >> 2)) * 16 + j) * 4 + (kgroup & 3)) * 16;
    __builtin_amdgcn_raw_buffer_store_b128(piece, rs, off, 0, 16);
}

// 16-byte payload load, sc1: served by L2 / the fabric, never by this CU's L1 (the producer stored sc1: no acquire fence needed).
// Plain loads were measured too (every hsplit address is written once per launch, so a cached copy cannot be stale): no faster.
__device__ __forceinline__ pm_u32x4 pm_ld(__amdgpu_buffer_rsrc_t rs, int off) { return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16); }

struct PmFwdItem {
    pm_u32x4 hh[PM_MAXS], hl[PM_MAXS];       // h_{l,t-1}: leading / residual plane, k-step s
    pm_u32x4 xh[PM_MAXS], xl[PM_MAXS];       // h_{l-1,t} (layers > 0)
    float e[4][4];                           // layer 0: the x-projection addend (gate, unit quad)
};

// products of k-steps [s_lo, s_hi) of one K segment: A = weight fragments from the LDS slice ([plane][KP / 8][64 columns][8 halves]),
// B = the loaded planes.  The eight fragments of a k-step are read one k-step ahead of their twelve MFMAs (hipcc left alone reads
// two fragments, waits, multiplies: the LDS latency of every pair was exposed), and consecutive MFMAs go to different accumulators.
__device__ __forceinline__ void pm_mma(const pm_u32x4 (&bh)[PM_MAXS], const pm_u32x4 (&bl)[PM_MAXS], const char* W, int KP, int s_lo,
                                       int s_hi, int j, int g, f32x4 (&acc)[4])
{
    const int plane_b = KP * 128;                                   // (KP / 8) groups x 64 columns x 16 bytes
    const char* w0 = W + (g * 64 + j) * 16;
    pm_u32x4 wh[2][4], wl[2][4];
    auto fetch = [&](int s, int buf) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            wh[buf][n] = *reinterpret_cast<const pm_u32x4*>(w0 + s * 4096 + 256 * n);
            wl[buf][n] = *reinterpret_cast<const pm_u32x4*>(w0 + s * 4096 + 256 * n + plane_b);
        }
    };
    if (s_lo < s_hi) fetch(s_lo, 0);
#pragma unroll
    for (int s = 0; s < PM_MAXS; ++s) {
        if (s >= s_lo && s < s_hi) {
            const int buf = (s - s_lo) & 1;
            if (s + 1 < s_hi) fetch(s + 1, buf ^ 1);
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pm_f16x8, wl[buf][n]), __builtin_bit_cast(pm_f16x8, bh[s]), acc[n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pm_f16x8, wh[buf][n]), __builtin_bit_cast(pm_f16x8, bl[s]), acc[n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pm_f16x8, wh[buf][n]), __builtin_bit_cast(pm_f16x8, bh[s]), acc[n], 0, 0, 0);
        }
    }
}

// copy this workgroup's 64 columns of every 16-byte-piece group of a plane pack into LDS
__device__ __forceinline__ void pm_fill(char* dst, const void* src, int ngroups, int Npad, int c0)
{
    const pm_u32x4* s = reinterpret_cast<const pm_u32x4*>(src);
    pm_u32x4* d = reinterpret_cast<pm_u32x4*>(dst);
    const int total = ngroups * 64;
    for (int e = threadIdx.x; e < total; e += 64 * PM_NW) {
        const int grp = e >> 6, c = e & 63;
        d[e] = s[(long)grp * Npad + c0 + c];
    }
}

template <bool L0>
__device__ __forceinline__ void pm_fwd_body(const PmFwdParams& p, const int l, const int rg, const int nb, const char* W0, const char* W1,
                                            char* keep)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int H = p.H, T = p.T, Bn = p.Bn, KP = p.KP, KG = KP >> 3, NS = KP >> 5, NRT = p.NRT;
    const int ntile = (NRT - rg + p.RG - 1) / p.RG;              // row tiles of this workgroup
    const int nq = (ntile > wave) ? (ntile - wave + PM_NW - 1) / PM_NW : 0;  // ... of this wave
    if (nq <= 0) return;
    const int unit0 = 16 * nb + 4 * g;
    const int nvalid = (H - unit0 > 4) ? 4 : (H - unit0 > 0 ? H - unit0 : 0);
    int uc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) uc[r] = (unit0 + r < H) ? unit0 + r : H - 1;
    // ONE accumulator for both segments: the W_ih products (scale 1 / inv1) are carried into W_hh's scale by an exact power of two
    float inv0, ratio = 1.f;
    {
        float sw0, sw1, inv1;
        gpe_h3_scale_of(p.s0[l][0], sw0, inv0);
        if (!L0) { gpe_h3_scale_of(p.s1[l][0], sw1, inv1); ratio = inv1 * sw0; }
        inv0 *= PM_INV_SA;
    }
    const long slot_bytes = 16L * NRT * KP * 4;
    const int al_sv = (H & 3) == 0 ? 16 : ((H & 1) == 0 ? 8 : 4);          // alignment of a unit quad inside [..][4H] / [..][H] rows
    int al_g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) al_g[q] = ((q * H) & 3) == 0 ? 16 : (((q * H) & 1) == 0 ? 8 : 4);      // rows are 4 H floats: 16-byte aligned
    const bool hs_quad = unit0 + 4 <= (int)p.hs_st;                         // the quad lies inside the padded state row
    unsigned* const flags_l = p.flags + (long)l * (T + 1) * NRT * PM_FS;
    const unsigned* const flags_dn = p.flags + (long)(l - 1) * (T + 1) * NRT * PM_FS;
    const unsigned need = (unsigned)p.NB;
    // per-lane state that outlives an item, in LDS (lane-contiguous 16-byte pieces: conflict-free, and no vector-memory instruction):
    //   c of the wave's tiles [wave][q][lane]; layer 0 with the same input row at every step: its x-projection addend [wave][q][gate][lane]
    float4* const c_keep = reinterpret_cast<float4*>(keep) + (wave * PM_MAXQ) * 64 + lane;
    // layer 0: [wave][q][gate][lane]; layers above: the bias row, the same for every tile and step: [gate][g] (sixteen lanes share a piece)
    float4* const e_keep = reinterpret_cast<float4*>(keep + 16 * 1024) + (L0 ? wave * PM_MAXQ * 256 + lane : g);
    const bool e_const = !L0 || p.xp0_st == 0;
    if (!L0) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const float* b = p.bias[l] + qq * H;
            e_keep[qq * 4] = make_float4(b[uc[0]], b[uc[1]], b[uc[2]], b[uc[3]]);       // (every wave writes the same values)
        }
    }

    // the storing wave drains its write-through stores, one lane counts the workgroup in
    auto publish = [&](int slot, int rt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(flags_l + ((long)slot * NRT + rt) * PM_FS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    // ---- slot 0: the start state of this workgroup's units, split and published like every later state ----
    for (int q = 0; q < nq; ++q) {
        const int rt = rg + p.RG * (wave + PM_NW * q);
        const int row = 16 * rt + j, rowc = row < Bn ? row : Bn - 1;
        const float* h0 = p.hs + l * p.hs_sl + (long)rowc * p.hs_sb;
        float h4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) h4[r] = (r < nvalid) ? h0[uc[r]] : 0.f;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.hsplit + ((long)l * (T + 1)) * slot_bytes, 0, (unsigned)slot_bytes, 0x00020000);
        pm_publish_split(rs, rt, j, NS, nb, g, h4);
        publish(0, rt);
        const float* c0 = p.cs + l * p.cs_sl + (long)rowc * H;
        c_keep[q * 64] = make_float4(c0[uc[0]], c0[uc[1]], c0[uc[2]], c0[uc[3]]);
        if (L0 && e_const) {
            const float* xp = p.xproj0 + (long)rowc * p.xp0_sb;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) e_keep[(q * 4 + qq) * 64] = make_float4(xp[qq * H + uc[0]], xp[qq * H + uc[1]], xp[qq * H + uc[2]], xp[qq * H + uc[3]]);
        }
    }

    auto tile_of = [&](int it, int& t, int& rt) {
        t = it / nq;
        rt = rg + p.RG * (wave + PM_NW * (it - t * nq));
    };
    // item (t, rt) reads h_{l,t-1} = slot t of this layer and h_{l-1,t} = slot t + 1 of the layer below
    auto wait_ready = [&](int t, int rt) {
        pm_spin(flags_l + ((long)t * NRT + rt) * PM_FS, need);
        if (!L0) pm_spin(flags_dn + ((long)(t + 1) * NRT + rt) * PM_FS, need);
    };
    auto issue = [&](PmFwdItem& I, int t, int rt) {
        asm volatile("" ::: "memory");                              // the payload loads stay below the polls
        const int toff = rt * (NS * 2048) + (j * 4 + g) * 16;      // this lane's piece of k-step 0, plane 0 of the tile
        if ((PM_PROBE & 1) && t > 0) return;
        {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.hsplit + ((long)l * (T + 1) + t) * slot_bytes, 0, (unsigned)slot_bytes, 0x00020000);
#pragma unroll
            for (int s = 0; s < PM_MAXS; ++s) {
                const int sc = s < NS ? s : NS - 1;
                I.hh[s] = pm_ld(rs, toff + sc * 1024);
                I.hl[s] = pm_ld(rs, toff + (NS + sc) * 1024);
            }
        }
        const int row = 16 * rt + j, rowc = row < Bn ? row : Bn - 1;
        if (!L0) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.hsplit + ((long)(l - 1) * (T + 1) + t + 1) * slot_bytes, 0, (unsigned)slot_bytes, 0x00020000);
#pragma unroll
            for (int s = 0; s < PM_MAXS; ++s) {
                const int sc = s < NS ? s : NS - 1;
                I.xh[s] = pm_ld(rs, toff + sc * 1024);
                I.xl[s] = pm_ld(rs, toff + (NS + sc) * 1024);
            }
        } else if (!e_const) {
            const float* xp = p.xproj0 + (long)rowc * p.xp0_sb + (long)t * p.xp0_st;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) I.e[q][r] = xp[q * H + uc[r]];
        }
    };

    const int nitems = T * nq;
    PmFwdItem I;
    int pend_slot = -1, pend_rt = 0;                 // a finished item whose stores are still draining (published under the next products)
    {
        int t, rt;
        tile_of(0, t, rt);
        wait_ready(t, rt);
        issue(I, t, rt);
    }
    for (int it = 0; it < nitems; ++it) {
        int t, rt, t2 = 0, rt2 = 0;
        tile_of(it, t, rt);
        const bool more = it + 1 < nitems;
        // the next item's flags: polled now, looked at after the products
        unsigned pv0 = need, pv1 = need;
        if (more) {
            tile_of(it + 1, t2, rt2);
            pv0 = __hip_atomic_load(flags_l + ((long)t2 * NRT + rt2) * PM_FS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!L0) pv1 = __hip_atomic_load(flags_dn + ((long)(t2 + 1) * NRT + rt2) * PM_FS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        PM_STAMP(0);
        const int row = 16 * rt + j;
        const bool rok = row < Bn;
        f32x4 accH[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) accH[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (PM_PROBE & 4) {
            PM_STAMP(1);
            if (pend_slot >= 0) { publish(pend_slot, pend_rt); pend_slot = -1; }
            PM_STAMP(2);
            accH[0][0] = __uint_as_float(I.hh[0][0]) + __uint_as_float(I.xl[7][3]);
        } else if (!L0) {
            pm_mma(I.xh, I.xl, W1, KP, 0, NS, j, g, accH);
            PM_STAMP(1);
            if (pend_slot >= 0) { publish(pend_slot, pend_rt); pend_slot = -1; }
            PM_STAMP(2);
#pragma unroll
            for (int n = 0; n < 4; ++n) accH[n] *= ratio;
            pm_mma(I.hh, I.hl, W0, KP, 0, NS, j, g, accH);
        } else {
            pm_mma(I.hh, I.hl, W0, KP, 0, NS >> 1, j, g, accH);
            PM_STAMP(1);
            if (pend_slot >= 0) { publish(pend_slot, pend_rt); pend_slot = -1; }
            PM_STAMP(2);
            pm_mma(I.hh, I.hl, W0, KP, NS >> 1, NS, j, g, accH);
        }
        const int qi = it - t * nq;                   // which of the wave's tiles
        float z[4][4], cp[4];
        {
            const float4 c4 = c_keep[qi * 64];
            cp[0] = c4.x; cp[1] = c4.y; cp[2] = c4.z; cp[3] = c4.w;
        }
        if (e_const) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 e4 = L0 ? e_keep[(qi * 4 + q) * 64] : e_keep[q * 4];
                I.e[q][0] = e4.x; I.e[q][1] = e4.y; I.e[q][2] = e4.z; I.e[q][3] = e4.w;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) z[q][r] = accH[q][r] * inv0 + I.e[q][r];
        PM_STAMP(3);
        const bool ok = more && __builtin_amdgcn_readfirstlane(pv0) >= need && __builtin_amdgcn_readfirstlane(pv1) >= need;
        if (ok) issue(I, t2, rt2);                    // the next item's loads fly under this item's cell update
        PM_STAMP(4);
        float gi[4], gf[4], gg[4], go[4], cn[4], h4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (PM_PROBE & 8) {
                gi[r] = z[0][r] * 0.01f; gf[r] = z[1][r] * 0.01f; gg[r] = z[2][r] * 0.01f; go[r] = z[3][r] * 0.01f;
                cn[r] = gf[r] * cp[r] + gi[r] * gg[r];
                h4[r] = (r < nvalid) ? go[r] * cn[r] : 0.f;
            } else {
                gi[r] = pm_sigmoid(z[0][r]); gf[r] = pm_sigmoid(z[1][r]); gg[r] = tanhf(z[2][r]); go[r] = pm_sigmoid(z[3][r]);
                cn[r] = gf[r] * cp[r] + gi[r] * gg[r];
                h4[r] = (r < nvalid) ? go[r] * tanhf(cn[r]) : 0.f;
            }
        }
        c_keep[qi * 64] = make_float4(cn[0], cn[1], cn[2], cn[3]);
        {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.hsplit + ((long)l * (T + 1) + t + 1) * slot_bytes, 0, (unsigned)slot_bytes, 0x00020000);
            pm_publish_split(rs, rt, j, NS, nb, g, h4);
        }
        if (rok && !(PM_PROBE & 2)) {
            float* sv = p.saved + l * p.sv_sl + (long)t * p.sv_st + (long)row * 4 * H + unit0;
            // (a gate's quads are 16-byte aligned when gate * H is a multiple of 4: gates 0 and 2 at H = 250)
            pm_store4(sv, gi, nvalid, al_g[0]); pm_store4(sv + H, gf, nvalid, al_g[1]);
            pm_store4(sv + 2 * H, gg, nvalid, al_g[2]); pm_store4(sv + 3 * H, go, nvalid, al_g[3]);
            pm_store4(p.cs + l * p.cs_sl + (long)(t + 1) * p.cs_st + (long)row * H + unit0, cn, nvalid, al_sv);
            float* hd = p.hs + l * p.hs_sl + (long)row * p.hs_sb + (long)(t + 1) * p.hs_st + unit0;
            if (hs_quad) *reinterpret_cast<float4*>(hd) = make_float4(h4[0], h4[1], h4[2], h4[3]);
            else pm_store4(hd, h4, nvalid, 4);
        }
        PM_STAMP(5);
        if (p.trace && lane == 0) p.trace[(((long)blockIdx.x * PM_NW + wave) * (T * PM_MAXQ) + it) * 8 + 7] = (ok ? 1u : 0u) | ((unsigned)t << 8) | ((unsigned)rt << 16);
        if (ok) { pend_slot = t + 1; pend_rt = rt; }
        else {
            // the next item is not up yet (or there is none): publish first — it may be waiting for this very state
            publish(t + 1, rt);
            if (more) { wait_ready(t2, rt2); issue(I, t2, rt2); }
        }
        PM_STAMP(6);
    }
    if (pend_slot >= 0) publish(pend_slot, pend_rt);
}

__global__ __launch_bounds__(64 * PM_NW) void gpe_rnn_pm_fwd_kernel(PmFwdParams p)
{
    extern __shared__ __align__(16) char pm_smem[];
    int bid = blockIdx.x;
    const int nb = bid % p.NB; bid /= p.NB;
    const int rg = bid % p.RG;
    const int l = bid / p.RG;
    const int wbytes = p.KP * 256;                   // one slice: 2 planes x (KP / 8) groups x 64 columns x 16 bytes
    char* W0 = pm_smem;
    char* W1 = pm_smem + wbytes;
    pm_fill(W0, p.w0[l], p.KP >> 2, 64 * p.NB, 64 * nb);
    if (l > 0) pm_fill(W1, p.w1[l], p.KP >> 2, 64 * p.NB, 64 * nb);
    __syncthreads();
    // behind the slices: what the waves keep per tile between steps (pm_fwd_body)
    char* keep = pm_smem + (p.L > 1 ? 2 : 1) * wbytes;
    if (l == 0) pm_fwd_body<true>(p, l, rg, nb, W0, W1, p.L > 1 ? pm_smem + wbytes : keep);
    else pm_fwd_body<false>(p, l, rg, nb, W0, W1, keep);
}

// =====================================================================================================================
// host side
// =====================================================================================================================
struct PmPlan { int NB, NRT, RG, KP, grid; };

// gpe_debug_set bit 131072 (measurement aid): never these kernels (the diagonal launches run instead)
static bool pm_plan(int gates, int L, int T, int Bn, int H, PmPlan& pl)
{
    if (gates != 4 || L < 1 || L > PM_MAXL || T < 1 || Bn < 1 || H < 1 || H > 256 || (gpe_debug_get() & (1024 | 131072))) return false;
    pl.NB = gpe_cdiv(H, 16);
    pl.NRT = gpe_cdiv(Bn, 16);
    pl.KP = gpe_round_up(H, 32);
    if (pl.KP != 16 * pl.NB) return false;            // every k-group of a published row is written by a unit block
    const int cus = gpe_num_cus();
    const int per = L * pl.NB;
    if (cus <= 0 || per > cus) return false;
    int rgmax = cus / per < pl.NRT ? cus / per : pl.NRT;
    // the critical wave walks ceil(ceil(NRT / RG) / 4) tiles per step: the fewest row groups that reach the minimum
    // (fewer readers of every state row, CUs left to other streams)
    static const int dbg_rg = gpe_dbg_env("GPE_PM_RG", 0);
    int best = rgmax;
    auto crit = [&](int rgv) { return gpe_cdiv(gpe_cdiv(pl.NRT, rgv), 4); };       // per SIMD (PM_NW / 4 waves each)
    for (int rgv = rgmax; rgv >= 1; --rgv) if (crit(rgv) <= crit(rgmax)) best = rgv;
    if (dbg_rg > 0 && dbg_rg <= rgmax) best = dbg_rg;
    pl.RG = best;
    pl.grid = per * pl.RG;
    return true;
}

static long pm_flag_bytes(int L, int T, const PmPlan& pl) { return (long)L * (T + 1) * pl.NRT * PM_FS * 4; }
static long pm_hsplit_bytes(int L, int T, const PmPlan& pl) { return (long)L * (T + 1) * 16 * pl.NRT * pl.KP * 4; }
static long pm_trace_bytes(int T, const PmPlan& pl) { return (gpe_debug_get() & 8192) ? (long)pl.grid * PM_NW * T * PM_MAXQ * 8 * 8 : 0; }

long gpe_rnn_pm_ws_bytes(int gates, int L, int T, int Bn, int H, int bwd)
{
    PmPlan pl;
    if (bwd || !pm_plan(gates, L, T, Bn, H, pl)) return 0;
    return pm_flag_bytes(L, T, pl) + pm_hsplit_bytes(L, T, pl) + pm_trace_bytes(T, pl);
}

// 1 = launched, 0 = not eligible (the caller runs the diagonal launches), < 0 = error
int gpe_rnn_pm_fwd(int L, int T, int Bn, int H, const float* xproj0, long xp0_sb, long xp0_st, const void* const* whh,
                   const void* const* wih, const void* const* bias, float* hs, long hs_sl, long hs_sb, long hs_st, float* cs,
                   long cs_sl, long cs_st, float* saved, long sv_sl, long sv_st, const void* const* whh_amax,
                   const void* const* wih_amax, void* ws, long ws_bytes, hipStream_t s)
{
    PmPlan pl;
    if (!pm_plan(4, L, T, Bn, H, pl)) return 0;
    const long nflag = pm_flag_bytes(L, T, pl), nsplit = pm_hsplit_bytes(L, T, pl), ntrace = pm_trace_bytes(T, pl);
    if (!ws || ws_bytes < nflag + nsplit + ntrace || (((uintptr_t)ws) & 15)) return 0;
    if (16L * pl.NRT * pl.KP * 4 >= (1L << 31) || (hs_sb & 3) || (hs_st & 3) || (hs_sl & 3) || (((uintptr_t)hs) & 15)) return 0;
    // LDS: the weight slices, behind them 16 KB of cell states; a layer-0 workgroup (one slice) also keeps 64 KB of addends there
    const size_t keep0 = (size_t)16 * 1024 * 5, keepn = (size_t)16 * 1024 + 256;
    size_t lds = (size_t)pl.KP * 256 + keep0;
    if (L > 1 && (size_t)2 * pl.KP * 256 + keepn > lds) lds = (size_t)2 * pl.KP * 256 + keepn;
    if (lds > 160 * 1024) return 0;
    if (gpe_cdiv(gpe_cdiv(pl.NRT, pl.RG), PM_NW) > PM_MAXQ) return 0;
    PmFwdParams p = {};
    p.L = L; p.T = T; p.Bn = Bn; p.H = H; p.NB = pl.NB; p.RG = pl.RG; p.NRT = pl.NRT; p.KP = pl.KP;
    p.xproj0 = xproj0; p.xp0_sb = xp0_sb; p.xp0_st = xp0_st;
    for (int l = 0; l < L; ++l) {
        if (!whh[l] || (((uintptr_t)whh[l]) & 15) || !whh_amax[l]) return 0;
        p.w0[l] = whh[l];
        p.s0[l] = (const unsigned*)whh_amax[l];
        if (l > 0) {
            if (!wih[l] || (((uintptr_t)wih[l]) & 15) || !bias[l] || !wih_amax[l]) return 0;
            p.w1[l] = wih[l];
            p.bias[l] = (const float*)bias[l];
            p.s1[l] = (const unsigned*)wih_amax[l];
        }
    }
    p.hs = hs; p.hs_sl = hs_sl; p.hs_sb = hs_sb; p.hs_st = hs_st;
    p.cs = cs; p.cs_sl = cs_sl; p.cs_st = cs_st;
    p.saved = saved; p.sv_sl = sv_sl; p.sv_st = sv_st;
    p.flags = (unsigned*)ws;
    p.hsplit = (char*)ws + nflag;
    p.trace = ntrace ? (unsigned long long*)((char*)ws + nflag + nsplit) : nullptr;
    if (hipMemsetAsync(ws, 0, (size_t)nflag, s) != hipSuccess) return GPE_ELAUNCH;
    if (ntrace && hipMemsetAsync(p.trace, 0, (size_t)ntrace, s) != hipSuccess) return GPE_ELAUNCH;
    GPE_ENSURE_MAX_LDS(gpe_rnn_pm_fwd_kernel);
    hipLaunchKernelGGL(gpe_rnn_pm_fwd_kernel, dim3(pl.grid), dim3(64 * PM_NW), lds, s, p);
    GPE_CHECK_LAUNCH();
    return 1;
}
