// Persistent execution of an LSTM stack on gfx950: ONE launch per direction instead of one (forward) or two (backward)
// launches per anti-diagonal (gpe_rnn_wave.hip).  Reference: nn.LSTM under the decoders, /root/reference/nn/net_blocks.py:363-402.
//
// Why: the shipped pattern decoder (32 rows, 2 layers, 23 steps, 250 units) is a chain of 24 dependent diagonals whose
// arithmetic is ~1 us each; as launches they cost 13 us forward and 26 us backward apiece — the time goes into re-reading the
// weight slices from L2 and into kernel boundaries, not into the matrix pipe.
//
// Structure.  Workgroup (layer l, row group rg, unit block nb) owns 16 units of layer l — the 64 gate columns i|f|g|o of those
// units — for the row tiles rt = rg, rg + RG, ... (16 rows each).  Its weight slices (W_hh_l and, above layer 0, W_ih_l:
// 64 KB each at 250 units) are copied into LDS ONCE and stay there for all T steps.  A step of a row tile is
//   wait for the producers' arrival counters -> read the 16 x H state rows straight from global memory into MFMA A fragments
//   (every element is read by exactly one wave: no LDS staging, no barrier) -> 4 waves = 4 K quarters of the gate products ->
//   partial C tiles meet in LDS -> thread (row, unit) applies the cell -> h is PUBLISHED.
// Cells are ordered by data flow, not by a grid barrier: counter (l, t, rt) counts the unit blocks that have published
// h_{l,t} of row tile rt; cell (l, t, rt) waits for (l, t-1, rt) and (l-1, t, rt).  Layer 0 depends on nothing but itself and
// runs ahead; the input-side product of the layers above is issued before the recurrent one, while the recurrent operand is
// still being published.
//
// Inter-workgroup visibility (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility", recipe R1
// of cdna_hip_programming.md Guideline 16): published words are written with sc1 (write-through) stores, every storing wave
// drains vmcnt before the workgroup's ONE relaxed agent-scope counter increment; consumers poll the counter with relaxed
// agent-scope loads and read the payload with sc1 buffer loads (L1 bypass; the producer stored sc1, so no acquire fence).
// Nothing depends on workgroup -> XCD placement or dispatch order.  All workgroups must be co-resident: the host sizes the grid
// to at most one workgroup per CU (the LDS image allows no second one); every spin is bounded and traps.  Counters are zeroed
// by a memset node in front of every launch.
//
// The backward kernel mirrors it: workgroup (l, rg, nb) owns dh columns of 16 units; its slices of W_hh_l^T and W_ih_{l+1}^T
// (K = 4H rows x 16 columns) stay in LDS; the exchanged operand is the pre-activation gradient row dG (4H wide, torch gate
// order, written straight into the caller's dgx rows, which the weight-gradient GEMMs read afterwards).
// Arithmetic: exact fp32 MFMA, or f16x3 (gpe_math_set(4)): weights from the caller's fp16 plane packs; the forward state rows
// enter scaled by 2^12 (|h| < 1, start states < 16: same contract as gpe_rnn_wave.hip); the backward dG fragments are
// normalised PER WAVE by the largest magnitude the wave just loaded (each wave owns its partial accumulator, so the scale is
// undone before the partials meet) — no amax words, no atomics.
#include "gpe_common.h"
#include <math.h>

extern "C" int gpe_debug_get(void);

#define PS_MAXL 4
#define PS_LDC 68                 // forward C tile pitch (64 gate columns + 4)
#define PS_LDB 20                 // backward C tile pitch (16 units + 4)
#define PS_SPIN_LIMIT (1u << 23)  // polls (>= 0.1 us each) before a stuck workgroup traps instead of hanging the queue
#define PS_SA 4096.f
#define PS_INV_SA (1.f / 4096.f)

typedef unsigned ps_u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 ps_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ps_f16x2 __attribute__((ext_vector_type(2)));
typedef float ps_f32x2 __attribute__((ext_vector_type(2)));

struct PsFwdParams {
    int L, T, Bn, H, NB, RG, NRT;
    const float* xproj0; long xp0_sb, xp0_st;
    const void* w0[PS_MAXL];                 // W_hh_l: gate-interleaved pack (fp32 [K/4][Npad][4]) or its fp16 plane pack
    const void* w1[PS_MAXL];                 // W_ih_l, layers > 0
    const unsigned* s0[PS_MAXL]; const unsigned* s1[PS_MAXL];        // amax words of the plane packs
    const float* bias[PS_MAXL];              // b_ih + b_hh of layers > 0
    float* hs; long hs_sl, hs_sb, hs_st;
    float* cs; long cs_sl, cs_st;
    float* saved; long sv_sl, sv_st;
    unsigned* flags;                         // [L][T][NRT] arrival counters
    unsigned long long* trace;               // measurement aid (gpe_debug_set 8192): [grid][T][8] wall-clock stamps of lane 0, else NULL
};

struct PsBwdParams {
    int L, T, Bn, H, NB, RG, NRT, KP;        // KP: padded K (= 4H) extent of the transposed packs
    const float* dtop; long dt_sb, dt_st;
    const float* d_hN; const float* d_cN;
    const void* w0[PS_MAXL];                 // W_hh_l^T: plain transposed pack or transposed plane pack
    const void* w1[PS_MAXL];                 // W_ih_{l+1}^T for layer l < L-1
    const unsigned* s0[PS_MAXL]; const unsigned* s1[PS_MAXL];
    const float* cs; long cs_sl, cs_st;
    const float* saved; long sv_sl, sv_st;
    float* dgx; long dg_sl, dg_sb, dg_st;
    float* carry;                            // [2][L][Bn][H]
    unsigned* flags;                         // [L][T][NRT]
    unsigned long long* trace;
};

// phase stamp of the step timeline (100 MHz wall clock): [workgroup][step][8], written by lane 0 of wave 0 when tracing is on
#define PS_STAMP(i)                                                                                         \
    do {                                                                                                    \
        if (p.trace && tid == 0) p.trace[((long)blockIdx.x * T + t) * 8 + (i)] = wall_clock64();            \
    } while (0)

__device__ __forceinline__ float ps_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// every wave polls for itself: one word, relaxed, agent scope (an sc1 load: served by L2 / the fabric, never by this CU's L1)
__device__ __forceinline__ void ps_wait(const unsigned* flag, unsigned need)
{
    unsigned spins = 0;
    for (;;) {
        const unsigned v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (v >= need) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > PS_SPIN_LIMIT) __builtin_trap();
    }
    asm volatile("" ::: "memory");           // payload loads stay below the poll
}

// publish: every storing wave has drained its sc1 stores; ONE lane counts the workgroup in
__device__ __forceinline__ void ps_arrive(unsigned* flag, unsigned long long* stamp = nullptr)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (stamp && threadIdx.x == 0) *stamp = wall_clock64();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void ps_split2(float a, float b, float s, unsigned& h, unsigned& l)
{
    const ps_f32x2 v = {a * s, b * s};
    const ps_f16x2 hh = __builtin_convertvector(v, ps_f16x2);
    const ps_f32x2 r = v - __builtin_convertvector(hh, ps_f32x2);
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, ps_f16x2));
}

// zero the elements of a loaded quad whose k index is past the operand's K extent (what follows a row in memory is another row)
__device__ __forceinline__ ps_u32x4 ps_mask4(ps_u32x4 v, int k0, int K)
{
    v[0] = (k0 < K) ? v[0] : 0u; v[1] = (k0 + 1 < K) ? v[1] : 0u;
    v[2] = (k0 + 2 < K) ? v[2] : 0u; v[3] = (k0 + 3 < K) ? v[3] : 0u;
    return v;
}

// One K segment of a 16-row tile, in two phases so that the loads of two segments can be in flight together:
//   ps_load: A = rows [arow][K] behind `rs` (sc1 loads, row pitch `pitch` floats) -> this wave's share of the K steps, in registers;
//   ps_mma : B = the workgroup's weight slice in LDS with NT 16-column tiles; multiplies the loaded steps into acc.
// sA: power of two applied to A before the fp16 split (H3); DYN: take it from the largest magnitude this wave loaded instead and
// return its inverse.
//   H3 slice layout  [plane][KP / 8][16 NT columns][8 halves]   step = 32 k, two quads per lane
//   fp32 slice layout [KP / 4][16 NT columns][4 floats]          step = 16 k, one quad per lane
template <bool H3, int MAXS>
struct PsA { ps_u32x4 v[H3 ? 2 * MAXS : MAXS]; };

template <bool H3, int MAXS>
__device__ __forceinline__ void ps_load(PsA<H3, MAXS>& A, __amdgpu_buffer_rsrc_t rs, int arow, long pitch, int K, int s_lo, int nsteps)
{
    const int g = (threadIdx.x & 63) >> 4;
    const int rowoff = (int)(arow * pitch) * 4;
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
        const int s = (s_lo + i < nsteps) ? s_lo + i : nsteps - 1;
        if constexpr (H3) {
            const int k0 = 32 * s + 8 * g;
            A.v[2 * i] = ps_mask4(__builtin_amdgcn_raw_buffer_load_b128(rs, rowoff + k0 * 4, 0, 16), k0, K);
            A.v[2 * i + 1] = ps_mask4(__builtin_amdgcn_raw_buffer_load_b128(rs, rowoff + k0 * 4 + 16, 0, 16), k0 + 4, K);
        } else {
            const int k0 = 16 * s + 4 * g;
            A.v[i] = ps_mask4(__builtin_amdgcn_raw_buffer_load_b128(rs, rowoff + k0 * 4, 0, 16), k0, K);
        }
    }
}

template <bool H3, int NT, int MAXS, bool DYN>
__device__ __forceinline__ float ps_mma(const PsA<H3, MAXS>& A, int KP, const char* W, int s_lo, int s_hi, float sA, f32x4 (&acc)[NT])
{
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
    float inv = 1.f;
    if constexpr (H3) {
        if constexpr (DYN) {
            unsigned m = 0u;
#pragma unroll
            for (int i = 0; i < 2 * MAXS; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const unsigned a = A.v[i][e] & 0x7fffffffu; m = m > a ? m : a; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o); m = m > t ? m : t; }
            gpe_h3_scale_of(m, sA, inv);
        }
        const int plane_b = KP * 32 * NT;                        // bytes per plane: (KP / 8) groups x 16 NT columns x 16
#pragma unroll
        for (int i = 0; i < MAXS; ++i) {
            const int s = s_lo + i;
            if (s < s_hi) {
                ps_u32x4 ah, al;
                const ps_u32x4 a0 = A.v[2 * i], a1 = A.v[2 * i + 1];
                { unsigned h, l; ps_split2(__uint_as_float(a0[0]), __uint_as_float(a0[1]), sA, h, l); ah[0] = h; al[0] = l; }
                { unsigned h, l; ps_split2(__uint_as_float(a0[2]), __uint_as_float(a0[3]), sA, h, l); ah[1] = h; al[1] = l; }
                { unsigned h, l; ps_split2(__uint_as_float(a1[0]), __uint_as_float(a1[1]), sA, h, l); ah[2] = h; al[2] = l; }
                { unsigned h, l; ps_split2(__uint_as_float(a1[2]), __uint_as_float(a1[3]), sA, h, l); ah[3] = h; al[3] = l; }
                const char* wb = W + ((4 * s + g) * 16 * NT + j) * 16;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const ps_u32x4 bh = *reinterpret_cast<const ps_u32x4*>(wb + 256 * n);
                    const ps_u32x4 bl = *reinterpret_cast<const ps_u32x4*>(wb + 256 * n + plane_b);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ps_f16x8, al), __builtin_bit_cast(ps_f16x8, bh), acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ps_f16x8, ah), __builtin_bit_cast(ps_f16x8, bl), acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ps_f16x8, ah), __builtin_bit_cast(ps_f16x8, bh), acc[n], 0, 0, 0);
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < MAXS; ++i) {
            const int s = s_lo + i;
            if (s < s_hi) {
                const char* wb = W + ((4 * s + g) * 16 * NT + j) * 16;
                float4 b4[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) b4[n] = *reinterpret_cast<const float4*>(wb + 256 * n);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const float bv = (t == 0) ? b4[n].x : (t == 1) ? b4[n].y : (t == 2) ? b4[n].z : b4[n].w;
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(A.v[i][t]), bv, acc[n], 0, 0, 0);
                    }
            }
        }
    }
    return inv;
}

// copy this workgroup's columns [c0, c0 + CW) of every 16-byte-piece group of a packed weight into LDS:
// piece (group, c) of the pack sits at (group * Npad + c0 + c) * 16 bytes
template <int CW>
__device__ __forceinline__ void ps_fill(char* dst, const void* src, int ngroups, int Npad, int c0)
{
    const ps_u32x4* s = reinterpret_cast<const ps_u32x4*>(src);
    ps_u32x4* d = reinterpret_cast<ps_u32x4*>(dst);
    const int total = ngroups * CW;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int grp = e / CW, c = e - grp * CW;
        d[e] = s[(long)grp * Npad + c0 + c];
    }
}

// =====================================================================================================================
// forward
// =====================================================================================================================
template <bool H3>
__global__ __launch_bounds__(256) void gpe_rnn_persist_fwd_kernel(PsFwdParams p)
{
    extern __shared__ __align__(16) char ps_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    int bid = blockIdx.x;
    const int nb = bid % p.NB; bid /= p.NB;
    const int rg = bid % p.RG;
    const int l = bid / p.RG;
    const int H = p.H, T = p.T, Bn = p.Bn;
    const int KP = H3 ? ((H + 31) & ~31) : ((H + 15) & ~15);
    const int wbytes = KP * 256;                     // one slice: 64 columns x KP x 4 bytes in either layout
    char* W0 = ps_smem;
    char* W1 = ps_smem + wbytes;
    float* Cs = reinterpret_cast<float*>(ps_smem + (p.L > 1 ? 2 : 1) * wbytes);       // [4 waves][16][PS_LDC]

    ps_fill<64>(W0, p.w0[l], KP >> 2, 64 * p.NB, 64 * nb);
    if (l > 0) ps_fill<64>(W1, p.w1[l], KP >> 2, 64 * p.NB, 64 * nb);
    float inv0 = 1.f, inv1 = 1.f;
    if constexpr (H3) {
        float sw;
        gpe_h3_scale_of(p.s0[l][0], sw, inv0); inv0 *= PS_INV_SA;
        if (l > 0) { gpe_h3_scale_of(p.s1[l][0], sw, inv1); inv1 *= PS_INV_SA; }
    }
    __syncthreads();

    const int er = tid >> 4, eu = tid & 15;
    const int unit = 16 * nb + eu;
    const bool uok = unit < H;
    const int unitc = uok ? unit : H - 1;
    float bi = 0.f, bf = 0.f, bg = 0.f, bo = 0.f;
    if (l > 0) { const float* b = p.bias[l]; bi = b[unitc]; bf = b[H + unitc]; bg = b[2 * H + unitc]; bo = b[3 * H + unitc]; }

    constexpr int MAXS = H3 ? 2 : 4;                 // H <= 256: 8 steps of 32 / 16 steps of 16 over four waves
    const int nsteps = H3 ? (KP >> 5) : (KP >> 4);
    const int spw = (nsteps + 3) >> 2;
    const int s_lo = wave * spw;
    const int s_hi = (s_lo + spw < nsteps) ? s_lo + spw : nsteps;
    const unsigned need = (unsigned)p.NB;
    const unsigned nrec = (unsigned)((((long)Bn - 1) * p.hs_sb + ((H + 3) & ~3)) * 4);

    for (int t = 0; t < T; ++t) {
        for (int rt = rg; rt < p.NRT; rt += p.RG) {
            const int row = 16 * rt + er;
            const bool rok = row < Bn;
            const int rowc = rok ? row : Bn - 1;
            // epilogue operands first: their latency hides under the waits and the products
            float e0 = bi, e1 = bf, e2 = bg, e3 = bo;
            if (l == 0) {
                const float* xp = p.xproj0 + (long)rowc * p.xp0_sb + (long)t * p.xp0_st;
                e0 = xp[unitc]; e1 = xp[H + unitc]; e2 = xp[2 * H + unitc]; e3 = xp[3 * H + unitc];
            }
            const float cprev = p.cs[l * p.cs_sl + (long)t * p.cs_st + (long)rowc * H + unitc];

            PS_STAMP(0);
            const int arow = (16 * rt + j < Bn) ? 16 * rt + j : Bn - 1;
            f32x4 accH[4], accX[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) { accH[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; accX[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            // both flags first, then the loads of both segments back to back (one latency instead of two), then the products
            PsA<H3, MAXS> A1, A0;
            if (l > 0) ps_wait(p.flags + ((long)(l - 1) * T + t) * p.NRT + rt, need);   // h_{l-1,t}: the layer below runs ahead
            PS_STAMP(1);
            if (t > 0) ps_wait(p.flags + ((long)l * T + t - 1) * p.NRT + rt, need);     // h_{l,t-1}
            PS_STAMP(3);
            if (l > 0)                               // h_{l-1,t} (slot t+1) x W_ih_l
                ps_load<H3, MAXS>(A1, __builtin_amdgcn_make_buffer_rsrc(p.hs + (l - 1) * p.hs_sl + (long)(t + 1) * p.hs_st, 0, nrec, 0x00020000),
                                  arow, p.hs_sb, H, s_lo, nsteps);
            ps_load<H3, MAXS>(A0, __builtin_amdgcn_make_buffer_rsrc(p.hs + l * p.hs_sl + (long)t * p.hs_st, 0, nrec, 0x00020000),
                              arow, p.hs_sb, H, s_lo, nsteps);                           // h_{l,t-1} (slot t) x W_hh_l
            if (l > 0) ps_mma<H3, 4, MAXS, false>(A1, KP, W1, s_lo, s_hi, PS_SA, accX);
            PS_STAMP(2);
            ps_mma<H3, 4, MAXS, false>(A0, KP, W0, s_lo, s_hi, PS_SA, accH);
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Cs[(wave * 16 + 4 * g + r) * PS_LDC + 16 * n + j] = H3 ? accH[n][r] * inv0 + accX[n][r] * inv1 : accH[n][r] + accX[n][r];
            PS_STAMP(4);
            __syncthreads();
            float z[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                z[q] = (Cs[er * PS_LDC + 16 * q + eu] + Cs[(16 + er) * PS_LDC + 16 * q + eu]) +
                       (Cs[(32 + er) * PS_LDC + 16 * q + eu] + Cs[(48 + er) * PS_LDC + 16 * q + eu]);
            if (rok && uok) {
                const float ig = ps_sigmoid(z[0] + e0);
                const float fg = ps_sigmoid(z[1] + e1);
                const float gg = tanhf(z[2] + e2);
                const float og = ps_sigmoid(z[3] + e3);
                const float cn = fg * cprev + ig * gg;
                float* go = p.saved + l * p.sv_sl + (long)t * p.sv_st + (long)row * 4 * H;
                go[unit] = ig; go[H + unit] = fg; go[2 * H + unit] = gg; go[3 * H + unit] = og;
                p.cs[l * p.cs_sl + (long)(t + 1) * p.cs_st + (long)row * H + unit] = cn;
                __hip_atomic_store(p.hs + l * p.hs_sl + (long)row * p.hs_sb + (long)(t + 1) * p.hs_st + unit, og * tanhf(cn),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            PS_STAMP(5);
            ps_arrive(p.flags + ((long)l * T + t) * p.NRT + rt,        // (its barrier also frees Cs for the next tile)
                      p.trace ? p.trace + ((long)blockIdx.x * T + t) * 8 + 6 : nullptr);
            PS_STAMP(7);
        }
    }
}

// =====================================================================================================================
// backward
// =====================================================================================================================
template <bool H3>
__global__ __launch_bounds__(256) void gpe_rnn_persist_bwd_kernel(PsBwdParams p)
{
    extern __shared__ __align__(16) char ps_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    int bid = blockIdx.x;
    const int nb = bid % p.NB; bid /= p.NB;
    const int rg = bid % p.RG;
    const int l = bid / p.RG;
    const int H = p.H, T = p.T, Bn = p.Bn, L = p.L, KP = p.KP, K4 = 4 * H;
    const long BH = (long)Bn * H;
    const int wbytes = KP * 64;                      // 16 columns x KP x 4 bytes
    char* W0 = ps_smem;
    char* W1 = ps_smem + wbytes;
    float* Cs = reinterpret_cast<float*>(ps_smem + (L > 1 ? 2 : 1) * wbytes);         // [4 waves][16][PS_LDB]
    const bool up = l < L - 1;                       // a layer above feeds dG_{l+1,t} x W_ih_{l+1}

    ps_fill<16>(W0, p.w0[l], KP >> 2, 16 * p.NB, 16 * nb);
    if (up) ps_fill<16>(W1, p.w1[l], KP >> 2, 16 * p.NB, 16 * nb);
    float invw0 = 1.f, invw1 = 1.f;
    if constexpr (H3) {
        float sw;
        gpe_h3_scale_of(p.s0[l][0], sw, invw0);
        if (up) gpe_h3_scale_of(p.s1[l][0], sw, invw1);
    }
    __syncthreads();

    const int er = tid >> 4, eu = tid & 15;
    const int unit = 16 * nb + eu;
    const bool uok = unit < H;
    const int unitc = uok ? unit : H - 1;
    constexpr int MAXS = H3 ? 8 : 16;                // KP <= 1024: 32 steps of 32 / 64 steps of 16 over four waves
    const int nsteps = H3 ? (KP >> 5) : (KP >> 4);
    const int spw = (nsteps + 3) >> 2;
    const int s_lo = wave * spw;
    const int s_hi = (s_lo + spw < nsteps) ? s_lo + spw : nsteps;
    const unsigned need = (unsigned)p.NB;
    const unsigned nrec = (unsigned)((((long)Bn - 1) * p.dg_sb + ((K4 + 3) & ~3)) * 4);

    for (int t = T - 1; t >= 0; --t) {
        for (int rt = rg; rt < p.NRT; rt += p.RG) {
            const int row = 16 * rt + er;
            const bool rok = row < Bn;
            const int rowc = rok ? row : Bn - 1;
            // pointwise operands first
            const long e = (long)rowc * H + unitc;
            float dh = 0.f;
            if (l == L - 1 && p.dtop) dh = p.dtop[(long)rowc * p.dt_sb + (long)t * p.dt_st + unitc];
            if (t == T - 1 && p.d_hN) dh += p.d_hN[(long)l * BH + e];
            float cin = 0.f;
            if (t == T - 1) { if (p.d_cN) cin = p.d_cN[(long)l * BH + e]; }
            else cin = p.carry[((long)((t + 1) & 1) * L + l) * BH + e];
            const float* sv = p.saved + l * p.sv_sl + (long)t * p.sv_st + (long)rowc * 4 * H;
            const float ig = sv[unitc], fg = sv[H + unitc], gg = sv[2 * H + unitc], og = sv[3 * H + unitc];
            const float ct = p.cs[l * p.cs_sl + (long)(t + 1) * p.cs_st + e];
            const float cp = p.cs[l * p.cs_sl + (long)t * p.cs_st + e];

            PS_STAMP(0);
            const int arow = (16 * rt + j < Bn) ? 16 * rt + j : Bn - 1;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            // both flags first, then the loads of both segments back to back (one latency instead of two), then the products
            PsA<H3, MAXS> A1, A0;
            const bool rec = t < T - 1;
            if (up) ps_wait(p.flags + ((long)(l + 1) * T + t) * p.NRT + rt, need);       // dG_{l+1,t}: the layer above runs ahead
            PS_STAMP(1);
            if (rec) ps_wait(p.flags + ((long)l * T + t + 1) * p.NRT + rt, need);        // dG_{l,t+1}
            PS_STAMP(3);
            if (up)                                  // dG_{l+1,t} x W_ih_{l+1}
                ps_load<H3, MAXS>(A1, __builtin_amdgcn_make_buffer_rsrc(p.dgx + (l + 1) * p.dg_sl + (long)t * p.dg_st, 0, nrec, 0x00020000),
                                  arow, p.dg_sb, K4, s_lo, nsteps);
            if (rec)                                 // dG_{l,t+1} x W_hh_l
                ps_load<H3, MAXS>(A0, __builtin_amdgcn_make_buffer_rsrc(p.dgx + l * p.dg_sl + (long)(t + 1) * p.dg_st, 0, nrec, 0x00020000),
                                  arow, p.dg_sb, K4, s_lo, nsteps);
            if (up) {
                f32x4 a1[1] = {{0.f, 0.f, 0.f, 0.f}};
                const float ia = ps_mma<H3, 1, MAXS, true>(A1, KP, W1, s_lo, s_hi, 1.f, a1);
                acc = a1[0] * (ia * invw1);
            }
            PS_STAMP(2);
            if (rec) {
                f32x4 a0[1] = {{0.f, 0.f, 0.f, 0.f}};
                const float ia = ps_mma<H3, 1, MAXS, true>(A0, KP, W0, s_lo, s_hi, 1.f, a0);
                acc += a0[0] * (ia * invw0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(wave * 16 + 4 * g + r) * PS_LDB + j] = acc[r];
            PS_STAMP(4);
            __syncthreads();
            dh += (Cs[er * PS_LDB + eu] + Cs[(16 + er) * PS_LDB + eu]) + (Cs[(32 + er) * PS_LDB + eu] + Cs[(48 + er) * PS_LDB + eu]);
            if (rok && uok) {
                const float tc = tanhf(ct);
                const float dc = dh * og * (1.f - tc * tc) + cin;
                float* gx = p.dgx + l * p.dg_sl + (long)row * p.dg_sb + (long)t * p.dg_st;
                __hip_atomic_store(gx + unit, dc * gg * ig * (1.f - ig), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gx + H + unit, dc * cp * fg * (1.f - fg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gx + 2 * H + unit, dc * ig * (1.f - gg * gg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gx + 3 * H + unit, dh * tc * og * (1.f - og), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                p.carry[((long)(t & 1) * L + l) * BH + (long)row * H + unit] = dc * fg;
            }
            PS_STAMP(5);
            ps_arrive(p.flags + ((long)l * T + t) * p.NRT + rt, p.trace ? p.trace + ((long)blockIdx.x * T + t) * 8 + 6 : nullptr);
            PS_STAMP(7);
        }
    }
}

// =====================================================================================================================
// host side
// =====================================================================================================================
struct PsPlan { int NB, NRT, RG, grid; };

// gpe_debug_set bits (measurement aids): 1024 = never the persistent kernels; 2048 / 4096 = allow them for a stack whose row tiles
// outnumber the row groups the chip can hold (forward / backward: several row tiles per workgroup)
static bool ps_plan(int gates, int L, int T, int Bn, int H, bool bwd, PsPlan& pl)
{
    const int dbg = gpe_debug_get();
    if (gates != 4 || L < 1 || L > PS_MAXL || T < 1 || Bn < 1 || H < 1 || H > 256 || (dbg & 1024)) return false;
    pl.NB = gpe_cdiv(H, 16);
    pl.NRT = gpe_cdiv(Bn, 16);
    const int cus = gpe_num_cus();
    const int per = L * pl.NB;
    if (cus <= 0 || per > cus) return false;
    pl.RG = cus / per < pl.NRT ? cus / per : pl.NRT;
    if (pl.RG < pl.NRT && !(dbg & (bwd ? 4096 : 2048))) return false;
    pl.grid = per * pl.RG;
    return true;
}

// bytes of the arrival counters (0: this stack does not run persistently)
static long ps_flag_bytes(int L, int T, const PsPlan& pl) { return (((long)L * T * pl.NRT * 4) + 255) & ~255L; }
static long ps_trace_bytes(int T, const PsPlan& pl) { return (gpe_debug_get() & 8192) ? (long)pl.grid * T * 8 * 8 : 0; }

long gpe_rnn_persist_ws_bytes(int gates, int L, int T, int Bn, int H, int bwd)
{
    PsPlan pl;
    if (!ps_plan(gates, L, T, Bn, H, bwd != 0, pl)) return 0;
    return ps_flag_bytes(L, T, pl) + ps_trace_bytes(T, pl);
}

// 1 = launched, 0 = not eligible (the caller runs the diagonal launches), < 0 = error
int gpe_rnn_persist_fwd(int L, int T, int Bn, int H, const float* xproj0, long xp0_sb, long xp0_st, const void* const* whh,
                        const void* const* wih, const void* const* bias, float* hs, long hs_sl, long hs_sb, long hs_st, float* cs,
                        long cs_sl, long cs_st, float* saved, long sv_sl, long sv_st, bool h3, const void* const* whh_amax,
                        const void* const* wih_amax, void* ws, long ws_bytes, hipStream_t s)
{
    PsPlan pl;
    if (!ps_plan(4, L, T, Bn, H, false, pl)) return 0;
    const long need = ps_flag_bytes(L, T, pl), ntrace = ps_trace_bytes(T, pl);
    if (!ws || ws_bytes < need + ntrace || (((uintptr_t)ws) & 7)) return 0;
    if ((long)Bn * hs_sb * 4 >= (1L << 31) || (hs_sb & 3) || (hs_st & 3) || (((uintptr_t)hs) & 15)) return 0;
    const int KP = h3 ? gpe_round_up(H, 32) : gpe_round_up(H, 16);
    const size_t lds = (size_t)(L > 1 ? 2 : 1) * KP * 256 + 4 * 16 * PS_LDC * 4;
    if (lds > 160 * 1024) return 0;
    PsFwdParams p = {};
    p.L = L; p.T = T; p.Bn = Bn; p.H = H; p.NB = pl.NB; p.RG = pl.RG; p.NRT = pl.NRT;
    p.xproj0 = xproj0; p.xp0_sb = xp0_sb; p.xp0_st = xp0_st;
    for (int l = 0; l < L; ++l) {
        p.w0[l] = whh[l];
        if (!whh[l] || (((uintptr_t)whh[l]) & 15)) return 0;
        if (h3) p.s0[l] = (const unsigned*)whh_amax[l];
        if (l > 0) {
            if (!wih[l] || (((uintptr_t)wih[l]) & 15) || !bias[l]) return 0;
            p.w1[l] = wih[l];
            p.bias[l] = (const float*)bias[l];
            if (h3) p.s1[l] = (const unsigned*)wih_amax[l];
        }
    }
    p.hs = hs; p.hs_sl = hs_sl; p.hs_sb = hs_sb; p.hs_st = hs_st;
    p.cs = cs; p.cs_sl = cs_sl; p.cs_st = cs_st;
    p.saved = saved; p.sv_sl = sv_sl; p.sv_st = sv_st;
    p.flags = (unsigned*)ws;
    p.trace = ntrace ? (unsigned long long*)((char*)ws + need) : nullptr;
    if (hipMemsetAsync(ws, 0, (size_t)(need + ntrace), s) != hipSuccess) return GPE_ELAUNCH;
    if (h3) {
        GPE_ENSURE_MAX_LDS(gpe_rnn_persist_fwd_kernel<true>);
        hipLaunchKernelGGL(gpe_rnn_persist_fwd_kernel<true>, dim3(pl.grid), dim3(256), lds, s, p);
    } else {
        GPE_ENSURE_MAX_LDS(gpe_rnn_persist_fwd_kernel<false>);
        hipLaunchKernelGGL(gpe_rnn_persist_fwd_kernel<false>, dim3(pl.grid), dim3(256), lds, s, p);
    }
    GPE_CHECK_LAUNCH();
    return 1;
}

int gpe_rnn_persist_bwd(int L, int T, int Bn, int H, const float* dtop, long dt_sb, long dt_st, const float* d_hN,
                        const float* d_cN, const void* const* whh_t, const void* const* wih_t, int KP, const float* cs, long cs_sl,
                        long cs_st, const float* saved, long sv_sl, long sv_st, float* dgx, long dg_sl, long dg_sb, long dg_st,
                        float* carry, bool h3, const void* const* whh_amax, const void* const* wih_amax, void* ws, long ws_bytes,
                        hipStream_t s)
{
    PsPlan pl;
    if (!ps_plan(4, L, T, Bn, H, true, pl)) return 0;
    const long need = ps_flag_bytes(L, T, pl), ntrace = ps_trace_bytes(T, pl);
    if (!ws || ws_bytes < need + ntrace || (((uintptr_t)ws) & 7)) return 0;
    if ((long)Bn * dg_sb * 4 >= (1L << 31) || (dg_sb & 3) || (dg_st & 3) || (((uintptr_t)dgx) & 15) || (dg_sl & 3)) return 0;
    if (KP < 4 * H || KP > 1024 || (KP & (h3 ? 31 : 15))) return 0;
    const size_t lds = (size_t)(L > 1 ? 2 : 1) * KP * 64 + 4 * 16 * PS_LDB * 4;
    if (lds > 160 * 1024) return 0;
    PsBwdParams p = {};
    p.L = L; p.T = T; p.Bn = Bn; p.H = H; p.NB = pl.NB; p.RG = pl.RG; p.NRT = pl.NRT; p.KP = KP;
    p.dtop = dtop; p.dt_sb = dt_sb; p.dt_st = dt_st; p.d_hN = d_hN; p.d_cN = d_cN;
    for (int l = 0; l < L; ++l) {
        if (!whh_t[l] || (((uintptr_t)whh_t[l]) & 15)) return 0;
        p.w0[l] = whh_t[l];
        if (h3) p.s0[l] = (const unsigned*)whh_amax[l];
        if (l < L - 1) {
            if (!wih_t[l + 1] || (((uintptr_t)wih_t[l + 1]) & 15)) return 0;
            p.w1[l] = wih_t[l + 1];
            if (h3) p.s1[l] = (const unsigned*)wih_amax[l + 1];
        }
    }
    p.cs = cs; p.cs_sl = cs_sl; p.cs_st = cs_st;
    p.saved = saved; p.sv_sl = sv_sl; p.sv_st = sv_st;
    p.dgx = dgx; p.dg_sl = dg_sl; p.dg_sb = dg_sb; p.dg_st = dg_st;
    p.carry = carry;
    p.flags = (unsigned*)ws;
    p.trace = ntrace ? (unsigned long long*)((char*)ws + need) : nullptr;
    if (hipMemsetAsync(ws, 0, (size_t)(need + ntrace), s) != hipSuccess) return GPE_ELAUNCH;
    if (h3) {
        GPE_ENSURE_MAX_LDS(gpe_rnn_persist_bwd_kernel<true>);
        hipLaunchKernelGGL(gpe_rnn_persist_bwd_kernel<true>, dim3(pl.grid), dim3(256), lds, s, p);
    } else {
        GPE_ENSURE_MAX_LDS(gpe_rnn_persist_bwd_kernel<false>);
        hipLaunchKernelGGL(gpe_rnn_persist_bwd_kernel<false>, dim3(pl.grid), dim3(256), lds, s, p);
    }
    GPE_CHECK_LAUNCH();
    return 1;
}
