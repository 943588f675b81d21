// Dense GEMMs of the decoders and of the [P|Q] projection on the 16-bit matrix pipe with fp32-grade products (round 6):
//   NT  Y[r][n] = act(sum_k A[r][k] W[n][k] + bias[n] + addend[r][n])      (gpe_linear: nn.Linear / the per-point projection
//       /root/reference/nn/net_blocks.py:45,156-158,373-376,397)
//   TN  G[m][n] = sum_r U[r][m] (V[r][n] - shift[n]),  colsum[m] = sum_r U[r][m]   (gpe_redgemm: their weight gradients)
// for the shapes that are neither latency-bound nor on the fused edge kernels' menu: at cfg 2 the five panel-decoder weight gradients
// (10304 rows x 1000 x 250: 71 us each on v_mfma_f32_16x16x4_f32, 0.46 of that pipe's peak) and the layer-2 [P|Q] projection with its
// two gradients (65536 rows x 400 x 150: 134 - 159 us each, 0.3 of peak) — 0.8 ms of a 9.5 ms step.
//
// Arithmetic: SplitBf16x3 ("bf16x6", gpe_edgegemm_split_kernel.h) — every fp32 operand x = h + m + l in three bf16 terms (24 mantissa
// bits, fp32's exponent range: NO scale, no amax word, nothing to overflow), six v_mfma_f32_16x16x32_bf16 per product block, small
// terms first, fp32 accumulate; dropped terms are O(2^-24) per product, the size of fp32's own rounding.  Both operands are split
// on the fly while they are staged (the weights of gpe_linear arrive as the plain fp32 pack), so the kernels take the same
// arguments as the exact ones and are selected by the arithmetic mode alone (gpe_math_set(4): the parity-grade fast mode).
//
// A workgroup computes a 128 x 128 output block, K in steps of 32.  TN (many steps per block): one 768-thread workgroup per CU, TWO
// ROLES (one multiplying and two staging waves per SIMD) —
//   waves 0-3  multiply: 2 x 2 waves, each 64 x 64 = 4 x 4 MFMA tiles, 96 MFMAs per step from the LDS image of the step;
//   waves 4-11 stage   : global loads three steps ahead of the MFMAs (three register sets: the loads of a step have two whole steps
//              to arrive), three-term split (~5.5 VALU per value, 4 cycles each: it issues under the partner wave's MFMAs instead
//              of behind the wave's own), LDS writes into the other of two buffers; one barrier per step.
//   (A first version did both jobs in every wave: MFMA pipe 0.26 - 0.31 busy, issue-stalled 0.33 - 0.41 — the splits sat between
//   the MFMA blocks; profiles/r06_b_dense_gemms.md.)
// LDS image of one operand of one step: [3 planes][128 rows][4 slots of 8 bf16], slot' = kgroup ^ (row & 8 ? 2 : 0): ds_read_b128 is
// serviced in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS) — with lane (j, g) reading
// row 16 t + j, k group g, the XOR puts the 16 lanes of every group on 16 different four-bank slots.
//   NT stager: thread -> (row, k quad): one float4 along k of each operand per 256 pieces, 8-byte LDS writes (conflict-free).
//   TN stager: the reduction index is the SLOW index of both operands.  Thread -> (4 consecutive rows, column quad): four float4
//              loads (128-byte runs along the columns), transposed in registers — column c of the four rows = half of the 8 k-values
//              of one MFMA fragment lane — split, one 8-byte LDS write per column and plane.  Staging waves 4-7 take U, 8-11 V.
// TN splits the rows over gridDim.x workgroups per output block; partial blocks go to the caller's `part` image
// [split][MgPad][NgPad] (+ fp64 column sums) and gpe_redgemm_finish adds them in fp64 (gpe_redgemm.hip: same image as the exact path).
#include "gpe_edgegemm_split_kernel.h"

#define GX_B 128                      // block edge (rows of A / columns of W; columns of U / of V)
#define GX_PLANE (GX_B * 64)          // bytes: 128 rows x 32 k x 2
#define GX_OP (3 * GX_PLANE)
#define GX_BUF (2 * GX_OP)            // one step: both operands, 49152 bytes
#define GX_LDC 132
#define GX_LDS (2 * GX_BUF)           // two steps; the epilogue's C image 128 x 132 x 4 = 67584 bytes (+ 4 KB of column sums) fits

typedef SplitBf16x3 GXS;
extern "C" int gpe_debug_get(void);

__device__ __forceinline__ int gx_slot(int row, int kg) { return kg ^ ((row >> 2) & 2); }

// the 96 MFMAs of one K step: this wave's 64 x 64 sub-block, operands from the LDS image of the step.  The A fragments of the four
// row tiles stay in registers (48); the B fragments of one column tile at a time (12): between two MFMAs of the same accumulator
// lie the three other row tiles
__device__ __forceinline__ void gx_compute(const char* Ab, const char* Bb, int wm, int wn, int j, int g, f32x4 (&acc)[4][4])
{
    x6_u32x4 a[4][3];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ra = 64 * wm + 16 * t + j;
        const char* pa = Ab + ra * 64 + 16 * gx_slot(ra, g);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) a[t][pl] = *reinterpret_cast<const x6_u32x4*>(pa + pl * GX_PLANE);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        x6_u32x4 b[3];
        const int rb = 64 * wn + 16 * nt + j;
        const char* pb = Bb + rb * 64 + 16 * gx_slot(rb, g);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const x6_u32x4*>(pb + pl * GX_PLANE);
#pragma unroll
        for (int t = 0; t < GXS::NPROD; ++t)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = GXS::mfma(a[mt][GXS::pa(t)], b[GXS::pw(t)], acc[mt][nt]);
    }
}

// a multiplying wave's accumulators -> the block's C image [128][GX_LDC] in LDS (the staging buffers are free by then)
__device__ __forceinline__ void gx_stage_c(float* Cs, int wm, int wn, int j, int g, const f32x4 (&acc)[4][4])
{
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(64 * wm + 16 * mt + 4 * g + r) * GX_LDC + 64 * wn + 16 * nt + j] = acc[mt][nt][r];
}

// =====================================================================================================================
// NT: gpe_linear
// =====================================================================================================================
struct GxLinParams {
    GpeRows a; const float* wp; int Npad, Kq;          // packed weight [Kq quads][Npad][4]
    const float* bias; GpeRows addend;
    float* y; long y_so, y_si; int y_inner; int act;
    long M; int N, K; int yvec;                         // yvec: y / addend rows take 16-byte accesses
};

// NT keeps BOTH jobs in every wave, 256 threads, ONE staging buffer (48 KB) and two workgroups per CU: its K is short (5 - 13 steps at
// the shipped shapes), so a block is mostly prologue + epilogue, and those only overlap with MFMAs through a second resident
// workgroup.  Measured (65536 x 400 x 150 / 65536 x 150 x 400, us): this form 97 / 106, the role split of the TN kernel 113 / 123,
// a second register set two steps ahead 115 / 133, the exact fp32 kernel 170 / 142.
__global__ __launch_bounds__(256, 2) void gpe_gemm_x6_nt_kernel(GxLinParams p)
{
    extern __shared__ __align__(16) char gx_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4, wm = wave >> 1, wn = wave & 1;
    const long row0 = (long)blockIdx.x * GX_B;
    const int col0 = blockIdx.y * GX_B;
    const int nsteps = (p.K + 31) >> 5;
    const int lr = tid >> 3, kq = tid & 7;              // piece i of this thread: row lr + 32 i, k quad kq, of both operands
    const float* arow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        long r = row0 + lr + 32 * i;
        if (r >= p.M) r = p.M - 1;
        arow[i] = gpe_row_ptr(p.a, r);
    }
    float4 ra[4], rb[4];
    // every load is unconditional (a load under a branch is waited for at the join): clamped address, value masked at commit time
    // (the first use of a loaded register is where the wave waits for it).  The rows are 16-byte loadable up to round4(K)
    // (host-checked), so a ragged last quad is loaded whole.
    const int K4 = (p.K + 3) & ~3;
    auto load = [&](int s) {
        const int k0 = 32 * s + 4 * kq;
        const int q = 8 * s + kq;
        const int kc = (k0 < K4) ? k0 : 0;
        const bool okq = q < p.Kq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = ld4(arow[i] + kc);
            const int n = col0 + lr + 32 * i;
            rb[i] = ld4(p.wp + ((long)(okq ? q : 0) * p.Npad + (n < p.Npad ? n : 0)) * 4);
        }
    };
    auto commit = [&](int s) {
        char* Ab = gx_smem;
        char* Bb = Ab + GX_OP;
        const int k0 = 32 * s + 4 * kq;
        const bool okq = 8 * s + kq < p.Kq;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = lr + 32 * i;
            const int off = row * 64 + 16 * gx_slot(row, kq >> 1) + 8 * (kq & 1);
            unsigned q0[3], q1[3];
            float4 x = ra[i];
            x.x = (k0 < p.K) ? x.x : 0.f; x.y = (k0 + 1 < p.K) ? x.y : 0.f;
            x.z = (k0 + 2 < p.K) ? x.z : 0.f; x.w = (k0 + 3 < p.K) ? x.w : 0.f;
            const float4 w = (okq && col0 + row < p.Npad) ? rb[i] : zero4;
            GXS::split2(x.x, x.y, q0);
            GXS::split2(x.z, x.w, q1);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(Ab + pl * GX_PLANE + off) = make_uint2(q0[pl], q1[pl]);
            GXS::split2(w.x, w.y, q0);
            GXS::split2(w.z, w.w, q1);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(Bb + pl * GX_PLANE + off) = make_uint2(q0[pl], q1[pl]);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    load(0);
    commit(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const bool more = s + 1 < nsteps;
        if (more) load(s + 1);                             // in flight under this step's MFMAs
        gx_compute(gx_smem, gx_smem + GX_OP, wm, wn, j, g, acc);
        __syncthreads();                                   // every wave has read its fragments
        if (more) commit(s + 1);
        __syncthreads();
    }
    // ---- rows out, in two halves of 64 (the C image of a half fits the staging buffer): a thread's 8 pieces of a half share one
    // column quad (bias loaded once, addend rows requested together, nothing waited for inside the loop) ----
    float* Cs = reinterpret_cast<float*>(gx_smem);
    const int cq = (tid & 31) << 2, col = col0 + cq;
    const int nv = (p.N - col < 4) ? p.N - col : 4;       // valid columns of this thread's quad (<= 0: none)
    float bq[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias)
        for (int t = 0; t < 4; ++t) bq[t] = p.bias[(col + t < p.N) ? col + t : p.N - 1];
    const bool vec = nv == 4 && p.yvec;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (wm == half) gx_stage_c(Cs, 0, wn, j, g, acc);
        float4 adq[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            long row = row0 + 64 * half + (tid >> 5) + 8 * i;
            if (row >= p.M) row = p.M - 1;
            adq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.addend.base) {                           // (uniform)
                const float* ad = gpe_row_ptr(p.addend, row) + ((col < p.N) ? col : 0);
                if (vec) adq[i] = ld4(ad);
                else { adq[i].x = ad[0]; adq[i].y = ad[nv > 1 ? 1 : 0]; adq[i].z = ad[nv > 2 ? 2 : 0]; adq[i].w = ad[nv > 3 ? 3 : 0]; }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = (tid >> 5) + 8 * i;
            const float4 v = ld4(&Cs[r * GX_LDC + cq]);
            float o[4] = {v.x + bq[0] + adq[i].x, v.y + bq[1] + adq[i].y, v.z + bq[2] + adq[i].z, v.w + bq[3] + adq[i].w};
            if (p.act == 1) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f); }
            const long row = row0 + 64 * half + r;
            if (row < p.M && nv > 0) {
                float* yr = (p.y_inner <= 0 ? p.y + row * p.y_so : p.y + (row / p.y_inner) * p.y_so + (row % p.y_inner) * p.y_si) + col;
                if (vec) st4(yr, make_float4(o[0], o[1], o[2], o[3]));
                else
                    for (int t = 0; t < nv; ++t) yr[t] = o[t];
            }
        }
        __syncthreads();
    }
}

// =====================================================================================================================
// TN: gpe_redgemm
// =====================================================================================================================
struct GxRedParams {
    GpeRows u, v; const float* v_shift;
    long rows, rows_per_split;
    int Mg, Ng, MgPad, NgPad;
    float* part; double* part_cs;
    unsigned long long* trace;              // measurement aid (scripts/gemm_trace.py): [2 roles][steps][4] stamps of workgroup 0, else NULL
};
#define GX_STAMP(role_, s_, i_)                                                                              \
    do {                                                                                                     \
        if (p.trace && blockIdx.x + blockIdx.y + blockIdx.z == 0 && (tid & 255) == 0 && (s_) < 64)             \
            p.trace[((role_) * 64 + (s_)) * 4 + (i_)] = wall_clock64();                                      \
    } while (0)

struct GxTnRegs { float4 r[4]; };

__global__ __launch_bounds__(768) void gpe_gemm_x6_tn_kernel(GxRedParams p)
{
    extern __shared__ __align__(16) char gx_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >= 4;                         // waves 0-3 multiply, waves 4-11 stage
    const int j = lane & 15, g = lane >> 4, wm = (wave >> 1) & 1, wn = wave & 1;
    const int split = blockIdx.x;
    const int m0 = blockIdx.y * GX_B, n0 = blockIdx.z * GX_B;
    const long r_begin = (long)split * p.rows_per_split;
    const long r_end = (r_begin + p.rows_per_split < p.rows) ? r_begin + p.rows_per_split : p.rows;
    const int nsteps = (r_end > r_begin) ? (int)((r_end - r_begin + 31) >> 5) : 0;
    f32x4 acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    double* red = reinterpret_cast<double*>(gx_smem + GX_B * GX_LDC * 4);          // [8 rq][128] column sums, behind the C image

    if (role == 1) {
        // ---- stager: waves 4-7 -> U (columns m0 ..), waves 8-11 -> V (columns n0 ..); thread -> (row quad rq of the step's 32 rows,
        // column quad mq): four float4 loads, 16 values to split (the step timeline of the first role-split version — four staging
        // waves, eight rows per thread — showed the multiplying waves waiting 45 % of every step for the stagers' ~800 VALU:
        // profiles/r06_b_dense_gemms.md) ----
        const bool isv = wave >= 8;
        const int t8 = (tid - 256) & 255;
        const int rq = t8 & 7, mq = t8 >> 3;
        const GpeRows op = isv ? p.v : p.u;
        const int width = isv ? p.Ng : p.Mg;
        const int c0 = (isv ? n0 : m0) + 4 * mq;           // first of this thread's four columns
        const int ncol = (width - c0 >= 4) ? 4 : (width - c0 > 0 ? width - c0 : 0);
        const int cc = ncol > 0 ? c0 : 0;                  // clamped: an all-invalid quad loads columns 0..3 and discards them
        float sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (isv && p.v_shift)
            for (int c = 0; c < ncol; ++c) sh[c] = p.v_shift[c0 + c];
        const bool do_cs = !isv && blockIdx.z == 0 && p.part_cs;
        double cs[4] = {0.0, 0.0, 0.0, 0.0};
        // one addressing form for both row layouts (no branch around the loads): row r = (o, i), o = r / inner, i = r % inner; plain
        // rows are the case inner = "infinite" (o = 0, i = r, inner stride = the row pitch).  One 32-bit division per step and
        // thread (rows < 2^31: host-checked), then a walk by selects (a branch between two loads makes the compiler wait for the
        // first).  Rows past the end re-read the last valid row and are masked at commit time.
        const unsigned inner_eff = op.inner > 0 ? (unsigned)op.inner : 0x7fffffffu;
        const long si_eff = op.inner > 0 ? op.stride_inner : op.stride_outer;
        const long r_last = r_end > r_begin ? r_end - 1 : r_begin;
        auto load = [&](GxTnRegs& R, int s) {
            if (s >= nsteps) s = nsteps - 1;
            const long rbase = r_begin + 32L * s + 4 * rq;
            const unsigned rb0 = (unsigned)(rbase < r_last ? rbase : r_last);
            unsigned o = rb0 / inner_eff, i = rb0 - o * inner_eff;
            const unsigned left = (unsigned)r_last - rb0;  // rows that may still advance
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                R.r[q] = ld4(op.base + (long)o * op.stride_outer + (long)i * si_eff + cc);
                const unsigned i2 = i + ((unsigned)q < left ? 1u : 0u);
                const bool wrap = i2 == inner_eff;
                i = wrap ? 0u : i2;
                o += wrap ? 1u : 0u;
            }
        };
        auto commit = [&](GxTnRegs& R, int s) {
            char* Ob = gx_smem + (s & 1) * GX_BUF + (isv ? GX_OP : 0);
            const long rbase = r_begin + 32L * s + 4 * rq;
            float v[4][4];
            float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool ok = rbase + q < r_end;
                const float x[4] = {R.r[q].x, R.r[q].y, R.r[q].z, R.r[q].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[q][c] = (ok && c < ncol) ? x[c] - sh[c] : 0.f;
                    s4[c] += v[q][c];
                }
            }
            if (do_cs)
                for (int c = 0; c < 4; ++c) cs[c] += (double)s4[c];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int row = 4 * mq + c;                // LDS row = output column inside the block
                unsigned k01[3], k23[3];
                GXS::split2(v[0][c], v[1][c], k01);
                GXS::split2(v[2][c], v[3][c], k23);
                char* dst = Ob + row * 64 + 16 * gx_slot(row, rq >> 1) + 8 * (rq & 1);   // k = 4 rq .. 4 rq + 3: half a slot
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(dst + pl * GX_PLANE) = make_uint2(k01[pl], k23[pl]);
            }
        };
        GxTnRegs R0, R1, R2;
        if (nsteps > 0) {
            load(R0, 0); load(R1, 1); load(R2, 2);
            commit(R0, 0);
        }
        __syncthreads();
        for (int s = 0; s < nsteps; s += 3) {
            GX_STAMP(1, s, 0);
            if (s + 1 < nsteps) commit(R1, s + 1);
            GX_STAMP(1, s, 1);
            load(R0, s + 3);
            GX_STAMP(1, s, 2);
            __syncthreads();
            GX_STAMP(1, s, 3);
            if (s + 1 >= nsteps) break;
            GX_STAMP(1, s + 1, 0);
            if (s + 2 < nsteps) commit(R2, s + 2);
            GX_STAMP(1, s + 1, 1);
            load(R1, s + 4);
            GX_STAMP(1, s + 1, 2);
            __syncthreads();
            GX_STAMP(1, s + 1, 3);
            if (s + 2 >= nsteps) break;
            GX_STAMP(1, s + 2, 0);
            if (s + 3 < nsteps) commit(R0, s + 3);
            GX_STAMP(1, s + 2, 1);
            load(R2, s + 5);
            GX_STAMP(1, s + 2, 2);
            __syncthreads();
            GX_STAMP(1, s + 2, 3);
        }
        // (the staging buffers are dead: every multiplying wave is past its last fragment read — the loop's last barrier)
        if (do_cs)
            for (int c = 0; c < 4; ++c) red[rq * GX_B + 4 * mq + c] = cs[c];
    } else {
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            const char* Ab = gx_smem + (s & 1) * GX_BUF;
            GX_STAMP(0, s, 0);
            gx_compute(Ab, Ab + GX_OP, wm, wn, j, g, acc);
            GX_STAMP(0, s, 1);
            __syncthreads();
            GX_STAMP(0, s, 3);
        }
        gx_stage_c(reinterpret_cast<float*>(gx_smem), wm, wn, j, g, acc);
    }
    __syncthreads();
    const float* Cs = reinterpret_cast<const float*>(gx_smem);
    float* dst = p.part + (size_t)split * p.MgPad * p.NgPad;
    for (int e = tid; e < GX_B * 32; e += 768) {
        const int r = e >> 5, cq = (e & 31) << 2;
        st4(dst + (size_t)(m0 + r) * p.NgPad + n0 + cq, ld4(&Cs[r * GX_LDC + cq]));
    }
    if (p.part_cs && blockIdx.z == 0 && tid < GX_B) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += red[q * GX_B + tid];
        p.part_cs[(size_t)split * p.MgPad + m0 + tid] = t;
    }
}

// =====================================================================================================================
// host side (called by gpe_linear / gpe_redgemm when the arithmetic mode allows it)
// =====================================================================================================================
static bool gx_rows16(const GpeRows& r, int cols)
{
    if (!r.base || (((uintptr_t)r.base) & 15) || (r.stride_outer & 3)) return false;
    if (r.inner > 0) return !(r.stride_inner & 3) && r.stride_inner >= ((cols + 3) & ~3);
    return r.stride_outer >= ((cols + 3) & ~3);
}

// 1 = launched, 0 = not on this kernel's menu
int gpe_gemm_x6_linear(const GpeRows& a, const float* wp, int Npad, int Kq, const float* bias, const GpeRows& addend, float* y,
                       long y_so, long y_si, int y_inner, long M, int N, int K, int act, hipStream_t s)
{
    // worth it from ~0.25 GFLOP with at least one full wave of blocks' worth of rows; A rows 16-byte loadable up to round4(K)
    if (M < 2048 || N < 48 || K < 32 || 2.0 * M * N * K < 2.5e8) return 0;
    if (!gx_rows16(a, K) || (((uintptr_t)wp) & 15)) return 0;
    GxLinParams p = {};
    p.a = a; p.wp = wp; p.Npad = Npad; p.Kq = Kq; p.bias = bias; p.addend = addend;
    p.y = y; p.y_so = y_so; p.y_si = y_si; p.y_inner = y_inner; p.act = act; p.M = M; p.N = N; p.K = K;
    p.yvec = !(((uintptr_t)y) & 15) && !(y_so & 3) && (y_inner <= 0 || !(y_si & 3)) && !(N & 3) &&
             (!addend.base || (!(((uintptr_t)addend.base) & 15) && !(addend.stride_outer & 3) && (addend.inner <= 0 || !(addend.stride_inner & 3))));
    GPE_ENSURE_MAX_LDS(gpe_gemm_x6_nt_kernel);
    hipLaunchKernelGGL(gpe_gemm_x6_nt_kernel, dim3((unsigned)gpe_cdiv(M, GX_B), gpe_cdiv(N, GX_B)), dim3(256), GX_BUF, s, p);
    GPE_CHECK_LAUNCH();
    return 1;
}

// workspace floats this path needs for an Mg x Ng product (gpe_redgemm_ws takes the maximum over the paths)
long gpe_gemm_x6_red_ws(int Mg, int Ng)
{
    const long MgPad = gpe_round_up(Mg, GX_B), NgPad = gpe_round_up(Ng, GX_B);
    return 32L * MgPad * NgPad + 2L * 32 * MgPad + 8 + 1024;      // (+ 4 KB: the step timeline of gpe_debug_set(32768))
}

long gpe_gemm_x6_red_ws(int Mg, int Ng);
// 1 = partial blocks written (the caller runs gpe_redgemm_finish over *nsplit partials of MgPad x NgPad), 0 = not on the menu
int gpe_gemm_x6_redgemm(const GpeRows& u, const GpeRows& v, const float* v_shift, long rows, int Mg, int Ng, float* part, bool want_cs,
                        int* nsplit, int* MgPad, int* NgPad, double** part_cs, hipStream_t s)
{
    // also the row-poor products (32 .. 736 rows: the exact kernels run those on ONE workgroup per column block — 39 - 53 us for
    // 2 - 90 MFLOP): here they are a handful of 128 x 128 blocks of one to six steps
    if (rows < 32 || rows >= (1L << 31) || Mg < 48 || Ng < 48 || 2.0 * rows * Mg * Ng < 2.0e6) return 0;
    if (!gx_rows16(u, Mg) || !gx_rows16(v, Ng)) return 0;
    const int mb = gpe_cdiv(Mg, GX_B), nb = gpe_cdiv(Ng, GX_B);
    int S = gpe_num_cus() / (mb * nb);
    if (S > 32) S = 32;
    if (S > rows / 128) S = (int)(rows / 128);             // >= 4 steps per workgroup
    if (S < 1) S = 1;
    GxRedParams p = {};
    p.u = u; p.v = v; p.v_shift = v_shift; p.rows = rows;
    p.rows_per_split = ((rows + S - 1) / S + 31) & ~31L;
    S = (int)((rows + p.rows_per_split - 1) / p.rows_per_split);
    p.Mg = Mg; p.Ng = Ng; p.MgPad = mb * GX_B; p.NgPad = nb * GX_B;
    p.part = part;
    size_t off = (size_t)S * p.MgPad * p.NgPad;
    off = (off + 1) & ~(size_t)1;
    p.part_cs = want_cs ? reinterpret_cast<double*>(part + off) : nullptr;
    if (gpe_debug_get() & 32768) {
        p.trace = reinterpret_cast<unsigned long long*>(part + gpe_gemm_x6_red_ws(Mg, Ng) - 1024);
        if (hipMemsetAsync(p.trace, 0, 4096, s) != hipSuccess) return GPE_ELAUNCH;
    }
    GPE_ENSURE_MAX_LDS(gpe_gemm_x6_tn_kernel);
    hipLaunchKernelGGL(gpe_gemm_x6_tn_kernel, dim3(S, mb, nb), dim3(768), GX_LDS, s, p);
    GPE_CHECK_LAUNCH();
    *nsplit = S; *MgPad = p.MgPad; *NgPad = p.NgPad; *part_cs = p.part_cs;
    return 1;
}
