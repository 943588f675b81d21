// Register-stationary-weight row GEMM for the per-edge MLP of DynamicEdgeConv on gfx950
// (/root/reference/nn/net_blocks.py:43-47,124-135 forward; its input-gradient half in backward).
//
// Why a second kernel next to gpe_rowgemm.hip: at the shipped sizes (EConv_hidden 200, EConv_feature 150) the
// layer weight (<= 208 x 208 fp32 = 169 KB) is too big for LDS next to the row tiles but fits the CU's register
// files (4 SIMDs x 128 KB).  So ONE persistent 256-thread workgroup per CU keeps the whole packed weight in VGPRs
// as ready-made MFMA B fragments for the entire kernel, and only the 64-row A tiles move:
//   * wave w owns N-tiles {AQ*w .. AQ*w+AQ-1} for all 64 rows, and the BQ left-over N-tiles are split by rows
//     (wave w takes rows 16w..16w+15 of each): 4*AQ+BQ = NT tiles -> every wave issues exactly NT MFMAs per k-step —
//     a perfectly balanced split of 13 (or 10) tiles over 4 SIMDs;
//   * A tiles are fetched global -> registers one tile ahead (for the gather producer: the dependent
//     neighbour-row loads too) and committed to the other half of a double-buffered LDS image: 2 barriers per tile
//     instead of 2 per 16-wide K chunk, no weight traffic at all in the main loop;
//   * per k-chunk a wave needs 5 ds_read_b128 for 52 MFMAs (v_mfma_f32_16x16x4_f32, 32-cycle issue).
// The epilogues (BN statistics in fp64, ReLU, max/min aggregation, BN/ReLU backward, per-point sums) are the
// shared ones of gpe_rowgemm.h, run on the accumulator tile staged through the just-consumed A buffer.
#include "gpe_rowgemm.h"

template <int AQ, int BQ, int KCH, int AMODE, int EMODE>
__global__ __launch_bounds__(256, 1) void gpe_edgegemm_kernel(RgParams p, int stats_nblk)
{
    constexpr int NT = 4 * AQ + BQ;
    constexpr int LDA = 16 * KCH + 4, LDC = 16 * NT + 4;
    constexpr int LDT = LDA > LDC ? LDA : LDC;
    constexpr int RQ = RG_BM / 4;                       // rows per wave in the staging map
    extern __shared__ __align__(16) float smem[];       // [2][64 * LDT]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;

    // ---- the weight, once, as MFMA B fragments: lane (j, g) holds k = 16*kc + 4*g .. +3 of column 16*n + j --------
    float4 wA[AQ][KCH];
    float4 wB[BQ > 0 ? BQ : 1][KCH];
#pragma unroll
    for (int i = 0; i < AQ; ++i) {
        const int col = 16 * (AQ * wave + i) + j;
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc)
            wA[i][kc] = (col < p.Npad) ? ld4(p.wp + (((long)(kc * 4 + g)) * p.Npad + col) * 4)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int b = 0; b < BQ; ++b) {
        const int col = 16 * (4 * AQ + b) + j;
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc)
            wB[b][kc] = (col < p.Npad) ? ld4(p.wp + (((long)(kc * 4 + g)) * p.Npad + col) * 4)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // ---- A staging --------------------------------------------------------------------------------------------
    // lane = column quad, rows r_q = wave + 4*q (q < 16).  Everything the address math needs is either constant per
    // kernel (the row's point slot r_q / k, since tiles hold whole points) or prefetched two tiles ahead (the
    // neighbour rows jg, one per lane q of a single VGPR), so no load in the MFMA stream depends on another load.
    // Loads are unconditional with clamped addresses; invalid rows / pad columns are zeroed when committed to LDS.
    // Rows are fetched two per K chunk during chunks 0..7 and committed CDEL chunks later: a register ring that the
    // allocator derives from the live ranges of the fully unrolled code.
    constexpr int CDEL = (KCH - 8 < 4) ? KCH - 8 : 4;
    const int cq = lane << 2;
    const bool col_ok = cq < p.K;                       // pad columns (K..16*KCH) are written as zeros
    const bool a_on = cq < 16 * KCH;
    const int cql = col_ok ? cq : 0;                    // clamped column for the address
    const int PT = p.R / p.k;
    int ptq[RQ];                                        // point slot of row r_q inside a tile (wave-uniform)
#pragma unroll
    for (int q = 0; q < RQ; ++q) ptq[q] = (int)(((float)(wave + 4 * q) + 0.5f) * (float)p.rcp_k);
    float4 ar[RQ], ar2[AMODE == A_GATHER ? RQ : 1];

    auto load_jg = [&](int tile) -> int {               // lane q < 16 holds jg of row r_q of `tile` (0 if out of range)
        int v = 0;
        if (AMODE == A_GATHER || EMODE == E_BWD_GATHER) {
            const long gr = (long)tile * p.R + wave + 4 * (lane & 15);
            const long grc = (tile < p.num_tiles && gr < p.M) ? gr : 0;
            v = p.jg[grc];
        }
        return v;
    };
    auto fetch_row = [&](int q, int tile, int rv, int jgv) {
        const int r = wave + 4 * q;
        const bool ok = r < rv;
        if (AMODE == A_DENSE) {
            const long gr = ok ? (long)tile * p.R + r : 0;
            const float* src = gpe_row_ptr(p.a, gr) + cql;
            ar[q] = ld4_guard(src, col_ok ? p.K - cq : 4, gpe_aligned16(src));
        } else {
            const long i = ok ? (long)tile * PT + ptq[q] : 0;
            const long jj = ok ? (long)__builtin_amdgcn_readlane(jgv, q) : 0;
            ar[q] = ld4(p.pq + i * p.ldpq + cql);                    // H % 4 == 0, ldpq % 4 == 0
            ar2[q] = ld4(p.pq + jj * p.ldpq + p.H + cql);
        }
    };
    auto commit_row = [&](int q, float* As, int rv) {
        if (!a_on) return;
        float4 v = ar[q];
        if (AMODE == A_GATHER) {
            v.x = fmaxf(v.x + ar2[q].x, 0.f); v.y = fmaxf(v.y + ar2[q].y, 0.f);
            v.z = fmaxf(v.z + ar2[q].z, 0.f); v.w = fmaxf(v.w + ar2[q].w, 0.f);
        }
        if (!(col_ok && wave + 4 * q < rv)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        else if (AMODE == A_DENSE && p.K - cq < 4) {                 // ragged last quad of a dense row
            const int nv = p.K - cq;
            if (nv < 2) v.y = 0.f;
            if (nv < 3) v.z = 0.f;
            v.w = 0.f;
        }
        st4(&As[(wave + 4 * q) * LDT + cq], v);
    };

    // E_EDGE_FWD statistics live in registers: lane (j, g==0) owns column 16*n + j of the wave's own N-tiles
    double stA_s[AQ], stA_q[AQ], stB_s[BQ > 0 ? BQ : 1], stB_q[BQ > 0 ? BQ : 1];
#pragma unroll
    for (int i = 0; i < AQ; ++i) { stA_s[i] = 0.0; stA_q[i] = 0.0; }
#pragma unroll
    for (int b = 0; b < (BQ > 0 ? BQ : 1); ++b) { stB_s[b] = 0.0; stB_q[b] = 0.0; }
    double dummy_s = 0.0, dummy_q = 0.0;

    int tile = blockIdx.x;
    int buf = 0;
    int jg0 = load_jg(tile);                            // neighbour rows of the tile being computed
    int jg1 = load_jg(tile + gridDim.x);                // ... of the tile being staged
    if (tile < p.num_tiles) {
        const int rv = (int)((p.M - (long)tile * p.R < p.R) ? (p.M - (long)tile * p.R) : p.R);
#pragma unroll
        for (int q = 0; q < RQ; ++q) fetch_row(q, tile, rv, jg0);
#pragma unroll
        for (int q = 0; q < RQ; ++q) commit_row(q, smem, rv);
    }
    __syncthreads();

    for (; tile < p.num_tiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        const bool has_next = next < p.num_tiles;
        const int nrv = has_next ? (int)((p.M - (long)next * p.R < p.R) ? (p.M - (long)next * p.R) : p.R) : 0;
        const float* As = smem + buf * (RG_BM * LDT);
        float* An = smem + (buf ^ 1) * (RG_BM * LDT);   // free: epilogue(tile-1) finished before this point
        const int jg2 = load_jg(next + gridDim.x);      // two tiles ahead, consumed next iteration

        f32x4 accA[4][AQ];
        f32x4 accB[BQ > 0 ? BQ : 1];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < AQ; ++i) accA[mt][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < (BQ > 0 ? BQ : 1); ++b) accB[b] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) {
            float a[4][4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float4 t4 = (p.dbg & 4) ? make_float4(1.f, 2.f, 3.f, 4.f)
                                              : ld4(&As[(16 * mt + j) * LDT + 16 * kc + 4 * g]);
                a[mt][0] = t4.x; a[mt][1] = t4.y; a[mt][2] = t4.z; a[mt][3] = t4.w;
            }
            float aw[4] = {0.f, 0.f, 0.f, 0.f};
            if (BQ > 0) {
                const float4 t4 = ld4(&As[(16 * wave + j) * LDT + 16 * kc + 4 * g]);
                aw[0] = t4.x; aw[1] = t4.y; aw[2] = t4.z; aw[3] = t4.w;
            }
            // staging ring for the NEXT tile: two rows fetched per chunk in chunks 0..7, committed CDEL chunks later
            if (kc < 8 && !(p.dbg & 1)) { fetch_row(2 * kc, next, nrv, jg1); fetch_row(2 * kc + 1, next, nrv, jg1); }
            if (kc >= CDEL && kc < 8 + CDEL && !(p.dbg & 1)) {
                commit_row(2 * (kc - CDEL), An, nrv);
                commit_row(2 * (kc - CDEL) + 1, An, nrv);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int i = 0; i < AQ; ++i) {
                    const float bv = (t == 0) ? wA[i][kc].x : (t == 1) ? wA[i][kc].y : (t == 2) ? wA[i][kc].z
                                                                                               : wA[i][kc].w;
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        accA[mt][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][t], bv, accA[mt][i], 0, 0, 0);
                }
#pragma unroll
                for (int b = 0; b < BQ; ++b) {
                    const float bv = (t == 0) ? wB[b][kc].x : (t == 1) ? wB[b][kc].y : (t == 2) ? wB[b][kc].z
                                                                                               : wB[b][kc].w;
                    accB[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[t], bv, accB[b], 0, 0, 0);
                }
            }
        }

        const long row0 = (long)tile * p.R;
        const int rv = (int)((p.M - row0 < p.R) ? (p.M - row0) : p.R);
        if (p.dbg & 2) {            // ablation: no epilogue at all (keep the accumulators alive)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int i = 0; i < AQ; ++i) asm volatile("" ::"v"(accA[mt][i]));
#pragma unroll
            for (int b = 0; b < (BQ > 0 ? BQ : 1); ++b) asm volatile("" ::"v"(accB[b]));
            if (!(p.dbg & 8)) { __syncthreads(); __syncthreads(); }
            jg0 = jg1; jg1 = jg2; buf ^= 1;
            continue;
        }

        if (EMODE == E_EDGE_FWD) {
            // bias + ReLU + BN statistics straight from the accumulators (rows >= rv masked out of the statistics):
            // <= 16 rows per lane summed in fp32, 4 lane groups combined with two xor-shuffles, fp64 across tiles
#pragma unroll
            for (int i = 0; i < AQ; ++i) {
                const int col = 16 * (AQ * wave + i) + j;
                const float bz = (p.bias && col < p.N) ? p.bias[col] : 0.f;
                float s32 = 0.f, q32 = 0.f;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = fmaxf(accA[mt][i][r] + bz, 0.f);
                        accA[mt][i][r] = v;
                        const float vm = (16 * mt + 4 * g + r < rv) ? v : 0.f;
                        s32 += vm; q32 = __builtin_fmaf(vm, vm, q32);
                    }
                s32 += __shfl_xor(s32, 16); q32 += __shfl_xor(q32, 16);
                s32 += __shfl_xor(s32, 32); q32 += __shfl_xor(q32, 32);
                stA_s[i] += (double)s32; stA_q[i] += (double)q32;
            }
#pragma unroll
            for (int b = 0; b < BQ; ++b) {
                const int col = 16 * (4 * AQ + b) + j;
                const float bz = (p.bias && col < p.N) ? p.bias[col] : 0.f;
                float s32 = 0.f, q32 = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = fmaxf(accB[b][r] + bz, 0.f);
                    accB[b][r] = v;
                    const float vm = (16 * wave + 4 * g + r < rv) ? v : 0.f;
                    s32 += vm; q32 = __builtin_fmaf(vm, vm, q32);
                }
                s32 += __shfl_xor(s32, 16); q32 += __shfl_xor(q32, 16);
                s32 += __shfl_xor(s32, 32); q32 += __shfl_xor(q32, 32);
                stB_s[b] += (double)s32; stB_q[b] += (double)q32;
            }
        }

        __syncthreads();       // every wave finished reading As[buf] (and An is fully committed)
        float* Cs = smem + buf * (RG_BM * LDT);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < AQ; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Cs[(16 * mt + 4 * g + r) * LDT + 16 * (AQ * wave + i) + j] = accA[mt][i][r];
#pragma unroll
        for (int b = 0; b < BQ; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(16 * wave + 4 * g + r) * LDT + 16 * (4 * AQ + b) + j] = accB[b][r];
        __syncthreads();

        const int c = lane << 2;
        if (EMODE == E_EDGE_FWD) {
            // Cs already holds relu(z + bias): coalesced whole-row stores + (optional) max/min over each point's rows
            if (c < p.N) {
                float4 v[RQ];
#pragma unroll
                for (int q = 0; q < RQ; ++q) v[q] = ld4(&Cs[(wave + 4 * q) * LDT + c]);
#pragma unroll
                for (int q = 0; q < RQ; ++q)
                    if (wave + 4 * q < rv) st4(p.out + (row0 + wave + 4 * q) * p.ldo + c, v[q]);
            }
            if (p.agg && tid < p.N) {
                const int pts = rv / p.k;
                const long pt0 = (long)tile * PT;
                for (int pt = 0; pt < pts; ++pt) {
                    const float* col = &Cs[(pt * p.k) * LDT + tid];
                    float vmx = col[0], vmn = col[0];
                    int imx = 0, imn = 0;
                    for (int s2 = 1; s2 < p.k; ++s2) {
                        const float v = col[s2 * LDT];
                        if (v > vmx) { vmx = v; imx = s2; }
                        if (v < vmn) { vmn = v; imn = s2; }
                    }
                    const long o = (pt0 + pt) * p.oldagg + tid;
                    p.mx[o] = vmx; p.mn[o] = vmn;
                    p.oamx[o] = (uint8_t)imx; p.oamn[o] = (uint8_t)imn;
                }
            }
        } else {
            // BN/ReLU backward: dz = (act>0) ? s*u - c1 - (act-mean)*k2 : 0 ; all activation loads issued first
            if (c < p.N) {
                const float4 cs4 = ld4(p.coef_out + c), c14 = ld4(p.coef_out + p.N + c);
                const float4 k24 = ld4(p.coef_out + 2 * p.N + c), mu4 = ld4(p.coef_out + 3 * p.N + c);
                float4 act[RQ], act2[EMODE == E_BWD_GATHER ? RQ : 1];
#pragma unroll
                for (int q = 0; q < RQ; ++q) {
                    const int r = wave + 4 * q;
                    const bool ok = r < rv;
                    if (EMODE == E_BWD_INPLACE) {
                        act[q] = ld4(p.out + (ok ? row0 + r : 0) * p.ldo + c);
                    } else {
                        const long i = ok ? (long)tile * PT + ptq[q] : 0;
                        const long jj = ok ? (long)__builtin_amdgcn_readlane(jg0, q) : 0;
                        act[q] = ld4(p.pq + i * p.ldpq + c);
                        act2[q] = ld4(p.pq + jj * p.ldpq + p.H + c);
                    }
                }
#pragma unroll
                for (int q = 0; q < RQ; ++q) {
                    const int r = wave + 4 * q;
                    float4 av = act[q];
                    if (EMODE == E_BWD_GATHER) {
                        av.x = fmaxf(av.x + act2[q].x, 0.f); av.y = fmaxf(av.y + act2[q].y, 0.f);
                        av.z = fmaxf(av.z + act2[q].z, 0.f); av.w = fmaxf(av.w + act2[q].w, 0.f);
                    }
                    const float4 u = ld4(&Cs[r * LDT + c]);
                    float4 o;
                    o.x = (av.x > 0.f) ? u.x * cs4.x - c14.x - (av.x - mu4.x) * k24.x : 0.f;
                    o.y = (av.y > 0.f) ? u.y * cs4.y - c14.y - (av.y - mu4.y) * k24.y : 0.f;
                    o.z = (av.z > 0.f) ? u.z * cs4.z - c14.z - (av.z - mu4.z) * k24.z : 0.f;
                    o.w = (av.w > 0.f) ? u.w * cs4.w - c14.w - (av.w - mu4.w) * k24.w : 0.f;
                    if (r < rv) st4(p.out + (row0 + r) * p.ldo + c, o);
                    if (EMODE == E_BWD_GATHER) st4(&Cs[r * LDT + c], (r < rv) ? o : make_float4(0.f, 0.f, 0.f, 0.f));
                }
            }
            if (EMODE == E_BWD_GATHER) {
                __syncthreads();
                if (tid < p.N) {
                    const int pts = rv / p.k;
                    const long pt0 = (long)tile * PT;
                    for (int pt = 0; pt < pts; ++pt) {
                        const float* col = &Cs[(pt * p.k) * LDT + tid];
                        float s2 = 0.f;
                        for (int t = 0; t < p.k; ++t) s2 += col[t * LDT];
                        p.dP[(pt0 + pt) * p.lddp + tid] = s2;
                    }
                }
            }
        }
        jg0 = jg1;
        jg1 = jg2;
        buf ^= 1;
    }

    if (EMODE == E_EDGE_FWD && p.stats_part) {
        // combine the per-wave register statistics through LDS in a fixed order, one partial row per workgroup
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem);          // [4 waves][2][16*NT]
        constexpr int NC = 16 * NT;
        for (int e = tid; e < 4 * 2 * NC; e += 256) red[e] = 0.0;
        __syncthreads();
        if (g == 0) {
#pragma unroll
            for (int i = 0; i < AQ; ++i) {
                const int col = 16 * (AQ * wave + i) + j;
                red[(wave * 2 + 0) * NC + col] = stA_s[i];
                red[(wave * 2 + 1) * NC + col] = stA_q[i];
            }
#pragma unroll
            for (int b = 0; b < BQ; ++b) {
                const int col = 16 * (4 * AQ + b) + j;
                red[(wave * 2 + 0) * NC + col] = stB_s[b];
                red[(wave * 2 + 1) * NC + col] = stB_q[b];
            }
        }
        __syncthreads();
        if (tid < p.N) {
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
static int eg_num_cus()
{
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

template <int AQ, int BQ, int KCH, int AMODE, int EMODE>
static int eg_launch(const RgParams& p, int stats_nblk, hipStream_t s)
{
    constexpr int NT = 4 * AQ + BQ;
    constexpr int LDA = 16 * KCH + 4, LDC = 16 * NT + 4;
    constexpr int LDT = LDA > LDC ? LDA : LDC;
    const size_t lds = (size_t)2 * RG_BM * LDT * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gpe_edgegemm_kernel<AQ, BQ, KCH, AMODE, EMODE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return GPE_ELAUNCH;
        attr_set = true;
    }
    int gx = eg_num_cus();
    if (gx > p.num_tiles) gx = p.num_tiles;
    if (stats_nblk > 0 && gx > stats_nblk) gx = stats_nblk;
    hipLaunchKernelGGL((gpe_edgegemm_kernel<AQ, BQ, KCH, AMODE, EMODE>), dim3(gx), dim3(256), lds, s, p, stats_nblk);
    GPE_CHECK_LAUNCH();
    return GPE_OK;
}

template <int AMODE, int EMODE>
static int eg_dispatch(int NT, int KCH, const RgParams& p, int stats_nblk, hipStream_t s)
{
    if (NT == 13 && KCH == 13) return eg_launch<3, 1, 13, AMODE, EMODE>(p, stats_nblk, s);
    if (NT == 13 && KCH == 10) return eg_launch<3, 1, 10, AMODE, EMODE>(p, stats_nblk, s);
    if (NT == 10 && KCH == 13) return eg_launch<2, 2, 13, AMODE, EMODE>(p, stats_nblk, s);
    if (NT == 10 && KCH == 10) return eg_launch<2, 2, 10, AMODE, EMODE>(p, stats_nblk, s);
    return GPE_EINVAL;
}

// Returns 1 and launches when the shape is on the register-stationary menu, 0 when the caller should use the generic
// LDS-streamed kernel, < 0 on a launch error.
int gpe_edgegemm_try(const RgParams& p, int amode, int emode, int stats_nblk, hipStream_t s)
{
    if (p.N <= 96 || p.N > 208 || p.K <= 96 || p.K > 208) return 0;
    if (emode != E_EDGE_FWD && (p.N & 3)) return 0;      // the backward epilogues use aligned 16-B coefficient loads
    const int NT = (p.N <= 160) ? 10 : 13;
    const int KCH = (p.K <= 160) ? 10 : 13;
    int rc = GPE_EINVAL;
    if (amode == A_GATHER && emode == E_EDGE_FWD) rc = eg_dispatch<A_GATHER, E_EDGE_FWD>(NT, KCH, p, stats_nblk, s);
    else if (amode == A_DENSE && emode == E_EDGE_FWD) rc = eg_dispatch<A_DENSE, E_EDGE_FWD>(NT, KCH, p, stats_nblk, s);
    else if (amode == A_DENSE && emode == E_BWD_INPLACE)
        rc = eg_dispatch<A_DENSE, E_BWD_INPLACE>(NT, KCH, p, stats_nblk, s);
    else if (amode == A_DENSE && emode == E_BWD_GATHER)
        rc = eg_dispatch<A_DENSE, E_BWD_GATHER>(NT, KCH, p, stats_nblk, s);
    return rc == GPE_OK ? 1 : rc;
}
