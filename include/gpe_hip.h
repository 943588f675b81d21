/*
 * libgpe_hip.so — C ABI of the MI355X (gfx950) encoder/decoder hot path.
 *
 * The reference (maria-korosteleva/Garment-Pattern-Estimation) has no FFI: its "plugin API" is class-name
 * lookup from YAML (nn/train.py:120, nn/nets.py:100,106,116).  The Python modules in
 * garment-pattern-estimation_amd/ keep that surface; underneath them every tensor op of the path is one of the
 * entry points below.  Each entry point names the reference line(s) whose arithmetic it replaces.
 *
 * Conventions (all functions):
 *   - plain C symbols; raw DEVICE pointers + dims + a hipStream_t passed as void*; the caller owns EVERY buffer — outputs,
 *     workspaces (`part`, `ws`: sizes from the gpe_*_ws* queries) and the f16x3 scale words below; the library never allocates
 *     device memory and keeps no record of tensors between calls.  Work is enqueued on `stream` of the CURRENT device and the
 *     call returns immediately.  Calls on different streams may run concurrently as long as they share no output / workspace.
 *     Process-global state is limited to (a) the arithmetic mode (gpe_math_set, gpe_f16x3_min_rows_set) and the profiling
 *     switches (gpe_debug_set), which apply to every stream and device of the process, and (b) per-device caches of the CU
 *     count and of the >64 KB LDS opt-in (hipFuncSetAttribute), keyed by device ordinal.  The mode setters are not
 *     thread-safe against concurrent launches; everything else is.
 *   - "amax word" (f16x3 mode): a uint32 in device memory holding the bit pattern of a non-negative float that is >= the
 *     largest magnitude of a tensor (a bound is as good as the value; tighter = more precise).  An entry point that writes an
 *     activation / dz tensor fills the word passed as `amax_out` (whatever kernel ran: if it could not track the maximum while
 *     storing, one extra streaming pass measures it); an entry point that reads the tensor through the fp16 pipe takes the word
 *     as `amax_a` / `amax_u` / `amax_v`.  All of them may be NULL: nothing is measured / the operand is measured in-call (edge
 *     GEMMs) or the exact fp32 kernel runs (reduce-GEMM).  The caller vouches for the words it passes: a word that is too small
 *     for its tensor overflows fp16 silently.  Modes other than f16x3 ignore `amax_a/u/v` and honour `amax_out`.
 *   - return 0 on success, -22 (EINVAL) on bad arguments, -5 (EIO) if the launch failed;
 *   - fp32 storage and arithmetic unless stated; matrix products run on v_mfma_f32_16x16x4_f32 (exact fp32);
 *     BatchNorm statistics are accumulated in fp64;
 *   - "ld" = leading dimension in elements; "packed weight" = the layout produced by gpe_pack_weight.
 */
#ifndef GPE_HIP_H
#define GPE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* version / capability probe (host only, no GPU needed) */
int gpe_abi_version(void);
/* profiling aid: ablation switches for the fused edge kernels (0 = production behaviour; results are WRONG otherwise, except
 * bit 512, which only makes gpe_edge_lazy_dz3_ok answer 0 — the eager in-place dz3 pass, same results to rounding) */
int gpe_debug_set(int flags);
int gpe_debug_get(void);
/* Multi-GPU: leave n compute units (0 .. 192; use a multiple of 8 = whole CUs per XCD, which keeps the cloud -> XCD pinning on) out
 * of every persistent launch.  The fused edge kernels, the reduce-GEMMs and the persistent LSTM size their grids to "one workgroup
 * per CU" and hold 100 - 160 KB of LDS each for the whole launch: with nothing reserved, a collective's kernel queued on another
 * stream (RCCL's all-reduce of a gradient bucket, launched from inside backward: garment-pattern-estimation_amd/parallel.py) only
 * gets onto the chip when one of them retires — the "overlap" degenerates into waiting at kernel boundaries.  Process-global like the
 * arithmetic mode; returns the previous reservation, -22 for a bad n. */
int gpe_reserve_cus_set(int n);
/* arithmetic of the fused per-edge GEMMs (gpe_edge_mlp_fwd / gpe_edge_mlp_bwd / gpe_edge_redgemm):
 *   0 = "f32"    exact fp32 matrix instruction (v_mfma_f32_16x16x4_f32)
 *   1 = "bf16x3" split-bf16: every fp32 operand x = hi + lo (two bf16), a*b ~= ah*bh + ah*bl + al*bh on the bf16 matrix
 *                pipe with fp32 accumulation (relative error ~1e-5 per product instead of ~1e-7).
 *   2 = "mixed"  split-bf16 for the per-edge ROW GEMMs (forward and the input-gradient half of backward, whose
 *                rounding errors are independent per element and average out), exact fp32 for the weight-gradient
 *                reduce-GEMM  G = dz^T (a - mean): G also feeds the BatchNorm-backward coefficients, residuals of large
 *                sums where a coherent 1e-5 product error would surface as a 1e-2 gradient error.
 *   3 = "bf16x6" THREE-term split x = h + m + l, six bf16 MFMAs per product (every term down to 2^-24 except m*l, l*m,
 *                l*l) for the row GEMMs whose shape fits the register file (10 output tiles; the others stay on the exact
 *                fp32 instruction); reduce-GEMM exact fp32.  PARITY-GRADE (meets the exact mode's bars in every test once the
 *                oracle stands on the build's ReLU / argmax decisions), measured 0.5 % faster than mode 0.
 *   4 = "f16x3"  TWO-term fp16 split of operands normalised per TENSOR by a power of two (largest magnitude -> [2^14, 2^15)),
 *                three fp16 MFMAs per product, fp32 accumulate: 23 mantissa bits per product, PARITY-GRADE (every test at the
 *                exact mode's bars; profiles/r03_i_f16x3_grad_errors.md).  All four shapes of the shipped edge MLPs stay
 *                resident (two weight planes); the weight-gradient reduce-GEMMs join when the caller passes both operand words.
 *                The scales come from caller-owned amax words (see Conventions): the packed weight is measured in-call (one
 *                workgroup), activations / dz tensors by the kernel that stores them, the gathered operand by a bound over the
 *                [P|Q] table (gpe_edge_pq_amax, or in-call).  Launches with fewer than gpe_f16x3_min_rows() rows run the exact
 *                kernels (the scale passes cost more than the fp16 pipe saves there).  The mode bench.py times.
 *                Mode 4 also (since round 4): the FORWARD gate products of the recurrences (gpe_rnn_seq_fwd with plane packs:
 *                state rows scaled by 2^12 — start states must stay below 16 in magnitude, the Python side checks) and, where
 *                the caller asks for it (out_half of gpe_edge_mlp_fwd), fp16 STORAGE of the aggregated block's activation:
 *                rows clamped at 65504, positives below 6e-8 flushed — the forward still reports the activation's true
 *                largest magnitude in amax_out, and a caller must not keep the fp16 copy when that word reaches 65504
 *                (gpe_amd.ops.set_half_act_guard does this on the Python side).  For k = 16 at the shipped widths (200 / 200 /
 *                150) the four edge launches run the two-waves-per-SIMD kernel (gpe_edgegemm_w8_kernel.h), else the
 *                single-role one.
 * Returns the previous mode, or -22 for an unknown one.  The kNN result is exact in every mode (its FILTER runs on fp16 planes
 * in every mode, followed by an exact fp32 recheck); BatchNorm statistics, the backward recurrences and every elementwise op
 * are fp32 (fp64 for reductions) in every mode. */
int gpe_math_set(int mode);
int gpe_math_get(void);
/* f16x3 size gate (part of the arithmetic mode): edge launches with fewer rows use the exact fp32 kernels.  Default 32768 (round 5: with the two-waves-per-SIMD kernels BASELINE cfg 1 — 40 960 edges — is faster on the fp16 pipe: kernel time 3.56 -> 3.48 ms per step; it was 65536 before);
 * the setter returns the previous value (tests set 0 to force the fp16-pipe kernels on small fixtures). */
long gpe_f16x3_min_rows(void);
long gpe_f16x3_min_rows_set(long rows);

/* ---- kNN graph: torch_cluster.knn as called by DynamicEdgeConv (nn/net_blocks.py:127-135,174) ------------
 * x [B][N][ldx>=C]; idx [B][N][k] int32, LOCAL to the cloud, ascending (dist, index); self included.
 * dist = fp32 fma chain over channels of (x_c - y_c)^2, ties -> lower index (same rules as oracle/knn_ref.c).
 * Limits: 1 <= k <= min(64, N) (the k-list of a query lives one entry per lane of its wavefront; torch_cluster's own device
 * kernel stops at k = 100, the reference's configurations use k = 5 .. 20), B*N*k < 2^31; anything else returns -22. */
long gpe_knn_ws_bytes(int B, int N, int C, int k);
int gpe_knn(const float* x, int B, int N, int C, int ldx, int k, int32_t* idx, int32_t* idx_glob, const int32_t* order_in,
            int32_t* order_out, void* ws, long ws_bytes, void* stream);
/* order_out (may be NULL) [B][N] int32: a LOCALITY ORDER of every cloud's points — position r of the order holds point
 * order_out[b][r].  The xyz search (C = 3) writes the Morton-curve order its sorted scan works in; every other path writes the
 * identity.  order_in (may be NULL): any permutation per cloud, taken as a hint by the matrix-pipe filter (16 <= C <= 256): it
 * lays its fp16 planes out in that order and starts every query tile's scan one tile before the queries' own tile — when the
 * order is spatially coherent (the second EdgeConv layer hands over the first layer's curve order: points that are close in
 * space are close in feature space) the first three tiles hold almost all true neighbours, the insertion bound is tight for
 * the rest of the scan and the ordered-list work of the filter drops (cfg 2, layer 2: 815 -> 699 us per search).  The RESULT
 * does not depend on the hint: idx is in point numbering, exact, for any permutation (tests/test_gpu_kernels.py). */
/* idx_glob (may be NULL) [B][N][k] = b*N + idx: the GLOBAL row of each neighbour, which is what the gather kernels
 * below take as `jg` (B*N*k must be < 2^31).  ws: gpe_knn_ws_bytes(B, N, C, k) bytes, 16-B aligned (squared norms, per-cloud
 * maxima, candidate lists of the matrix-pipe filter / of a split candidate range; for C = 3 the spatially sorted copy of the
 * clouds and its tile boxes); NULL or too small: the all-exact all-pairs kernel without candidate split runs (same result, slower).
 * Paths (all produce the identical, defined result): C = 3 and 128 <= N <= 8192 — the cloud sorted along a Morton curve and
 * scanned tile by tile with an exact box bound that prunes tiles (gpe_knn3.hip; GPE_KNN_SORTED=0 disables); 16 <= C <= 256 and
 * k <= 48 — fp16-pipe filter + exact recheck; otherwise the all-pairs VALU kernel. */

/* reverse adjacency of the kNN graph (needed by the gather's backward = scatter-add into x_j rows):
 * rev_off [B][N+1] int32 (local offsets), rev_edge [B][N*k] int32 = local edge ids (i*k+s) sorted ascending
 * inside each bucket, so the pull-style accumulation order is deterministic. */
int gpe_knn_reverse(const int32_t* idx, int B, int N, int k, int32_t* rev_off, int32_t* rev_edge, void* stream);

/* ---- weight packing (+ BatchNorm folding) ------------------------------------------------------------------
 * w [N][ldw] row-major (nn.Linear layout, nn/net_blocks.py:45) or, if transpose != 0, w is [K][ldw] and the packed
 * operand is its transpose.  Optional col_scale[K]: packed(n,k) = w(n,k)*col_scale[k]  (folds the previous
 * BatchNorm's scale s=gamma*rstd into this Linear).  wp holds gpe_packed_size(N,K) floats: K in whole 16-k chunks, and a K in
 * (96, 208] filled up with zeros to the 10 or 13 chunks the register-stationary edge kernels keep resident. */
long gpe_packed_size(int N, int K);
int gpe_pack_weight(const float* w, int ldw, int N, int K, int transpose, const float* col_scale,
                    float* wp, void* stream);
/* every weight-derived operand of a model refreshed by ONE launch (after an optimizer step): `jobs_dev` is a device array
 * of njobs 64-byte records {const float* w, w2; float* out; long total, first_block; int ldw, N, K, kind, Npad, aux}
 * sorted by first_block; a job occupies gpe_pack_job_blocks(kind, total, Npad, K) 256-thread blocks (ABI version 6: row-major sources —
 * kinds 0, 2, 8 — are packed in 64 x 64 tiles, kind 9 reads 4096 elements per block, everything else writes 1024 outputs per block;
 * versions 3 - 5: 1024 outputs per block throughout).  kind 0 plain pack, 1 transposed pack, 2 gate-interleaved pack (aux = H),
 * 3 pack of [W1a-W1b ; W1b] from W1 [H][2C] (gpe_w1_split + pack; aux = H), 4 its transpose, 5 out = w + w2 (N floats),
 * 6 out = [w[0:aux] | 0] (N floats), 7 out = w + [w2[0:aux] | 0] (N floats; GRU input-side bias b_ih + [b_hr | b_hz | 0]).
 * f16x3 operands of the recurrences (ABI version 4): kind 9 = largest |w| of the [N][K] matrix, atomicMax into the uint32 word at
 * `out` (zeroed by the caller; total = N*K); kind 8 / 10 = two-term fp16 PLANES of the gate-interleaved pack (element map of
 * kind 2) / of the plain transpose (kind 1) in the B-fragment order of v_mfma_f32_16x16x32_f16 — out = [plane h | plane l], a
 * plane = [KP/8][Npad][8 halves], KP = K rounded up to 32; w * 2^sh = h + l with 2^sh from the amax word at `w2`
 * (total = 2 * (KP/8) * Npad * 4; gpe_packed_planes_size).  A kind-9 job must run in an EARLIER launch than the planes that
 * read its word. */
long gpe_packed_planes_size(int Npad, int K);        /* floats (4-byte units) of a kind-8 / kind-10 output */
long gpe_pack_job_blocks(int kind, long total, int Npad, int K);
int gpe_pack_multi(const void* jobs_dev, int njobs, long total_blocks, void* stream);
/* (ABI version 7) what a forward block of the edge MLP needs from the BatchNorm in front of it, in one launch instead of three:
 * wp = pack of w [N][K] with col_scale (gpe_pack_weight), bias_out[n] = bias[n] + sum_k w[n][k] t[k] (gpe_fold_bias) and, with the
 * caller's edge workspace `ws` (gpe_edge_ws_bytes; NULL = skip), the packed weight's largest magnitude in its f16x3 slot + the clears
 * the edge launch's own pass performs (`clear_word`: the amax_out word of the coming gpe_edge_mlp_fwd, or NULL) — that launch is then
 * told so through bit 1 of its `out_half` argument.  `ticket`: one uint32 the caller zeroes ONCE and reuses (left zero; not shared
 * by concurrent calls; only needed with `ws`).  Results bit-identical to the separate entry points.
 * Replaces nothing of the reference by itself: it is the BatchNorm fold of DESIGN.md 4 (nn/net_blocks.py:43-47). */
int gpe_pack_fold(const float* w, int ldw, int N, int K, const float* col_scale, const float* t, const float* bias, float* wp,
                  float* bias_out, void* ws, long ws_bytes, uint32_t* clear_word, uint32_t* ticket, void* stream);
/* folded bias: out[n] = bias[n] + sum_k w[n][k]*t[k]   (t = beta - mean*s of the previous BatchNorm) */
int gpe_fold_bias(const float* w, int ldw, int N, int K, const float* bias, const float* t, float* out,
                  void* stream);

/* ---- dense Linear: nn.Linear / LSTM gate projections (nn/net_blocks.py:45,158,373-376,397) -----------------
 * Y[r][n] = act( sum_k A[r][k]*W[n][k] + bias[n] + addend[r][n] ),  r<M, n<N.   A rows use 2-level addressing
 * (a_inner<=0: row r at a + r*a_so; else row r at a + (r/a_inner)*a_so + (r%a_inner)*a_si); same for Y and addend.
 * act: 0 none, 1 relu.  bias/addend may be NULL. */
int gpe_linear(const float* a, long a_so, long a_si, int a_inner,
               const float* wp, const float* bias,
               const float* addend, long ad_so, long ad_si, int ad_inner,
               float* y, long y_so, long y_si, int y_inner,
               int M, int N, int K, int act, void* stream);

/* reduce-GEMM: G[m][n] (+)= sum_r U[r][m]*V[r][n]  (weight gradients, nn.Linear backward), colsum[m] = sum_r U[r][m].
 * part: workspace of gpe_redgemm_ws(Mg,Ng) floats.  accumulate != 0 adds into G / colsum. */
long gpe_redgemm_ws(int Mg, int Ng);
int gpe_redgemm(const float* u, long u_so, long u_si, int u_inner,
                const float* v, long v_so, long v_si, int v_inner, const float* v_shift,
                long rows, int Mg, int Ng, float* G, int ldg, float* colsum, float* part, int accumulate,
                void* stream);
/* v_shift (may be NULL) [Ng]: V rows are centred on the fly, G = sum_r U^T (V - shift) — used with shift = BatchNorm
 * batch mean so the BN-backward covariance term is accumulated without cancellation. */

/* ---- EdgeConv block (PyG DynamicEdgeConv.message + MLP + max aggregation; nn/net_blocks.py:43-47,124-135) ---
 * Algebra used: W1.[x_i, x_j-x_i] = (W1a-W1b).x_i + W1b.x_j = P_i + Q_j, with PQ = [P|Q] [B*N][2H] produced by
 * gpe_linear on the per-point features; every BatchNorm is folded into the next Linear once its batch statistics
 * are known (training mode needs the global reduction between layers; eval mode uses running stats).           */

/* stats of a1 = relu(P_i + Q_j) over all E=B*N*k edges: part [nblk][2][H] fp64 partial (sum, sumsq); nblk returned
 * by gpe_stats_blocks().  This is the EdgeConv neighbourhood gather. */
int gpe_stats_blocks(void);
int gpe_edge_gather_stats(const float* pq, int ldpq, int H, const int32_t* jg, int B, int N, int k,
                          double* part, void* stream);

/* BatchNorm finalisation (nn.BatchNorm1d in training mode, nn/net_blocks.py:45): from fp64 partials over `count`
 * rows -> mean, biased var; scale s = gamma/sqrt(var+eps), shift t = beta - mean*s; running stats updated with
 * `momentum` (unbiased var) and num_batches_tracked += 1 when running_mean != NULL.
 * stats_out [4][C] fp32 = {mean, rstd, s, t}. */
int gpe_bn_finalize(const double* part, int nblk, int C, double count, const float* gamma, const float* beta,
                    float eps, float momentum, float* running_mean, float* running_var, int64_t* num_batches,
                    float* stats_out, void* stream);
/* eval mode: stats_out from running stats */
int gpe_bn_from_running(const float* running_mean, const float* running_var, int C, const float* gamma,
                        const float* beta, float eps, float* stats_out, void* stream);

/* fused per-edge Linear+ReLU (+BN statistics, + max/min aggregation over the k messages of each point).
 * a_mode 0: A rows = relu(P_i+Q_j) gathered through the global neighbour rows jg (layer 2 of the edge MLP);
 * a_mode 1: A rows = a_in[e][*] dense (layer 3).
 * out [E][ldo] = relu(A.Wp^T + bias');  stats_part [nblk][2][Cout] fp64 (NULL in eval mode);
 * if agg != 0: mx,mn [B*N][ldagg] = max/min over the k rows of each point, amx/amn uint8 argmax/argmin slot. */
int gpe_edge_mlp_fwd(int a_mode, const float* pq, int ldpq, const int32_t* jg, const float* a_in, int lda,
                     int B, int N, int k, int Cin, int Cout, const float* wp, const float* bias,
                     float* out, int ldo, double* stats_part,
                     int agg, float* mx, float* mn, uint8_t* amx, uint8_t* amn, int ldagg,
                     const uint32_t* amax_a, uint32_t* amax_out, void* ws, long ws_bytes, int out_half, void* stream);
/* out_half: bit 0 (below) and bit 1 = "the packed weight's f16x3 scale is already in `ws`" (written by gpe_pack_fold on the same stream,
 * same workspace, nothing of this library in between: the launch then skips its own pass over wp).
 * out_half & 1: `out` is a _Float16 [E][ldo] tensor (ldo in halves, % 4 == 0; values rounded to nearest even, clamped at 65504) — the storage of the
 * aggregated block's activation when its backward forms dz3 lazily (below): that backward only needs the ReLU side of a3 and
 * the term (a3 - mean) * k2 with a coefficient of order 1e-3, gradients move by 2e-6 / 5e-6 of their maximum
 * (profiles/r04_h_row_g_probe.txt), and the step loses 0.96 GB of traffic per layer.  Only where gpe_edge_lazy_dz3_ok(B, N, k, Cout,
 * Cin) == 1 with a_mode 1 and agg != 0; anything else returns -22.  mx / mn, the statistics and amax_out are taken from the fp32
 * values before rounding. */
/* amax_a: amax word of the A operand (a_mode 0: of relu(P_i+Q_j), e.g. from gpe_edge_pq_amax; a_mode 1: of a_in), amax_out: word
 * that receives the largest |out| — see Conventions.  ws: gpe_edge_ws_bytes(B, N, k, ldagg) bytes, 16-B aligned: the store image
 * of the straight-line kernels, the in-call f16x3 words, the per-pseudo-point rows of a k > 16 launch.  NULL / too small: the
 * variants that need none run (slower; f16x3 falls back to exact fp32). */
long gpe_edge_ws_bytes(int B, int N, int k, int ldmax);   /* ldmax: the largest ldagg / lddp the workspace will be used with */
/* amax[0] = bits of a bound of relu(P_i + Q_j) over the [rows][ldpq >= 2H] table (max P + max Q, rounded up); ws as above */
int gpe_edge_pq_amax(const float* pq, int ldpq, int H, long rows, uint32_t* amax, void* ws, long ws_bytes, void* stream);
/* amax[0] = bits of the largest |x| over [rows][cols] (row pitch ldx, 16-B aligned rows) */
int gpe_absmax(const float* x, int ldx, long rows, int cols, uint32_t* amax, void* stream);

/* layer output: y[i][c] = s[c]*(s[c]>=0 ? mx : mn)[i][c] + t[c]  (BN applied after the max, sign-aware) */
int gpe_edge_finish(const float* mx, const float* mn, int ldagg, const float* stats, long rows, int C,
                    float* y, int ldy, void* stream);

/* ---- EdgeConv backward ------------------------------------------------------------------------------------ */
/* per-channel sums for the last BN's backward from per-point data, as fp64 partials
 * part [gpe_point_sums_blocks()][2][C]: sum_i g, sum_i g*xhat_sel */
int gpe_point_sums_blocks(void);
int gpe_edge_bwd_point_sums(const float* g, int ldg, const float* mx, const float* mn, int ldagg,
                            const float* stats, long rows, int C, double* part, uint32_t* amax_sg, void* stream);
/* amax_sg (may be NULL): receives the bits of max |s_c * g_ic| (s = the BatchNorm scale in `stats`), measured while g is read —
 * the data term of gpe_edge_dz3_bound */
/* coefficient vectors for "dz = (a>0) ? s*dy - c1 - (a-mean)*k2 : 0":  coef [4][C] = {s, c1 = s*mean(dy),
 * k2 = s*rstd*mean(dy*xhat), mean} from partial sums part [nblk][2][C]; also the BatchNorm parameter gradients
 * dgamma = sum dy*xhat, dbeta = sum dy (may be NULL) */
int gpe_bn_bwd_coef(const double* part, int nblk, const float* stats, int C, double count, float* coef,
                    float* dgamma, float* dbeta, void* stream);
/* sums for an inner BN from the next layer's CENTRED weight-gradient product: G [Cn][ldG] = dz_next^T (a - mean)
 * (gpe_edge_redgemm with v_shift = mean), db [Cn] = colsum dz_next, w_next [Cn][ldw] the UNFOLDED next Linear:
 *   sums[0][c] = sum_f w[f][c]*db[f];   sums[1][c] = rstd_c * sum_f w[f][c]*G[f][c];
 * also emits the true weight gradient dW_next[f][c] = G[f][c]*s_c + db[f]*beta_c into dw (ld lddw). */
int gpe_bn_bwd_from_G(const float* G, int ldG, const float* db, const float* w_next, int ldw, int Cn, int C,
                      const float* stats, double* sums, float* dw, int lddw, void* stream);

/* dz3 = (a3>0) ? [slot==argsel]*s*g - k1 - a3*k2 : 0 written IN PLACE over the stored activation a3 [E][lda3]
 * (BN-after-max backward + ReLU backward of the last edge-MLP block; g [B*N][ldg] is the layer-output gradient,
 * amx/amn the argmax/argmin slots saved by gpe_edge_mlp_fwd, coef [4][F] from gpe_bn_bwd_coef). */
int gpe_edge_dz3(float* a3, int lda3, const float* g, int ldg, const uint8_t* amx, const uint8_t* amn, int ldagg,
                 const float* coef, int B, int N, int k, int F, uint32_t* amax_out, void* stream);

/* reduce-GEMM over the E edges: G[Mg][Ng] = sum_e U[e][:]^T (V[e][:] - v_shift), colsum[Mg] = sum_e U[e][:];
 * U dense [E][ldu]; v_mode 0: V rows = relu(P_i+Q_j) gathered through jg (Ng = H) ; v_mode 1: V rows dense [E][ldv];
 * v_shift [Ng] may be NULL */
int gpe_edge_redgemm(const float* u, int ldu, int v_mode, const float* v, int ldv, const float* pq, int ldpq,
                     const int32_t* jg, const float* v_shift, int B, int N, int k, int Mg, int Ng, float* G, int ldG,
                     float* colsum, float* part, const uint32_t* amax_u, const uint32_t* amax_v, void* ws, long ws_bytes,
                     const float* lz_g, int lz_ldg, const uint8_t* lz_amx, const uint8_t* lz_amn, int lz_ldagg,
                     const float* lz_coef, void* stream);
/* amax_u / amax_v: amax words of U and of V (v_mode 0: of relu(P_i+Q_j); NULL there = the bound passes run in `ws`).  f16x3 mode
 * runs the fp16-pipe kernel only when both are known, else the exact fp32 kernel. */

/* propagate + BN/ReLU backward:  u = A.Wp^T ; dz = (act>0) ? s*u - c1 - (act-mean)*k2 : 0 ; written to dz_out
 * (A dense [E][lda]; coef_out [4][Cout] = {s,c1,k2,mean} of the BatchNorm being crossed; Wp = packed TRANSPOSE of the
 * unfolded Linear).
 * act_mode 0: act = dz_out's previous contents (in place over the stored activation);
 * act_mode 1: act = relu(P_i+Q_j) gathered, and dP[i] = sum_s dz[(i,s)] is also written (ld lddp).
 * Aliasing: act_mode 1 may run IN PLACE (dz_out == a, ldo == lda); any other overlap of dz_out with a, pq or dP — and, in
 * gpe_edge_mlp_fwd, of out with a_in, pq, mx or mn — is refused with -22 (several kernels store rows through a buffer descriptor
 * and load through plain pointers: aliased, the stores would lose their order against the loads). */
int gpe_edge_mlp_bwd(const float* a, int lda, int act_mode, const float* pq, int ldpq, const int32_t* jg,
                     int B, int N, int k, int Cin, int Cout, const float* wp, const float* coef_out,
                     float* dz_out, int ldo, float* dP, int lddp, const uint32_t* amax_a, uint32_t* amax_out, void* ws,
                     long ws_bytes, const float* lz_g, int lz_ldg, const uint8_t* lz_amx, const uint8_t* lz_amn, int lz_ldagg,
                     const float* lz_coef, void* stream);
/* ---- lazy dz3 (ABI version 4): gpe_edge_dz3's in-place pass folded into its two consumers ---------------------------------
 * With lz_g != NULL, gpe_edge_mlp_bwd (act_mode 0: `a`) and gpe_edge_redgemm (v_mode 1: `u`) take the STORED ACTIVATION a3 of the
 * block under the max aggregation — as the _Float16 [E][lda / ldu] tensor gpe_edge_mlp_fwd stored with out_half = 1 (pitch in
 * halves, % 4 == 0, rows 8-B aligned) — instead of dz3 and form dz3 = (a3>0) ? [slot==argsel]*s*g - c1 - (a3-mean)*k2 : 0 while staging
 * it: lz_g [B*N][lz_ldg] the layer-output gradient as it arrives (any pitch >= F, no alignment requirement), lz_amx / lz_amn
 * [B*N][lz_ldagg] the slots saved by gpe_edge_mlp_fwd, lz_coef [4][F] from gpe_bn_bwd_coef.  a3 is not modified.  Only where
 * gpe_edge_lazy_dz3_ok(...) == 1 (f16x3 arithmetic, k = 16, widths on the two-plane kernels' menu, above the size gate); amax_a /
 * amax_u must then be a bound of |dz3| (gpe_edge_dz3_bound) and amax_v the word of V.  Anything else returns -22. */
int gpe_edge_lazy_dz3_ok(int B, int N, int k, int F, int Cprev);
/* amax_out = bits of a bound of |dz3| = max |s*g| (amax_sg, from gpe_edge_bwd_point_sums) + max_c (|c1| + (amax(a3) + |mean|) |k2|)
 * with coef [4][F] from gpe_bn_bwd_coef and amax_a3 the word of the stored activation (gpe_edge_mlp_fwd's amax_out) */
int gpe_edge_dz3_bound(const uint32_t* amax_sg, const float* coef, int F, const uint32_t* amax_a3, uint32_t* amax_out, void* stream);

/* dQ[j] = sum over incoming edges e of dz1[e]  (deterministic pull through the reverse adjacency) */
int gpe_edge_pull_dq(const float* dz, int lddz, const int32_t* rev_off, const int32_t* rev_edge,
                     int B, int N, int k, int H, float* dQ, int lddq, void* stream);

/* ---- pooling (torch_geometric global_mean_pool, nn/net_blocks.py:148,184) --------------------------------- */
int gpe_segment_mean_fwd(const float* x, int ldx, int B, int N, int C, float* y, int ldy, void* stream);
int gpe_segment_mean_bwd(const float* gy, int ldgy, int B, int N, int C, float* gx, int ldgx, int accumulate,
                         void* stream);

/* ---- LSTM cell pointwise (nn.LSTM, gate order i,f,g,o; nn/net_blocks.py:373,393) --------------------------- */
/* gates_pre [Bn][4H] (pre-activation, overwritten with activated gates), c_prev [Bn][H] (row stride ldc_prev),
 * writes c [Bn][H] and h rows (2-level addressing via h + b*h_stride). */
int gpe_lstm_cell_fwd(float* gates, const float* c_prev, long ldc_prev, float* c, float* h, long h_stride,
                      int Bn, int H, void* stream);
/* dh_total = dh_out + sum of the n_rec partials dh_rec[z][Bn][H] (gpe_linear_splitk); computes dgates (pre-activation grads) [Bn][4H] rows at dgates + b*dg_stride,
 * dc_prev; inputs: activated gates, c (this step), c_prev. */
int gpe_lstm_cell_bwd(const float* dh_out, long dho_stride, const float* dh_rec, int n_rec, const float* dc_next,
                      const float* gates, const float* c, const float* c_prev, long ldc_prev,
                      float* dgates, long dg_stride, float* dc_prev, int Bn, int H, void* stream);

/* fused LSTM step (nn.LSTM recurrence, gate order i,f,g,o): gates = h_prev . W_hh^T + xproj (xproj = x_t . W_ih^T + b_ih
 * + b_hh, rows at xproj + b*xp_stride), then the cell update, in ONE launch.  W_hh must be packed gate-interleaved by
 * gpe_pack_weight_gates (gpe_packed_gates_size(H, K) floats).  Writes the ACTIVATED gates [Bn][4H] (for backward),
 * c_out [Bn][H] and h rows at h_out + b*h_stride. */
long gpe_packed_gates_size(int H, int K);
int gpe_pack_weight_gates(const float* w, int ldw, int H, int K, float* wp, void* stream);
int gpe_lstm_step_fwd(const float* h_prev, long hp_stride, const float* whh_gates_packed, const float* xproj,
                      long xp_stride, const float* c_prev, long ldc_prev, float* gates, float* c_out, float* h_out,
                      long h_stride, int Bn, int H, void* stream);

/* ---- GRU decoder (nn.GRU, gate order r,z,n; GRUDecoderModule, nn/net_blocks.py:457-497) ------------------------------
 * gate-interleaved packing for G gates (G = 3 here; gpe_pack_weight_gates is the G = 4 case) */
long gpe_packed_ngates_size(int H, int G, int K);
int gpe_pack_weight_ngates(const float* w, int ldw, int H, int G, int K, float* wp, void* stream);
/* fused step: r = s(xr + W_hr.h), z = s(xz + W_hz.h), n = tanh(xn + r*(W_hn.h + b_hn)), h' = (1-z)*n + z*h in ONE launch.
 * xproj rows [Bn][3H] = x.W_ih^T + b_ih (+ b_hr, b_hz); saved [Bn][4H] = {r, z, n, W_hn.h + b_hn} (for backward). */
int gpe_gru_step_fwd(const float* h_prev, long hp_stride, const float* whh_gates_packed, const float* xproj,
                     long xp_stride, const float* bhn, float* saved, float* h_out, long h_stride, int Bn, int H,
                     void* stream);
/* pointwise backward of one step: dh = dh_out + dh_dir_next + sum of the n_rec partials dh_rec[z][Bn][H];
 * dgx [..][3H] = input-side pre-activation gradients {dr, dz, dn}, dgh = recurrent-side {dr, dz, dn*r}; dh_dir_prev = dh*z */
int gpe_gru_cell_bwd(const float* dh_out, long dho_stride, const float* dh_rec, int n_rec, const float* dh_dir_next,
                     const float* saved, const float* h_prev, long hp_stride, float* dgx, float* dgh, long dg_stride,
                     float* dh_dir_prev, int Bn, int H, void* stream);

/* ---- whole recurrences, wavefront order (gpe_rnn_wave.hip) ---------------------------------------------------------------
 * A stack of L layers over T steps (nn.LSTM gates = 4 / nn.GRU gates = 3, batch_first, no dropout) executed diagonal by
 * diagonal: all cells (l, t) with l + t = d run in one launch.  Buffers (device, fp32):
 *   hs    h history, element (l, b, slot) at hs + l*hs_sl + b*hs_sb + slot*hs_st, slot 0 = h0, slot t+1 = h_t (pitches % 4 == 0)
 *   cs    LSTM c history, (l, slot) at cs + l*cs_sl + slot*cs_st, each [Bn][H]
 *   saved activated gates per cell, (l, t) at saved + l*sv_sl + t*sv_st, each [Bn][4H] (GRU: {r, z, n, W_hn.h + b_hn})
 *   xproj0  layer-0 input projection x.W_ih^T + bias, row (b, t) at xproj0 + b*xp0_sb + t*xp0_st (xp0_st = 0: same input at
 *           every step, the decoders' case)
 * whh / wih / bias / bhn are HOST arrays of L device pointers: gate-interleaved packed W_hh_l; gate-interleaved packed W_ih_l
 * (entry 0 unused); the addend row [gates*H] of layers > 0 (LSTM b_ih + b_hh; GRU b_ih + [b_hr | b_hz | 0]; entry 0 unused);
 * GRU b_hn [H] per layer (NULL for LSTM). */
int gpe_rnn_seq_fwd(int gates, int L, int T, int Bn, int H, const float* xproj0, long xp0_sb, long xp0_st,
                    const void* const* whh, const void* const* wih, const void* const* bias, const void* const* bhn,
                    float* hs, long hs_sl, long hs_sb, long hs_st, float* cs, long cs_sl, long cs_st, float* saved,
                    long sv_sl, long sv_st, const void* const* whh_pl, const void* const* wih_pl,
                    const void* const* whh_amax, const void* const* wih_amax, void* ws, long ws_bytes, void* stream);
/* ws: gpe_rnn_seq_fwd_ws bytes of scratch (4-byte aligned; may be NULL when the query answers 0).  An LSTM stack of <= 256
 * units whose 16-row tiles fit the chip (layers x ceil(H/16) x row tiles <= compute units: the pattern decoder, 32 rows x 2
 * layers) then runs as ONE persistent launch (gpe_rnn_persist.hip): every workgroup keeps its weight slices in LDS for all T
 * steps and cells hand their state rows on through arrival counters in `ws` (zeroed in-call) instead of kernel boundaries.
 * Same arithmetic per product as the diagonal launches (K is split over four waves instead of two: results agree to fp32
 * rounding, not bit for bit). */
long gpe_rnn_seq_fwd_ws(int gates, int L, int T, int Bn, int H);
/* whh_pl / wih_pl / whh_amax / wih_amax (host arrays of L device pointers, or NULL; entry 0 of the wih arrays unused): the fp16
 * plane packs (gpe_pack_multi kind 8) of W_hh_l / W_ih_l and their amax words (kind 9).  With all of them present and the f16x3
 * arithmetic selected (gpe_math_set(4)) the gate products run on the fp16 pipe: the state rows enter scaled by 2^12 and split in
 * two fp16 terms (|h| < 1 for every state a cell produces; start states must satisfy |h0| < 16), three MFMAs per product, fp32
 * accumulate — same bars as the exact kernel in every test.  Otherwise: exact fp32. */
/* backward through the same recurrence.  dtop: gradient of the top layer's outputs, row (b, t) at dtop + b*dt_sb + t*dt_st
 * (may be NULL); d_hN / d_cN [L][Bn][H]: gradients of the final states (may be NULL).  whh_t / wih_t: host arrays of the
 * plain TRANSPOSED packs (gpe_pack_weight(.., transpose = 1)).  Outputs: dgx / dgh [L][Bn][T][ld], element (l, b, t) at
 * + l*dg_sl + b*dg_sb + t*dg_st (pitches % 4 == 0): pre-activation gradients on the input side / recurrent side (LSTM: pass
 * the same buffer twice).  part: workspace of gpe_rnn_seq_bwd_ws floats; carry: [2][L][Bn][H] scratch — on return
 * carry[0][l] holds dc_0 (LSTM) / the z-gated part of dh_0 (GRU) of layer l. */
long gpe_rnn_seq_bwd_ws(int gates, int L, int T, int Bn, int H);
int gpe_rnn_seq_bwd(int gates, int L, int T, int Bn, int H, const float* dtop, long dt_sb, long dt_st, const float* d_hN,
                    const float* d_cN, const void* const* whh_t, const void* const* wih_t, const float* hs, long hs_sl,
                    long hs_sb, long hs_st, const float* cs, long cs_sl, long cs_st, const float* saved, long sv_sl,
                    long sv_st, float* dgx, float* dgh, long dg_sl, long dg_sb, long dg_st, float* part, float* carry,
                    const void* const* whh_tpl, const void* const* wih_tpl, const void* const* whh_amax,
                    const void* const* wih_amax, void* stream);
/* whh_tpl / wih_tpl / whh_amax / wih_amax (host arrays of L device pointers, or NULL; entry 0 of the wih arrays unused): the
 * TRANSPOSED fp16 plane packs (gpe_pack_multi kind 10) of W_hh_l / W_ih_l and their amax words — used by the persistent
 * backward launch (same eligibility as the forward's) in f16x3 mode: the dG rows are normalised per wave by the largest
 * magnitude the wave loaded, three fp16 MFMAs per product, fp32 accumulate.  Absent: exact fp32. */

/* ---- attention variant (GarmentSegmentPattern3D, nn/nets.py:187-299) ------------------------------------------ */
/* sparsemax.Sparsemax(dim=1) over rows of width W <= 32 (nn/nets.py:225): forward and backward */
int gpe_sparsemax_fwd(const float* z, int ldz, long rows, int W, float* out, int ldo, void* stream);
int gpe_sparsemax_bwd(const float* out, int ldo, const float* g, int ldg, long rows, int W, float* gz, int ldgz,
                      void* stream);
/* entmax.SparsemaxLoss() on the attention weights (nn/metrics/composed_loss.py:196,323-332; third-party, un-vendored: the
 * published sparsemax Fenchel-Young loss restated): loss[0] = mean_r (1 - |p|^2)/2 + <p - e_t, x>, p = sparsemax(x_r), W <= 32;
 * gx [rows][ldg] = (p - e_t)/rows (the gradient of loss[0]); part: ceil(rows/256) doubles of scratch; bad[0] |= 1 when a
 * target lies outside [0, W) (the caller zeroes it before the call and raises after). */
int gpe_sparsemax_loss(const float* x, int ldx, const int32_t* target, long rows, int W, float* gx, int ldg, double* part,
                       float* loss, int* bad, void* stream);
/* y = s*a + t with {s,t} = stats rows 2,3: BatchNorm of a stored post-ReLU activation (last block of a dense MLP) */
int gpe_bn_apply(const float* a, int lda, const float* stats, long rows, int C, float* y, int ldy, void* stream);

/* y = s*(a*a_scale) + t*t_scale: BatchNorm of a SUM (t_scale = k, EdgeConv aggr 'add') or MEAN (a_scale = 1/k) over the k
 * messages of a point, applied after the aggregation (nn/net_blocks.py:129 with EConv_aggr != 'max') */
int gpe_bn_apply_scaled(const float* a, int lda, const float* stats, long rows, int C, float a_scale, float t_scale,
                        float* y, int ldy, void* stream);
/* out[i][c] = sum over the k messages of point i of a[i*k+s][c] */
int gpe_edge_sum_k(const float* a, int lda, long npts, int k, int F, float* out, int ldo, void* stream);
/* explicit message inputs of DynamicEdgeConv (nn/net_blocks.py:124-135: cat[x_i, x_j - x_i]) for first-block widths the fused
 * P|Q path does not take (EConv_hidden not a multiple of 4, or > 256): out [npts*k][ldo >= 2C], row e = i*k + s holds
 * [x_i | x_{jg[e]} - x_i] (pad columns zero).  bwd: gx [B*N][C] from g [B*N*k][ldg >= 2C] through the transposed graph of
 * gpe_knn_reverse (rev_off [B][N+1], rev_edge [B][N*k]) — a deterministic pull, no atomics. */
int gpe_edge_inputs_fwd(const float* x, int ldx, int C, const int32_t* jg, long npts, int k, float* out, int ldo,
                        void* stream);
int gpe_edge_inputs_bwd(const float* g, int ldg, int C, const int32_t* rev_off, const int32_t* rev_edge, int B, int N, int k,
                        float* gx, int ldgx, void* stream);
/* gpe_edge_dz3 for aggr 'add' / 'mean': every message of a point receives gscale * g[i] (no arg-slot selection) */
int gpe_edge_dz3_all(float* a3, int lda3, const float* g, int ldg, float gscale, const float* coef, int B, int N, int k,
                     int F, uint32_t* amax_out, void* stream);

/* ---- pooling variants (torch_geometric global_max_pool / global_add_pool, nn/net_blocks.py:145-150) -----------------
 * mode 1 = max (arg [B][C] int32 = the winning point, first maximum), 2 = add. */
int gpe_segment_pool_fwd(const float* x, int ldx, int B, int N, int C, int mode, float* y, int ldy, int32_t* arg,
                         void* stream);
int gpe_segment_pool_bwd(const float* gy, int ldgy, const int32_t* arg, int B, int N, int C, int mode, float* gx,
                         int ldgx, void* stream);

/* attention pooling (nn/nets.py:263-276: `w[:, p] * features -> global_pool`, for all P panels at once):
 * out[b][p][c] = pool_n w[b*N+n][p] * feat[b*N+n][c];  mode 0 mean, 1 max, 2 add (the encoder's global_pool).
 * part / part_arg: workspaces of gpe_attn_pool_ws(B,N,P,C) floats / int32 (part_arg and arg only for max). */
long gpe_attn_pool_ws(int B, int N, int P, int C);
int gpe_attn_pool_fwd(const float* w, int ldw, const float* feat, int ldf, int B, int N, int P, int C, int mode,
                      float* out, int32_t* arg, float* part, int32_t* part_arg, void* stream);
/* g [B][P][C] -> gw [B*N][ldgw], gf [B*N][ldgf] (both fully written) */
int gpe_attn_pool_bwd(const float* w, int ldw, const float* feat, int ldf, const float* g, const int32_t* arg, int B,
                      int N, int P, int C, int mode, float* gw, int ldgw, float* gf, int ldgf, void* stream);

/* ---- PointNet++ set abstraction (PointNetPlusPlus, nn/net_blocks.py:10-88: torch_geometric fps / radius / PointConv) ------
 * pos [B][N][ldp>=C] (C <= 8).  gpe_fps: farthest point sampling of M points per cloud starting at LOCAL point start[b] (device
 * array [B]; NULL = point 0.  PyG's fps(random_start=True) draws it at random: the host mirror draws from torch's generator),
 * idx [B][M] LOCAL indices in selection order (ties -> lower index).  gpe_radius: for centroid s = (b, m) the first `maxn`
 * points of cloud b, ascending index, with squared distance <= r^2: nbr [B*M][maxn] local indices, cnt [B*M]. */
int gpe_fps(const float* pos, int ldp, int B, int N, int C, int M, const int32_t* start, int32_t* idx, void* stream);
int gpe_radius(const float* pos, int ldp, const int32_t* cidx, int B, int N, int C, int M, float r, int maxn,
               int32_t* nbr, int32_t* cnt, void* stream);
/* PyG PointNetConv(add_self_loops=True) on that edge list (nn/net_blocks.py:17,24 use the default): remove_self_loops drops the
 * edge whose flat source point number equals the flat centroid number s (drop[s] = its slot in nbr[s], -1 if none) and
 * add_self_loops appends s -> s; cnt_out[s] = cnt[s] - (drop[s] >= 0) + 1. */
int gpe_pointconv_self_loops(const int32_t* nbr, const int32_t* cnt, int B, int N, int M, int maxn, int32_t* cnt_out,
                             int32_t* drop, void* stream);
/* PointConv message inputs over the COMPACT edge list (edges of centroid s at rows off[s]..off[s+1]-1, off = exclusive scan of
 * the edge counts, int64 [B*M+1]): msg[e] = [x_j (Cx values, optional) | pos_j - pos_centroid]; seg_of_row[e] = s.  drop == NULL:
 * the edges are the ball-query neighbours (off from cnt); drop != NULL: PyG's re-indexed list (off from cnt_out; neighbour slot
 * drop[s] skipped, last edge = the loop from flat point s). */
int gpe_ball_messages(const float* pos, int ldp, const float* x, int ldx, int Cx, const int32_t* cidx, const int32_t* nbr,
                      const int64_t* off, const int32_t* drop, int B, int N, int C, int M, int maxn, float* msg, int ldm,
                      int32_t* seg_of_row, void* stream);
/* max over each ragged segment of rows (PointConv aggr = 'max'): y [S][ldy], arg [S][C] = winning row or -1 (empty -> 0) */
int gpe_ragged_max_fwd(const float* x, int ldx, const int64_t* off, long S, int C, float* y, int ldy, int64_t* arg,
                       void* stream);
int gpe_ragged_max_bwd(const float* gy, int ldgy, const int64_t* off, const int64_t* arg, const int32_t* seg_of_row, long E,
                       int C, float* gx, int ldgx, void* stream);

/* ---- loss (nn/metrics/composed_loss.py:294-334 main terms; nn/metrics/losses.py:19-51 PanelLoopLoss) ----------------
 * Predictions are strided views of the decoder outputs: outlines (b,p,l,c<4) at ol + b*ol_sb + p*ol_sp + l*ol_sl + c,
 * rotations (b,p,c<R) at rot + (b*P+p)*rot_s + c, translations likewise.  Ground truth dense fp32; num_edges int32 [B*P].
 * flags: 1 shape | 2 loop | 4 rotation | 8 translation.  pad0/pad1: the standardised padding vector's first two entries.
 * fwd: part [B][4] fp64 and loop_sums [B*P][2] are workspaces (loop_sums is re-used by bwd);
 *      out5 = {total, shape, loop, rotation, translation} (each term a mean, total = shape + loop_w*loop + rot + tr).
 * bwd: gradients of `total`, times the device scalar *gscale (NULL = 1), dense: g_ol [B,P,L,4], g_rot [B,P,R], g_tr [B,P,T]. */
int gpe_pattern_loss_fwd(const float* ol, long ol_sb, long ol_sp, long ol_sl, const float* rot, long rot_s,
                         const float* tr, long tr_s, const float* gt_ol, const float* gt_rot, const float* gt_tr,
                         const int32_t* num_edges, int B, int P, int L, int R, int T, int flags, float pad0, float pad1,
                         float loop_w, double* part, float* loop_sums, float* out5, void* stream);
int gpe_pattern_loss_bwd(const float* ol, long ol_sb, long ol_sp, long ol_sl, const float* rot, long rot_s,
                         const float* tr, long tr_s, const float* gt_ol, const float* gt_rot, const float* gt_tr,
                         const int32_t* num_edges, int B, int P, int L, int R, int T, int flags, float pad0, float pad1,
                         float loop_w, const float* loop_sums, const float* gscale, float* g_ol, float* g_rot,
                         float* g_tr, void* stream);
/* panel-origin matching (composed_loss.py:656-703): gt_out [B,P,L,D] = each GT panel's edge loop cyclically shifted (first
 * num_edges rows only) to the FIRST shift with the smallest squared distance to the prediction; lead [B*P] = that shift. */
int gpe_origin_match(const float* ol, long ol_sb, long ol_sp, long ol_sl, const float* gt_ol, int D,
                     const int32_t* num_edges, int B, int P, int L, float* gt_out, int32_t* lead, void* stream);
/* greedy panel-order matching (composed_loss.py:530-570) on feature rows [B][P][D]: perm [B][P] int64 with
 * perm[b][pred panel] = gt panel; *fail set to 1 if a finite distance is left unmatched (the reference raises). */
int gpe_order_match(const float* pred_feat, const float* gt_feat, int B, int P, int D, int64_t* perm, int32_t* fail,
                    void* stream);

/* ---- stitch terms, active from `epoch_with_stitches` on (nn/metrics/composed_loss.py:336-362; PatternStitchLoss
 * nn/metrics/losses.py:54-180) --------------------------------------------------------------------------------------------
 * tags (b,p,l,d<D) at tags + b*t_sb + p*t_sp + l*t_sl + d and free-edge logits (b,p,l) at logit + b*m_sb + p*m_sp + l*m_sl
 * are strided views of the panel decoder's [B,P,L,8] output.  stitches int64 [B][2][S]: pattern-level edge ids
 * (panel * L + edge) of the two sides of every stitch, the first nums[b] (int64 [B]) of them valid; S <= 64, D <= 8.
 * gt_mask fp32 [B,P,L] (1 = free edge), gt_tags fp32 [B,P,L,D] (supervised variant only).
 * flags: 1 stitch (similarity + negative term) | 2 HardNet negative (closest other tag only, losses.py:148-180; default:
 * every other tag, :112-146) | 4 free-edge BCE-with-logits | 8 supervised tag MSE.
 * fwd: part [B][6] fp64 workspace (re-used by bwd); out5 = {total, similarity, negative, supervised, free} with
 *      total = (similarity + negative) + sup_w * supervised + free.  A pattern without stitches gives NaN (the reference
 *      divides by its zero stitch count as well).
 * bwd: gradient of `total` times the device scalar *gscale (NULL = 1), dense: g_tags [B,P,L,D], g_mask [B,P,L]. */
int gpe_stitch_loss_fwd(const float* tags, long t_sb, long t_sp, long t_sl, int D, const float* logit, long m_sb, long m_sp,
                        long m_sl, const int64_t* stitches, const int64_t* nums, int S, const float* gt_mask,
                        const float* gt_tags, int B, int P, int L, int flags, float margin, float sup_w, double* part,
                        float* out5, void* stream);
int gpe_stitch_loss_bwd(const float* tags, long t_sb, long t_sp, long t_sl, int D, const float* logit, long m_sb, long m_sp,
                        long m_sl, const int64_t* stitches, const int64_t* nums, int S, const float* gt_mask,
                        const float* gt_tags, int B, int P, int L, int flags, float margin, float sup_w, const double* part,
                        const float* gscale, float* g_tags, float* g_mask, void* stream);
/* stitched-edge re-numbering of the ground truth (composed_loss.py:592-620 after the panel-order permutation `perm` int64
 * [B][P], then :727-755 for the panel-origin shift `lead` int32 [B*P] with `num_edges` int32 [B*P] of the permuted panels);
 * either step is skipped when its pointer is NULL.  out int64 [B][2][S]; entries past nums[b] are copied. */
int gpe_stitch_renumber(const int64_t* stitches, const int64_t* nums, int B, int S, int P, int L, const int64_t* perm,
                        const int32_t* lead, const int32_t* num_edges, int64_t* out, void* stream);
/* per-edge ground truth [npanels][L][D] follows its panel's new loop origin (composed_loss.py:705-725 `_per_panel_shift`):
 * rows l < n of a panel with n >= 3 edges and lead != 0 become feat[(l + lead) mod n], everything else is copied. */
int gpe_panel_shift(const float* feat, int D, const int32_t* lead, const int32_t* num_edges, long npanels, int L, float* out,
                    void* stream);


/* ---- optimizer / input side (nn/trainer.py:162-185; nn/data/transforms.py:35-50) ----------------------------------- */
/* one torch.optim.Adam step (amsgrad off) over a flat arena of n floats (16-B aligned p, g, m, v); `step` counts from 1;
 * the gradient is read as g*gscale; zero_grad != 0 clears g afterwards.  The OneCycleLR value is passed in as lr. */
int gpe_adam_step(float* p, float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, long step, float gscale, int zero_grad, void* stream);
/* the same step with the two step-dependent scalars read from DEVICE memory — hyper[0] = lr / (1 - beta1^step),
 * hyper[1] = 1 / sqrt(1 - beta2^step), as gpe_adam_hyper writes them into a HOST pair — for a step that is replayed from a
 * captured hipGraph, whose kernel arguments are frozen (gpe_amd/graph.py; nn/trainer.py:162-185 runs Adam under OneCycleLR) */
int gpe_adam_step_dev(float* p, float* g, float* m, float* v, long n, const float* hyper, float beta1, float beta2, float eps,
                      float weight_decay, float gscale, int zero_grad, void* stream);
int gpe_adam_hyper(float lr, float beta1, float beta2, long step, float* out_host);
/* out = (x - shift) / scale per column; shift_host / scale_host are HOST arrays of C <= 8 floats */
int gpe_standardize(const float* x, long rows, int C, const float* shift_host, const float* scale_host, float* out,
                    void* stream);

/* split-K product y[z][M][N] = A[:, z*256:(z+1)*256] . W[:, z*256:...]^T for z < ceil(K/256): one K slab per workgroup
 * (latency-bound long-K shapes: the LSTM backward recurrence).  The partials are summed by the consumer. */
int gpe_linear_splitk(const float* a, long a_so, const float* wp, float* y, int M, int N, int K, void* stream);

/* ---- small helpers --------------------------------------------------------------------------------------- */
/* y[r][c] (+)= sum over inner index t of x[r][t][c]   (x rows: r*x_so + t*x_si) */
int gpe_reduce_inner(const float* x, long x_so, long x_si, int T, int R, int C, float* y, int ldy,
                     int accumulate, void* stream);
/* dW1 [H][2C] from dWpq [2H][C]:  dW1[:, :C] = dWp ; dW1[:, C:] = dWq - dWp */
int gpe_w1_grad_from_pq(const float* dwpq, int ld, int H, int C, float* dw1, int lddw1, void* stream);
/* Wpq [2H][C] from W1 [H][2C]: rows 0..H-1 = W1a - W1b, rows H..2H-1 = W1b; bias_pq = [b1 | 0] */
int gpe_w1_split(const float* w1, int ldw1, const float* b1, int H, int C, float* wpq, int ldwpq, float* bias_pq,
                 void* stream);
/* out = alpha * x (n floats; out may alias x) */
int gpe_scale(const float* x, float alpha, float* out, long n, void* stream);
/* out = alpha[0] * x, alpha a DEVICE scalar (the upstream gradient of a scalar loss term; no host read-back) */
int gpe_scale_dev(const float* x, const float* alpha, float* out, long n, void* stream);
/* out = a + b (n floats) */
int gpe_add(const float* a, const float* b, float* out, long n, void* stream);
/* inter-layer dropout of nn.LSTM / nn.GRU (nn/net_blocks.py:346,374,418-420,469): out [Bn,T,H] dense = x (strided [Bn,T,H] view:
 * element (b,t,h) at x + b*x_sb + t*x_st + h) * mask [Bn,T,H] dense (Bernoulli(1-p)/(1-p), drawn by the caller). */
int gpe_mul_rows(const float* x, long x_sb, long x_st, const float* mask, long Bn, int T, int H, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GPE_HIP_H */
