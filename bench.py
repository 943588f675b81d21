#!/usr/bin/env python3
"""bench.py — garments/sec of one training step (fwd -> loss -> bwd [-> grad all-reduce] -> Adam) of the
NeuralTailor LSTM model (GarmentFullPattern3D, reference nn/nets.py:41-184 driven as nn/trainer.py:92-99) on
synthetic point clouds, BASELINE.json config 2 per GPU:  N=2048 points, batch 32, k=16, EdgeConv encoder + LSTM
decoders.  Arithmetic of the timed region = --math, default 'f16x3': every product of the fused edge GEMMs, of their
weight-gradient reduce-GEMMs and of the forward recurrences as three fp16 MFMAs on tensor-normalised two-term splits (23 mantissa
bits), fp32 accumulate; the kNN filter runs on fp16 planes in every mode (exact after its rerank) — the parity-grade mode (every
-m gpu test holds it to the exact mode's bars).  Storage: fp32 tensors in HBM, except the aggregated block's activation a3, which
the f16x3 mode keeps in fp16 where its backward forms dz3 lazily (k = 16 above the size gate; DESIGN.md 8 row g).  The
exact-fp32-MFMA step is measured in the same run and reported as `exact_f32` and inside `config` (for every world size).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` = garments processed by all ranks / max-over-ranks wall time of exactly K
steps (barrier + synchronize on both sides).  Extra objects:
  roofline      — the dominant kernel of the step, priced as SURVEY.md 8(d) prescribes: t_roof = max(ALGORITHMIC HBM bytes / 8 TB/s,
                  EXECUTED matrix-pipe FLOPs / peak of the instruction it runs) (f16x3: 3 fp16 MFMA FLOPs per algorithmic one vs
                  2.5 PF; exact: 1 vs 157.3 TF), frac = t_roof / launch duration (HIP events on the launch stream; rocprofv3's
                  average of the same kernel beside it), `achieved` = algorithmic bytes / duration, `bound` = the larger roof.
                  `traffic` = counter bytes of the committed PMC pass of THESE kernel sources, `hbm_util` = traffic / duration / peak,
                  `refetch` = traffic / algorithmic bytes (re-fetched bytes are waste, not achievement).
  roofline_per_kernel — the same for every kernel family of the step, and `roofline_step`: whole-step HBM and pipe fractions;
  roofline_gather — the EdgeConv neighbourhood gather: bytes_gather = N*k*(C*s+4) + N*C*s + N*F*s
                  per garment and layer, for (a) the kernel that carries the layer-2 gather (the fused gather->GEMM
                  forward: its time is NOT a bandwidth measurement) and (b) the stand-alone gather + BN
                  statistics pass (bandwidth-bound), each next to the counter-derived HBM bytes of the committed PMC pass;
  cpu_baseline  — the CPU oracle (oracle/ref_path.py, kind "port") timed on this box's host cores on a bounded sample
                  (the benchmarked shape at the benchmarked batch, and the reference's own cfg-1 shape).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch  # noqa: E402

PEAK_F32_TFLOPS = 157.3     # MI355X fp32 vector = fp32-input MFMA peak (MI355X_MICROARCH.md)
PEAK_F16_TFLOPS = 2516.6    # dense fp16 / bf16 MFMA peak (256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz; no sparsity)
PEAK_HBM_GBS = 8000.0
PEAK_L2_GBS = 34500.0        # aggregate of the eight 4 MiB XCD L2s (MI355X_MICROARCH.md, 'L2 (per XCD)')


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='garments per GPU')
    ap.add_argument('--points', type=int, default=2048)
    ap.add_argument('--k', type=int, default=16)
    ap.add_argument('--model', choices=['lstm', 'att'], default='lstm',
                    help="'lstm' = GarmentFullPattern3D (BASELINE cfg 1/2/3/5, the default line); 'att' = GarmentSegmentPattern3D "
                         "(cfg 4: --model att --points 4096 --k 20) — an extra measurement, never the default")
    ap.add_argument('--epoch', type=int, default=0,
                    help='epoch handed to the loss: 0 (default, SURVEY.md 8d) = the four main terms; >= 40 = the shipped YAML\'s '
                         'stitch + free-class terms are active as well (synthetic stitches) — an extra measurement')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=0, help='batch of the CPU baseline sample; 0 = the benchmarked batch (capped at 32)')
    ap.add_argument('--cpu-steps', type=int, default=3)
    ap.add_argument('--cpu-threads', type=int, default=0, help='0 = all host cores (nproc)')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--math', choices=['f32', 'f16x3', 'bf16x6', 'mixed', 'bf16x3'], default='f16x3',
                    help="arithmetic of the fused edge GEMMs for the timed region (gpe_math_set); 'f32' = exact")
    ap.add_argument('--no-fast-math-line', action='store_true', help='skip the extra mixed / bf16x3 measurements')
    ap.add_argument('--call-shapes', default=None, metavar='FILE', help='write per-(entry, int args) launch counts and mean durations')
    ap.add_argument('--edge-dbg', type=int, default=int(os.environ.get('GPE_EDGE_DBG', '0')), help='measurement aid: gpe_debug_set flags for the edge kernels (0 = product path)')
    ap.add_argument('--graph', action='store_true',
                    help='the timed steps replay ONE captured hipGraph of the step (gpe_amd/graph.py StepGraph: same launches, no Python '
                         'between them); single GPU only.  The per-kernel timing afterwards runs eager steps')
    ap.add_argument('--f16x3-min-rows', type=int, default=-1, help='measurement aid: override the f16x3 size gate (gpe_f16x3_min_rows_set); -1 = library default')
    ap.add_argument('--torch-adam', action='store_true', help='torch.optim.Adam instead of the fused arena optimizer')
    ap.add_argument('--reserve-cus', type=int, default=-1, help='compute units left out of every persistent launch (room for RCCL under the '
                    'edge kernels); -1 = the package default: 16 when N > 1, else 0')
    return ap.parse_args()


def parse_defaults():
    import sys
    argv, sys.argv = sys.argv, sys.argv[:1]
    try:
        return parse()
    finally:
        sys.argv = argv


def synthetic(B, N, data_config, seed, device):
    g = torch.Generator().manual_seed(seed)
    P, Lp = data_config['max_pattern_len'], data_config['max_panel_len']
    feats = torch.randn(B, N, 3, generator=g)
    gt = {'outlines': torch.randn(B, P, Lp, 4, generator=g), 'rotations': torch.randn(B, P, 4, generator=g),
          'translations': torch.randn(B, P, 3, generator=g),
          'num_edges': torch.randint(3, Lp + 1, (B, P), generator=g)}
    # ground truth of the stitch terms (only read when --epoch >= epoch_with_stitches): S stitches per pattern between distinct
    # real edges (every panel has >= 3), the free-edge mask derived from them as the dataset does
    S = data_config['max_num_stitches']
    st = torch.zeros(B, 2, S, dtype=torch.long)
    free = torch.ones(B, P, Lp, dtype=torch.bool)
    for b in range(B):
        pick = torch.randperm(P * 3, generator=g)[:2 * S]            # edges 0..2 of every panel exist
        ids = (pick // 3) * Lp + pick % 3
        st[b, 0], st[b, 1] = ids[:S], ids[S:]
        free[b].view(-1)[ids] = False
    gt.update(stitches=st, num_stitches=torch.full((B,), S, dtype=torch.long), free_edges_mask=free)
    return feats.to(device), {k: v.to(device) for k, v in gt.items()}


# algorithmic work per C-ABI call, from its integer arguments (see include/gpe_hip.h for the argument order; `a` holds the
# int / float arguments of the call in order — pointers are not recorded)
def call_work(name, a):
    """-> (flops, bytes) for one launch: algorithmic FLOPs of its matrix product and the bytes it has to move once (fp32)"""
    if name == 'gpe_edge_mlp_fwd':          # a_mode, ldpq, lda, B, N, k, Cin, Cout, ...
        B, N, k, Cin, Cout = a[3], a[4], a[5], a[6], a[7]
        E = float(B) * N * k
        so = 2 if (a[-1] & 1) else 4        # last int = out_half (bit 0): the aggregated block's activation stored in fp16 (row g)
        by = E * Cout * so + (B * N * 2.0 * Cin * 4 + E * 4 if a[0] == 0 else E * Cin * 4)
        return 2.0 * E * Cin * Cout, by
    if name == 'gpe_edge_mlp_bwd':          # lda, act_mode, ldpq, B, N, k, Cin, Cout, ldo, lddp, ws_bytes, lz_ldg, lz_ldagg
        B, N, k, Cin, Cout = a[3], a[4], a[5], a[6], a[7]
        E = float(B) * N * k
        si = 2 if a[-1] > 0 else 4          # lazy dz3: the A operand is the fp16 activation
        by = E * Cin * si + E * Cout * 4 + (B * N * 2.0 * Cout * 4 + E * 4 if a[1] == 1 else E * Cout * 4)
        return 2.0 * E * Cin * Cout, by
    if name == 'gpe_edge_redgemm':          # ldu, v_mode, ldv, ldpq, B, N, k, Mg, Ng, ldG, ws_bytes, lz_ldg, lz_ldagg
        B, N, k, Mg, Ng = a[4], a[5], a[6], a[7], a[8]
        E = float(B) * N * k
        su = 2 if a[-1] > 0 else 4
        return 2.0 * E * Mg * Ng, E * Mg * su + (B * N * 2.0 * Ng * 4 + E * 4 if a[1] == 0 else E * Ng * 4)
    if name == 'gpe_edge_dz3':              # lda3, ldg, ldagg, B, N, k, F : the activation read and overwritten
        B, N, k, F = a[3], a[4], a[5], a[6]
        return 0.0, 2.0 * B * N * k * F * 4
    if name == 'gpe_knn':                   # B, N, C, ldx, k : the distance form 2 N^2 C per cloud; the table once + the lists
        B, N, C, k = a[0], a[1], a[2], a[4]
        return 2.0 * B * N * N * C, float(B) * N * (C * 4 + 2 * k * 4)
    if name == 'gpe_linear':                # strides..., M, N, K, act  (last four ints)
        M, N, K = a[-4], a[-3], a[-2]
        return 2.0 * M * N * K, 4.0 * M * (N + K)
    if name == 'gpe_redgemm':               # ..., rows, Mg, Ng, ldg, accumulate
        rows, Mg, Ng = a[-5], a[-4], a[-3]
        return 2.0 * rows * Mg * Ng, 4.0 * rows * (Mg + Ng)
    if name == 'gpe_edge_gather_stats':     # ldpq, H, B, N, k : HBM side = the [P|Q] table once + the neighbour indices (the k-fold
        H, B, N, k = a[1], a[2], a[3], a[4]  # gather itself is served by the XCD L2s: priced in roofline_gather)
        return 0.0, float(B) * N * (2 * H * 4 + k * 4)
    if name == 'gpe_rnn_seq_fwd':           # gates, L, T, Bn, H, ... : cell (l, t) multiplies [Bn, H or 2H] x [., gates*H]
        G, L_, T, Bn, H = a[0], a[1], a[2], a[3], a[4]
        return 2.0 * Bn * T * G * H * H * (2 * L_ - 1), 4.0 * Bn * T * L_ * (6 * H + G * H)
    if name == 'gpe_rnn_seq_bwd':           # the same products transposed (dh = dG W_hh + dG_above W_ih)
        G, L_, T, Bn, H = a[0], a[1], a[2], a[3], a[4]
        return 2.0 * Bn * T * G * H * H * (2 * L_ - 1), 4.0 * Bn * T * L_ * (8 * H + 2 * G * H)
    if name == 'gpe_edge_pull_dq':          # lddz, B, N, k, H, lddq : dz rows read once through the reversed graph
        B, N, k, H = a[1], a[2], a[3], a[4]
        return 0.0, float(B) * N * (k * (H * 4 + 4) + H * 4)
    return 0.0, 0.0


def call_family(name, a):
    """kernel family of a launch: the entry point plus the variant its arguments select (they run different kernels)"""
    if name == 'gpe_edge_mlp_fwd':
        return name + (':gather' if a[0] == 0 else ':dense')
    if name == 'gpe_edge_mlp_bwd':
        return name + (':gather' if a[1] == 1 else ':inplace')
    if name == 'gpe_edge_redgemm':
        return name + (':gather' if a[1] == 0 else ':dense')
    if name == 'gpe_knn':
        return name + (':filter' if a[2] >= 16 else ':exact')
    return name


# HBM traffic per launch from the committed PMC passes (profiles/*_hbm_traffic.json, made by scripts/collect_profiles.sh
# from two `rocprofv3 --pmc` runs of this same command); C-ABI entry -> device kernels it launches
_TRAFFIC_KERNELS = {
    # template tails: rowgemm <NT, AMODE, EMODE>; edgegemm (paired) <.., AMODE, EMODE, MATH>; edgegemm_sr <.., AMODE, EMODE, KC, HALF>
    # split <Policy, AQ, BQ, KCH, AMODE, EMODE, K16, PSEUDO>;  AMODE 1 = gather;  EMODE 1 forward, 2 in-place backward, 3 gathered
    # backward;  redgemm_pc / _b3 <MT, NT, VMODE(, F16)>: VMODE 0 = gathered V
    # w8 (two waves per SIMD, round 5) <NT, KCH, AMODE, EMODE, AGGT, LAZY, KK>
    'gpe_edge_mlp_fwd:gather': r'gpe_edgegemm(_sr)?_kernel<\d+, \d+, \d+, 1, 1(, [-\w]+)+>$|gpe_edgegemm_split_kernel<\w+, \d+, \d+, \d+, 1, 1(, [-\w]+)+>$|gpe_edgegemm_w8_kernel<\d+, \d+, 1, 1(, [-\w]+)+>$',
    'gpe_edge_mlp_fwd:dense': r'gpe_edgegemm(_sr)?_kernel<\d+, \d+, \d+, 0, 1(, [-\w]+)+>$|gpe_edgegemm_split_kernel<\w+, \d+, \d+, \d+, 0, 1(, [-\w]+)+>$|gpe_edgegemm_w8_kernel<\d+, \d+, 0, 1(, [-\w]+)+>$',
    'gpe_edge_mlp_bwd:inplace': r'gpe_edgegemm(_sr)?_kernel<\d+, \d+, \d+, 0, 2(, [-\w]+)+>$|gpe_edgegemm_split_kernel<\w+, \d+, \d+, \d+, 0, 2(, [-\w]+)+>$|gpe_edgegemm_w8_kernel<\d+, \d+, 0, 2(, [-\w]+)+>$',
    'gpe_edge_mlp_bwd:gather': r'gpe_edgegemm(_sr)?_kernel<\d+, \d+, \d+, 0, 3(, [-\w]+)+>$|gpe_edgegemm_split_kernel<\w+, \d+, \d+, \d+, 0, 3(, [-\w]+)+>$|gpe_edgegemm_w8_kernel<\d+, \d+, 0, 3(, [-\w]+)+>$',
    'gpe_edge_redgemm:gather': r'gpe_redgemm_(pc|b3)_kernel<\d+, \d+, 0(, \w+)*>$',
    'gpe_edge_redgemm:dense': r'gpe_redgemm_(pc|b3)_kernel<\d+, \d+, 1(, \w+)*>$',
    'gpe_edge_gather_stats': r'gpe_gather_stats_kernel',
    'gpe_edge_dz3': r'gpe_dz3_kernel',
    'gpe_knn:filter': r'gpe_knn_mfma_kernel|gpe_knn_h3_kernel|gpe_knn_planes_kernel|gpe_knn_rerank_kernel|gpe_knn_norms_kernel|gpe_knn_cmax_kernel',
    'gpe_knn:exact': r'gpe_knn_kernel|gpe_knn3_sort_kernel|gpe_knn3_query_kernel',
    'gpe_edge_pull_dq': r'gpe_pull_dq_kernel',
}


def csrc_sha():
    """sha1 over the kernel sources (csrc/*.hip, *.h): a PMC summary is only quoted for the code that produced it."""
    import hashlib
    d = os.path.join(REPO, 'garment-pattern-estimation_amd', 'csrc')
    h = hashlib.sha1()
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


# the committed PMC passes are runs of the DEFAULT workload (scripts/collect_profiles.sh): any other shape / model / loss epoch /
# arithmetic quotes algorithmic bytes instead (set by main() from the arguments)
_PMC_APPLIES = [True, None]


def _pmc_doc():
    """(doc, source, why-not) of the newest profiles/*_hbm_traffic.json — only if it records the csrc hash of THIS tree
    (profiles/summarize_pmc.py writes it) and this run is the workload it was measured on; anything else is refused."""
    import glob
    if not _PMC_APPLIES[0]:
        return None, None, _PMC_APPLIES[1]
    files = sorted(glob.glob(os.path.join(REPO, 'profiles', '*_hbm_traffic.json')))
    if not files:
        return None, None, 'no profiles/*_hbm_traffic.json'
    doc = json.load(open(files[-1]))
    if doc.get('csrc_sha') != csrc_sha():
        return None, None, 'refused: profiles/%s was measured on csrc %s, this tree is %s' % (
            os.path.basename(files[-1]), doc.get('csrc_sha'), csrc_sha())
    return doc, 'profiles/' + os.path.basename(files[-1]), None


def pmc_traffic(family, launches_per_step=None):
    """HBM bytes per launch of the C-ABI entry `family` (all device kernels it runs, summed), from the committed PMC pass."""
    import re
    doc, src, why = _pmc_doc()
    if doc is None or family not in _TRAFFIC_KERNELS:
        return None, why
    steps = float(doc.get('steps_profiled', 3))
    per_step = 0.0
    hit = False
    for name, v in doc['kernels'].items():
        if re.search(_TRAFFIC_KERNELS[family], name):
            per_step += v['launches'] * v['hbm_bytes'] / steps
            hit = True
    if not hit:
        return None, None
    if launches_per_step is None:                       # one device-kernel launch per entry launch: the main kernel's count
        n = max(v['launches'] for name, v in doc['kernels'].items() if re.search(_TRAFFIC_KERNELS[family], name)) / steps
    else:
        n = launches_per_step
    return per_step / n, src


def rocprof_avg_ms(family):
    """rocprofv3 --kernel-trace average duration (ms) of the main device kernel of `family`, from the newest
    profiles/*_kernel_avg.json (profiles/summarize_rocpd.py) of THIS tree's kernel sources; (None, why) otherwise."""
    import glob
    import re
    if not _PMC_APPLIES[0] or family not in _TRAFFIC_KERNELS:
        return None, _PMC_APPLIES[1]
    files = sorted(glob.glob(os.path.join(REPO, 'profiles', '*_kernel_avg.json')))
    if not files:
        return None, 'no profiles/*_kernel_avg.json'
    doc = json.load(open(files[-1]))
    if doc.get('csrc_sha') != csrc_sha():
        return None, 'refused: profiles/%s is a trace of csrc %s, this tree is %s' % (os.path.basename(files[-1]), doc.get('csrc_sha'), csrc_sha())
    hits = [v for k, v in doc['kernels'].items() if re.search(_TRAFFIC_KERNELS[family], k)]
    if not hits:
        return None, None
    best = max(hits, key=lambda v: v['total_us'])
    return best['avg_us'] * 1e-3, 'profiles/' + os.path.basename(files[-1])


def pmc_step_bytes():
    """counter HBM bytes of one whole training step (every kernel of the PMC pass), or None"""
    doc, src, _ = _pmc_doc()
    if doc is None:
        return None, None
    steps = float(doc.get('steps_profiled', 3))
    return sum(v['launches'] * v['hbm_bytes'] for v in doc['kernels'].values()) / steps, src



def _workload_name(args):
    """Names the BASELINE.json configuration the arguments amount to; anything else is labelled as a custom shape."""
    shape = (args.model, args.points, args.batch, args.k)
    named = {('lstm', 1024, 8, 5): 'BASELINE cfg 1', ('lstm', 2048, 32, 16): 'BASELINE cfg 2 (cfg 3 per GPU)',
             ('att', 4096, 32, 20): 'BASELINE cfg 4', ('lstm', 8192, 64, 16): 'BASELINE cfg 5 per-GPU share (fp32)'}
    model = ('GarmentSegmentPattern3D (attention)' if args.model == 'att'
             else 'GarmentFullPattern3D, EdgeConv encoder + LSTM decoders')
    # BASELINE cfg 2 says "bf16", cfg 5 "fp16 with fp32 accumulate": the fp16 matrix pipe with fp32 accumulate is what f16x3 runs;
    # tensors stay fp32 in HBM (DESIGN.md section 8, row g: which kernels sit near their HBM bound and what was done about it)
    arith = {'f32': 'exact fp32 MFMA arithmetic',
             'f16x3': 'fp32-grade arithmetic: edge-GEMM products as 3 fp16 MFMAs on tensor-normalised two-term splits, fp32 '
                      'accumulate (parity-grade: the exact mode\'s test bars)'}.get(getattr(args, 'math', 'f16x3'), 'APPROXIMATE arithmetic (%s)' % getattr(args, 'math', '?'))
    return '%s: %s, N=%d, batch %d/GPU, k=%d; fp32 storage, %s' % (named.get(shape, 'custom shape'), model, args.points,
                                                                  args.batch, args.k, arith)


def _cpu_name():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return ''


def cpu_baseline(args, data_config, nn_cfg):
    """The CPU oracle (pure torch restatement of the reference path; kNN by cdist+topk, a fair CPU kNN — the parity
    oracle's scalar C loop is a definition, not a baseline) on this box's host cores: all of them (nproc) unless
    --cpu-threads says otherwise, 1 warm-up + >= 3 timed steps of fwd + loss + bwd."""
    import copy
    from oracle import ref_path as O
    O.KNN_IMPL = 'torch'

    def timed(B, N, k, steps):
        cfg = copy.deepcopy(nn_cfg)
        cfg['k_neighbors'] = k
        torch.manual_seed(0)
        model = O.GarmentFullPattern3D(data_config, copy.deepcopy(cfg), copy.deepcopy(cfg['loss'])).train()
        feats, gt = O.synthetic_batch(B, N, data_config, seed=0)
        times = []
        for step in range(1 + steps):
            t0 = time.perf_counter()
            model.zero_grad(set_to_none=True)
            O.train_step(model, feats, {k_: v.clone() for k_, v in gt.items()}, epoch=0, seed=step)
            times.append(time.perf_counter() - t0)
        return sum(times[1:]) / len(times[1:])

    def timed_once(B, N, k):
        cfg = copy.deepcopy(nn_cfg)
        cfg['k_neighbors'] = k
        torch.manual_seed(0)
        model = O.GarmentFullPattern3D(data_config, copy.deepcopy(cfg), copy.deepcopy(cfg['loss'])).train()
        feats, gt = O.synthetic_batch(B, N, data_config, seed=0)
        t0 = time.perf_counter()
        O.train_step(model, feats, {k_: v.clone() for k_, v in gt.items()}, epoch=0, seed=0)
        return time.perf_counter() - t0

    # Thread count: torch's intra-op pool does not scale to a 256-thread host on this op mix, so the baseline picks its
    # thread count by MEASUREMENT inside this run, AT THE BENCHMARKED BATCH (round 4 probed a B=2 sample, which favours few
    # threads): one timed step (after one warm-up at the first setting) at 16 / 64 / nproc threads, then >= 3 timed steps at the
    # fastest.  --cpu-threads overrides.  The scan is bounded (below).
    nproc = os.cpu_count() or 1
    probe = {}
    cpu_batch = args.cpu_batch if args.cpu_batch > 0 else min(args.batch, 32)
    if args.cpu_threads > 0:
        ncores = args.cpu_threads
    else:
        # candidates in increasing order, ONE timed step each (one warm-up before the first); the scan stops as soon as a setting is
        # slower than the best so far — on the 256-thread EPYC 9575F a step takes 10 s at 16 threads, 12 s at 64 and FIVE MINUTES at
        # 256 (measured, r05_z): probing nproc blindly would cost this line ten minutes
        best = None
        for i, nt in enumerate(sorted({min(n, nproc) for n in (16, 64, min(nproc, 128))})):
            torch.set_num_threads(nt)
            t = timed(cpu_batch, args.points, args.k, 1) if i == 0 else timed_once(cpu_batch, args.points, args.k)
            probe[str(nt)] = round(t, 3)
            if best is not None and t > best:
                break
            best = t
        ncores = int(min(probe, key=probe.get))
        probe['note'] = ('s/step at B=%d, N=%d, k=%d (one timed step per setting, one warm-up before the first; the scan stops at the '
                         'first setting slower than its predecessor), measured in this run' % (cpu_batch, args.points, args.k))
    torch.set_num_threads(ncores)
    t2 = timed(cpu_batch, args.points, args.k, max(args.cpu_steps, 3))
    t1 = timed(8, 1024, 5, max(args.cpu_steps, 5))          # BASELINE cfg 1: the reference's own CPU-runnable case
    O.KNN_IMPL = 'c'
    return {'value': cpu_batch / t2, 'unit': 'garments/s', 'cores': ncores, 'kind': 'port',
            'sample': 'oracle/ref_path.py (kNN by cdist+topk) fwd+loss+bwd, B=%d N=%d k=%d fp32, 1 warm-up + %d timed '
                      'steps, %.2f s/step' % (cpu_batch, args.points, args.k, max(args.cpu_steps, 3), t2),
            'cfg1': {'value': 8 / t1, 'unit': 'garments/s',
                     'sample': 'BASELINE cfg 1 (N=1024, B=8, k=5), 1 warm-up + %d timed steps, %.3f s/step'
                               % (max(args.cpu_steps, 5), t1)},
            'cpu': _cpu_name(), 'nproc': nproc,
            'thread_probe': probe}


def roofline_tables(rec, nsteps, args, step_s, f16_rows):
    """-> (roofline of the dominant family, per-family table, whole-step table) from the recorded launches `rec` =
    [(entry, int args, start event, end event)] of `nsteps` training steps; step_s = seconds per step of the timed region."""
    # ---- roofline per kernel family: t_roof = max(HBM bytes / 8 TB/s, executed pipe FLOPs / peak(dtype)) ----
    fam = {}
    for name, ints, e0, e1 in rec:
        fl, by = call_work(name, ints)
        d = fam.setdefault(call_family(name, ints), [0, 0.0, 0.0, 0.0, name])
        d[0] += 1
        d[1] += e0.elapsed_time(e1)
        d[2] += fl
        d[3] += by
    # which matrix instruction a family's product runs on: the fused edge GEMMs follow --math (above the f16x3 size gate),
    # everything else (Linear, dense reduce-GEMMs, the kNN filter) is the exact fp32 instruction
    on_f16 = args.math == 'f16x3' and args.batch * args.points * args.k >= f16_rows
    per_kernel = {}
    for key, (n_l, ms, fl, by, entry) in fam.items():
        if not (fl or by):
            continue
        t = ms * 1e-3 / n_l
        edge = entry in ('gpe_edge_mlp_fwd', 'gpe_edge_mlp_bwd', 'gpe_edge_redgemm')
        # the kNN filter (16 <= C <= 256) runs on the fp16 pipe in EVERY arithmetic mode: it only has to stay inside a proven bound,
        # the exact recheck behind it makes the result bit-exact (csrc/gpe_knn.hip); the forward recurrences follow --math (plane
        # packs from the model's PackPlan)
        f16 = (edge and on_f16) or key == 'gpe_knn:filter' or (entry == 'gpe_rnn_seq_fwd' and args.math == 'f16x3')
        pipe_peak = PEAK_F16_TFLOPS if f16 else PEAK_F32_TFLOPS
        exec_fl = (3.0 if f16 else 1.0) * fl / n_l
        # frac = t_roof / t with t_roof from the ALGORITHMIC bytes (what the launch has to move once) and the matrix-pipe
        # FLOPs it executes; the counter traffic of the committed PMC pass is reported beside it (hbm_util = counter bytes / t / peak,
        # refetch = counter bytes / algorithmic bytes): re-fetched bytes are waste, not achievement
        traffic, tsrc = pmc_traffic(key, n_l / nsteps)
        alg = by / n_l
        t_hbm, t_pipe = alg / (PEAK_HBM_GBS * 1e9), exec_fl / (pipe_peak * 1e12)
        bound = 'hbm' if t_hbm >= t_pipe else 'mfma'
        per_kernel[key] = {
            'launches_per_step': n_l / nsteps, 'avg_launch_ms': ms / n_l, 'ms_per_step': ms / nsteps,
            'bound': bound, 'frac': max(t_hbm, t_pipe) / t,
            'algorithmic_bytes_per_launch': alg, 'hbm_GBs': alg / t / 1e9, 'hbm_frac': t_hbm / t,
            'traffic': traffic, 'traffic_source': tsrc if traffic else 'none (no PMC pass of these sources / this workload)',
            'hbm_util': (traffic / t / (PEAK_HBM_GBS * 1e9)) if traffic else None,
            'refetch': (traffic / alg) if (traffic and alg) else None,
            'flops_per_launch': fl / n_l, 'pipe': ('v_mfma_f32_16x16x32_f16 x3 per product' if f16 else 'v_mfma_f32_16x16x4_f32') if fl else None,
            'pipe_TFLOPs': exec_fl / t / 1e12 if fl else None, 'pipe_peak': pipe_peak if fl else None,
            'pipe_frac': t_pipe / t if fl else None}
    per_kernel = dict(sorted(per_kernel.items(), key=lambda kv: -kv[1]['ms_per_step']))
    # dominant KERNEL of the step = the family with the most time per step among the entries that are ONE device kernel per launch
    # (SURVEY.md 8d prices a kernel against its roof and asks rocprof's average duration of that kernel to agree).  gpe_rnn_seq_fwd /
    # _bwd (40 / 80 dependent launches behind two C-ABI calls each) and gpe_knn (norms + planes + filter + rerank) are sequences:
    # they are priced in roofline_per_kernel like everything else and named in `largest_sequence` when one of them outweighs the
    # dominant kernel.
    SEQUENCES = ('gpe_rnn_seq_fwd', 'gpe_rnn_seq_bwd', 'gpe_knn:filter', 'gpe_knn:exact', 'gpe_redgemm', 'gpe_linear')
    dom = next((k for k in per_kernel if k not in SEQUENCES), next(iter(per_kernel)))
    pk = per_kernel[dom]
    first = next(iter(per_kernel))
    if pk['bound'] == 'hbm':
        roof = {'kernel': dom, 'bound': 'hbm', 'achieved': pk['hbm_GBs'], 'peak': PEAK_HBM_GBS, 'unit': 'GB/s'}
    else:
        roof = {'kernel': dom, 'bound': 'mfma', 'achieved': pk['pipe_TFLOPs'], 'peak': pk['pipe_peak'], 'unit': 'TFLOP/s'}
    rp_ms, rp_src = rocprof_avg_ms(dom)
    roof.update(frac=pk['frac'], traffic=pk['traffic'], traffic_unit='HBM bytes/launch (PMC counters)', traffic_source=pk['traffic_source'],
                hbm_util=pk['hbm_util'], refetch=pk['refetch'],
                launches_per_step=pk['launches_per_step'], avg_launch_ms=pk['avg_launch_ms'],
                rocprof_avg_launch_ms=rp_ms, rocprof_source=rp_src,
                flops_per_launch=pk['flops_per_launch'], algorithmic_bytes_per_launch=pk['algorithmic_bytes_per_launch'],
                hbm_frac=pk['hbm_frac'], pipe=pk['pipe'], pipe_frac=pk['pipe_frac'],
                rule='SURVEY.md 8(d): frac = max(ALGORITHMIC bytes / 8 TB/s, executed matrix-pipe FLOPs / pipe peak) / launch duration '
                     '(HIP events on the launch stream; rocprof_avg_launch_ms = rocprofv3\'s average of the same kernel from the committed '
                     'trace of these sources); `achieved` = algorithmic bytes / duration; hbm_util prices the PMC counter bytes instead, '
                     'refetch = counter / algorithmic bytes; the dominant KERNEL = the single-kernel entry with the most time per step',
                largest_sequence=(None if first == dom else
                                  {'entry': first, 'ms_per_step': per_kernel[first]['ms_per_step'],
                                   'launches_per_call': 'a sequence of dependent device kernels behind one C-ABI call',
                                   'frac': per_kernel[first]['frac'], 'bound': per_kernel[first]['bound']}))
    # whole step: counter bytes / step time against HBM, executed pipe FLOPs against the two pipes
    sb, ssrc = pmc_step_bytes()
    alg_b = sum(v['algorithmic_bytes_per_launch'] * v['launches_per_step'] for v in per_kernel.values())
    t_f16 = sum(v['flops_per_launch'] * 3 * v['launches_per_step'] for v in per_kernel.values()
                if v['pipe'] and 'f16' in v['pipe']) / (PEAK_F16_TFLOPS * 1e12)
    t_f32 = sum(v['flops_per_launch'] * v['launches_per_step'] for v in per_kernel.values()
                if v['pipe'] and 'f16' not in v['pipe']) / (PEAK_F32_TFLOPS * 1e12)
    roof_step = {'ms_per_step': step_s * 1e3, 'hbm_bytes_per_step': sb, 'hbm_bytes_source': ssrc,
                 'hbm_frac': sb / step_s / (PEAK_HBM_GBS * 1e9) if sb else None,
                 'algorithmic_bytes_per_step': alg_b, 'algorithmic_hbm_frac': alg_b / step_s / (PEAK_HBM_GBS * 1e9),
                 'refetch': (sb / alg_b) if (sb and alg_b) else None,
                 'pipe_time_ms': {'fp16 MFMA (x3)': t_f16 * 1e3, 'fp32 MFMA': t_f32 * 1e3},
                 'pipe_frac': (t_f16 + t_f32) / step_s}
    return roof, per_kernel, roof_step


def main():
    args = parse()
    d = parse_defaults()
    off = [k for k in ('batch', 'points', 'k', 'model', 'epoch', 'math') if getattr(args, k) != getattr(d, k)]
    if off:
        _PMC_APPLIES[0] = False
        _PMC_APPLIES[1] = 'not quoted: the committed PMC passes measure the default workload, this run changes ' + ', '.join('--' + k for k in off)
    import gpe_amd
    from gpe_amd import _lib, configs, nets, ops, parallel
    if args.edge_dbg:
        _lib.query('gpe_debug_set', args.edge_dbg)
    if args.f16x3_min_rows >= 0:
        gpe_amd.set_f16x3_min_rows(args.f16x3_min_rows)

    rank, local, world = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit('launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)' % (args.gpus, world))
    dev = torch.device('cuda', local)
    data_config = configs.data_config()
    nn_cfg = (configs.att_model_config if args.model == 'att' else configs.lstm_model_config)(k_neighbors=args.k)

    gpe_amd.set_math(args.math)
    torch.manual_seed(0)                               # identical replicas on every rank
    model_cls = nets.GarmentSegmentPattern3D if args.model == 'att' else nets.GarmentFullPattern3D
    model = model_cls(data_config, dict(nn_cfg), dict(nn_cfg['loss'])).to(dev).train()
    model.loss.with_quality_eval = False
    from gpe_amd import optim
    if args.torch_adam:
        arena = optim.FlatArena(model, register_sink=False)
        opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    else:
        # parameters + gradients in one flat arena: backward kernels write gradients in place, the all-reduce buckets are
        # slices of it, Adam (+ zero_grad) is one launch (nn/trainer.py:162-185: Adam, lr 0.002)
        arena = optim.FlatArena(model)
        opt = optim.FusedAdam(arena, lr=2e-3)
    wrapped = parallel.DistributedHotPath(model, device_ids=[dev], arena=arena, reserve_cus=None if args.reserve_cus < 0 else args.reserve_cus)
    feats, gt = synthetic(args.batch, args.points, data_config, seed=1000 + rank, device=dev)

    def step(i):
        torch.manual_seed(i * 131 + rank)              # the decoder draws random LSTM states every forward
        preds = wrapped(feats, log_step=i, epoch=args.epoch)
        loss, _, _ = model.loss(preds, gt, epoch=args.epoch)
        loss.backward()
        wrapped.finish_gradient_sync()
        opt.step()
        if args.torch_adam:
            arena.zero_grad()
        return loss

    eager_step = step
    sg = None
    if args.graph:
        if world > 1 or args.torch_adam:
            raise SystemExit('--graph: single GPU and the fused Adam only')
        from gpe_amd import graph as gpe_graph
        sg = gpe_graph.StepGraph(lambda f, g: model.loss(wrapped(f, log_step=0, epoch=args.epoch), g, epoch=args.epoch)[0], opt, warmup=2)

        def step(i):
            torch.manual_seed(i * 131 + rank)
            return sg.step(feats, gt)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(max(args.warmup, 4) if args.graph else args.warmup):      # (graph: two eager steps, the capture, one replay)
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = loss.item()

    # ---- per-kernel durations: HIP events on the launch stream, same workload, right after the timed region ----
    roof, roof_gather, breakdown, per_kernel, roof_step = None, None, None, None, None
    rec = None
    if not args.no_kernel_timing:
        # every rank runs these extra steps (they contain the gradient all-reduce); only rank 0 records events
        nsteps = min(args.steps, 5)
        if rank == 0:
            _lib.TIMING = []
        # (one stream for these steps: a launch that shares the chip with a side-stream launch — ops.side_grads — would be charged
        # the other's time; `value` above is measured with the side stream at work, this pass prices every kernel on its own)
        side_was, ops.SIDE_GRADS = ops.SIDE_GRADS, False
        for i in range(nsteps):
            eager_step(args.warmup + args.steps + i)
        barrier()
        ops.SIDE_GRADS = side_was
        rec, _lib.TIMING = _lib.TIMING, None
    if rank == 0 and rec is not None:
        agg = {}
        for name, ints, e0, e1 in rec:
            fl, by = call_work(name, ints)
            d = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
            d[0] += 1
            d[1] += e0.elapsed_time(e1)
            d[2] += fl
            d[3] += by
        breakdown = {n: {'launches_per_step': v[0] / nsteps, 'ms_per_step': v[1] / nsteps} for n, v in
                     sorted(agg.items(), key=lambda kv: -kv[1][1])}
        if args.call_shapes:
            # per (entry point, integer arguments): launches and mean duration — which SHAPES the time goes to
            shp = {}
            for name, ints, e0, e1 in rec:
                d = shp.setdefault((name, ints), [0, 0.0])
                d[0] += 1
                d[1] += e0.elapsed_time(e1)
            with open(args.call_shapes, 'w') as f:
                for (name, ints), v in sorted(shp.items(), key=lambda kv: -kv[1][1]):
                    f.write('%-24s x%-4d %8.1f us  %s\n' % (name, v[0] // nsteps, 1e3 * v[1] / v[0], ints))
        roof, per_kernel, roof_step = roofline_tables(rec, nsteps, args, elapsed / args.steps, _lib.query('gpe_f16x3_min_rows'))
        # ---- the EdgeConv gather, SURVEY.md §8(d) row 5: bytes_gather(layer) per garment, s = 4 (fp32) -------------
        H, F = nn_cfg['EConv_hidden'], nn_cfg['EConv_feature']

        def bytes_gather(C):
            return float(args.batch) * (args.points * args.k * (C * 4 + 4) + args.points * C * 4 + args.points * F * 4)

        roof_gather = {'definition': 'bytes_gather(layer) = N*k*(C_in*s+4) + N*C_in*s + N*F*s per garment (SURVEY.md 8d), '
                                     'x batch; layer 1 C_in=3, layer 2 C_in=%d; s=4' % F,
                       'bytes_layer1': bytes_gather(3), 'bytes_layer2': bytes_gather(F), 'peak': PEAK_HBM_GBS,
                       'peak_l2': PEAK_L2_GBS, 'unit': 'GB/s', 'bound': 'hbm'}
        if 'gpe_edge_gather_stats' in agg:
            # The stand-alone gather + BN-statistics pass is the only kernel whose time IS the gather.  Two honest roofs:
            #   HBM side: counter bytes (FETCH+WRITE of the committed PMC pass of THIS csrc) / time / 8 TB/s
            #   L2 side : algorithmic k-fold bytes (what the gather touches, served by the XCD L2s) / time / 34.5 TB/s
            n_l, ms, _, _ = agg['gpe_edge_gather_stats']
            by = n_l * float(args.batch) * (args.points * args.k * (H * 4 + 4) + args.points * H * 4)   # k-fold bytes_gather (C = H)
            t = ms * 1e-3 / n_l
            traffic, tsrc = pmc_traffic('gpe_edge_gather_stats')
            l2_rate = by / n_l / t / 1e9
            hbm_rate = traffic / t / 1e9 if traffic else None
            roof_gather['stats_pass'] = {
                'kernel': 'gpe_edge_gather_stats (k-fold neighbour gather of the [P|Q] rows + fp64 BN statistics)',
                'algorithmic_bytes_per_launch': by / n_l, 'avg_launch_ms': ms / n_l,
                'l2_achieved': l2_rate, 'l2_frac': l2_rate / PEAK_L2_GBS,
                'traffic': traffic, 'traffic_source': tsrc,
                'hbm_achieved': hbm_rate, 'hbm_frac': hbm_rate / PEAK_HBM_GBS if hbm_rate else None,
                'note': 'the Q table of a cloud is pinned to one XCD L2, so the k-fold gather is an L2 workload: l2_frac prices '
                        'the algorithmic bytes against the L2 aggregate, hbm_frac the counter bytes against HBM.  Neither '
                        'reaches 0.6: north_star\'s ">= 60 % of HBM roofline on the gather" is NOT met by this pass '
                        '(0.1 ms of the step); the gather that matters is fused into the MFMA-bound forward below'}
            roof_gather['achieved'] = hbm_rate
            roof_gather['frac'] = roof_gather['stats_pass']['hbm_frac']
            roof_gather['traffic'] = traffic
        # the kernel that carries the layer-2 gather into the matrix pipe (gather -> GEMM -> ReLU -> stats, fused)
        gl = [(ints, e0.elapsed_time(e1)) for name, ints, e0, e1 in rec if name == 'gpe_edge_mlp_fwd' and ints[0] == 0]
        if gl:
            ms = sum(t for _, t in gl) / len(gl)
            by = float(args.batch) * (args.points * args.k * (H * 4 + 4) + args.points * H * 4)
            roof_gather['fused_forward'] = {
                'kernel': 'gpe_edge_mlp_fwd a_mode=0 (gather -> LDS tile -> MFMA): the gather hides under the GEMM; priced in '
                          'roofline_per_kernel', 'bound': per_kernel['gpe_edge_mlp_fwd:gather']['bound'], 'gather_bytes_per_launch': by, 'avg_launch_ms': ms,
                'gather_rate_GBs': by / (ms * 1e-3) / 1e9, 'l2_frac': by / (ms * 1e-3) / 1e9 / PEAK_L2_GBS,
                'traffic': pmc_traffic('gpe_edge_mlp_fwd:gather')[0]}

    # the other arithmetic modes, measured the same way on the same workload (reported beside `value`, never as `value`)
    fast = None
    if args.math in ('f32', 'f16x3') and not args.no_fast_math_line:
        fast = {}
        only = None if world == 1 else ('f32', 'f16x3')        # N > 1: the two parity-grade modes (both are first-class numbers)
        for mode, note in (('f32', 'exact fp32: every product on v_mfma_f32_16x16x4_f32 (gpe_math_set(0))'),
                           ('f16x3', 'two-term split-fp16 products (3 fp16 MFMAs per product, fp32 accumulate) on tensor-normalised '
                                     'operands in all four single-role edge kernels and, where both operand scales are known without a '
                                     'pass over the tensor, the weight-gradient reduce-GEMMs.  PARITY-GRADE: every -m gpu test runs it at the exact mode\'s bars; gpe_math_set(4)'),
                           ('bf16x6', 'three-term split-bf16 products (6 bf16 MFMAs per product) in the 10-tile single-role edge '
                                      'kernels; parity-grade (the exact mode\'s bars in every test); gpe_math_set(3)'),
                           ('mixed', 'split-bf16 row GEMMs (forward + input-gradient half) on the bf16 matrix pipe, exact-fp32 '
                                     'weight-gradient / BN-coefficient products: forward within 1e-4, parameter gradients '
                                     '1e-3 .. 1.5e-2 of max|grad| (approximate, tests TOL 3e-2); gpe_math_set(2) / GPE_MATH=mixed'),
                           ('bf16x3', 'every fused edge GEMM split-bf16: forward within 1e-4 of the reference, encoder '
                                      'gradients within ~1e-2 (tests/test_gpu_kernels.py TOL); gpe_math_set(1)')):
            if mode == args.math or (only and mode not in only):
                continue
            gpe_amd.set_math(mode)
            n_f = max(3, min(args.steps, 10))
            for i in range(2):
                eager_step(10_000 + i)
            barrier()
            t1 = time.perf_counter()
            for i in range(n_f):
                eager_step(10_002 + i)
            barrier()
            dt = time.perf_counter() - t1
            if world > 1:
                tt = torch.tensor([dt], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
                dt = tt.item()
            fast[mode] = {'value': args.batch * world * n_f / dt, 'unit': 'garments/s', 'steps': n_f,
                          'ms_per_step': dt / n_f * 1e3, 'note': note}
        gpe_amd.set_math(args.math)

    # ---- the exchange step, for N > 1: what RCCL saw and what it costs -------------------------------------------------
    exchange = None
    if world > 1:
        exchange = wrapped.measure_exchange(iters=10)      # every rank participates; rank 0 reports
        exchange['exposed_ms_per_step'] = wrapped.exposed_ms()
        exchange['reserved_cus'] = wrapped.reserved_cus

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, data_config, nn_cfg) if args.model == 'lstm' else None

    if rank == 0:
        garments = args.batch * world * args.steps
        out = {
            'metric': 'garments/sec (fwd+bwd) at N=%d pts, batch %d per GPU' % (args.points, args.batch),
            'value': garments / elapsed, 'unit': 'garments/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            # the arithmetic type of the timed region (not a storage claim): 'f16x3' = fp16 MFMA on two-term splits, fp32 accumulate
            'dtype': {'f32': 'f32', 'f16x3': 'f16x3 (fp16 MFMA x3 per product, fp32 accumulate)', 'bf16x6': 'bf16x6',
                      'mixed': 'bf16x3/f32', 'bf16x3': 'bf16x3'}[args.math],
            'math_mode': args.math, 'data': 'synthetic',
            # (the driver's parsed record keeps `config`, `roofline` and `cpu_baseline` and drops other top-level keys: what a
            # reader needs to interpret `value` is repeated here)
            'config': {'workload': _workload_name(args),
                       'global_batch': args.batch * world, 'parallelism': 'dp%d' % world,
                       'math_mode': args.math,
                       'storage': 'fp32 tensors in HBM' + ('; the aggregated block\'s activation a3 in fp16 (lazy dz3, k = 16 above the '
                                                            'size gate)' if args.math == 'f16x3' and args.k == 16 else ''),
                       # (flat scalars: the driver's parsed record drops nested objects inside `config`)
                       'exact_f32_value': fast['f32']['value'] if fast and fast.get('f32') else None,
                       'exact_f32_ms': fast['f32']['ms_per_step'] if fast and fast.get('f32') else None,
                       'loss_epoch': args.epoch, 'reserved_cus': wrapped.reserved_cus,
                       'launch': ('one captured hipGraph per step (StepGraph: %d capture(s), %d replays)' % (sg.captures, sg.replays)) if sg else 'eager (one Python call per launch)',
                       'side_stream': bool(ops.SIDE_GRADS and not sg and ops._LAST_EDGES[0] >= ops.SIDE_MIN_EDGES),
                       'step': 'fwd + ComposedPatternLoss + bwd' + (' + RCCL grad all-reduce' if world > 1 else '')
                               + (' + Adam (torch)' if args.torch_adam else ' + fused Adam (flat arena)'),
                       'final_loss': final_loss},
            'roofline': roof, 'roofline_step': roof_step, 'roofline_per_kernel': per_kernel,
            'roofline_gather': roof_gather, 'cpu_baseline': cpu,
            'exact_f32': (fast or {}).get('f32'), 'fast_math': fast,
            'dist_world': (torch.distributed.get_world_size() if world > 1 else 1),
            'allreduce_ms_per_step': exchange and exchange['ms_per_step'],
            'allreduce_bytes': exchange and exchange['bytes_per_step'], 'exchange': exchange,
            'kernel_ms_per_step': breakdown}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
