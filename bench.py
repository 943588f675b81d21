#!/usr/bin/env python3
"""bench.py — garments/sec of one training step (fwd -> loss -> bwd [-> grad all-reduce] -> Adam) of the
NeuralTailor LSTM model (GarmentFullPattern3D, reference nn/nets.py:41-184 driven as nn/trainer.py:92-99) on
synthetic point clouds, BASELINE.json config 2 per GPU:  N=2048 points, batch 32, k=16, EdgeConv encoder + LSTM
decoders.  fp32 storage, exact-fp32 MFMA (the parity-proven mode; BASELINE's "bf16" is a storage option not built
yet — see DESIGN.md).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` = garments processed by all ranks / max-over-ranks wall time of exactly K
steps (barrier + synchronize on both sides).  Extra objects:
  roofline      — the dominant kernel family of the step (MFMA-bound fp32 edge-MLP GEMMs): algorithmic FLOPs per
                  launch / average launch duration from HIP events on the launch stream, vs 157.3 TFLOP/s;
  roofline_gather — the EdgeConv neighbourhood gather (HBM/L2-bound): algorithmic bytes / duration vs 8 TB/s;
  cpu_baseline  — the CPU oracle (oracle/ref_path.py, kind "port") timed on this box's host cores on a bounded sample.
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
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='garments per GPU')
    ap.add_argument('--points', type=int, default=2048)
    ap.add_argument('--k', type=int, default=16)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=8)
    ap.add_argument('--cpu-steps', type=int, default=2)
    ap.add_argument('--cpu-threads', type=int, default=32)
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--math', choices=['f32', 'bf16x3'], default='f32',
                    help="arithmetic of the fused edge GEMMs for the timed region (gpe_math_set); 'f32' = exact")
    ap.add_argument('--no-fast-math-line', action='store_true', help='skip the extra bf16x3 measurement')
    return ap.parse_args()


def synthetic(B, N, data_config, seed, device):
    g = torch.Generator().manual_seed(seed)
    P, Lp = data_config['max_pattern_len'], data_config['max_panel_len']
    feats = torch.randn(B, N, 3, generator=g)
    gt = {'outlines': torch.randn(B, P, Lp, 4, generator=g), 'rotations': torch.randn(B, P, 4, generator=g),
          'translations': torch.randn(B, P, 3, generator=g),
          'num_edges': torch.randint(3, Lp + 1, (B, P), generator=g)}
    return feats.to(device), {k: v.to(device) for k, v in gt.items()}


# algorithmic work per C-ABI call, from its integer arguments (see include/gpe_hip.h for the argument order)
def call_work(name, a):
    """-> (flops, bytes) for one launch"""
    if name == 'gpe_edge_mlp_fwd':          # a_mode, ldpq, lda, B, N, k, Cin, Cout, ...
        B, N, k, Cin, Cout = a[3], a[4], a[5], a[6], a[7]
        return 2.0 * B * N * k * Cin * Cout, 0.0
    if name == 'gpe_edge_mlp_bwd':          # lda, act_mode, ldpq, B, N, k, Cin, Cout
        B, N, k, Cin, Cout = a[3], a[4], a[5], a[6], a[7]
        return 2.0 * B * N * k * Cin * Cout, 0.0
    if name == 'gpe_edge_redgemm':          # ldu, v_mode, ldv, ldpq, B, N, k, Mg, Ng
        B, N, k, Mg, Ng = a[4], a[5], a[6], a[7], a[8]
        return 2.0 * B * N * k * Mg * Ng, 0.0
    if name == 'gpe_linear':                # strides..., M, N, K, act  (last four ints)
        M, N, K = a[-4], a[-3], a[-2]
        return 2.0 * M * N * K, 0.0
    if name == 'gpe_redgemm':               # ..., rows, Mg, Ng, ldg, accumulate
        rows, Mg, Ng = a[-5], a[-4], a[-3]
        return 2.0 * rows * Mg * Ng, 0.0
    if name == 'gpe_edge_gather_stats':     # ldpq, H, B, N, k : compulsory bytes = PQ once + the kNN graph once
        H, B, N, k = a[1], a[2], a[3], a[4]  # (the k-fold neighbour re-reads are L2/MALL hits, not HBM work)
        return 0.0, float(B) * N * (2 * H * 4 + k * 4)
    if name == 'gpe_edge_pull_dq':          # lddz, B, N, k, H, lddq : dz rows read once through the reversed graph
        B, N, k, H = a[1], a[2], a[3], a[4]
        return 0.0, float(B) * N * (k * (H * 4 + 4) + H * 4)
    return 0.0, 0.0


# HBM traffic per launch from the committed PMC passes (profiles/*_hbm_traffic.json, made by scripts/collect_profiles.sh
# from two `rocprofv3 --pmc` runs of this same command); C-ABI entry -> device kernels it launches
_TRAFFIC_KERNELS = {
    # template tails: rowgemm <NT, AMODE, EMODE>; edgegemm (paired) <.., AMODE, EMODE, MATH>; edgegemm_sr <.., AMODE, EMODE, K16>
    'gpe_edge_mlp_fwd': r'gpe_rowgemm_kernel<.*, 1>$|gpe_edgegemm(_sr)?_kernel<.*, 1, \w+>$',
    'gpe_edge_mlp_bwd': r'gpe_rowgemm_kernel<.*, [23]>$|gpe_edgegemm(_sr)?_kernel<.*, [23], \w+>$',
    'gpe_edge_redgemm': r'gpe_redgemm_pc_kernel<',
    'gpe_edge_gather_stats': r'gpe_gather_stats_kernel',
}


def pmc_traffic(entry):
    import glob
    import re
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', '*_hbm_traffic.json')))
    if not files or entry not in _TRAFFIC_KERNELS:
        return None, None
    kern = json.load(open(files[-1]))['kernels']
    n = b = 0.0
    for name, v in kern.items():
        if re.search(_TRAFFIC_KERNELS[entry], name):
            n += v['launches']
            b += v['launches'] * v['hbm_bytes']
    return (b / n, 'profiles/' + os.path.basename(files[-1])) if n else (None, None)


def cpu_baseline(args, data_config, nn_cfg):
    """The CPU oracle (pure torch restatement of the reference path + C kNN) on this box's host cores."""
    import copy
    from oracle import ref_path as O
    # torch's intra-op pool stops scaling (and then collapses) long before a 2-socket EPYC's core count on this
    # op mix; 32 threads is the measured sweet spot class for oneDNN/MKL at these sizes
    ncores = min(os.cpu_count() or 1, args.cpu_threads)
    torch.set_num_threads(ncores)
    O.KNN_IMPL = 'torch'          # cdist+topk: a fair CPU kNN (the parity oracle's scalar C loop is a definition, not a baseline)
    torch.manual_seed(0)
    model = O.GarmentFullPattern3D(data_config, copy.deepcopy(nn_cfg), copy.deepcopy(nn_cfg['loss'])).train()
    feats, gt = O.synthetic_batch(args.cpu_batch, args.points, data_config, seed=0)
    times = []
    for step in range(1 + args.cpu_steps):
        t0 = time.perf_counter()
        model.zero_grad(set_to_none=True)
        O.train_step(model, feats, {k: v.clone() for k, v in gt.items()}, epoch=0, seed=step)
        times.append(time.perf_counter() - t0)
    t = sum(times[1:]) / len(times[1:])
    cpu_name = ''
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    cpu_name = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    O.KNN_IMPL = 'c'
    return {'value': args.cpu_batch / t, 'unit': 'garments/s', 'cores': ncores, 'kind': 'port',
            'sample': 'oracle/ref_path.py (kNN by cdist+topk) fwd+loss+bwd, B=%d N=%d k=%d fp32, 1 warm-up + %d timed steps, %.2f s/step'
                      % (args.cpu_batch, args.points, args.k, args.cpu_steps, t),
            'cpu': cpu_name}


def main():
    args = parse()
    import gpe_amd
    from gpe_amd import _lib, configs, nets, parallel

    rank, local, world = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit('launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)' % (args.gpus, world))
    dev = torch.device('cuda', local)
    data_config = configs.data_config()
    nn_cfg = configs.lstm_model_config(k_neighbors=args.k)

    gpe_amd.set_math(args.math)
    torch.manual_seed(0)                               # identical replicas on every rank
    model = nets.GarmentFullPattern3D(data_config, dict(nn_cfg), dict(nn_cfg['loss'])).to(dev).train()
    model.loss.with_quality_eval = False
    wrapped = parallel.DistributedHotPath(model, device_ids=[dev])
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    feats, gt = synthetic(args.batch, args.points, data_config, seed=1000 + rank, device=dev)

    def step(i):
        torch.manual_seed(i * 131 + rank)              # the decoder draws random LSTM states every forward
        preds = wrapped(feats, log_step=i, epoch=0)
        loss, _, _ = model.loss(preds, gt, epoch=0)
        loss.backward()
        wrapped.finish_gradient_sync()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
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
    roof, roof_gather, breakdown = None, None, None
    rec = None
    if not args.no_kernel_timing:
        # every rank runs these extra steps (they contain the gradient all-reduce); only rank 0 records events
        nsteps = min(args.steps, 5)
        if rank == 0:
            _lib.TIMING = []
        for i in range(nsteps):
            step(args.warmup + args.steps + i)
        barrier()
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
        # dominant = the fused per-edge GEMM family (forward + backward + weight-gradient kernels)
        fam = ['gpe_edge_mlp_fwd', 'gpe_edge_mlp_bwd', 'gpe_edge_redgemm']
        dom = max(fam, key=lambda n: agg.get(n, [0, 0, 0, 0])[1])
        n_l, ms, fl, _ = agg[dom]
        ach = fl / (ms * 1e-3) / 1e12
        traffic, tsrc = pmc_traffic(dom)
        roof = {'kernel': dom, 'bound': 'mfma', 'achieved': ach, 'peak': PEAK_F32_TFLOPS, 'unit': 'TFLOP/s',
                'frac': ach / PEAK_F32_TFLOPS, 'traffic': traffic, 'traffic_unit': 'HBM bytes/launch',
                'traffic_source': tsrc, 'launches_per_step': n_l / nsteps,
                'avg_launch_ms': ms / n_l, 'flops_per_launch': fl / n_l}
        if 'gpe_edge_gather_stats' in agg:
            n_l, ms, _, by = agg['gpe_edge_gather_stats']
            ach = by / (ms * 1e-3) / 1e9
            roof_gather = {'kernel': 'gpe_edge_gather_stats', 'bound': 'hbm', 'achieved': ach, 'peak': PEAK_HBM_GBS,
                           'unit': 'GB/s', 'frac': ach / PEAK_HBM_GBS,
                           'traffic': pmc_traffic('gpe_edge_gather_stats')[0],
                           'avg_launch_ms': ms / n_l, 'bytes_per_launch': by / n_l}

    # the opt-in fast mode, measured the same way on the same workload (reported beside `value`, never as `value`)
    fast = None
    if world == 1 and args.math == 'f32' and not args.no_fast_math_line:
        gpe_amd.set_math('bf16x3')
        n_f = max(3, min(args.steps, 10))
        for i in range(2):
            step(10_000 + i)
        barrier()
        t1 = time.perf_counter()
        for i in range(n_f):
            step(10_002 + i)
        barrier()
        dt = time.perf_counter() - t1
        gpe_amd.set_math('f32')
        fast = {'math': 'bf16x3', 'value': args.batch * n_f / dt, 'unit': 'garments/s', 'steps': n_f,
                'ms_per_step': dt / n_f * 1e3,
                'note': 'split-bf16 edge GEMMs on the bf16 matrix pipe, fp32 accumulate: forward within 1e-4 of the '
                        'reference, encoder gradients within ~1e-2 (tests/test_gpu_kernels.py TOL); opt-in via '
                        'gpe_math_set(1) / GPE_MATH=bf16x3'}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, data_config, nn_cfg)

    if rank == 0:
        garments = args.batch * world * args.steps
        out = {
            'metric': 'garments/sec (fwd+bwd) at N=%d pts, batch %d per GPU' % (args.points, args.batch),
            'value': garments / elapsed, 'unit': 'garments/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.math, 'data': 'synthetic',
            'config': {'workload': 'BASELINE cfg 2: GarmentFullPattern3D, N=%d, batch %d/GPU, k=%d, EdgeConv encoder'
                                   ' + LSTM decoders' % (args.points, args.batch, args.k),
                       'global_batch': args.batch * world, 'parallelism': 'dp%d' % world,
                       'step': 'fwd + ComposedPatternLoss + bwd' + (' + RCCL grad all-reduce' if world > 1 else '')
                               + ' + Adam', 'final_loss': final_loss},
            'roofline': roof, 'roofline_gather': roof_gather, 'cpu_baseline': cpu, 'fast_math': fast,
            'kernel_ms_per_step': breakdown}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
