"""Runs one training step of the other BASELINE.json configurations at full size (they are parity-test shapes, not
bench lines): checks that they run, that outputs are finite, and prints the step time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpe_amd as gpe

cfg = gpe.configs
cases = [
    ('cfg1-shape  full3d  N=1024 B=8  k=5 ', 'full', 8, 1024, 5),
    ('cfg4        segment N=4096 B=32 k=20', 'seg', 32, 4096, 20),
    ('cfg5/GPU    full3d  N=8192 B=64 k=16', 'full', 64, 8192, 16),
]
for name, kind, B, N, k in cases:
    torch.manual_seed(0)
    if kind == 'full':
        nn_cfg = cfg.lstm_model_config(k_neighbors=k)
        model = gpe.nets.GarmentFullPattern3D(cfg.data_config(), dict(nn_cfg), dict(nn_cfg['loss']))
    else:
        nn_cfg = cfg.att_model_config(k_neighbors=k)
        model = gpe.nets.GarmentSegmentPattern3D(cfg.data_config(), dict(nn_cfg), dict(nn_cfg['loss']))
    model = model.cuda().train()
    g = torch.Generator().manual_seed(1)
    feats = torch.randn(B, N, 3, generator=g).cuda()
    def step():
        torch.manual_seed(3)
        p = model(feats)
        loss = sum(v.float().square().mean() for v in p.values() if v.dtype.is_floating_point)
        loss.backward()
        model.zero_grad(set_to_none=True)
        return p, loss
    p, loss = step(); torch.cuda.synchronize()
    t0 = time.perf_counter(); p, loss = step(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ok = all(torch.isfinite(v).all().item() for v in p.values() if v.dtype.is_floating_point)
    print('%s  fwd+bwd %.1f ms  (%.0f garments/s)  finite=%s  peak mem %.1f GB' % (
        name, dt * 1e3, B / dt, ok, torch.cuda.max_memory_allocated() / 2**30))
    del model, feats, p, loss
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
