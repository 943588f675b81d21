#!/usr/bin/env python3
"""Race detector for the side stream (ops.side_grads / EdgeConvFn forks / idle-stretch jobs): the same training run twice at cfg 2 with
the side stream ON — N steps of forward + loss + backward + FusedAdam from the same seed — must give bit-identical losses and
parameters, and must equal the run with everything on one stream.    python scripts/side_stream_soak.py [steps] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpe_amd
from gpe_amd import configs, nets, optim, ops
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = 2048
dev = torch.device('cuda', 0)
gpe_amd.set_math('f16x3')
dc = configs.data_config(); cfg = configs.lstm_model_config(k_neighbors=16)


def run(side):
    ops.SIDE_GRADS = side
    torch.manual_seed(0)
    model = nets.GarmentFullPattern3D(dc, dict(cfg), dict(cfg['loss'])).to(dev).train()
    model.loss.with_quality_eval = False
    opt = optim.FusedAdam(optim.FlatArena(model), lr=1e-3)
    losses = []
    for s in range(steps):
        feats, gt = bench.synthetic(B, N, dc, seed=1000 + s, device=dev)
        torch.manual_seed(100 + s)
        loss = model.loss(model(feats), gt, epoch=0)[0]
        loss.backward()
        opt.step()
        losses.append(loss.detach().clone())
    torch.cuda.synchronize()
    return torch.stack(losses).cpu(), opt.arena.flat.clone().cpu()


a = run(True); b = run(True); c = run(False)
ok = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
ok1 = torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
print('side stream on, twice: losses / parameters bit-identical: %s; vs one stream: %s (%d steps, batch %d, last loss %.6f)'
      % (ok, ok1, steps, B, a[0][-1].item()))
sys.exit(0 if ok and ok1 else 1)
