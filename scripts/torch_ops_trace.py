#!/usr/bin/env python3
"""Which torch (aten) operators a training step dispatches besides the C-ABI launches: name, output shape, Python call site."""
import sys, traceback, collections
import torch
import torch.utils._python_dispatch as pd
import gpe_amd
from gpe_amd import configs, nets, optim
import bench
dev = torch.device('cuda', 0)
gpe_amd.set_math('f16x3')
k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B, N = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4, 512)
dc = configs.data_config()
cfg = configs.lstm_model_config(k_neighbors=k)
torch.manual_seed(0)
model = nets.GarmentFullPattern3D(dc, dict(cfg), dict(cfg['loss'])).to(dev).train()
model.loss.with_quality_eval = False
feats, gt = bench.synthetic(B, N, dc, seed=1000, device=dev)
opt = optim.FusedAdam(optim.FlatArena(model), lr=2e-3)
def step():
    loss = model.loss(model(feats), gt, epoch=0)[0]
    loss.backward(); opt.step()
for _ in range(2): step()
seen = collections.OrderedDict()
class Mode(pd.TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(s in name for s in ('empty', 'view', 'as_strided', 'slice', 'select', 'reshape', 'detach', 'alias', 'expand', 'unsqueeze', 'squeeze', 'permute', 'transpose', 't.default', 'unbind', 'split', '_unsafe_view', 'is_pinned', 'stride', 'size', 'sym_', 'lift_fresh', 'record_stream', 'narrow', 'contiguous', 'unflatten')):
            if not ('contiguous' in name or 'reshape' in name):
                return out
        st = [f for f in traceback.extract_stack() if 'garment-pattern-estimation_amd' in f.filename or 'gpe_amd' in f.filename]
        site = '%s:%d' % (st[-1].filename.split('/')[-1], st[-1].lineno) if st else 'autograd/engine'
        shape = tuple(out.shape) if isinstance(out, torch.Tensor) else None
        key = (name, site, shape)
        seen[key] = seen.get(key, 0) + 1
        return out
with Mode():
    step()
torch.cuda.synchronize()
tot = 0
for (name, site, shape), n in seen.items():
    print('%3d x %-38s %-28s %s' % (n, name, site, shape)); tot += n
print('total', tot)
