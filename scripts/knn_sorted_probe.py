#!/usr/bin/env python3
"""Layer-2 kNN (C = 150) on the features of a real layer-1 forward at cfg 2: rows in original order vs rows pre-sorted by the Morton
order of the points, with GPE_KNN_ROT (tile visit order rotated to start at the queries' own tile).  GPE_DEBUG=1 GPE_KNN_ROT=0/1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpe_amd
from gpe_amd import ops, configs, nets

B, N, k = 32, 2048, 16
torch.manual_seed(0)
dc = configs.data_config(); cfg = configs.lstm_model_config(k_neighbors=k)
gpe_amd.set_math('f16x3')
model = nets.GarmentFullPattern3D(dc, dict(cfg), dict(cfg['loss'])).cuda().train()
pos = torch.randn(B, N, 3, generator=torch.Generator().manual_seed(1)).cuda()
conv = model.feature_extractor.conv_layers[0]
with torch.no_grad():
    f1 = conv(pos.reshape(-1, 3), B, N)                      # [B*N, 150]


def morton(p):
    lo, hi = p.min(1, keepdim=True).values, p.max(1, keepdim=True).values
    c = ((p - lo) / (hi - lo + 1e-9) * 16).long().clamp(0, 15)
    code = torch.zeros(p.shape[:2], dtype=torch.long, device=p.device)
    for b in range(4):
        for a in range(3):
            code |= ((c[..., a] >> b) & 1) << (3 * b + a)
    return code


order = morton(pos).argsort(dim=1, stable=True)                 # [B, N] sorted position -> original index
fs = f1.view(B, N, -1).gather(1, order[..., None].expand(-1, -1, f1.shape[1])).reshape(B * N, -1).contiguous()
pad = lambda t: torch.nn.functional.pad(t, (0, (-t.shape[1]) % 4)).contiguous()[:, :t.shape[1]]
for name, x in (('original order', pad(f1)), ('Morton-sorted rows', pad(fs))):
    idx = ops.knn(x, B, N, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        idx = ops.knn(x, B, N, k)
    e1.record(); torch.cuda.synchronize()
    print('ROT=%s %-20s %.1f us per search' % (os.environ.get('GPE_KNN_ROT', '0'), name, e0.elapsed_time(e1) * 100))
# same neighbour SETS either way (the sorted search answers in sorted numbering)
i0 = ops.knn(pad(f1), B, N, k).long()
i1 = ops.knn(pad(fs), B, N, k).long()
back = order.gather(1, i1.view(B, N * k)).view(B, N, k)     # sorted numbering -> original indices
back_q = torch.empty_like(back); back_q.scatter_(1, order[..., None].expand(-1, -1, k), back)
print('neighbour sets equal:', bool((back_q.sort(-1).values == i0.sort(-1).values).all()))
