#!/usr/bin/env python3
"""Layer-2 search (C = 150) at cfg 2 on the features of a real layer-1 forward, in the first layer's curve order: time per launch under
GPE_KNN_PROBE (1 = no selection after the first tile, 4 = no MFMAs; wrong results) — what the staging skeleton costs.
    for p in 0 1 4 5; do GPE_DEBUG=1 GPE_KNN_PROBE=$p python scripts/knn_l2_probe.py; done"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpe_amd
from gpe_amd import ops, configs, nets, _lib

B, N, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 2048, 16)
torch.manual_seed(0)
dc = configs.data_config(); cfg = configs.lstm_model_config(k_neighbors=k)
gpe_amd.set_math('f16x3')
model = nets.GarmentFullPattern3D(dc, dict(cfg), dict(cfg['loss'])).cuda().train()
pos = torch.randn(B, N, 3, generator=torch.Generator().manual_seed(1)).cuda()
conv = model.feature_extractor.conv_layers[0]
with torch.no_grad():
    f1 = conv(pos.reshape(-1, 3), B, N)
order = conv.last_order
x = torch.zeros(B * N, 152, device='cuda'); x[:, :150] = f1
x = x[:, :150]
_lib.TIMING = []
for it in range(6):
    ops.knn(x, B, N, k, order=order)
torch.cuda.synchronize()
rec = _lib.TIMING[1:]; _lib.TIMING = None
print('PROBE=%s order: %.1f us per search' % (os.environ.get('GPE_KNN_PROBE', '0'), 1e3 * sum(e0.elapsed_time(e1) for _, _, e0, e1 in rec) / len(rec)))
