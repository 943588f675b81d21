#!/usr/bin/env python3
"""Where a wave of gpe_knn_ft_kernel spends its cycles (library built with KNN_FT_TIMING=1: the first query of every wave carries six
counters instead of its list): commit + prefetch issue | LDS reads + MFMAs + distances | appends + tightens | barrier wait | tightens | tiles with a hit."""
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
for it in range(2):
    ops.knn(x, B, N, k, order=order)
torch.cuda.synchronize()
nws = _lib.query('gpe_knn_ws_bytes', B, N, 150, k)
ws = ops._workspace(nws, x.device)
part = ws.view(torch.int64)[: B * N * 64].view(B, N, 64).cpu()
inv = torch.empty(B, N, dtype=torch.long)
o = order.cpu().long()
# the lists are stored by POINT; wave w of workgroup qt owns plane rows 128 qt + 16 w .. -> its first query is point order[b][row]
rows = torch.arange(0, N, int(os.environ.get("ROWS_PER_WAVE", "16")))
keys = part[torch.arange(B)[:, None], o[:, rows]]          # [B, N/16, 64]
cyc = (keys[..., :6] >> 32).double()
names = ['commit+prefetch', 'lds+mfma+dist', 'append+tighten', 'barrier', 'tightens', 'tiles with a hit']
tot = cyc[..., :4].sum(-1)
print('waves %d; cycles per wave: total %.0f' % (cyc.shape[0] * cyc.shape[1], tot.mean()))
for i, n in enumerate(names):
    print('  %-18s mean %9.1f  max %9.1f' % (n, cyc[..., i].mean(), cyc[..., i].max()))
