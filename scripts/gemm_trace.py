#!/usr/bin/env python3
"""Step timeline of workgroup 0 of the bf16x6 reduce-GEMM (csrc/gpe_gemm_x6.hip, gpe_debug_set(32768)): per step, what the multiplying
wave 0 and the staging wave 4 did when (10 ns wall-clock stamps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gpe_amd
from gpe_amd import ops, _lib as L

gpe_amd.set_math('f16x3')
g = torch.Generator().manual_seed(0)
rows, Mg, Ng = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (10304, 1000, 250)
u = torch.randn(rows, Mg, generator=g).cuda()
v = torch.randn(rows, (Ng + 3) // 4 * 4, generator=g).cuda()
G = torch.empty(Mg, Ng, device='cuda')
cs = torch.empty(Mg, device='cuda')
n = L.query('gpe_redgemm_ws', Mg, Ng)
ws = torch.empty(n, device='cuda')
L.query('gpe_debug_set', 32768)
for it in range(3):
    L.call('gpe_redgemm', u, Mg, 0, 0, v, v.stride(0), 0, 0, None, rows, Mg, Ng, G, Ng, cs, ws, 0)
torch.cuda.synchronize()
from math import ceil
x6 = 32 * ceil(Mg / 128) * 128 * ceil(Ng / 128) * 128 + 64 * ceil(Mg / 128) * 128 + 8 + 1024
tr = ws[x6 - 1024:x6].view(torch.uint8).cpu().numpy().view(np.uint64).reshape(2, 64, 4).astype(np.float64) * 0.01
t0 = tr[tr > 0].min()
print('rows %d, %d x %d: step | consumer: start, MFMAs issued, past barrier | stager: start, committed, loads issued, past barrier  (us from t0)' % (rows, Mg, Ng))
for s in range(64):
    if tr[0, s, 0] == 0 and tr[1, s, 0] == 0:
        break
    c, st = tr[0, s] - t0, tr[1, s] - t0
    print('%2d | %6.2f %6.2f %6.2f | %6.2f %6.2f %6.2f %6.2f' % (s, c[0], c[1], c[3], st[0], st[1], st[2], st[3]))
