#!/usr/bin/env python3
"""Step timeline of the persistent LSTM kernels (csrc/gpe_rnn_persist.hip) at the pattern decoder's shape: gpe_debug_set(8192) makes
every workgroup stamp eight phases per step into the tail of its workspace; this prints the median duration of each phase per layer.
    python scripts/rnn_trace.py [Bn In H T L]"""
import sys
import numpy as np
import torch
import gpe_amd
from gpe_amd import ops, net_blocks, _lib as Lb

Bn, In, H, T, L = [int(a) for a in sys.argv[1:6]] if len(sys.argv) > 5 else (32, 250, 250, 23, 2)
torch.manual_seed(0)
rnn = torch.nn.LSTM(In, H, L, batch_first=True).cuda()
params = net_blocks._rnn_params(rnn, L)
plan = ops.PackPlan()
net_blocks._register_rnn_packs(plan, rnn, L, H, 4)
x = torch.randn(Bn, In).cuda()
h0 = (torch.randn(L, Bn, H) * 0.3).cuda()
c0 = (torch.randn(L, Bn, H) * 0.3).cuda()
gpe_amd.set_math(sys.argv[6] if len(sys.argv) > 6 else 'f16x3')
plan.refresh()
Lb.query('gpe_debug_set', 8192)
got = {}
ops.RNN_WS_HOOK = lambda kind, ws: got.__setitem__(kind, ws)
NB, NRT = (H + 15) // 16, (Bn + 15) // 16
grid = L * NB * NRT
flag_bytes = (L * T * NRT * 4 + 255) & ~255
NAMES = ['0 start->seg1 poll', '1 seg1 poll->seg1 done', '2 seg1 done..', '3 seg0 poll done->partials written', '4 barrier+epilogue+stores issued',
         '5 stores issued->drained', '6 drained->arrived']
for rep in range(3):
    xd = x.clone().requires_grad_()
    top, _, _ = ops.rnn_stack(xd, h0, c0, T, L, 'lstm', params)
    top.sum().backward()
    torch.cuda.synchronize()
for kind in ('fwd', 'bwd'):
    raw = got[kind].view(torch.uint8).cpu().numpy()[flag_bytes:flag_bytes + grid * T * 64]
    st = raw.view(np.uint64).reshape(grid, T, 8).astype(np.float64) * 0.01          # us (100 MHz)
    print('== %s: Bn=%d H=%d T=%d L=%d, %d workgroups; kernel span %.1f us' % (kind, Bn, H, T, L, grid, st[st > 0].max() - st[st > 0].min()))
    for l in range(L):
        s = st[l * NRT * NB:(l + 1) * NRT * NB]                                      # [wg][t][8]
        order = range(T) if kind == 'fwd' else range(T - 1, -1, -1)
        seq = [t for t in order]
        inner = seq[2:-1]
        per_step = np.median([s[:, seq[i + 1], 0] - s[:, seq[i], 0] for i in range(1, T - 2)])
        print(' layer %d: median step period %.2f us' % (l, per_step))
        def d(a, b):
            v = s[:, inner, b] - s[:, inner, a]
            ok = (s[:, inner, a] > 0) & (s[:, inner, b] > 0)
            return np.median(v[ok]) if ok.any() else float('nan')
        print('   start->flag 1 seen %.2f | ->flag 0 seen %.2f | loads of both segments + products of segment 1 %.2f | products of segment 0 + '
              'partials %.2f | barrier+epilogue+stores issued %.2f | drain %.2f | barrier+atomic %.2f'
              % (d(0, 1), d(1, 3), d(3, 2), d(2, 4), d(4, 5), d(5, 6), d(6, 7)))
