#!/usr/bin/env python3
"""Timing probes of the multi-tile persistent LSTM forward: gpe_debug_set bits 19..22 switch phases off (numbers are wrong then)."""
import sys
import torch
import gpe_amd
from gpe_amd import ops, net_blocks, _lib as Lb
Bn, In, H, T, L = (736, 250, 250, 14, 3)
torch.manual_seed(0)
rnn = torch.nn.LSTM(In, H, L, batch_first=True).cuda()
params = net_blocks._rnn_params(rnn, L)
plan = ops.PackPlan()
net_blocks._register_rnn_packs(plan, rnn, L, H, 4)
x = torch.randn(Bn, In).cuda(); h0 = (torch.randn(L, Bn, H) * 0.3).cuda(); c0 = (torch.randn(L, Bn, H) * 0.3).cuda()
gpe_amd.set_math('f16x3')
plan.refresh()
def t(dbg, reps=20):
    Lb.query('gpe_debug_set', dbg)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for r in range(reps + 3):
        torch.cuda.synchronize(); e0.record()
        with torch.no_grad():
            ops.rnn_stack(x, h0, c0, T, L, 'lstm', params, h0_bounded=True)
        e1.record(); torch.cuda.synchronize()
        if r >= 3: tot += e0.elapsed_time(e1)
    return tot / reps * 1e3
for name, bits in (('diagonal launches', 131072), ('persistent', 0), ('no payload loads', 1 << 19), ('no plain stores', 2 << 19), ('no products', 4 << 19), ('cheap cell', 8 << 19),
                   ('no loads + no stores', 3 << 19), ('no loads, stores, products', 7 << 19), ('nothing', 15 << 19)) + tuple(sys.argv[1:] and [('custom', int(sys.argv[1]))]):
    print('%-30s %.1f us' % (name, t(bits)))
Lb.query('gpe_debug_set', 0)
