#!/usr/bin/env python3
"""The multi-tile persistent LSTM kernels (csrc/gpe_rnn_persist_mt.hip) against the diagonal launches at the panel decoder's shape:
numbers (forward and every gradient), run-to-run bit identity, and HIP-event time of forward / backward.
    python scripts/rnn_mt_check.py [Bn In H T L]"""
import sys
import torch
import gpe_amd
from gpe_amd import ops, net_blocks, _lib as Lb

Bn, In, H, T, L = [int(a) for a in sys.argv[1:6]] if len(sys.argv) > 5 else (736, 250, 250, 14, 3)
torch.manual_seed(0)
rnn = torch.nn.LSTM(In, H, L, batch_first=True).cuda()
params = net_blocks._rnn_params(rnn, L)
plan = ops.PackPlan()
net_blocks._register_rnn_packs(plan, rnn, L, H, 4)
g = torch.Generator().manual_seed(3)
x = torch.randn(Bn, In, generator=g).cuda()
h0 = (torch.randn(L, Bn, H, generator=g) * 0.3).cuda()
c0 = (torch.randn(L, Bn, H, generator=g) * 0.3).cuda()
wgt = torch.randn(Bn, T, H, generator=g).cuda()
gpe_amd.set_math('f16x3')
plan.refresh()


def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def run():
    for p in rnn.parameters():
        p.grad = None
    xd = x.clone().requires_grad_()
    top, hN, cN = ops.rnn_stack(xd, h0, c0, T, L, 'lstm', params, want_state=True, h0_bounded=True)
    ((top * wgt).sum() + hN.sum() * 0.5 + cN.sum() * 0.25).backward()
    return [top.detach().clone(), hN.clone(), cN.clone(), xd.grad.clone()] + [p.grad.clone() for p in rnn.parameters()]


def timed(dbg, reps=20):
    Lb.query('gpe_debug_set', dbg)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for r in range(reps + 3):
        xd = x.clone().requires_grad_()
        torch.cuda.synchronize()
        ev[0].record()
        top, hN, cN = ops.rnn_stack(xd, h0, c0, T, L, 'lstm', params, want_state=True, h0_bounded=True)
        ev[1].record()
        loss = (top * wgt).sum()
        torch.cuda.synchronize()
        ev[0].synchronize()
        e2 = torch.cuda.Event(enable_timing=True); e3 = torch.cuda.Event(enable_timing=True)
        e2.record()
        loss.backward()
        e3.record()
        torch.cuda.synchronize()
        if r >= 3:
            tf += ev[0].elapsed_time(ev[1]); tb += e2.elapsed_time(e3)
    return tf / reps * 1e3, tb / reps * 1e3


Lb.query('gpe_debug_set', 131072)
ref = run()
Lb.query('gpe_debug_set', 0)
first = run()
names = ['top', 'hN', 'cN', 'dx'] + [n for n, _ in rnn.named_parameters()]
worst = 0.0
for n, a, b in zip(names, first, ref):
    e = relerr(a, b)
    worst = max(worst, e)
    if e > 2e-5:
        print('MISMATCH', n, e)
print('worst relative difference vs the diagonal launches: %.2e; same bits as them: %s' % (worst, torch.equal(first[0], ref[0])))
same = True
for rep in range(5):
    again = run()
    same = same and all(torch.equal(a, b) for a, b in zip(again, first))
print('run-to-run bit identity over 5 repetitions:', same)
for dbg, name in ((131072, 'diagonal launches'), (0, 'persistent (waves own row tiles)'), (262144, 'persistent, sc1 payload loads')):
    f, b = timed(dbg)
    print('%-36s rnn_stack forward %.1f us, backward (incl. weight-gradient GEMMs) %.1f us' % (name, f, b))
Lb.query('gpe_debug_set', 0)
