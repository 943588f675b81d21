#!/usr/bin/env python3
"""gpe_bn_finalize alone (512 partial blocks, C = 200 / 150): microseconds per launch by HIP events, 200 launches back to back."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpe_amd
from gpe_amd import ops, _lib as L
dev = torch.device('cuda')
nblk = L.query('gpe_stats_blocks')
for C in (200, 150):
    part = torch.randn(nblk, 2, C, dtype=torch.float64, device=dev).abs()
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    rm, rv, nb = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.tensor(0, device=dev)
    for _ in range(20):
        ops.bn_finalize(part, nblk, C, 1e6, g, b, 1e-5, 0.1, rm, rv, nb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.bn_finalize(part, nblk, C, 1e6, g, b, 1e-5, 0.1, rm, rv, nb)
    e1.record(); torch.cuda.synchronize()
    print('C = %d: %.2f us per launch' % (C, e0.elapsed_time(e1) * 1e3 / 200))
