#!/usr/bin/env python3
"""The dense GEMM shapes of the cfg-2 step (decoder weight gradients, layer-2 [P|Q] projection and its gradients) a few times each, in
f16x3 mode, for counter passes / timing of csrc/gpe_gemm_x6.hip:   rocprofv3 --pmc ... -- python scripts/pmc_gemm.py [debug flags]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpe_amd
from gpe_amd import ops, _lib

flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
gpe_amd.set_math('f16x3')
_lib.lib().gpe_debug_set(flags)
g = torch.Generator().manual_seed(0)
x = torch.randn(65536, 152, generator=g).cuda()
w = torch.randn(400, 150, generator=g).cuda()
b = torch.randn(400, generator=g).cuda()
wp, wpt = ops.pack_weight(w), ops.pack_weight(w, transpose=True)
dpq = torch.randn(65536, 400, generator=g).cuda()
y = torch.empty(65536, 400, device='cuda')
dx = torch.empty(65536, 150, device='cuda')
dg = torch.randn(736, 14, 1000, generator=g).cuda()
hs = torch.randn(736, 15, 252, generator=g).cuda()


def run():
    ops.linear_raw((x, 152, 0, 0), wp, b, 65536, 400, 150, (y, 400, 0, 0))
    ops.linear_raw((dpq, 400, 0, 0), wpt, None, 65536, 150, 400, (dx, 150, 0, 0))
    ops.redgemm_raw((dpq, 400, 0, 0), (x, 152, 0, 0), 65536, 400, 150)
    ops.redgemm_raw(ops._rows3d(dg), ops._rows3d(hs[:, :14, :250]), 10304, 1000, 250)


for it in range(3):
    run()
torch.cuda.synchronize()
if len(sys.argv) > 2:
    names = ['linear 65536x400x150', 'linear 65536x150x400', 'redgemm 65536 rows 400x150', 'redgemm 10304 rows 1000x250']
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(10)]
    for it in range(10):
        ev[it][0].record()
        ops.linear_raw((x, 152, 0, 0), wp, b, 65536, 400, 150, (y, 400, 0, 0)); ev[it][1].record()
        ops.linear_raw((dpq, 400, 0, 0), wpt, None, 65536, 150, 400, (dx, 150, 0, 0)); ev[it][2].record()
        ops.redgemm_raw((dpq, 400, 0, 0), (x, 152, 0, 0), 65536, 400, 150); ev[it][3].record()
        ops.redgemm_raw(ops._rows3d(dg), ops._rows3d(hs[:, :14, :250]), 10304, 1000, 250); ev[it][4].record()
    torch.cuda.synchronize()
    for i, n in enumerate(names):
        print('flags %d %-30s %.1f us' % (flags, n, 1e3 * sum(ev[it][i].elapsed_time(ev[it][i + 1]) for it in range(10)) / 10))
