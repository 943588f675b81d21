#!/usr/bin/env python3
"""One EdgeConv layer (cfg-2 layer-2 shape: B=32, N=2048, C=150, H=200, F=150, k=16) forward+backward a few times, for
counter passes on the gather kernels:   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- \
    python scripts/pmc_f2.py [debug-flags]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpe_amd
from gpe_amd import _lib

flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
_lib.lib().gpe_debug_set(flags)
B, N, C, H, F, k = 32, 2048, 150, 200, 150, 16
torch.manual_seed(0)
conv = gpe_amd.net_blocks.DynamicEdgeConv(gpe_amd.net_blocks.MLP([2 * C, H, H, F]), k=k).cuda().train()
x = torch.randn(B * N, C, device='cuda', requires_grad=True)
for it in range(3):
    out = conv(x, B, N)
    out.square().mean().backward()
torch.cuda.synchronize()
if len(sys.argv) > 2:
    import time
    t0 = time.perf_counter()
    for it in range(10):
        out = conv(x, B, N)
        out.square().mean().backward()
    torch.cuda.synchronize()
    print('flags %d: %.3f ms per fwd+bwd' % (flags, (time.perf_counter() - t0) * 100))
