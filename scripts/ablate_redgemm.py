"""Profiling aid: the f16x3 edge reduce-GEMM (gpe_redgemm_b3_kernel) at the BASELINE cfg-2 size under the timing-only switches of
GPE_RD_DBG (1 = no commit, 2 = no row loads, 4 = no consumer MFMAs, 8 = no left-over MFMAs; results are wrong).  One process per
setting (the switch is read once):  for d in 0 1 2 3 4 8 15; do GPE_RD_DBG=$d python scripts/ablate_redgemm.py; done"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpe_amd
from gpe_amd import ops, _lib as L

B, N, k, H, Fo = 32, 2048, 16, 200, 150
dev = 'cuda'
torch.manual_seed(0)
x = torch.randn(B * N, 3, device=dev)
idx, jg = ops.knn(x, B, N, k, want_global=True)
PQ = torch.randn(B * N, 2 * H, device=dev)
E = B * N * k
dz2 = torch.randn(E, H, device=dev)
a2 = torch.randn(E, H, device=dev).abs()
dz3 = torch.randn(E, 152, device=dev)
G = torch.empty(H, H, device=dev)
cs = torch.empty(H, device=dev)
ws = torch.empty(L.query('gpe_redgemm_ws', H, H), device=dev)
shift = torch.randn(H, device=dev)
EWS, NWS = ops.edge_workspace(B, N, k, 2 * H, dev)
gpe_amd.set_math('f16x3')
words = torch.zeros(4, dtype=torch.int32, device=dev)
for i, t in enumerate((dz2, a2, dz3)):
    L.call('gpe_absmax', t, t.stride(0), t.shape[0], t.shape[1], words[i:i + 1])
L.call('gpe_edge_pq_amax', PQ, 2 * H, H, B * N, words[3:4], EWS, NWS)


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def rg():   # gathered V: G1 = dz2^T relu(P_i + Q_j)
    L.call('gpe_edge_redgemm', dz2, H, 0, None, 0, PQ, 2 * H, jg, shift, B, N, k, H, H, G, H, cs, ws, words[0:1], words[3:4], EWS, NWS,
           None, 0, None, None, 0, None)


def rd():   # dense V: G2 = dz3^T a2
    L.call('gpe_edge_redgemm', dz3, 152, 1, a2, H, None, 0, None, shift, B, N, k, Fo, H, G[:Fo], H, cs, ws, words[2:3], words[1:2], EWS, NWS,
           None, 0, None, None, 0, None)


print('GPE_RD_DBG=%s  gathered 13x13: %.0f us   dense 10x13: %.0f us' % (os.environ.get('GPE_RD_DBG', '0'), timeit(rg), timeit(rd)))
