"""Profiling aid: times the fused edge kernels at the BASELINE cfg-2 size with ablation switches (gpe_debug_set)."""
import sys, os
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
W2 = torch.randn(H, H, device=dev) / 14
b2 = torch.randn(H, device=dev)
W3 = torch.randn(Fo, H, device=dev) / 14
b3 = torch.randn(Fo, device=dev)
E = B * N * k
a2 = torch.empty(E, H, device=dev)
a3 = torch.empty(E, 152, device=dev)
nblk = L.query('gpe_stats_blocks')
EWS, NWS = ops.edge_workspace(B, N, k, 2 * H, dev)      # caller-owned workspace of the edge entry points
part = torch.empty(nblk, 2, H, device=dev, dtype=torch.float64)
mx = torch.empty(B * N, 152, device=dev); mn = torch.empty_like(mx)
amx = torch.empty(B * N, 152, device=dev, dtype=torch.uint8); amn = torch.empty_like(amx)
w2p, w3p = ops.pack_weight(W2), ops.pack_weight(W3)
w3t = ops.pack_weight(W3, transpose=True)
coef = torch.randn(4, H, device=dev)
G = torch.empty(H, H, device=dev); cs = torch.empty(H, device=dev)
ws = torch.empty(L.query('gpe_redgemm_ws', H, H), device=dev)
shift = torch.randn(H, device=dev)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def f2(): L.call('gpe_edge_mlp_fwd', 0, PQ, 2 * H, jg, None, 0, B, N, k, H, H, w2p, b2, a2, H, part, 0, None, None, None, None, 0, None, None, EWS, NWS, 0)
def f3(): L.call('gpe_edge_mlp_fwd', 1, None, 0, None, a2, H, B, N, k, H, Fo, w3p, b3, a3, 152, part, 1, mx, mn, amx, amn, 152, None, None, EWS, NWS, 0)
def b2a(): L.call('gpe_edge_mlp_bwd', a3, 152, 0, None, 0, None, B, N, k, Fo, H, w3t, coef, a2, H, None, 0, None, None, EWS, NWS, None, 0, None, None, 0, None)
def rg(): L.call('gpe_edge_redgemm', a2, H, 0, None, 0, PQ, 2 * H, jg, shift, B, N, k, H, H, G, H, cs, ws, None, None, EWS, NWS, None, 0, None, None, 0, None)
def rd(): L.call('gpe_edge_redgemm', a3, 152, 1, a2, H, None, 0, None, shift, B, N, k, Fo, H, G[:Fo], H, cs, ws, None, None, EWS, NWS, None, 0, None, None, 0, None)

MODE = sys.argv[1] if len(sys.argv) > 1 else 'f32'
gpe_amd.set_math(MODE)
BASE = 128 if MODE == 'f16x3' else 0          # f16x3: measure the GEMM kernel alone (scales of the first launch reused)
if BASE:
    for fn in (f2, f3, b2a):
        fn()
flops = {'f2': 2.0 * E * H * H, 'f3': 2.0 * E * H * Fo, 'b2a': 2.0 * E * H * Fo, 'rg': 2.0 * E * H * H, 'rd': 2.0 * E * H * Fo}
for flags, label in [(0, 'production'), (1, 'no staging'), (2, 'no epilogue'), (3, 'no staging, no epilogue'),
                     (7, 'MFMA only + barriers'), (15, 'MFMA only, no barriers'), (32, 'no producer MFMAs'), (16, 'producers only (no consumer MFMA)')]:
    if BASE and flags > 3:
        continue
    L.query('gpe_debug_set', flags | BASE)
    out = []
    for name, fn in [('f2', f2), ('f3', f3), ('b2a', b2a)]:
        ms = timeit(fn)
        out.append('%s %.3f ms (%.0f TF)' % (name, ms, flops[name] / ms / 1e9))
    print('dbg=%2d %-26s ' % (flags, label) + ' | '.join(out))
L.query('gpe_debug_set', 0)
for name, fn in [('rg(gather)', rg), ('rd(dense)', rd)]:
    ms = timeit(fn)
    print('%s %.3f ms (%.0f TF)' % (name, ms, flops['rg' if 'rg' in name else 'rd'] / ms / 1e9))
