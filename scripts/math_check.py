"""Accuracy + speed of the bf16x3 edge kernels against the exact-fp32 ones on the same inputs (BASELINE cfg-2 sizes)."""
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
nblk = L.query('gpe_stats_blocks')
EWS, NWS = ops.edge_workspace(B, N, k, 2 * H, dev)      # caller-owned workspace of the edge entry points
w2p, w3p = ops.pack_weight(W2), ops.pack_weight(W3)
w3t = ops.pack_weight(W3, transpose=True)
coef = torch.randn(4, H, device=dev)
dz3 = torch.randn(E, 152, device=dev) * 1e-3
dz3[:, 150:] = 0


def run():
    a2 = torch.zeros(E, H, device=dev)
    a3 = torch.zeros(E, 152, device=dev)
    part2 = torch.zeros(nblk, 2, H, device=dev, dtype=torch.float64)
    part3 = torch.zeros(nblk, 2, Fo, device=dev, dtype=torch.float64)
    mx = torch.zeros(B * N, 152, device=dev); mn = torch.zeros_like(mx)
    amx = torch.zeros(B * N, 152, device=dev, dtype=torch.uint8); amn = torch.zeros_like(amx)
    L.call('gpe_edge_mlp_fwd', 0, PQ, 2 * H, jg, None, 0, B, N, k, H, H, w2p, b2, a2, H, part2, 0, None, None, None, None, 0, None, None, EWS, NWS, 0)
    L.call('gpe_edge_mlp_fwd', 1, None, 0, None, a2, H, B, N, k, H, Fo, w3p, b3, a3, 152, part3, 1, mx, mn, amx, amn, 152, None, None, EWS, NWS, 0)
    d2 = a2.clone()
    L.call('gpe_edge_mlp_bwd', dz3, 152, 0, None, 0, None, B, N, k, Fo, H, w3t, coef, d2, H, None, 0, None, None, EWS, NWS, None, 0, None, None, 0, None)
    return a2, a3, mx, part2.sum(0), d2


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


gpe_amd.set_math('f32'); ref = run()
gpe_amd.set_math('bf16x3'); got = run()
for name, r, g in zip(['a2', 'a3', 'mx', 'stats2', 'dz2'], ref, got):
    r, g = r.double(), g.double()
    print('%-7s scale %.3e  max abs err %.3e  rms err %.3e  (rel-to-rms %.2e)' % (
        name, r.pow(2).mean().sqrt().item(), (r - g).abs().max().item(), (r - g).pow(2).mean().sqrt().item(),
        ((r - g).pow(2).mean().sqrt() / r.pow(2).mean().sqrt()).item()))

a2 = torch.empty(E, H, device=dev); a3 = torch.empty(E, 152, device=dev)
part = torch.empty(nblk, 2, H, device=dev, dtype=torch.float64)
mx = torch.empty(B * N, 152, device=dev); mn = torch.empty_like(mx)
amx = torch.empty(B * N, 152, device=dev, dtype=torch.uint8); amn = torch.empty_like(amx)
def f2(): L.call('gpe_edge_mlp_fwd', 0, PQ, 2 * H, jg, None, 0, B, N, k, H, H, w2p, b2, a2, H, part, 0, None, None, None, None, 0, None, None, EWS, NWS, 0)
def f3(): L.call('gpe_edge_mlp_fwd', 1, None, 0, None, a2, H, B, N, k, H, Fo, w3p, b3, a3, 152, part, 1, mx, mn, amx, amn, 152, None, None, EWS, NWS, 0)
def b2a(): L.call('gpe_edge_mlp_bwd', a3, 152, 0, None, 0, None, B, N, k, Fo, H, w3t, coef, a2, H, None, 0, None, None, EWS, NWS, None, 0, None, None, 0, None)
for mode in ['f32', 'bf16x3']:
    gpe_amd.set_math(mode)
    for flags in [0, 16, 3]:
        L.query('gpe_debug_set', flags)
        print(mode, 'dbg=%d' % flags, ' '.join('%s %.3f ms' % (n, timeit(f)) for n, f in [('f2', f2), ('f3', f3), ('b2a', b2a)]))
L.query('gpe_debug_set', 0)

# determinism: the same launch twice must be bit-identical in both modes
for mode in ['f32', 'bf16x3']:
    gpe_amd.set_math(mode)
    r1 = run(); r2 = run()
    print(mode, 'bit-identical reruns:', [bool(torch.equal(a, b)) for a, b in zip(r1, r2)],
          [(a != b).sum().item() for a, b in zip(r1, r2)])

# ---- reduce-GEMM (weight gradients) ----------------------------------------------------------------------------------
G = torch.empty(H, H, device=dev); cs = torch.empty(H, device=dev)
ws = torch.empty(L.query('gpe_redgemm_ws', H, H), device=dev)
shift = torch.randn(H, device=dev)
a2r = torch.randn(E, H, device=dev); a3r = torch.randn(E, 152, device=dev); a3r[:, 150:] = 0
def rg(): L.call('gpe_edge_redgemm', a2r, H, 0, None, 0, PQ, 2 * H, jg, shift, B, N, k, H, H, G, H, cs, ws, None, None, EWS, NWS, None, 0, None, None, 0, None)
def rd(): L.call('gpe_edge_redgemm', a3r, 152, 1, a2r, H, None, 0, None, shift, B, N, k, Fo, H, G[:Fo], H, cs, ws, None, None, EWS, NWS, None, 0, None, None, 0, None)
res = {}
for mode in ['f32', 'bf16x3']:
    gpe_amd.set_math(mode)
    out = []
    for name, fn, rows in [('rg(gather)', rg, H), ('rd(dense)', rd, Fo)]:
        fn(); torch.cuda.synchronize()
        g1, c1 = G[:rows].clone(), cs[:rows].clone()
        fn(); torch.cuda.synchronize()
        assert torch.equal(g1, G[:rows]) and torch.equal(c1, cs[:rows]), 'non-deterministic ' + name
        res[(mode, name)] = (g1.double(), c1.double())
        out.append('%s %.3f ms' % (name, timeit(fn)))
    print(mode, ' '.join(out))
for name in ['rg(gather)', 'rd(dense)']:
    r, g_ = res[('f32', name)], res[('bf16x3', name)]
    print(name, 'G scale %.3e max abs diff %.3e (rel %.2e); colsum max diff %.3e' % (
        r[0].abs().max().item(), (r[0] - g_[0]).abs().max().item(), ((r[0] - g_[0]).abs().max() / r[0].abs().max()).item(),
        (r[1] - g_[1]).abs().max().item()))
# exact check of the gather variant against fp64 torch on a subset of columns
vfull = (torch.relu(PQ[:, :H].repeat_interleave(k, 0) + PQ[jg.view(-1).long(), H:]) - shift).double()
Gref = a2r.double().t() @ vfull
for mode in ['f32', 'bf16x3']:
    print(mode, 'rg vs fp64: max abs err %.3e of scale %.3e' % ((res[(mode, 'rg(gather)')][0] - Gref).abs().max().item(), Gref.abs().max().item()))
