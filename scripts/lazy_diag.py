"""Diagnostic (GPU): lazy dz3 consumers against the eager dz3 pass + the same consumers, by direct C-ABI calls."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpe_amd as gpe
ops, L = gpe.ops, gpe._lib
gpe.set_math('f16x3')
def rel(a, b): return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300)).item()
B, N, k, F, Cp = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 512, 16, 150, 200
E, BN = B * N * k, B * N
g = torch.Generator().manual_seed(3)
a3 = torch.relu(torch.randn(E, 152, generator=g)).cuda(); a3[:, F:] = 0
gp = torch.zeros(BN, 152); gp[:, :F] = torch.randn(BN, F, generator=g) * 1e-3; gp = gp.cuda()
amx = torch.randint(0, k, (BN, 152), generator=g).to(torch.uint8).cuda()
amn = torch.randint(0, k, (BN, 152), generator=g).to(torch.uint8).cuda()
coef = torch.randn(4, F, generator=g); coef[1] *= 1e-5; coef[2] *= 1e-5; coef = coef.cuda().contiguous()
a2 = torch.randn(E, Cp, generator=g).abs().cuda()
shift = torch.randn(Cp, generator=g).abs().cuda()
W = (torch.randn(F, Cp, generator=g) * 0.1).cuda()
coef_p = torch.randn(4, Cp, generator=g); coef_p[1] *= 1e-5; coef_p[2] *= 1e-5; coef_p = coef_p.cuda().contiguous()
ws, nws = ops.edge_workspace(B, N, k, 2 * Cp, 'cuda')
part = torch.empty(L.query('gpe_redgemm_ws', F, Cp)).cuda()
words = torch.zeros(8, dtype=torch.int32, device='cuda')
print('lazy ok', L.query('gpe_edge_lazy_dz3_ok', B, N, k, F, Cp))
# fp64 dz3
slot = torch.arange(k, device='cuda').repeat(BN).view(E, 1)
sel = torch.where(coef[0] >= 0, amx[:, :F].long(), amn[:, :F].long()).repeat_interleave(k, 0)
sg = (coef[0].double() * gp[:, :F].double()).repeat_interleave(k, 0)
a3d = a3[:, :F].double()
dzr = torch.where(a3d > 0, torch.where(sel == slot, sg, torch.zeros_like(sg)) - coef[1].double() - (a3d - coef[3].double()) * coef[2].double(), torch.zeros_like(sg))
# eager
dz = a3.clone()
L.call('gpe_edge_dz3', dz, 152, gp, 152, amx, amn, 152, coef, B, N, k, F, words[0:1])
print('eager dz3 vs fp64', rel(dz[:, :F], dzr))
L.call('gpe_absmax', a2, Cp, E, Cp, words[1:2])
L.call('gpe_absmax', a3, 152, E, F, words[2:3])
Ge, cse = torch.zeros(F, Cp).cuda(), torch.zeros(F).cuda()
L.call('gpe_edge_redgemm', dz, 152, 1, a2, Cp, None, 0, None, shift, B, N, k, F, Cp, Ge, Cp, cse, part, words[0:1], words[1:2], ws, nws,
       None, 0, None, None, 0, None)
Gr = dzr.t() @ (a2.double() - shift.double())
print('eager redgemm G', rel(Ge, Gr), 'cs', rel(cse, dzr.sum(0)))
# lazy (reads the activation as the fp16 tensor the forward stores with out_half = 1)
a3h = a3.half()
# the reference for the lazy consumers: dz3 from the ROUNDED activation
a3d = a3h[:, :F].double()
dzr = torch.where(a3d > 0, torch.where(sel == slot, sg, torch.zeros_like(sg)) - coef[1].double() - (a3d - coef[3].double()) * coef[2].double(), torch.zeros_like(sg))
Gr = dzr.t() @ (a2.double() - shift.double())
gu = gp[:, :F].contiguous()                                   # the gradient as autograd hands it over: pitch F, no padding
stats = torch.zeros(4, F, device='cuda'); stats[2] = coef[0]; stats[1] = 1
mxx = torch.zeros(BN, 152, device='cuda')
psb = L.query('gpe_point_sums_blocks')
pp = torch.empty(psb, 2, F, device='cuda', dtype=torch.float64)
L.call('gpe_edge_bwd_point_sums', gu, F, mxx, mxx, 152, stats, BN, F, pp, words[4:5])
print('max|s g| word', torch.tensor([words[4].item()], dtype=torch.int32).view(torch.float32).item(), 'true', (coef[0] * gu).abs().max().item())
L.call('gpe_edge_dz3_bound', words[4:5], coef, F, words[2:3], words[3:4])
wv = lambda w: torch.tensor([w.item()], dtype=torch.int32).view(torch.float32).item()
print('bound', wv(words[3]), 'measured', wv(words[0]), 'true', dzr.abs().max().item())
Gl, csl = torch.zeros(F, Cp).cuda(), torch.zeros(F).cuda()
L.call('gpe_edge_redgemm', a3h, 152, 1, a2, Cp, None, 0, None, shift, B, N, k, F, Cp, Gl, Cp, csl, part, words[3:4], words[1:2], ws, nws,
       gu, F, amx, amn, 152, coef)
print('lazy  redgemm G', rel(Gl, Gr), 'cs', rel(csl, dzr.sum(0)))
# propagation
wt = ops.pack_weight(W, transpose=True)
oe = a2.clone(); ol = a2.clone()
L.call('gpe_edge_mlp_bwd', dz, 152, 0, None, 0, None, B, N, k, F, Cp, wt, coef_p, oe, Cp, None, 0, words[0:1], words[5:6], ws, nws,
       None, 0, None, None, 0, None)
L.call('gpe_edge_mlp_bwd', a3h, 152, 0, None, 0, None, B, N, k, F, Cp, wt, coef_p, ol, Cp, None, 0, words[3:4], words[6:7], ws, nws,
       gu, F, amx, amn, 152, coef)
y = dzr @ W.double()
ref = torch.where(a2.double() > 0, coef_p[0].double() * y - coef_p[1].double() - (a2.double() - coef_p[3].double()) * coef_p[2].double(), torch.zeros_like(y))
print('eager bwd', rel(oe, ref), 'lazy bwd', rel(ol, ref), 'amax words', wv(words[5]), wv(words[6]), ref.abs().max().item())
bad = ((ol.double() - ref).abs().view(BN, k, Cp).amax(2) > 1e-3 * ref.abs().max())
print('bad rows per slot', bad.sum(0).tolist(), 'bad points', bad.any(1).sum().item(), 'first bad points', bad.any(1).nonzero().view(-1)[:10].tolist())
# which tensor did the lazy reduce-GEMM sum?  candidates for its column sums
cand = {'dz3': dzr.sum(0), 'raw a3': a3d.sum(0),
        'no hit': torch.where(a3d > 0, -coef[1].double() - (a3d - coef[3].double()) * coef[2].double(), torch.zeros_like(a3d)).sum(0),
        'relu mask off': (torch.where(sel == slot, sg, torch.zeros_like(sg)) - coef[1].double() - (a3d - coef[3].double()) * coef[2].double()).sum(0)}
for n, v in cand.items():
    print('cs vs %-14s %.3e' % (n, rel(csl, v)))
print('cs lazy ', csl[:8].tolist(), csl[140:150].tolist())
print('cs ref  ', dzr.sum(0)[:8].tolist(), dzr.sum(0)[140:150].tolist())
print('cs a3   ', a3d.sum(0)[:8].tolist())
