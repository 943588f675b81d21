import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpe_amd
from gpe_amd import ops, _lib as L
B, N, k, H, Fo = 2, 128, 16, 200, 150
dev='cuda'
torch.manual_seed(0)
x = torch.randn(B*N, 3, device=dev)
idx, jg = ops.knn(x, B, N, k, want_global=True)
PQ = torch.randn(B*N, 2*H, device=dev)
W2 = torch.randn(H, H, device=dev)/14; b2 = torch.randn(H, device=dev)
W3 = torch.randn(Fo, H, device=dev)/14; b3 = torch.randn(Fo, device=dev)
E = B*N*k
a2 = torch.empty(E, H, device=dev); a3 = torch.empty(E, 152, device=dev)
nblk = L.query('gpe_stats_blocks')
part = torch.empty(nblk, 2, H, device=dev, dtype=torch.float64)
part3 = torch.empty(nblk, 2, Fo, device=dev, dtype=torch.float64)
mx = torch.empty(B*N, 152, device=dev); mn = torch.empty_like(mx)
amx = torch.empty(B*N, 152, device=dev, dtype=torch.uint8); amn = torch.empty_like(amx)
import os
L.query('gpe_debug_set', int(os.environ.get('GPE_DBG', '0')))
def step(name, fn):
    print(name, '...', flush=True); fn(); torch.cuda.synchronize(); print('   ok', flush=True)
step('F2', lambda: L.call('gpe_edge_mlp_fwd', 0, PQ, 2*H, jg, None, 0, B, N, k, H, H, ops.pack_weight(W2), b2, a2, H, part, 0, None, None, None, None, 0))
ref = torch.relu(torch.relu(PQ[:, :H].repeat_interleave(k, 0) + PQ[jg.view(-1).long(), H:]) @ W2.t() + b2)
print('F2 err', (a2-ref).abs().max().item(), 'stats', (part.sum(0)[0].float() - ref.sum(0)).abs().max().item())
step('F3', lambda: L.call('gpe_edge_mlp_fwd', 1, None, 0, None, a2, H, B, N, k, H, Fo, ops.pack_weight(W3), b3, a3, 152, part3, 1, mx, mn, amx, amn, 152))
ref3 = torch.relu(a2 @ W3.t() + b3)
print('F3 err', (a3[:, :Fo]-ref3).abs().max().item(), 'mx', (mx[:, :Fo] - ref3.view(B*N, k, Fo).max(1).values).abs().max().item())
coef = torch.randn(4, H, device=dev)
a2c = a2.clone()
step('B2a', lambda: L.call('gpe_edge_mlp_bwd', a3, 152, 0, None, 0, None, B, N, k, Fo, H, ops.pack_weight(W3, transpose=True), coef, a2c, H, None, 0))
dPQ = torch.empty(B*N, 2*H, device=dev)
a2d = a2.clone()
step('B1a', lambda: L.call('gpe_edge_mlp_bwd', a2, H, 1, PQ, 2*H, jg, B, N, k, H, H, ops.pack_weight(W2, transpose=True), coef, a2d, H, dPQ, 2*H))
print('done')
