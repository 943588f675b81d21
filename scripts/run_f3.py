import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpe_amd
from gpe_amd import ops, _lib as L
B, N, k, H, Fo = 32, 2048, 16, 200, 150
dev='cuda'
E = B*N*k
a2 = torch.randn(E, H, device=dev).relu_(); a3 = torch.empty(E, 152, device=dev)
W3 = torch.randn(Fo, H, device=dev)/14; b3 = torch.randn(Fo, device=dev)
nblk = L.query('gpe_stats_blocks')
part = torch.empty(nblk, 2, Fo, device=dev, dtype=torch.float64)
mx = torch.empty(B*N, 152, device=dev); mn = torch.empty_like(mx)
amx = torch.empty(B*N, 152, device=dev, dtype=torch.uint8); amn = torch.empty_like(amx)
w3p = ops.pack_weight(W3)
for flags in (0, 3, 16):
    L.query('gpe_debug_set', flags)
    for _ in range(3):
        L.call('gpe_edge_mlp_fwd', 1, None, 0, None, a2, H, B, N, k, H, Fo, w3p, b3, a3, 152, part, 1, mx, mn, amx, amn, 152)
    torch.cuda.synchronize()
