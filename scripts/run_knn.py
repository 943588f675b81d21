"""kNN timing at the layer shapes of BASELINE cfg 2 (B=32, N=2048, k=16): C=3 (contiguous) and C=150 in 152-float rows (the
layer-2 feature tensor).  GPE_KNN_EXACT=1 forces the all-exact kernel.  argv: [N] [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpe_amd
from gpe_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
k = 16
torch.manual_seed(0)
for C, ld in ((3, 3), (150, 152)):
    buf = torch.randn(B * N, ld, device='cuda')
    x = buf[:, :C]
    for _ in range(3):
        ops.knn(x, B, N, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.knn(x, B, N, k)
    e1.record(); torch.cuda.synchronize()
    print('C=%d ld=%d N=%d B=%d  %.3f ms' % (C, ld, N, B, e0.elapsed_time(e1) / 5))
