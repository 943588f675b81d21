import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpe_amd
from gpe_amd import ops
B, N, k = 32, 2048, 16
for C in (3, 150):
    x = torch.randn(B * N, C, device='cuda')
    for _ in range(3):
        ops.knn(x, B, N, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.knn(x, B, N, k)
    e1.record(); torch.cuda.synchronize()
    print('C=%d  %.3f ms' % (C, e0.elapsed_time(e1) / 5))
