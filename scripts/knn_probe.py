"""Phase breakdown of gpe_knn_kernel at the shipped sizes (GPE_KNN_PROBE bits: 1 no selection after the first tile,
2 no staging after the first step, 4 no distance arithmetic).  Run as one process per setting:
    for p in 0 1 2 4 3 6 7; do GPE_KNN_PROBE=$p python scripts/knn_probe.py; done"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gpe = importlib.import_module('garment-pattern-estimation_amd')
B, N, k = int(os.environ.get('KNN_PROBE_B', 32)), int(os.environ.get('KNN_PROBE_N', 2000)), 16
for C in (3, 150):
    ld = (C + 3) // 4 * 4 if C >= 32 else C        # the training step's feature rows are padded to 4 columns
    x = torch.randn(B * N, ld, device='cuda')[:, :C]
    for _ in range(3): gpe.ops.knn(x, B, N, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): gpe.ops.knn(x, B, N, k)
    e1.record(); torch.cuda.synchronize()
    print('probe=%s C=%d  %.3f ms' % (os.environ.get('GPE_KNN_PROBE', '0'), C, e0.elapsed_time(e1) / 10))
