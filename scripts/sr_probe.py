"""Where the single-role edge kernels spend their time: the shipped layer-2 EdgeConv block (B=32, N=2048, k=16,
150 -> 200 -> 200 -> 150) timed per C-ABI call with the kernel's phase switches (gpe_debug_set bits: 1 = no operand
staging, 2 = no epilogue).  Outputs are garbage with a switch on; only the durations are meaningful."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gpe = importlib.import_module('garment-pattern-estimation_amd')
_lib = importlib.import_module('garment-pattern-estimation_amd._lib')
B, N, k, C, H, F = 32, 2048, 16, 150, 200, 150
torch.manual_seed(0)
conv = gpe.net_blocks.DynamicEdgeConv(gpe.net_blocks.MLP([2 * C, H, H, F]), k=k).cuda().train()
x = torch.randn(B * N, C, device='cuda', requires_grad=True)


def run():
    y = conv(x, B, N)
    y.square().mean().backward()


for _ in range(2):
    run()
for dbg in (0, 1, 2, 3):
    _lib.lib().gpe_debug_set(dbg)
    run()
    torch.cuda.synchronize()
    _lib.TIMING = []
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    rec, _lib.TIMING = _lib.TIMING, None
    agg = {}
    for name, ints, e0, e1 in rec:
        if name.startswith('gpe_edge_mlp'):
            d = agg.setdefault((name, ints[:2] + ints[-4:]), [0, 0.0])
            d[0] += 1
            d[1] += e0.elapsed_time(e1)
    print('dbg=%d' % dbg, '  '.join('%s%s %.0f us' % (n[0][9:], n[1][-3:], 1e3 * v[1] / v[0]) for n, v in sorted(agg.items())))
_lib.lib().gpe_debug_set(0)
