"""Debug helper for the f16x3 mode: runs one EdgeConv layer forward + backward and reports, per C-ABI call, whether any
float tensor argument holds a non-finite value AFTER the call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpe_amd
from gpe_amd import _lib as L

B, N, C, H, Fo, k = 2, 128, 3, 200, 150, 16
mode = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
gpe_amd.set_math(mode)
torch.manual_seed(0)
from oracle import ref_path as O   # debug script only
torch.manual_seed(B + N + C)
oconv = O.DynamicEdgeConv(O.MLP([2 * C, H, H, Fo]), k=k)
with torch.no_grad():
    for blk in oconv.nn:
        blk[2].weight.uniform_(0.5, 1.5)
        blk[2].bias.uniform_(-0.3, 0.3)
    oconv.nn[2][2].weight[::5] *= -1
conv = gpe_amd.net_blocks.DynamicEdgeConv(gpe_amd.net_blocks.MLP([2 * C, H, H, Fo]), k=k)
conv.load_state_dict(oconv.state_dict())
conv = conv.cuda().train()
g = torch.Generator().manual_seed(1)
x = torch.randn(B * N, C, generator=g).cuda().requires_grad_()
wgt = torch.randn(B * N, Fo, generator=g).cuda()
SYNC = os.environ.get('H3_SYNC', '1') == '1'
orig = L.call
def call(name, *args):
    rc = orig(name, *args)
    if not SYNC:
        return rc
    torch.cuda.synchronize()
    bad = []
    for i, a in enumerate(args):
        if torch.is_tensor(a) and a.is_floating_point() and a.numel() and not torch.isfinite(a).all():
            bad.append((i, tuple(a.shape), int((~torch.isfinite(a)).sum())))
    print('%-28s %s' % (name, ('NONFINITE ' + str(bad)) if bad else 'ok'))
    return rc
L.call = call
gpe_amd.ops.L.call = call
out = conv(x, B, N)
print('--- backward')
(out * wgt).sum().backward()
print('dx finite', torch.isfinite(x.grad).all().item(), [(n, torch.isfinite(p.grad).all().item()) for n, p in conv.named_parameters() if not torch.isfinite(p.grad).all()])
