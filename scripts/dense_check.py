import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpe_amd as gpe
from oracle import ref_path as O
M, chans = 1000, [153, 153, 153, 23]
torch.manual_seed(M)
omlp = O.MLP(chans)
with torch.no_grad():
    for blk in omlp:
        blk[2].weight.uniform_(0.5, 1.5); blk[2].bias.uniform_(-0.3, 0.3)
    omlp[-1][2].weight[::3] *= -1
x = torch.randn(M, chans[0], generator=torch.Generator().manual_seed(1))
wgt = torch.randn(M, chans[-1], generator=torch.Generator().manual_seed(2))
o64 = copy.deepcopy(omlp).double().train()
xr = x.double().requires_grad_()
yr = o64(xr); (yr * wgt.double()).sum().backward()
for mode in ['f32', 'bf16x3']:
    gpe.set_math(mode)
    pmlp = gpe.net_blocks.MLP(chans); pmlp.load_state_dict(omlp.state_dict()); pmlp = pmlp.cuda().train()
    xd = x.cuda().requires_grad_()
    y = gpe.ops.dense_mlp(xd, pmlp, True); (y * wgt.cuda()).sum().backward()
    e = (xd.grad.cpu().double() - xr.grad).abs()
    sc = xr.grad.abs().max()
    rows = (e.amax(1) > 3e-4 * sc).nonzero().flatten().tolist()
    print(mode, 'fwd max err', (y.cpu().double() - yr).abs().max().item(), 'grad max err/scale', (e.max() / sc).item(), 'bad rows', rows[:10], len(rows),
          'median row err', (e.amax(1).median() / sc).item())
