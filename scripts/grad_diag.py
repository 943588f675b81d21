"""Where does the encoder-gradient error of a fixture come from?  Runs the product on a golden fixture with every C-ABI
call of the EdgeConv backward snapshotted, runs the fp64 oracle with hooks on the same graphs, and prints, per EdgeConv
layer and MLP block, the relative error of dz_l, G_l, db_l, the BN-backward sums and the coefficients.
usage: python scripts/grad_diag.py [fixture tag]"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gpe_amd
from gpe_amd import _lib as L
from oracle import ref_path as O

tag = sys.argv[1] if len(sys.argv) > 1 else 'full3d_k16'
fx = torch.load(os.path.join('tests', 'golden', tag + '.pt'), weights_only=False)

CAP = []            # (name, [cloned tensor args])
WATCH = {'gpe_edge_bwd_point_sums', 'gpe_bn_bwd_coef', 'gpe_edge_dz3', 'gpe_edge_redgemm', 'gpe_bn_bwd_from_G',
         'gpe_edge_mlp_bwd', 'gpe_edge_pull_dq', 'gpe_redgemm', 'gpe_w1_grad_from_pq', 'gpe_edge_mlp_fwd'}
orig_call = L.call


def spy(name, *args):
    orig_call(name, *args)
    if name in WATCH:
        torch.cuda.synchronize()
        CAP.append((name, [a.detach().clone() if isinstance(a, torch.Tensor) else a for a in args]))


L.call = spy
import gpe_amd.ops as ops
ops.L.call = spy

torch.manual_seed(fx['seed'])
model = getattr(gpe_amd.nets, fx['model'])(fx['data_config'], copy.deepcopy(fx['nn_config']),
                                           copy.deepcopy(fx['loss_config'])).cuda().train()
torch.manual_seed(fx['seed'] + 2)
preds = model(fx['features'].cuda(), log_step=0, epoch=0)
loss, _, _ = model.loss(preds, {k: v.clone() for k, v in fx['gt'].items()}, epoch=0)
loss.backward()
torch.cuda.synchronize()
knn = [c.last_knn.cpu().view(-1, c.k).long() for c in model.feature_extractor.conv_layers]


def run_oracle(dtype):
    torch.manual_seed(fx['seed'])
    o = getattr(O, fx['model'])(fx['data_config'], copy.deepcopy(fx['nn_config']), copy.deepcopy(fx['loss_config']))
    o = (o.double() if dtype == torch.float64 else o).train()
    rec = {}
    for li, conv in enumerate(o.feature_extractor.conv_layers):
        conv.knn_override = knn[li]
        for bi, blk in enumerate(conv.nn):
            def mk(li, bi):
                def hook_lin(m, i, out):
                    out.retain_grad(); rec[(li, bi, 'z')] = out
                def hook_relu(m, i, out):
                    out.retain_grad(); rec[(li, bi, 'a')] = out
                def hook_bn(m, i, out):
                    out.retain_grad(); rec[(li, bi, 'y')] = out
                return hook_lin, hook_relu, hook_bn
            h1, h2, h3 = mk(li, bi)
            blk[0].register_forward_hook(h1); blk[1].register_forward_hook(h2); blk[2].register_forward_hook(h3)
    torch.manual_seed(fx['seed'] + 2)
    feats = fx['features'].double() if dtype == torch.float64 else fx['features']
    p = o(feats)
    l, _, _ = o.loss(p, {k: v.clone() for k, v in fx['gt'].items()}, epoch=0)
    l.backward()
    return o, rec


o64, r64 = run_oracle(torch.float64)
o32, r32 = run_oracle(torch.float32)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-300)).item()


# ---- forward activations: error and ReLU-mask flips vs fp64 -------------------------------------------------
fl, fb = 0, 0
for name, a in CAP:
    if name != 'gpe_edge_mlp_fwd':
        continue
    fb += 1
    if fb == len(o64.feature_extractor.conv_layers[0].nn):
        fl, fb = fl + 1, 1
    act_b = a[13]
    a64, a32 = r64[(fl, fb, 'a')], r32[(fl, fb, 'a')]
    z64 = r64[(fl, fb, 'z')]
    C = a64.shape[1]
    ab = act_b[:, :C].double().cpu()
    flips_b = ((ab > 0) != (a64 > 0)).sum().item()
    flips_32 = ((a32 > 0) != (a64 > 0)).sum().item()
    print('fwd layer %d block %d: a err build %.2e | torch32 %.2e (abs, max|a| %.2e); mask flips build %d | torch32 %d of %d; '
          'mean/std max %.1f' % (fl, fb, (ab - a64).abs().max().item(), (a32.double() - a64).abs().max().item(),
                                 a64.abs().max().item(), flips_b, flips_32, a64.numel(),
                                 (r64[(fl, fb - 1, 'a')].mean(0).abs() / r64[(fl, fb - 1, 'a')].std(0)).max().item()))

# ---- walk the captured calls: backward runs layer nl-1 first -------------------------------------------
nl = len(model.feature_extractor.conv_layers)
layer = nl
blk = None
nb = len(o64.feature_extractor.conv_layers[0].nn)
B, N = fx['features'].shape[:2]
k = model.feature_extractor.conv_layers[0].k
E = B * N * k
print('fixture %s  B=%d N=%d k=%d E=%d  blocks/layer=%d' % (tag, B, N, k, E, nb))
for name, a in CAP:
    if name == 'gpe_edge_bwd_point_sums':
        layer -= 1
        blk = nb - 1
        print('=== layer %d ===' % layer)
    if layer >= nl or layer < 0:
        continue
    key = lambda b, w: (layer, b, w)
    if name == 'gpe_edge_dz3':
        dz = a[0]
        C = r64[key(nb - 1, 'z')].shape[1]
        print('  dz_%d (in place, dz3 kernel): build %.2e | torch32 %.2e' % (
            nb - 1, rel(dz[:, :C], r64[key(nb - 1, 'z')].grad), rel(r32[key(nb - 1, 'z')].grad, r64[key(nb - 1, 'z')].grad)))
    elif name == 'gpe_bn_bwd_coef':
        part, nblk_, stats, C, count, coef = a[0], a[1], a[2], a[3], a[4], a[5]
        s = part.double().sum(0)                       # [2, C]: sum dy, sum dy*xhat
        # oracle: dy = grad wrt the BN output; xhat from the oracle's activation
        b_of = blk if blk is not None else nb - 1
        # which BN? after point_sums -> last BN (sums over points of the aggregated rows); after from_G -> BN of block blk-1
        print('  bn_bwd_coef (C=%d count=%g): sums captured' % (C, count), 'blk', b_of)
        y, act = r64[key(b_of, 'y')], r64[key(b_of, 'a')]
        dy = y.grad
        mean = act.mean(0)
        var = act.var(0, unbiased=False)
        bn = o64.feature_extractor.conv_layers[layer].nn[b_of][2]
        xhat = (act - mean) / torch.sqrt(var + bn.eps)
        S1, S2 = dy.sum(0), (dy * xhat).sum(0)
        y32, act32 = r32[key(b_of, 'y')], r32[key(b_of, 'a')]
        m32 = act32.mean(0); v32 = act32.var(0, unbiased=False)
        xh32 = (act32 - m32) / torch.sqrt(v32 + bn.eps)
        S1_32, S2_32 = y32.grad.sum(0), (y32.grad * xh32).sum(0)
        print('     sum dy      : build %.2e | torch32 %.2e   (max|S1| %.3e, sum|dy| max %.3e)' % (
            rel(s[0], S1), rel(S1_32, S1), S1.abs().max().item(), dy.abs().sum(0).max().item()))
        print('     sum dy*xhat : build %.2e | torch32 %.2e   (max|S2| %.3e)' % (
            rel(s[1], S2), rel(S2_32, S2), S2.abs().max().item()))
        print('     mean / rstd in stats: %.2e / %.2e' % (
            rel(stats[0], mean), rel(stats[1], 1 / torch.sqrt(var + bn.eps))))
    elif name == 'gpe_edge_redgemm':
        dzin = a[0]
        Cl, Cp = a[12], a[13]
        G, db = a[14], a[16]
        dz_o = r64[key(blk, 'z')].grad
        act = r64[key(blk - 1, 'a')]
        Gt = dz_o.t() @ (act - act.mean(0))
        dz32 = r32[key(blk, 'z')].grad
        act32 = r32[key(blk - 1, 'a')]
        G32 = dz32.t() @ (act32 - act32.mean(0))
        print('  block %d redgemm: dz_in %.2e | G %.2e (torch32-style %.2e) | db %.2e (torch32 %.2e)  max|G| %.3e max|db| %.3e' % (
            blk, rel(dzin[:, :Cl], dz_o), rel(G, Gt), rel(G32, Gt), rel(db, dz_o.sum(0)), rel(dz32.sum(0), dz_o.sum(0)),
            Gt.abs().max().item(), dz_o.sum(0).abs().max().item()))
        # the SAME product from the build's own dz / a in fp64: isolates the MFMA accumulation from the input error
        Gb = dzin[:, :Cl].double().cpu().t() @ (act - act.mean(0))
        print('     G vs fp64 product of the BUILD\'s dz with the oracle activation: %.2e' % rel(G, Gb))
    elif name == 'gpe_bn_bwd_from_G':
        sums = a[8]
        dw = a[9]
        lin = o64.feature_extractor.conv_layers[layer].nn[blk][0]
        lin32 = o32.feature_extractor.conv_layers[layer].nn[blk][0]
        print('  block %d dW %.2e (torch32 %.2e)' % (blk, rel(dw, lin.weight.grad), rel(lin32.weight.grad, lin.weight.grad)))
        blk -= 1
    elif name == 'gpe_edge_mlp_bwd':
        dst = a[13]
        Cp = a[10]
        dz_o = r64[key(blk, 'z')].grad
        print('  dz_%d after mlp_bwd: build %.2e | torch32 %.2e   max|dz| %.3e' % (
            blk, rel(dst[:, :Cp], dz_o), rel(r32[key(blk, 'z')].grad, dz_o), dz_o.abs().max().item()))
        mb, mo = dst[:, :Cp].cpu() != 0, dz_o != 0
        same = mb == mo
        d = (dst[:, :Cp].double().cpu() - dz_o).abs()
        print('     zero-pattern mismatches %d; max err on matching elements %.2e; on mismatching %.2e (rel max|dz|)' % (
            (~same).sum().item(), (d * same).max().item() / dz_o.abs().max().item(), (d * ~same).max().item() / dz_o.abs().max().item()))
        print('     colsum(dz_%d): build-from-its-dz %.2e | torch32 %.2e  (max|colsum| %.3e, max colsum|dz| %.3e)' % (
            blk, rel(dst[:, :Cp].double().sum(0), dz_o.sum(0)), rel(r32[key(blk, 'z')].grad.sum(0), dz_o.sum(0)),
            dz_o.sum(0).abs().max().item(), dz_o.abs().sum(0).max().item()))

print('--- final parameter gradients (encoder) ---')
pn = dict(model.named_parameters())
g32 = dict(o32.named_parameters())
for n, p in o64.named_parameters():
    if not n.startswith('feature_extractor') or p.grad is None:
        continue
    print('%-55s build %.2e | torch32 %.2e | max|g| %.2e' % (n, rel(pn[n].grad, p.grad), rel(g32[n].grad, p.grad),
                                                           p.grad.abs().max().item()))
