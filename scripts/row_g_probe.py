"""Row g, measured: how much gradient accuracy does the backward lose when the stored activation of the aggregated
block (a3) is kept in fp16 instead of fp32?  With lazy dz3 its only readers are the two kernels that form
dz3 = (a3>0) ? hit - c1 - (a3-mean)*k2 : 0 from it, so a 16-bit a3 would take 0.96 GB per layer off the step (write in the
forward, two reads in the backward).  This probe EMULATES the storage: it rounds the saved a3 to fp16 between forward and backward
(positive values that underflow are kept at the smallest subnormal so the ReLU mask is unchanged) and compares the layer's
gradients with the unrounded run and with the fp64 oracle.  No product code stores fp16; this is the measurement the decision
rests on."""
import os, sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpe_amd as gpe
from oracle import ref_path as O
L = gpe._lib

def rel_max(a, b): return ((a.double().cpu() - b.double().cpu()).abs().max() / (b.double().abs().max() + 1e-300)).item()
def rel_fro(a, b): return ((a.double().cpu() - b.double().cpu()).norm() / (b.double().norm() + 1e-300)).item()

B, N, k = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (4, 2048, 16)))
for C in (3, 150):                                               # layer 1 and layer 2 of the encoder
    torch.manual_seed(11 + C)
    oconv = O.DynamicEdgeConv(O.MLP([2 * C, 200, 200, 150]), k=k)
    with torch.no_grad():
        for blk in oconv.nn:
            blk[2].weight.uniform_(0.5, 1.5); blk[2].bias.uniform_(-0.3, 0.3)
        oconv.nn[2][2].weight[::5] *= -1
    conv = gpe.net_blocks.DynamicEdgeConv(gpe.net_blocks.MLP([2 * C, 200, 200, 150]), k=k)
    conv.load_state_dict(oconv.state_dict())
    conv = conv.cuda().train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B * N, C, generator=g)
    wgt = torch.randn(B * N, 150, generator=g)
    gpe.set_math('f16x3')
    assert L.query('gpe_edge_lazy_dz3_ok', B, N, k, 150, 200) == 1
    res = {}
    for mode in ('fp32 a3', 'fp16 a3', 'bf16 a3'):
        for p_ in conv.parameters(): p_.grad = None
        xd = x.cuda().requires_grad_()
        y = conv(xd, B, N)
        if mode != 'fp32 a3':
            sv = [t for t in y.grad_fn.saved_tensors if t is not None and t.dim() == 2 and t.shape[0] == B * N * k and t.shape[1] == 152]
            assert len(sv) == 1, [tuple(t.shape) for t in y.grad_fn.saved_tensors if t is not None]
            a3 = sv[0]
            if True:
                lo = a3.half() if mode == 'fp16 a3' else a3.bfloat16()
                r = lo.float()
                tiny = 2.0 ** -24 if mode == 'fp16 a3' else 1e-38
                r = torch.where((a3 > 0) & (r == 0), torch.full_like(r, tiny), r)
                a3.data.copy_(r)                                  # (.data: the autograd version counter must not see the emulation)
        (y * wgt.cuda()).sum().backward()
        res[mode] = (xd.grad.clone(), {n: p_.grad.clone() for n, p_ in conv.named_parameters()})
    o64 = copy.deepcopy(oconv).double().train()
    o64.knn_override = conv.last_knn.cpu().view(-1, k).long()
    xr = x.double().requires_grad_()
    yr = o64(xr, torch.arange(B).repeat_interleave(N))
    (yr * wgt.double()).sum().backward()
    ref = dict(o64.named_parameters())
    print('C_in=%d  B=%d N=%d k=%d  (max-norm | Frobenius relative error against the fp64 oracle)' % (C, B, N, k))
    for mode in res:
        dx, gp = res[mode]
        worst = max(rel_fro(gp[n], ref[n].grad) for n in gp)
        worstm = max(rel_max(gp[n], ref[n].grad) for n in gp)
        print('  %-8s dx %.2e | %.2e   worst parameter gradient %.2e | %.2e   vs fp32-a3 run: dx %.2e, params %.2e' % (
            mode, rel_max(dx, xr.grad), rel_fro(dx, xr.grad), worstm, worst, rel_max(dx, res['fp32 a3'][0]),
            max(rel_max(gp[n], res['fp32 a3'][1][n]) for n in gp)), flush=True)
