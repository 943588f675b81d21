"""Two identical training-mode steps must give bit-identical outputs and gradients (every kernel is deterministic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpe_amd as gpe

mode = sys.argv[1] if len(sys.argv) > 1 else 'f32'
gpe.set_math(mode)
cfg = gpe.configs
nn_cfg = cfg.lstm_model_config(k_neighbors=16)
torch.manual_seed(0)
model = gpe.nets.GarmentFullPattern3D(cfg.data_config(), dict(nn_cfg), dict(nn_cfg['loss'])).cuda().train()
g = torch.Generator().manual_seed(5)
feats = torch.randn(4, 2048, 3, generator=g).cuda()
outs = []
for _ in range(3):
    torch.manual_seed(3)
    p = model(feats)
    (p['outlines'].square().mean() + p['rotations'].square().mean()).backward()
    outs.append({n: q.grad.clone() for n, q in model.named_parameters() if q.grad is not None})
    outs[-1]['__out'] = p['outlines'].detach().clone()
    model.zero_grad(set_to_none=True)
for n in outs[0]:
    d = [(outs[0][n] != outs[i][n]).sum().item() for i in (1, 2)]
    if any(d):
        print(mode, 'DIFFERS', n, tuple(outs[0][n].shape), d)
print(mode, 'checked', len(outs[0]))
