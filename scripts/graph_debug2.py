import sys, ctypes
import torch
import gpe_amd
from gpe_amd import _lib as L, ops, optim, graph, configs, nets
import bench
dev = torch.device('cuda', 0)
gpe_amd.set_math(sys.argv[1] if len(sys.argv) > 1 else 'f32')
gpe_amd.set_f16x3_min_rows(0)
data_config = configs.data_config()
nn_cfg = configs.lstm_model_config(k_neighbors=5)
torch.manual_seed(0)
model = nets.GarmentFullPattern3D(data_config, dict(nn_cfg), dict(nn_cfg['loss'])).to(dev).train()
model.loss.with_quality_eval = False
feats, gt = bench.synthetic(4, 256, data_config, seed=1000, device=dev)
opt = optim.FusedAdam(optim.FlatArena(model), lr=2e-3)
host = graph.HostDrawn(dev)
s = torch.cuda.Stream()
def stage(upto):
    preds = model(feats)
    if upto == 'fwd': return
    loss = model.loss(preds, gt, epoch=0)[0]
    if upto == 'loss': return
    loss.backward()
    if upto == 'bwd':
        opt.arena.begin_step(); return
    opt.step()
for upto in ('fwd', 'loss', 'bwd'):
    with torch.cuda.stream(s):
        ops.HOST_DRAWN = host
        for i in range(2):
            host.begin_step(); stage('all')
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        host.begin_step(); host.capturing = True
        class C: guards = []
        ops.CAPTURE = C()
        try:
            with torch.cuda.graph(g, stream=s):
                stage(upto)
            print(upto, 'captured ok')
        except Exception as e:
            print(upto, 'FAILED', str(e)[:90])
            break
        finally:
            ops.CAPTURE = None; host.capturing = False; ops.HOST_DRAWN = None
            opt.arena.begin_step()
