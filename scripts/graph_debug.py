"""Which call invalidates a stream capture: runs a small captured step with every C-ABI call followed by hipStreamIsCapturing."""
import ctypes, sys, copy
import torch
import gpe_amd
from gpe_amd import _lib as L, ops, optim, graph, configs, nets
import bench
hip = ctypes.CDLL('libamdhip64.so')
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
orig = L.call
state = {'bad': None}
def status():
    st = ctypes.c_int(0)
    h = torch.cuda.current_stream().cuda_stream
    hip.hipStreamIsCapturing(ctypes.c_void_p(h), ctypes.byref(st))
    return st.value
def call(name, *a):
    before = status()
    orig(name, *a)
    after = status()
    if before == 1 and after != 1 and state['bad'] is None:
        state['bad'] = name
        print('capture invalidated by', name, [x if not isinstance(x, torch.Tensor) else tuple(x.shape) for x in a][:12])
L.call = call
ops.L.call = call
import torch.utils._python_dispatch as pd
class Mode(pd.TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        before = status()
        out = func(*args, **(kwargs or {}))
        after = status()
        if before == 1 and after != 1 and state['bad'] is None:
            state['bad'] = str(func)
            print('capture invalidated by torch op', func)
        return out
sg = graph.StepGraph(lambda f, g: model.loss(model(f), g, epoch=0)[0], opt, warmup=2)
try:
    with Mode():
        for i in range(4):
            torch.manual_seed(i)
            sg.step(feats, gt)
    torch.cuda.synchronize()
    print('ok', sg.captures, sg.replays)
except Exception as e:
    print('failed:', type(e).__name__, str(e)[:200])
