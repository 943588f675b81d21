"""gpe_amd/graph.py StepGraph: a training step captured once as a hipGraph and replayed.  The replayed steps must be THE SAME steps:
same CPU-generator draws for the LSTM start states (nn/net_blocks.py:391-392), same Adam schedule, same numbers bit for bit."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gpe():
    if not torch.cuda.is_available():
        pytest.skip('needs the MI355X')
    import gpe_amd
    return gpe_amd


def _setup(gpe, model_kind='lstm', k=5, batch=4, points=256, seed=0):
    from gpe_amd import configs, nets, optim
    import bench
    dev = torch.device('cuda', 0)
    data_config = configs.data_config()
    nn_cfg = (configs.att_model_config if model_kind == 'att' else configs.lstm_model_config)(k_neighbors=k)
    torch.manual_seed(seed)
    cls = nets.GarmentSegmentPattern3D if model_kind == 'att' else nets.GarmentFullPattern3D
    model = cls(data_config, dict(nn_cfg), dict(nn_cfg['loss'])).to(dev).train()
    model.loss.with_quality_eval = False
    feats, gt = bench.synthetic(batch, points, data_config, seed=1000, device=dev)
    return model, feats, gt


@pytest.mark.parametrize('mode,model_kind', [('f32', 'lstm'), ('f16x3', 'lstm'), ('f16x3', 'att')])
def test_step_graph_replays_the_eager_steps(gpe, mode, model_kind):
    from gpe_amd import optim, graph
    prev = gpe.set_math(mode)
    prev_rows = gpe.set_f16x3_min_rows(0)
    try:
        model_a, feats, gt = _setup(gpe, model_kind)
        model_b = copy.deepcopy(model_a)
        sched = lambda: optim.OneCycle(2e-3, 40)
        opt_a = optim.FusedAdam(optim.FlatArena(model_a), lr=2e-3, schedule=sched())
        opt_b = optim.FusedAdam(optim.FlatArena(model_b), lr=2e-3, schedule=sched())
        nsteps = 7
        losses_a = []
        for i in range(nsteps):
            torch.manual_seed(100 + i)
            loss = model_a.loss(model_a(feats), gt, epoch=0)[0]
            loss.backward()
            opt_a.step()
            losses_a.append(loss.detach().clone())
        sg = graph.StepGraph(lambda f, g: model_b.loss(model_b(f), g, epoch=0)[0], opt_b, warmup=2)
        losses_b = []
        for i in range(nsteps):
            torch.manual_seed(100 + i)
            losses_b.append(sg.step(feats, gt).detach().clone())
        sg.synchronize()
        torch.cuda.synchronize()
        assert sg.captures == 1 and sg.replays == nsteps - 2
        for i, (a, b) in enumerate(zip(losses_a, losses_b)):
            assert torch.equal(a, b), (i, float(a), float(b))
        for (n, p), q in zip(model_a.named_parameters(), model_b.parameters()):
            assert torch.equal(p, q), n
        for (n, p), q in zip(model_a.named_buffers(), model_b.buffers()):
            assert torch.equal(p, q), n
        assert opt_a.t == opt_b.t and opt_a.steps == opt_b.steps and opt_a.last_lr == opt_b.last_lr
    finally:
        gpe.set_f16x3_min_rows(prev_rows)
        gpe.set_math(prev)


def test_step_graph_recaptures_when_the_half_activation_guard_trips(gpe):
    """The fp16 storage of the aggregated EdgeConv activation is watched in graph mode too: the amax words of the captured forward are
    read after every replay; a layer that outgrows fp16 (here: its last Linear scaled by 1e5 between two replays) warns, switches to
    fp32 rows, and the next step is captured again."""
    import warnings
    from gpe_amd import optim, graph, ops
    prev = gpe.set_math('f16x3')
    prev_rows = gpe.set_f16x3_min_rows(0)
    mode0 = ops.set_half_act_guard('fallback')
    try:
        model, feats, gt = _setup(gpe, 'lstm', k=16, batch=8, points=512)
        convs = [m for m in model.modules() if hasattr(m, 'half_act_guard')]
        assert convs and gpe._lib.query('gpe_edge_lazy_dz3_ok', 8, 512, 16, 150, 200) == 1
        opt = optim.FusedAdam(optim.FlatArena(model), lr=1e-4)
        sg = graph.StepGraph(lambda f, g: model.loss(model(f), g, epoch=0)[0], opt, warmup=1)
        for i in range(3):
            torch.manual_seed(i)
            sg.step(feats, gt)
        sg.synchronize()
        assert sg.captures == 1 and sg.replays == 2 and len(sg.guards) == len(convs) and not any(c.half_act_guard.disabled for c in convs)
        with torch.no_grad():
            convs[0].nn[2][0].weight.mul_(1e5)
        with warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter('always')
            for i in range(3, 9):
                torch.manual_seed(i)
                loss = sg.step(feats, gt)
                sg.synchronize()                       # (lets the asynchronous read of the words land before the next step looks)
        assert convs[0].half_act_guard.disabled and convs[0].half_act_guard.last_amax > 65504
        assert any('fp16 storage' in str(w.message) for w in wl)
        assert sg.captures == 2 and bool(torch.isfinite(loss))
    finally:
        ops.set_half_act_guard(mode0)
        gpe.set_f16x3_min_rows(prev_rows)
        gpe.set_math(prev)
