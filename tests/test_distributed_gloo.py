"""not-gpu: the multi-GPU exchange step (bucketed gradient all-reduce from autograd hooks, parallel.py) on 2 CPU
processes over gloo — the same code path the driver runs over RCCL."""
import os
import socket
import subprocess
import sys
import textwrap

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.nn as nn, torch.distributed as dist
    import gpe_amd
    from gpe_amd.parallel import DistributedHotPath, init_distributed
    rank, local, world = init_distributed(backend='gloo')
    torch.manual_seed(0)
    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.enc = nn.Linear(6, 32); self.mid = nn.Linear(32, 32); self.unused = nn.Linear(32, 4)
            self.dec = nn.Linear(32, 3)
        def forward(self, x):
            return self.dec(torch.relu(self.mid(torch.relu(self.enc(x)))))
    full_x = torch.randn(8, 6, generator=torch.Generator().manual_seed(1))
    full_y = torch.randn(8, 3, generator=torch.Generator().manual_seed(2))
    ref = Net()
    ((ref(full_x) - full_y) ** 2).mean().backward()
    model = Net()
    model.load_state_dict(ref.state_dict())
    ddp = DistributedHotPath(model, device_ids=[], bucket_bytes=2048)      # several buckets
    assert len(ddp._buckets) > 1
    g0 = ddp.arena.grad
    for step in range(2):                                                  # hooks must re-arm after each step
        ddp.arena.zero_grad()
        sl = slice(rank * 4, rank * 4 + 4)
        ((ddp(full_x[sl]) - full_y[sl]) ** 2).mean().backward()
        ddp.finish_gradient_sync()
        for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            # gradients are permanent views into the arena: reduced in place, no per-parameter copies
            assert p.grad.untyped_storage().data_ptr() == g0.untyped_storage().data_ptr(), n
            if q.grad is None:
                assert not p.grad.any(), n                                  # never produced: stays zero on every rank
            else:
                assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-7), (n, step)
    try:                                                                   # dropping the views must fail loudly
        model.zero_grad(set_to_none=True)
        ((ddp(full_x[:4]) - full_y[:4]) ** 2).mean().backward()
        raise SystemExit('expected the lost-view check to fire')
    except RuntimeError as e:
        assert 'arena gradient view' in str(e)
    dist.barrier()
    dist.destroy_process_group()
    print('rank', rank, 'ok')
''') % REPO


# The REAL model's protocol, without kernels: gradients are not accumulated by autograd but written IN PLACE into the
# arena by backward code that asks ops._gbuf for the buffer and reports through ops._gret -> FlatArena.mark_written ->
# DistributedHotPath._on_written (what every autograd.Function of ops.py does on the GPU).  The stub Functions below do
# their arithmetic in torch and follow exactly that protocol, in the order ops.py produces gradients (decoder first,
# inside a Function: weight then bias), including a parameter that never receives a gradient
# (feature_extractor.lin of the attention model) whose bucket only leaves in finish_gradient_sync().
WORKER_SINK = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.nn as nn, torch.distributed as dist
    import gpe_amd
    from gpe_amd import ops, optim
    from gpe_amd.parallel import DistributedHotPath, init_distributed
    rank, local, world = init_distributed(backend='gloo')

    class SinkLinear(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, b):
            ctx.save_for_backward(x, w, b)
            return x @ w.t() + b
        @staticmethod
        def backward(ctx, gy):
            x, w, b = ctx.saved_tensors
            gw, gb = ops._gbuf(w), ops._gbuf(b)          # arena views when the parameters are registered
            gw.copy_(gy.t() @ x); gb.copy_(gy.sum(0))    # "kernel" writes in place
            return gy @ w, ops._gret(w, gw), ops._gret(b, gb)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.enc = nn.Linear(6, 40); self.lin_unused = nn.Linear(40, 8); self.mid = nn.Linear(40, 40)
            self.dec = nn.Linear(40, 3)
        def forward(self, x):
            h = torch.relu(SinkLinear.apply(x, self.enc.weight, self.enc.bias))
            h = torch.relu(SinkLinear.apply(h, self.mid.weight, self.mid.bias))
            return SinkLinear.apply(h, self.dec.weight, self.dec.bias)

    torch.manual_seed(0)
    full_x = torch.randn(8, 6, generator=torch.Generator().manual_seed(1))
    full_y = torch.randn(8, 3, generator=torch.Generator().manual_seed(2))
    ref = Net()
    ((ref(full_x) - full_y) ** 2).mean().backward()     # no arena: _gbuf hands out fresh tensors, autograd accumulates
    model = Net()
    model.load_state_dict(ref.state_dict())
    arena = optim.FlatArena(model, register_sink=False)
    arena.register_sink()                                # CPU arena as gradient sink (the GPU default)
    ddp = DistributedHotPath(model, device_ids=[], bucket_bytes=1024, arena=arena)
    assert len(ddp._buckets) >= 3, ddp._buckets
    # bucket order = gradient-ready order: the decoder's parameters sit in bucket 0
    assert ddp._bucket_of[arena.index[model.dec.weight.data_ptr()]] == 0
    launched_in_backward = []
    orig_launch = ddp._launch
    def spy(bi):
        launched_in_backward.append(bi)
        orig_launch(bi)
    ddp._launch = spy
    for step in range(3):
        sl = slice(rank * 4, rank * 4 + 4)
        del launched_in_backward[:]
        ((ddp(full_x[sl]) - full_y[sl]) ** 2).mean().backward()
        early = list(launched_in_backward)
        assert early and early == sorted(early), early   # buckets leave during backward, decoder first
        unused_b = ddp._bucket_of[arena.index[model.lin_unused.weight.data_ptr()]]
        assert unused_b not in early                     # the None-grad bucket cannot complete on its own ...
        ddp.finish_gradient_sync()
        assert unused_b in launched_in_backward          # ... finish_gradient_sync sends it
        assert sorted(launched_in_backward) == list(range(len(ddp._buckets)))
        for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            assert p.grad.untyped_storage().data_ptr() == arena.grad.untyped_storage().data_ptr(), n
            if 'unused' in n:
                assert not p.grad.any(), n
            else:
                assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-7), (n, step)
        assert arena.touched == arena.written and len(arena.written) == 6
        arena.zero_grad()                                # what FusedAdam.step() does after consuming the gradients
    # forgetting begin_step()/zero_grad() between two backward passes is reported, not silently double-counted
    ((ddp(full_x[:4]) - full_y[:4]) ** 2).mean().backward()
    try:
        ((ddp(full_x[:4]) - full_y[:4]) ** 2).mean().backward()
        raise SystemExit('expected the second-gradient check to fire')
    except RuntimeError as e:
        assert 'second gradient' in str(e)
    ddp._reset()
    ex = ddp.measure_exchange(iters=2)                   # what bench.py prints for N > 1
    assert ex['dist_world'] == 2 and ex['backend'] == 'gloo' and ex['buckets'] == len(ddp._buckets)
    assert ex['bytes_per_step'] == sum((hi - lo) * 4 for _, _, lo, hi in ddp._buckets) and ex['ms_per_step'] > 0
    assert ddp.exposed_ms() is not None
    dist.barrier()
    dist.destroy_process_group()
    print('rank', rank, 'ok')
''') % REPO


def _run_two_ranks(tmp_path, source):
    script = tmp_path / 'worker.py'
    script.write_text(source)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), OMP_NUM_THREADS='1')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert 'rank %d ok' % rank in out


def test_two_rank_arena_sink_protocol(tmp_path):
    _run_two_ranks(tmp_path, WORKER_SINK)


def test_two_rank_gradient_average_matches_full_batch(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), OMP_NUM_THREADS='1')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert 'rank %d ok' % rank in out
