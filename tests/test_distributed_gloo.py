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
