"""Data parallelism for the path: one process per GPU, local BatchNorm statistics (as the reference's
nn.DataParallel replicas, nn/train.py:124 — no SyncBN), and ONE exchange per step: a bucketed RCCL all-reduce
(average) of the gradients over xGMI, launched while backward is still running so the decoder's 91 % of the bytes
travel under the encoder backward.  `backend="nccl"` IS RCCL on ROCm; the CPU tests run the same
code over gloo.

Gradients live in the flat arena of optim.FlatArena: a bucket is a contiguous slice of the arena's gradient buffer, the
all-reduce runs in place on that slice — no torch.cat into a staging tensor, no copy back per parameter."""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from .optim import FlatArena


class DistributedHotPath(nn.Module):
    """Wrapper exposing what the reference's trainer touches on an nn.DataParallel object: `.module`, `.device_ids`,
    `__call__`.  Call `finish_gradient_sync()` after `loss.backward()` and before the optimizer step.

    A gradient reaches the arena in one of two ways, both tracked here: written in place by a backward kernel of ops.py
    (FlatArena.mark_written -> `_on_written`), or accumulated by autograd into the parameter's permanent `.grad` view
    (post-accumulate hook -> `_on_hook`; used by plain torch modules, e.g. in the CPU tests).  When the last parameter of
    a bucket has its gradient, the bucket's slice leaves for the all-reduce.  The arena gradient buffer must be zero when
    a backward pass starts (FusedAdam.step clears it; otherwise call `arena.zero_grad()`)."""

    def __init__(self, module, device_ids=None, bucket_bytes=8 << 20, process_group=None, arena=None, reserve_cus=None):
        """reserve_cus: compute units left out of every persistent launch of this process while the model trains (room for the
        collective's kernels under the fused edge kernels: include/gpe_hip.h gpe_reserve_cus_set).  None = 16 when the world has
        more than one rank (two CUs per XCD: measured cost at N = 1, cfg 2: profiles/r06_z_reserved_cus.md), else 0."""
        super().__init__()
        self.module = module
        self.device_ids = list(device_ids) if device_ids is not None else []
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # ReduceOp.AVG exists on the nccl (= RCCL) backend only
        self._avg_op = dist.is_initialized() and dist.get_backend(process_group) == 'nccl'
        self.arena = arena if arena is not None else FlatArena(module)
        a = self.arena
        # buckets = consecutive parameter ranges of the arena (already in gradient-ready order)
        self._buckets, start, size = [], 0, 0
        for i, p in enumerate(a.params):
            size += p.numel() * p.element_size()
            if size >= bucket_bytes or i == len(a.params) - 1:
                lo = a.offsets[start]
                hi = a.offsets[i] + (a.params[i].numel() + 3) // 4 * 4
                self._buckets.append((start, i + 1, lo, hi))
                start, size = i + 1, 0
        self._bucket_of = [0] * len(a.params)
        for bi, (s, e, _, _) in enumerate(self._buckets):
            for i in range(s, e):
                self._bucket_of[i] = bi
        if self.world > 1:
            a.listeners.append(self._on_written)
            for i, p in enumerate(a.params):
                p.register_post_accumulate_grad_hook(lambda _p, i=i: self._on_hook(_p, i))
        self._exposed = []
        self._reset()
        if reserve_cus is None:
            reserve_cus = 16 if self.world > 1 else 0
        self.reserved_cus = int(reserve_cus)
        if torch.cuda.is_available():
            from . import _lib
            _lib.set_reserved_cus(self.reserved_cus)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def measure_exchange(self, iters=10):
        """The exchange step on its own (collective on every rank): all buckets all-reduced back to back on a scratch copy
        of the gradient arena, `iters` times between two synchronisations.  -> {dist_world (what the process group
        reports), backend, buckets, bytes_per_step, ms_per_step}.  bench.py prints it so that a reader can check that RCCL
        really connected N ranks and what the un-overlapped exchange costs; the step itself overlaps it with backward."""
        import time
        scratch = torch.zeros_like(self.arena.grad)
        sync = torch.cuda.synchronize if scratch.is_cuda else (lambda: None)
        nbytes = sum((hi - lo) * scratch.element_size() for _, _, lo, hi in self._buckets)
        out = {'dist_world': self.world, 'backend': dist.get_backend(self.group) if dist.is_initialized() else None,
               'buckets': len(self._buckets), 'bytes_per_step': nbytes, 'ms_per_step': 0.0}
        if self.world == 1:
            return out
        for it in range(iters + 2):
            if it == 2:
                dist.barrier(group=self.group)
                sync()
                t0 = time.perf_counter()
            works = [dist.all_reduce(scratch[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                     for _, _, lo, hi in self._buckets]
            for w in works:
                w.wait()
        sync()
        out['ms_per_step'] = (time.perf_counter() - t0) / iters * 1e3
        return out

    def exposed_ms(self):
        """Mean time finish_gradient_sync() spent between the end of backward and the arrival of the last bucket (event
        pair on the compute stream around the waits), over the steps since construction: the part of the exchange that
        backward did NOT hide."""
        if not self._exposed:
            return None
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        ms = [a.elapsed_time(b) if not isinstance(a, float) else (b - a) * 1e3 for a, b in self._exposed]
        return sum(ms) / len(ms)

    def _reset(self):
        self._pending = [e - s for s, e, _, _ in self._buckets]
        self._seen = set()
        self._inflight = []
        self._launched = set()

    def _launch(self, bi):
        self._launched.add(bi)
        _, _, lo, hi = self._buckets[bi]
        flat = self.arena.grad[lo:hi]
        if flat.is_cuda:
            # "written" means "queued": on the stream of the launch that told the arena.  The collective's stream waits for the CURRENT
            # stream only, and gradients of this bucket may sit on the other one (ops.side_grads runs the leaf weight-gradient launches
            # on a side stream): inside a side block the side stream is behind everything the main stream had queued at the fork —
            # which is every gradient written so far — and on the main stream the side stream is joined here.
            from . import ops
            if ops._SIDE_DIRTY[0] and torch.cuda.current_stream() != ops._side_stream():
                torch.cuda.current_stream().wait_stream(ops._side_stream())
        if self._avg_op:
            # RCCL averages inside the collective: no extra elementwise launch per bucket and step
            self._inflight.append(dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
        else:
            # gloo (the CPU tests) has no AVG: pre-divide, then SUM
            flat.div_(self.world)
            self._inflight.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _on_written(self, i):
        self._on_ready(i)

    def _on_hook(self, p, i):
        o, n = self.arena.segment(i)
        if p.grad is None or p.grad.data_ptr() != self.arena.grad.data_ptr() + o * self.arena.grad.element_size():
            raise RuntimeError('DistributedHotPath: a parameter lost its arena gradient view (zero_grad(set_to_none=True) '
                               'or `p.grad = None`?) — clear gradients with arena.zero_grad() / FusedAdam.step()')
        self._on_ready(i)

    def _on_ready(self, i):
        if i in self._seen:
            return
        self._seen.add(i)
        bi = self._bucket_of[i]
        self._pending[bi] -= 1
        if self._pending[bi] == 0 and bi not in self._launched:
            self._launch(bi)

    def finish_gradient_sync(self):
        if self.world > 1:
            # buckets holding parameters that received no gradient (e.g. feature_extractor.lin in the attention
            # variant) never complete on their own; every rank has the same pattern, so the order is consistent
            for bi in range(len(self._buckets)):
                if bi not in self._launched:
                    self._launch(bi)
            cuda = self.arena.grad.is_cuda
            if cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            else:
                import time
                e0 = time.perf_counter()
            for work in self._inflight:
                work.wait()
            if cuda:
                e1.record()
            else:
                e1 = time.perf_counter()
            self._exposed.append((e0, e1))
            if len(self._exposed) > 64:
                del self._exposed[0]
        self._reset()


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run) and binds this process to its GPU."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if torch.cuda.is_available():
        # GPE_SHARE_DEVICE=1 (testing only): all ranks use cuda:0, for exercising the N>1 code path on a 1-GPU box
        # together with backend gloo (RCCL rejects two ranks on one device)
        if os.environ.get('GPE_SHARE_DEVICE') == '1':
            local = 0
        torch.cuda.set_device(local)
    if (world > 1 or os.environ.get('GPE_FORCE_DIST') == '1') and not dist.is_initialized():
        backend = backend or os.environ.get('GPE_DIST_BACKEND')
        dist.init_process_group(backend or ('nccl' if torch.cuda.is_available() else 'gloo'),
                                rank=rank, world_size=world)
    return rank, local, world
