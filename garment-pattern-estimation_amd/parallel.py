"""Data parallelism for the path: one process per GPU, local BatchNorm statistics (as the reference's
nn.DataParallel replicas, nn/train.py:124 — no SyncBN), and ONE exchange per step: a bucketed RCCL all-reduce
(average) of the gradients over xGMI, launched from autograd hooks so the decoder's 91 % of the bytes travel
while the encoder backward is still running.  `backend="nccl"` IS RCCL on ROCm; the CPU
tests run the same code over gloo."""
import torch
import torch.distributed as dist
import torch.nn as nn


class DistributedHotPath(nn.Module):
    """Wrapper exposing what the reference's trainer touches on an nn.DataParallel object: `.module`,
    `.device_ids`, `__call__`.  Call `finish_gradient_sync()` after `loss.backward()` and before
    `optimizer.step()`."""

    def __init__(self, module, device_ids=None, bucket_bytes=8 << 20, process_group=None):
        super().__init__()
        self.module = module
        self.device_ids = list(device_ids) if device_ids is not None else []
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # buckets in reverse registration order ~= the order gradients become ready (decoder first)
        params = [p for p in module.parameters() if p.requires_grad]
        self._buckets, cur, size = [], [], 0
        for p in reversed(params):
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self._buckets.append(cur)
                cur, size = [], 0
        if cur:
            self._buckets.append(cur)
        self._bucket_of = {}
        for bi, b in enumerate(self._buckets):
            for p in b:
                self._bucket_of[p] = bi
        self._pending = [0] * len(self._buckets)
        self._inflight = []
        self._launched = set()
        if self.world > 1:
            for p in params:
                p.register_post_accumulate_grad_hook(self._on_grad)
        self._reset()

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def _reset(self):
        self._pending = [len(b) for b in self._buckets]
        self._inflight = []
        self._launched = set()

    def _launch(self, bi):
        ps = [p for p in self._buckets[bi] if p.grad is not None]
        self._launched.add(bi)
        if not ps:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        flat.div_(self.world)
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._inflight.append((work, flat, ps))

    def _on_grad(self, p):
        bi = self._bucket_of[p]
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
            for work, flat, ps in self._inflight:
                work.wait()
                off = 0
                for p in ps:
                    n = p.numel()
                    p.grad.copy_(flat[off:off + n].view_as(p.grad))
                    off += n
        self._reset()


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run) and binds this process to its GPU."""
    import os
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if torch.cuda.is_available():
        # GPE_SHARE_DEVICE=1 (testing only): all ranks use cuda:0, for exercising the N>1 code path on a 1-GPU box
        # together with backend gloo (RCCL rejects two ranks on one device)
        if os.environ.get('GPE_SHARE_DEVICE') == '1':
            local = 0
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        backend = backend or os.environ.get('GPE_DIST_BACKEND')
        dist.init_process_group(backend or ('nccl' if torch.cuda.is_available() else 'gloo'),
                                rank=rank, world_size=world)
    return rank, local, world
