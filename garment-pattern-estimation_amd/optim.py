"""The optimizer side of the training step (reference: nn/trainer.py:162-185 — torch.optim.Adam + OneCycleLR) built for
this path: all parameters of a model live in ONE flat arena, so that

  * the Adam update of the ~70 parameter tensors is a single launch (gpe_adam_step) instead of torch's multi-tensor
    foreach chains;
  * gradients are produced IN PLACE: the backward kernels of ops.py write each weight gradient straight into its slice
    of the arena's gradient buffer (ops._gbuf / ops._gret) — no per-parameter gradient tensors, no zero_grad pass, and
    the data-parallel all-reduce (parallel.DistributedHotPath) works on contiguous slices of that buffer: no torch.cat
    into a bucket and no copy back;
  * `zero_grad` is folded into the Adam kernel.

FlatArena is plain torch (it also runs on CPU tensors: the gloo tests use it); FusedAdam needs the HIP library.
"""
import math

import torch

from . import ops
from . import _lib as L


class FlatArena:
    """Re-homes every parameter of `module` that requires grad into one flat fp32 buffer (16-byte aligned segments) and gives
    each a permanent `.grad` view into a flat gradient buffer.  Order: REVERSE registration order, i.e. roughly the order
    in which backward produces the gradients (decoder first) — consecutive ranges of the gradient buffer therefore become
    complete one after another, which is what the bucketed all-reduce wants."""

    def __init__(self, module, register_sink=True):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError('FlatArena: the module has no trainable parameters')
        dev, dt = params[0].device, params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in params):
            raise ValueError('FlatArena: all parameters must share one device and dtype')
        self.params = list(reversed(params))
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.numel = off
        self.flat = torch.zeros(off, device=dev, dtype=dt)
        self.grad = torch.zeros(off, device=dev, dtype=dt)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
        self.index = {p.data_ptr(): i for i, p in enumerate(self.params)}
        self.listeners = []            # callables(param index) run when a gradient has been written by a kernel
        self.written = set()
        self.is_sink = False
        if register_sink and dev.type == 'cuda':
            self.register_sink()
        ops.bump_weights_epoch()       # parameters moved: packed operands must be rebuilt

    # ---- gradient sink protocol (ops._gbuf / ops._gret) ---------------------------------------------------
    def register_sink(self):
        for p in self.params:
            ops._SINK[p.data_ptr()] = (self, p.grad)
        self.is_sink = True

    def unregister_sink(self):
        for p in self.params:
            ops._SINK.pop(p.data_ptr(), None)
        self.is_sink = False

    def mark_written(self, param):
        i = self.index[param.data_ptr()]
        if i in self.written:
            raise RuntimeError('FlatArena: a second gradient for the same parameter in one backward pass — the in-place '
                               'gradient sink needs every parameter to be used once per step (unregister_sink() for '
                               'weight sharing / gradient accumulation)')
        self.written.add(i)
        for fn in self.listeners:
            fn(i)

    def begin_step(self):
        """Forget which gradients were written (call after the optimizer consumed them)."""
        self.written.clear()

    def zero_grad(self):
        self.grad.zero_()
        self.begin_step()

    def segment(self, i):
        return self.offsets[i], self.params[i].numel()


class OneCycle:
    """torch.optim.lr_scheduler.OneCycleLR(max_lr, epochs, steps_per_epoch, cycle_momentum=False) with torch's defaults
    (pct_start 0.3, cosine annealing, div_factor 25, final_div_factor 1e4, two phases) — nn/trainer.py:175-181 —
    as a host-side function of the step number."""

    def __init__(self, max_lr, total_steps, pct_start=0.3, div_factor=25., final_div_factor=1e4):
        self.max_lr, self.total = float(max_lr), int(total_steps)
        self.initial = self.max_lr / div_factor
        self.min_lr = self.initial / final_div_factor
        self.end1 = float(pct_start * self.total) - 1
        self.end2 = self.total - 1

    @staticmethod
    def _cos(start, end, pct):
        return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1)

    def lr(self, step):
        """learning rate used BY optimizer step number `step` (0-based: step 0 runs at the initial rate)."""
        if step > self.end2:
            raise ValueError('Tried to step {} times. The specified number of total steps is {}'.format(step, self.total))
        if step <= self.end1:
            return self._cos(self.initial, self.max_lr, step / self.end1)
        return self._cos(self.max_lr, self.min_lr, (step - self.end1) / (self.end2 - self.end1))


class FusedAdam:
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) over a FlatArena in one launch per step, optionally driven by
    a OneCycle schedule (then `step()` also advances the schedule, like calling scheduler.step() after optimizer.step()).

        arena = FlatArena(model); opt = FusedAdam(arena, lr=2e-3)            # or FusedAdam(model, ...)
        loss.backward(); opt.step()                                          # gradients are cleared by the same launch
    """

    def __init__(self, arena_or_module, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, schedule=None):
        self.arena = arena_or_module if isinstance(arena_or_module, FlatArena) else FlatArena(arena_or_module)
        if self.arena.flat.device.type != 'cuda':
            raise RuntimeError('FusedAdam runs on the MI355X (libgpe_hip.so); there is no CPU path')
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), betas, float(eps), float(weight_decay)
        self.schedule = schedule
        self.m = torch.zeros_like(self.arena.flat)
        self.v = torch.zeros_like(self.arena.flat)
        self.t = 0
        self.last_lr = self.lr if schedule is None else schedule.lr(0)

    def step(self, grad_scale=1.0):
        a = self.arena
        lr = self.lr if self.schedule is None else self.schedule.lr(self.t)
        self.t += 1
        L.call('gpe_adam_step', a.flat, a.grad, self.m, self.v, a.numel, float(lr), float(self.betas[0]),
               float(self.betas[1]), self.eps, self.weight_decay, self.t, float(grad_scale), 1)
        self.last_lr = lr
        a.begin_step()
        ops.bump_weights_epoch()       # parameters changed through raw pointers: torch's version counters did not move

    def zero_grad(self, set_to_none=False):
        """Gradients are cleared inside step(); this only exists for trainer loops that call it unconditionally."""
        self.arena.begin_step()

    def state_dict(self):
        return {'t': self.t, 'm': self.m.clone(), 'v': self.v.clone(), 'lr': self.lr, 'betas': self.betas,
                'eps': self.eps, 'weight_decay': self.weight_decay}

    def load_state_dict(self, sd):
        self.t = sd['t']
        self.m.copy_(sd['m'])
        self.v.copy_(sd['v'])
