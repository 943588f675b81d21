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
        # parameters that received a gradient since the last optimizer step, through either route (kernel-written:
        # mark_written; autograd-accumulated: the hook below).  FusedAdam skips the others, like torch.optim.Adam skips
        # parameters whose .grad is None.
        self.touched = set()
        for i, p in enumerate(self.params):
            p.register_post_accumulate_grad_hook(lambda _p, i=i: self.touched.add(i))
        self.is_sink = False
        if register_sink and dev.type == 'cuda':
            self.register_sink()
        ops.bump_weights_epoch()       # parameters moved: packed operands must be rebuilt

    # ---- gradient sink protocol (ops._gbuf / ops._gret) ---------------------------------------------------
    def register_sink(self):
        import weakref
        me = weakref.ref(self)
        keys = [p.data_ptr() for p in self.params]
        for k, p in zip(keys, self.params):
            ops._SINK[k] = (me, p.grad)
        # the registry must not keep a dead model's arena (2 x 11 MB) alive, nor answer for recycled addresses
        weakref.finalize(self, lambda: [ops._SINK.pop(k, None) for k in keys if (ops._SINK.get(k) or (None,))[0] is me])
        self.is_sink = True

    def unregister_sink(self):
        for p in self.params:
            ops._SINK.pop(p.data_ptr(), None)
        self.is_sink = False

    def mark_written(self, param):
        i = self.index[param.data_ptr()]
        if i in self.written:
            raise RuntimeError('FlatArena: a second gradient for the same parameter since the last begin_step() — either the '
                               'optimizer step was not taken through FusedAdam.step() / arena.zero_grad() (torch optimizers: '
                               'call arena.zero_grad() instead of optimizer.zero_grad()), or a parameter is used twice per '
                               'step (weight sharing / gradient accumulation: unregister_sink())')
        self.written.add(i)
        self.touched.add(i)
        for fn in self.listeners:
            fn(i)

    def begin_step(self):
        """Forget which gradients were written (call after the optimizer consumed them)."""
        self.written.clear()
        self.touched.clear()

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

    def state_dict(self):
        """What a resumed run needs (FusedAdam owns the step counter): the schedule's shape."""
        return {'max_lr': self.max_lr, 'total_steps': self.total, 'initial_lr': self.initial, 'min_lr': self.min_lr,
                'end1': self.end1, 'end2': self.end2}

    def load_state_dict(self, sd):
        """Accepts this class's own dict, or torch's OneCycleLR.state_dict() (nn/trainer.py:283 saves that one).  torch keeps the
        schedule's RATES (max_lr / initial_lr / min_lr) in the optimizer's param group, not here: of the scheduler dict
        `total_steps` and, through `_schedule_phases`, the phase boundaries are taken, and `last_epoch` (the step counter) is
        returned so that the caller can hand it to FusedAdam; the rates arrive with FusedAdam.load_state_dict (set_rates)."""
        if 'total_steps' in sd and 'max_lr' in sd:
            self.max_lr, self.total = float(sd['max_lr']), int(sd['total_steps'])
            self.initial, self.min_lr = float(sd['initial_lr']), float(sd['min_lr'])
            self.end1, self.end2 = float(sd['end1']), float(sd['end2'])
            return None
        self.total = int(sd['total_steps'])
        ph = sd.get('_schedule_phases')
        if ph:
            self.end1, self.end2 = float(ph[0]['end_step']), float(ph[1]['end_step'])
        return sd.get('last_epoch')

    def set_rates(self, group):
        """takes max_lr / initial_lr / min_lr from an optimizer param group (where torch's OneCycleLR stores them)"""
        if 'max_lr' in group:
            self.max_lr = float(group['max_lr'])
        if 'initial_lr' in group:
            self.initial = float(group['initial_lr'])
        if 'min_lr' in group:
            self.min_lr = float(group['min_lr'])

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
        self.t = 0                                     # optimizer steps taken (drives the schedule)
        # torch.optim.Adam counts steps PER PARAMETER (state[p]['step']): a parameter that received no gradient in a step keeps
        # its count, and its bias correction continues from there when a gradient shows up later.  Same here.
        self.steps = [0] * len(self.arena.params)
        self.last_lr = self.lr if schedule is None else schedule.lr(0)

    def _runs(self, live, advance):
        """Maximal runs of consecutive live arena segments that share one step count -> [(lo, hi, step)].  torch.optim.Adam leaves a
        parameter alone in a step in which it received no gradient (grad None): no moment decay, no weight decay, no move, no step
        count.  Same here (normally ONE run = the whole arena; the attention model's unused feature_extractor.lin splits it in
        two).  advance=False only looks (the step counts a run WOULD take)."""
        a = self.arena
        runs, start, cur, end = [], None, None, None
        for i in range(len(a.params)):
            if i in live:
                st = self.steps[i] + 1
                if advance:
                    self.steps[i] = st
                if start is not None and st != cur:
                    runs.append((start, end, cur))
                    start = None
                if start is None:
                    start, cur = a.offsets[i], st
                end = a.offsets[i] + (a.params[i].numel() + 3) // 4 * 4
            elif start is not None:
                runs.append((start, end, cur))
                start = None
        if start is not None:
            runs.append((start, end, cur))
        return runs

    def _live(self):
        # nothing recorded at all (gradients written by hand into arena.grad) = everything is live
        a, n = self.arena, len(self.arena.params)
        return set(a.touched) if (a.touched and len(a.touched) < n) else set(range(n))

    def step(self, grad_scale=1.0):
        a = self.arena
        ops.join_side()                # weight gradients written on the side stream (ops.side_grads)
        lr = self.lr if self.schedule is None else self.schedule.lr(self.t)
        self.t += 1
        runs = self._runs(self._live(), advance=True)
        for lo, hi, st in runs:
            L.call('gpe_adam_step', a.flat[lo:hi], a.grad[lo:hi], self.m[lo:hi], self.v[lo:hi], hi - lo, float(lr),
                   float(self.betas[0]), float(self.betas[1]), self.eps, self.weight_decay, st, float(grad_scale), 1)
        if len(runs) != 1 or runs[0][:2] != (0, a.numel):
            a.grad.zero_()             # untouched segments may still hold stale values written by hand
        self.last_lr = lr
        a.begin_step()
        ops.bump_weights_epoch()       # parameters changed through raw pointers: torch's version counters did not move

    # ---- a step inside a captured hipGraph (graph.StepGraph): kernel arguments are frozen, so the learning rate and the bias
    # corrections reach the kernel through a two-float device buffer per run, rewritten in front of every replay ----
    def step_captured(self, ctx, grad_scale=1.0):
        """Called INSIDE the capture, after backward: queues the launches, advances nothing.  -> the runs' (lo, hi)."""
        a = self.arena
        ops.join_side()
        self._cap_live = self._live()
        runs = self._runs(self._cap_live, advance=False)
        for r, (lo, hi, _) in enumerate(runs):
            L.call('gpe_adam_step_dev', a.flat[lo:hi], a.grad[lo:hi], self.m[lo:hi], self.v[lo:hi], hi - lo, ctx.hyper_slot(r)[0],
                   float(self.betas[0]), float(self.betas[1]), self.eps, self.weight_decay, float(grad_scale), 1)
        if len(runs) != 1 or runs[0][:2] != (0, a.numel):
            a.grad.zero_()
        a.begin_step()
        return [(lo, hi) for lo, hi, _ in runs]

    def advance_captured(self, ctx, layout):
        """Called in front of every replay: the host side of step() — schedule, step counts — and the scalars of each run."""
        lr = self.lr if self.schedule is None else self.schedule.lr(self.t)
        self.t += 1
        runs = self._runs(self._cap_live, advance=True)
        if [(lo, hi) for lo, hi, _ in runs] != layout:
            raise RuntimeError('FusedAdam: the runs of the captured step no longer match the step counts (parameters that took '
                               'different numbers of steps since the capture): capture again')
        for r, (_, _, st) in enumerate(runs):
            ctx.write_hyper(r, lr, self.betas[0], self.betas[1], st)
        self.last_lr = lr
        ops.bump_weights_epoch()

    def zero_grad(self, set_to_none=False):
        """Gradients are cleared inside step(); this only exists for trainer loops that call it unconditionally."""
        self.arena.begin_step()

    # ---- checkpoints: torch.optim.Adam's format (nn/trainer.py:281-285 saves optimizer.state_dict(), _restore_run loads it) ----
    def _module_order(self):
        """arena index of the i-th parameter in `[p for p in module.parameters() if p.requires_grad]` order — the order
        torch.optim.Adam(model.parameters()) numbers its state by (the arena stores the reverse)."""
        n = len(self.arena.params)
        return [n - 1 - i for i in range(n)]

    def state_dict(self):
        """torch.optim.Adam.state_dict() layout: state[i] = {step, exp_avg, exp_avg_sq} per parameter that has taken a step
        (index = position in model.parameters(); like torch, no entry for a parameter that never received a gradient),
        param_groups[0] with the hyper-parameters — and, under a OneCycle schedule, the `initial_lr` / `max_lr` / `min_lr` keys
        torch's OneCycleLR keeps in the group, so that the reference's restore flow (build Adam + OneCycleLR, then
        optimizer.load_state_dict: nn/trainer.py _restore_run) can step its scheduler on this checkpoint.  A checkpoint written
        here loads into torch.optim.Adam over the same model and vice versa."""
        a = self.arena
        state = {}
        for i, ai in enumerate(self._module_order()):
            if self.steps[ai] == 0:
                continue
            o, n = a.segment(ai)
            shape = a.params[ai].shape
            state[i] = {'step': torch.tensor(float(self.steps[ai])),
                        'exp_avg': self.m[o:o + n].view(shape).clone(),
                        'exp_avg_sq': self.v[o:o + n].view(shape).clone()}
        # under a schedule torch's group holds the rate of the NEXT step (scheduler.step() ran after optimizer.step())
        lr_now = self.lr
        if self.schedule is not None:
            lr_now = self.schedule.lr(self.t) if self.t <= self.schedule.end2 else self.last_lr
        group = {'lr': lr_now, 'betas': tuple(self.betas), 'eps': self.eps,
                 'weight_decay': self.weight_decay, 'amsgrad': False, 'maximize': False, 'foreach': None,
                 'capturable': False, 'differentiable': False, 'fused': None, 'decoupled_weight_decay': False,
                 'params': list(range(len(a.params)))}
        if self.schedule is not None:
            group['initial_lr'] = self.schedule.initial
            group['max_lr'] = self.schedule.max_lr
            group['min_lr'] = self.schedule.min_lr
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        """Loads torch.optim.Adam's layout (this class's own output, or a checkpoint of the reference trainer).  Validates the
        parameter count and every shape; parameters without an entry (torch keeps none for a parameter that never received
        a gradient) start from zero moments and step 0.  Hyper-parameters come from param_groups[0]; under a OneCycle schedule
        its rates (max_lr / initial_lr / min_lr) are taken from there as well.  The schedule position `t` becomes the largest
        per-parameter step (pass OneCycle.load_state_dict's return value to `self.t` when the scheduler dict is at hand)."""
        a = self.arena
        if 't' in sd and 'm' in sd:                    # round-2 private format: flat buffers in arena order
            if sd['m'].numel() != self.m.numel():
                raise ValueError('FusedAdam: flat moment buffer of %d elements does not fit this arena (%d)'
                                 % (sd['m'].numel(), self.m.numel()))
            self.t = int(sd['t'])
            self.steps = [self.t] * len(a.params)
            self.m.copy_(sd['m'])
            self.v.copy_(sd['v'])
            return
        groups = sd['param_groups']
        if len(groups) != 1:
            raise ValueError('FusedAdam: expected ONE param group (nn/trainer.py:172 builds one), got %d' % len(groups))
        g = groups[0]
        if len(g['params']) != len(a.params):
            raise ValueError('FusedAdam: checkpoint has %d parameters, the model %d' % (len(g['params']), len(a.params)))
        if g.get('amsgrad') or g.get('maximize'):
            raise ValueError('FusedAdam: amsgrad / maximize checkpoints are not supported')
        order = self._module_order()
        self.m.zero_()
        self.v.zero_()
        steps = [0] * len(a.params)
        for key, st in sd['state'].items():
            i = g['params'].index(key) if key in g['params'] else int(key)
            ai = order[i]
            o, n = a.segment(ai)
            shape = a.params[ai].shape
            for name, dst in (('exp_avg', self.m), ('exp_avg_sq', self.v)):
                if tuple(st[name].shape) != tuple(shape):
                    raise ValueError('FusedAdam: %s of parameter %d has shape %s, the model\'s is %s'
                                     % (name, i, tuple(st[name].shape), tuple(shape)))
                dst[o:o + n].copy_(st[name].reshape(-1))
            steps[ai] = int(float(st['step']))
        self.steps = steps
        self.t = max(steps) if steps else 0
        self.lr = float(g['lr']) if self.schedule is None else self.lr
        self.betas = tuple(float(b) for b in g['betas'])
        self.eps, self.weight_decay = float(g['eps']), float(g['weight_decay'])
        self.last_lr = float(g['lr'])
        if self.schedule is not None:
            self.schedule.set_rates(g)
