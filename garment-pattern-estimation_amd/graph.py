"""A training step captured ONCE as a hipGraph and replayed (host-bound shapes: the reference's shipped training shape — BASELINE
cfg 1, N = 1024, batch 8, k = 5 — is ~200 launches of 3 - 30 us each; the Python side of a launch costs more than the kernel).

    sg = StepGraph(lambda feats, gt: model.loss(model(feats), gt)[0], optimizer)      # optimizer: optim.FusedAdam
    for feats, gt in loader:
        loss = sg.step(feats, gt)          # the first `warmup` calls run eagerly; then one capture; then replays

What a replay cannot carry in frozen kernel arguments is fed through device memory in front of the graph launch:
  * the reference draws the LSTM start states (and dropout masks) on the CPU generator inside forward() (nn/net_blocks.py:391-392):
    every such tensor has a static device buffer; before each replay the host draws them IN THE SAME ORDER from the same generator
    into pinned memory and queues the copies (HostDrawn below; net_blocks._init_tenzor and ops._dropout_mask go through it);
  * Adam's step-dependent scalars (OneCycle learning rate, bias corrections) live in a two-float device buffer per arena run
    (include/gpe_hip.h gpe_adam_step_dev);
  * the inputs are copied into static buffers.
The fp16-activation guard (ops.set_half_act_guard) keeps working in its default 'fallback' mode: the amax words of the captured forward
are read asynchronously after every replay; a layer that trips it invalidates the graph, and the next step is captured again on the
fp32 path.  'strict' (a host read inside the forward) cannot be captured and raises.

One process per GPU, world size 1: a gradient exchange inside the captured region is not supported (RCCL launches are not captured
here).  The captured step is single-stream; the tensors a replay returns are static buffers, overwritten by the next replay."""
import ctypes

import torch

from . import _lib as L
from . import ops


class HostDrawn:
    """Host-drawn device tensors of a step (start states, dropout masks), drawn in call order from the CPU generator."""

    def __init__(self, device):
        self.device = device
        self.entries = []            # [shape, fill_fn, device tensor, [pinned x 2], [event x 2], flip]
        self.cursor = 0
        self.capturing = False

    def begin_step(self):
        """Draw every registered tensor (registration order = call order = the reference's draw order) and queue its copy."""
        self.cursor = 0
        for e in self.entries:
            self._fill(e)

    def _fill(self, e):
        e[5] ^= 1
        pin, ev = e[3][e[5]], e[4][e[5]]
        ev.synchronize()                       # the copy that last read this pinned buffer (two steps ago) has run
        e[1](pin)
        e[2].copy_(pin, non_blocking=True)
        ev.record()

    def get(self, shape, fill_fn):
        shape = tuple(shape)
        i = self.cursor
        self.cursor += 1
        if i == len(self.entries):
            if self.capturing:
                raise RuntimeError('StepGraph: the captured step draws a host tensor the warm-up steps did not draw (shape %s): the '
                                   'step must issue the same sequence of launches every time' % (shape,))
            e = [shape, fill_fn, torch.empty(shape, device=self.device),
                 [torch.empty(shape, pin_memory=True) for _ in range(2)], [torch.cuda.Event() for _ in range(2)], 0]
            self.entries.append(e)
            self._fill(e)
        e = self.entries[i]
        if e[0] != shape:
            raise RuntimeError('StepGraph: host-drawn tensor %d changed shape (%s -> %s)' % (i, e[0], shape))
        return e[2]


class StepGraph:
    def __init__(self, forward_loss, optimizer, warmup=2, device=None):
        """forward_loss(*inputs) -> scalar loss tensor (forward + loss; backward and optimizer.step() are run here).
        optimizer: optim.FusedAdam.  warmup: eager steps before the capture (>= 1: lazily built state — pack plans, workspaces,
        kernel attributes — must exist before a capture)."""
        if not (hasattr(optimizer, 'step_captured') and hasattr(optimizer, 'advance_captured')):
            raise TypeError('StepGraph needs an optimizer whose step-dependent scalars live in device memory (optim.FusedAdam); a '
                            'torch optimizer bakes them into its kernels\' arguments')
        self.forward_loss = forward_loss
        self.opt = optimizer
        self.warmup = max(int(warmup), 1)
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.host = HostDrawn(self.device)
        self.calls = 0
        self.graph = None
        self.static_in = None
        self.loss = None
        self.guards = []             # (HalfActGuard, amax word) pairs met during the capture
        self.hyper = []              # per arena run: (device pair, [pinned pair x 2], [event x 2])
        self.runs = None
        self.replays = 0
        self.captures = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            raise RuntimeError('StepGraph captures a single-process step (world size 1)')

    # ---- inputs: tensors, or (nested) dicts / lists / tuples of tensors ----
    def _flat(self, obj, out):
        if isinstance(obj, torch.Tensor):
            out.append(obj)
        elif isinstance(obj, dict):
            for k in obj:
                self._flat(obj[k], out)
        elif isinstance(obj, (list, tuple)):
            for v in obj:
                self._flat(v, out)
        return out

    def _like(self, obj):
        if isinstance(obj, torch.Tensor):
            return torch.empty_like(obj, device=self.device) if obj.dtype.is_floating_point or True else obj
        if isinstance(obj, dict):
            return {k: self._like(v) for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            return type(obj)(self._like(v) for v in obj)
        return obj

    def _stage(self, inputs):
        if self.static_in is None:
            self.static_in = self._like(inputs)
        dst, src = self._flat(self.static_in, []), self._flat(inputs, [])
        if len(dst) != len(src) or any(d.shape != s.shape or d.dtype != s.dtype for d, s in zip(dst, src)):
            raise RuntimeError('StepGraph: the inputs changed shape or type; a captured step is one fixed shape')
        for d, s in zip(dst, src):
            if d.data_ptr() != s.data_ptr():
                d.copy_(s, non_blocking=True)

    def _eager(self):
        loss = self.forward_loss(*self.static_in)
        loss.backward()
        self.opt.step()
        return loss

    def _capture(self):
        if ops._HALF_ACT_GUARD == 'strict':
            raise RuntimeError("StepGraph: the 'strict' half-activation guard reads the device inside the forward and cannot be captured")
        self.guards = []
        self.hyper_slot(7)                 # pinned memory cannot be allocated inside a capture: eight runs are plenty
        self.graph = torch.cuda.CUDAGraph()
        self.host.capturing = True
        ops.CAPTURE = self
        try:
            with torch.cuda.graph(self.graph, stream=self.stream):
                loss = self.forward_loss(*self.static_in)
                loss.backward()
                self.runs = self.opt.step_captured(self)
        finally:
            ops.CAPTURE = None
            self.host.capturing = False
        self.loss = loss.detach()
        self.captures += 1

    def step(self, *inputs):
        """One training step on `inputs`; returns the loss (a static device tensor from the first replay on).  The caller's
        stream is made to wait for the step, so the returned tensor can be used there right away."""
        outer = torch.cuda.current_stream(self.device)
        try:
            return self._step(inputs)
        finally:
            if outer != self.stream:
                outer.wait_stream(self.stream)

    def _step(self, inputs):
        outer = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.stream):
            if outer != self.stream:
                self.stream.wait_stream(outer)       # the inputs were produced there
            self._stage(inputs)
            ops.HOST_DRAWN = self.host
            try:
                self.host.begin_step()
                self.calls += 1
                if self.calls <= self.warmup:
                    return self._eager()
                if self.graph is None:
                    self._capture()
                    self.host.cursor = 0
                self.opt.advance_captured(self, self.runs)
                self.graph.replay()
                self.replays += 1
                tripped = False
                for guard, word in self.guards:
                    was = guard.disabled
                    guard.watch(word)
                    guard.allow()
                    tripped = tripped or (guard.disabled and not was)
                if tripped:
                    self.graph = None            # the next step is captured again, on the fp32 rows of the layer that tripped
                return self.loss
            finally:
                ops.HOST_DRAWN = None

    def synchronize(self):
        self.stream.synchronize()

    # ---- Adam's step-dependent scalars ----
    def hyper_slot(self, i):
        if i >= len(self.hyper) and ops.CAPTURE is not None:
            raise RuntimeError('StepGraph: more optimizer runs than pre-allocated scalar slots')
        while len(self.hyper) <= i:
            self.hyper.append((torch.zeros(2, device=self.device), [torch.zeros(2, pin_memory=True) for _ in range(2)],
                               [torch.cuda.Event() for _ in range(2)], [0]))
        return self.hyper[i]

    def write_hyper(self, i, lr, beta1, beta2, step):
        dev, pins, evs, flip = self.hyper_slot(i)
        flip[0] ^= 1
        pin, ev = pins[flip[0]], evs[flip[0]]
        ev.synchronize()
        rc = L.lib().gpe_adam_hyper(float(lr), float(beta1), float(beta2), int(step), ctypes.c_void_p(pin.data_ptr()))
        if rc != 0:
            raise RuntimeError('gpe_adam_hyper failed with code %d' % rc)
        dev.copy_(pin, non_blocking=True)
        ev.record()
