"""Input side of a training step (reference: nn/trainer.py:93 `features = batch['features'].to(device)` from pageable
memory, after nn/data/transforms.py:35-50 standardised every sample on the CPU).

BatchStager keeps two pinned host buffers and a copy stream per device: batch i+1 is copied while step i computes, the
compute stream only waits on the copy's event, and the per-axis standardisation `(x - shift) / scale` runs as one kernel
on the device instead of per sample on the host."""
import torch

from . import ops


class BatchStager:
    def __init__(self, device, shift=None, scale=None, slots=2):
        self.device = torch.device(device)
        self.shift, self.scale = shift, scale
        self.stream = torch.cuda.Stream(device=self.device)
        self.slots = [None] * slots          # (pinned buffer, event of the last copy out of it)
        self.turn = 0

    def stage(self, features):
        """features: CPU tensor [B, N, C] (any float dtype) -> fp32 device tensor, standardised if stats were given."""
        if features.is_cuda:
            out = features.float()
        else:
            i = self.turn
            self.turn = (i + 1) % len(self.slots)
            slot = self.slots[i]
            if slot is None or slot[0].shape != features.shape:
                slot = (torch.empty(features.shape, dtype=torch.float32, pin_memory=True), torch.cuda.Event())
                self.slots[i] = slot
            buf, ev = slot
            ev.synchronize()                 # the previous copy out of this buffer has finished
            buf.copy_(features)
            compute = torch.cuda.current_stream(self.device)
            with torch.cuda.stream(self.stream):
                out = torch.empty(buf.shape, device=self.device, dtype=torch.float32)
                out.copy_(buf, non_blocking=True)
                ev.record(self.stream)
            compute.wait_event(ev)
            out.record_stream(compute)
        if self.shift is not None:
            out = ops.standardize(out.view(-1, out.shape[-1]), self.shift, self.scale).view(out.shape)
        return out
