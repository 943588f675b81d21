"""Test infrastructure: evaluate the oracle on the SAME side of every ReLU as the build.

A gradient is discontinuous where a pre-activation crosses zero, exactly like a kNN graph is discontinuous at a distance
tie.  The parity tests already evaluate the fp64 oracle "on the build's own kNN graphs"; this module does the same for the
ReLU decisions and for the winners of the max aggregation (two messages of a point within 1e-6 of each other: the gradient
of that channel goes to a different edge — the bias gradient does not notice, the weight gradient does): the build's masks (a > 0 of every [Linear -> ReLU -> BatchNorm] block, in forward order) are captured from
the C-ABI calls, and the oracle's nn.ReLU modules are made to take the build's branch wherever the two disagree — which is
only legitimate, and is asserted, where the oracle's own pre-activation is within the forward tolerance of zero.

Why it is needed (profiles/r03_a_grad_diag.md): on `full3d_k16` ONE of 1.6 M pre-activations of the second EdgeConv layer
is 1e-6 from zero; the build rounds it to the other side than fp64 does, and that single decision moves the first-block
gradients by 3e-3 of max|grad| (the fp32 torch oracle has the same kind of flip in the first layer: 3.5e-4).  With the
decisions aligned the build's gradients agree with fp64 to ~1e-6.
"""
import contextlib

import torch

FLIP_Z_TOL = 5e-5      # a build/oracle ReLU disagreement is accepted only where |z_fp64| is below this
ARGSEL_TOL = 5e-5      # ... and a different max-aggregation winner only where the two messages are this close


class Decisions(list):
    """The build's ReLU masks in forward order (the list itself) + `.argsel`: per max-aggregated EdgeConv layer the slots
    (argmax, argmin) of the pre-BatchNorm activation over each point's k messages, LongTensors [B*N, F]."""

    def __init__(self, *a):
        super().__init__(*a)
        self.argsel = []

    def absorb(self, other):
        self.extend(other)
        self.argsel.extend(other.argsel)


@contextlib.contextmanager
def capture_relu_masks():
    """Yields a list that receives, for every gpe_edge_mlp_fwd launch of the product path (EdgeConv MLP blocks and dense
    MLPs alike), the boolean masks `a > 0` of the blocks it evaluates: the gather variant covers block 0 (recomputed from
    P_i + Q_j with the kernel's own single fp32 add) and block 1, the dense variant one block."""
    from gpe_amd import _lib as L
    masks = Decisions()
    orig = L.call

    def spy(name, *args):
        orig(name, *args)
        if name != 'gpe_edge_mlp_fwd':
            return
        mode, PQ, ldpq, jg = args[0], args[1], args[2], args[3]
        k, Cin, Cout, a = args[8], args[9], args[10], args[13]
        if mode == 0:
            H = Cin
            rows = jg.numel()
            i = torch.arange(rows, device=PQ.device) // k
            z0 = PQ[i, :H] + PQ[jg.view(-1).long(), H:2 * H]
            masks.append((z0 > 0).cpu())
        masks.append((a[:, :Cout] > 0).cpu())
        if args[16]:                                   # aggregated block: the winners of the max / min over the k messages
            masks.argsel.append((args[19][:, :Cout].long().cpu(), args[20][:, :Cout].long().cpu()))

    L.call = spy
    try:
        yield masks
    finally:
        L.call = orig


def align_relus(model, masks):
    """Hooks every nn.ReLU of the oracle `model`: the i-th ReLU evaluated takes the i-th captured mask.  Returns a dict
    that is filled during the forward: flips (count), max_z (largest |z| of a flipped element), n (elements seen)."""
    stats = {'flips': 0, 'max_z': 0.0, 'n': 0, 'used': 0, 'per_relu': []}
    queue = list(masks)

    def hook(_m, inp, _out):
        z = inp[0]
        assert queue, 'the oracle evaluated more ReLUs than the build launched MLP blocks'
        m = queue.pop(0)
        assert tuple(m.shape) == tuple(z.shape), ('ReLU order mismatch between build and oracle', m.shape, z.shape)
        stats['used'] += 1
        flip = (z.detach() > 0) != m
        nf = int(flip.sum())
        stats['n'] += z.numel()
        mz = 0.0
        if nf:
            stats['flips'] += nf
            mz = z.detach()[flip].abs().max().item()
            stats['max_z'] = max(stats['max_z'], mz)
        # per-ReLU record (printed by the whole-batch tests): elements, flips, largest |z| among them, rms of z
        stats['per_relu'].append((stats['used'], z.numel(), nf, mz, float(z.detach().double().pow(2).mean().sqrt())))
        return z * m.to(z.dtype)

    handles = [mod.register_forward_hook(hook) for mod in model.modules() if isinstance(mod, torch.nn.ReLU)]
    stats['handles'] = handles
    stats['pending'] = queue
    return stats


def check_alignment(stats, n_masks, z_tol=FLIP_Z_TOL, max_frac=2e-5):
    """After the oracle forward: every mask was consumed and every flip sat inside the forward tolerance.  (The multi-step
    trajectory test passes the distance it allows between the two parameter sets as `z_tol`: there the pre-activations of
    build and oracle differ by the drift of the weights, not only by rounding.)"""
    for h in stats['handles']:
        h.remove()
    for i, n, nf, mz, rms in stats['per_relu']:
        if nf:
            print('relu_align (%s): ReLU #%d  %d elements, %d decisions differ from the build, largest |z| among them %.2e (rms z %.2e)' % (stats.get('tag', 'oracle'), i, n, nf, mz, rms))
    assert stats['used'] == n_masks and not stats['pending'], (stats['used'], n_masks)
    assert stats['max_z'] < z_tol, 'a ReLU decision differs where |z_fp64| = %.2e' % stats['max_z']
    assert stats['flips'] <= max(4, max_frac * stats['n']), (stats['flips'], stats['n'])
