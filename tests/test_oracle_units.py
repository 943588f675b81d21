"""not-gpu: unit checks of the oracle's restated third-party ops against their published definitions."""
import numpy as np
import torch

from oracle import ref_path as O


def test_knn_oracle_against_float64_bruteforce():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3 * 50, 5, generator=g)
    idx = O.knn_local(x, 3, 7).view(3, 50, 7)
    xd = x.double().view(3, 50, 5)
    for b in range(3):
        d = ((xd[b][:, None, :] - xd[b][None, :, :]) ** 2).sum(-1)
        ref = d.argsort(dim=1, stable=True)[:, :7]
        assert torch.equal(idx[b], ref)          # random data: no near-ties at fp32 resolution
        assert torch.equal(idx[b][:, 0], torch.arange(50))   # self is the nearest neighbour


def test_knn_oracle_tie_rule_and_bad_args():
    x = torch.zeros(6, 2)                       # all points identical: every distance ties at 0
    idx = O.knn_local(x, 1, 4)
    assert torch.equal(idx, torch.arange(4).expand(6, 4))    # lower index wins
    try:
        O.knn_local(torch.randn(4, 2), 1, 5)
        assert False
    except ValueError:
        pass


def test_sparsemax_definition():
    g = torch.Generator().manual_seed(1)
    z = torch.randn(64, 23, generator=g, dtype=torch.float64, requires_grad=True)
    p = O.Sparsemax(dim=1)(z)
    assert torch.allclose(p.sum(1), torch.ones(64, dtype=torch.float64))
    assert (p >= 0).all() and (p == 0).any()
    # projection onto the simplex: p = max(z - tau, 0) with a single tau per row
    tau = (z - p)[p > 0]
    rows = torch.nonzero(p > 0)[:, 0]
    for r in range(64):
        t = tau[rows == r]
        assert (t - t[0]).abs().max() < 1e-12
    assert torch.autograd.gradcheck(lambda t: O.Sparsemax(dim=1)(t), (z[:4].detach().requires_grad_(),))


def test_global_pools():
    x = torch.arange(24.).view(6, 4)
    batch = torch.tensor([0, 0, 0, 1, 1, 1])
    assert torch.equal(O.global_mean_pool(x, batch, 2), x.view(2, 3, 4).mean(1))
    assert torch.equal(O.global_max_pool(x, batch, 2), x.view(2, 3, 4).max(1).values)
    assert torch.equal(O.global_add_pool(x, batch, 2), x.view(2, 3, 4).sum(1))
