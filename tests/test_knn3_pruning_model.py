"""CPU model of the tile pruning of csrc/gpe_knn3.hip (the xyz kNN of the first EdgeConv layer) against the C oracle.

The HIP kernel itself is held bit-exact on the GPU (tests/test_gpu_kernels.py::test_knn_xyz_sorted_cloud_bit_exact).  This file
pins the ARGUMENT it rests on where no GPU is needed: with the oracle's own distances (oracle/knn_ref.c through
ref_path.sqdist_one_cloud) and the kernel's float32 bound arithmetic restated in numpy, a scan that
  * visits the 64-point tiles of a Morton-sorted cloud in ascending order of a box-to-box lower bound,
  * stops at the first tile whose bound (scaled by 1 - 2^-17) exceeds the largest k-th distance among the wave's queries, and
  * lets a query skip a visited tile whose point-to-box bound exceeds its own k-th distance
returns exactly the oracle's neighbour lists, on data with exact ties, duplicated points, a zero-width axis and far clusters —
and really prunes (a sheet-like cloud visits a fraction of its tiles)."""
import numpy as np
import pytest
import torch

from oracle import ref_path as O

F32 = np.float32
QW = 4                       # queries per wave (K3_QW)
SHRINK = F32(0.99999237)     # 1 - 2^-17


def _spread(v):
    return (v & 1) | ((v & 2) << 2) | ((v & 4) << 4) | ((v & 8) << 6)


def _bound_bits(gap):
    """gap [.., 3] float32 >= 0 -> bit patterns of the scaled squared bound, float32 operation by operation"""
    lb = (gap[..., 0] * gap[..., 0]).astype(F32)
    lb = (lb + (gap[..., 1] * gap[..., 1]).astype(F32)).astype(F32)
    lb = (lb + (gap[..., 2] * gap[..., 2]).astype(F32)).astype(F32)
    lb = (lb * SHRINK).astype(F32)
    return lb.view(np.uint32).astype(np.int64)


def pruned_knn(x, k, rng):
    """x [N, 3] float32 -> (indices [N, k], mean tiles visited per query, tiles)"""
    N = x.shape[0]
    tiles = (N + 63) // 64
    dist = O.sqdist_one_cloud(torch.from_numpy(x)).numpy()                  # the oracle's chain, [N, N] float32
    lo, hi = x.min(0), x.max(0)
    r = (hi - lo).astype(F32)
    inv = np.where((r > 0) & np.isfinite(r), F32(16.) / np.where(r > 0, r, F32(1)), F32(0)).astype(F32)
    cell = ((x - lo).astype(F32) * inv).astype(F32).astype(np.int64).clip(0, 15)
    code = _spread(cell[:, 0]) | (_spread(cell[:, 1]) << 1) | (_spread(cell[:, 2]) << 2)
    perm = np.lexsort((rng.random(N), code))                                # arbitrary order inside a cell (the kernel's atomics)
    xs = x[perm]
    tl = np.stack([xs[64 * t:64 * t + 64].min(0) for t in range(tiles)])
    th = np.stack([xs[64 * t:64 * t + 64].max(0) for t in range(tiles)])
    out = np.zeros((N, k), dtype=np.int64)
    pairs = 0
    ALL = np.uint64(2 ** 64 - 1)
    for q0 in range(0, N, QW):
        qs = np.arange(q0, min(N, q0 + QW))
        bl, bh = xs[qs].min(0), xs[qs].max(0)
        lists = [np.full(k, ALL, dtype=np.uint64) for _ in qs]

        def kth_bits(a):
            return int(lists[a][k - 1] >> np.uint64(32))

        def visit(t, first):
            nonlocal pairs
            seg = np.arange(64 * t, min(N, 64 * t + 64))
            for a, qq in enumerate(qs):
                if not first and lbq[a][t] > kth_bits(a):
                    continue
                pairs += 1
                d = dist[perm[qq], perm[seg]]
                key = (d.view(np.uint32).astype(np.uint64) << np.uint64(32)) | perm[seg].astype(np.uint64)
                lists[a] = np.sort(np.concatenate([lists[a], key[key < lists[a][k - 1]]]))[:k]

        lbq = [_bound_bits(np.maximum(np.maximum(tl - xs[qq], xs[qq] - th), F32(0)).astype(F32)) for qq in qs]
        t0 = q0 >> 6
        visit(t0, True)
        lbb = _bound_bits(np.maximum(np.maximum(tl - bh, bl - th), F32(0)).astype(F32))
        for t in np.lexsort((np.arange(tiles), lbb)):
            if t == t0:
                continue
            if lbb[t] > max(kth_bits(a) for a in range(len(qs))):
                break                                                        # sorted: every later tile is pruned too
            visit(t, False)
        for a, qq in enumerate(qs):
            out[perm[qq]] = (lists[a] & np.uint64(0xffffffff)).astype(np.int64)
    return out, pairs / N, tiles


def _cloud(kind, N, rng):
    if kind == 'gauss':
        x = rng.standard_normal((N, 3))
    elif kind == 'lattice':                                  # exact ties, duplicated points
        x = rng.integers(0, 4, (N, 3)).astype(float)
    elif kind == 'planar':                                   # a zero-width axis
        x = rng.standard_normal((N, 3))
        x[:, 2] = 0.5
    elif kind == 'clusters':                                 # tiny distance differences far from the origin
        x = (rng.standard_normal((8, 3)) * 20)[rng.integers(0, 8, N)] + 1e-2 * rng.standard_normal((N, 3))
    else:                                                    # 'sheet': what a garment scan looks like
        u, v = rng.random(N) * 2 * np.pi, rng.random(N) * 1.5
        x = np.stack([0.3 * np.cos(u) * (1 + 0.2 * np.sin(3 * v)), v, 0.2 * np.sin(u)], 1) + 0.003 * rng.standard_normal((N, 3))
    return np.ascontiguousarray(x.astype(F32))


@pytest.mark.parametrize('kind', ['gauss', 'lattice', 'planar', 'clusters', 'sheet'])
@pytest.mark.parametrize('N,k', [(577, 9), (300, 16), (130, 64)])
def test_pruned_scan_returns_the_oracle_lists(kind, N, k):
    rng = np.random.default_rng(N + k)
    x = _cloud(kind, N, rng)
    ref = O.knn_local(torch.from_numpy(x), 1, k).numpy()
    got, per_query, tiles = pruned_knn(x, k, rng)
    assert (got == ref).all(), '%d queries differ' % int((got != ref).any(1).sum())
    assert per_query <= tiles


def test_pruning_prunes_on_a_sheet():
    rng = np.random.default_rng(3)
    x = _cloud('sheet', 1024, rng)
    got, per_query, tiles = pruned_knn(x, 16, rng)
    assert (got == O.knn_local(torch.from_numpy(x), 1, 16).numpy()).all()
    assert per_query < 0.5 * tiles, (per_query, tiles)      # measured: ~5 of 16
