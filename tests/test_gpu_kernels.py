"""-m gpu: kernel-level parity of libgpe_hip.so (through the C ABI via ops.py) against the CPU oracle.

Bars: kNN indices BIT-EXACT vs oracle/knn_ref.c; fp32 tensors compared with the fp64 oracle evaluated on the
SAME graph, tolerance written next to each check (matrix products are exact-fp32 MFMA fma chains, so the error
budget is summation-order roundoff: ~1e-6 relative for K<=256, amplified by 1/sqrt(var+eps) through BatchNorm)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gpe():
    if not torch.cuda.is_available():
        pytest.skip('needs the MI355X')
    import gpe_amd
    return gpe_amd


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def relerr_fro(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


# Tolerances per arithmetic of the fused edge GEMMs.  'f32' = exact-fp32 MFMA (summation-order roundoff only): this is
# the library default and the mode every parity claim is made in.  'bf16x3' = opt-in fast mode, split-bf16 products
# (~1e-5 relative per GEMM): the forward still meets the north-star 1e-4 bar, but gradients do not inherit it — the
# BatchNorm backward coefficients are residuals of large sums (taken from the weight-gradient product G),
# which turns the 1e-5 product error into a coherent ~1e-2 relative error of the encoder gradients, and a forward
# perturbation of 1e-5 flips a handful of ReLU masks per 1e5 activations (each flip moves ONE row of a per-row gradient
# by its full value, hence the Frobenius norm).  Measured worst cases: 1.3e-2 (layer dx), 6.5e-3 (dense MLP dx).
TOL = {
    'f32': dict(fwd=5e-5, dx=2e-4, dparam=3e-4, mlp_dx=3e-4, mlp_dw=5e-4, mlp_db=5e-3, norm=relerr),
    # three-term split-bf16 products (24 mantissa bits): held to the SAME bars as the exact fp32 instruction, one layer at a
    # time here and end to end in tests/test_gpu_model.py (parity-grade once the oracle stands on the build's decisions,
    # DESIGN.md 5.2; superseded by f16x3)
    'bf16x6': dict(fwd=5e-5, dx=2e-4, dparam=3e-4, mlp_dx=3e-4, mlp_dw=5e-4, mlp_db=5e-3, norm=relerr),
    # two-term split-fp16 products on tensor-normalised operands (23 mantissa bits): the exact mode's bars
    'f16x3': dict(fwd=5e-5, dx=2e-4, dparam=3e-4, mlp_dx=3e-4, mlp_dw=5e-4, mlp_db=5e-3, norm=relerr),
    'bf16x3': dict(fwd=1e-4, dx=5e-2, dparam=5e-2, mlp_dx=5e-2, mlp_dw=5e-2, mlp_db=5e-2, norm=relerr_fro),
    # 'mixed' = split-bf16 row GEMMs + exact-fp32 weight-gradient / BN-coefficient products.  Measured (r02_a): forward as
    # bf16x3; parameter gradients 1e-3 .. 1.5e-2 of max|grad| — better than bf16x3 but NOT the exact mode's 2e-3: the
    # BatchNorm backward makes first-block gradients residuals of cancelling sums, which amplify the 2^-17 representation
    # error of a two-term bf16 split by ~1e3.  So this is a second opt-in approximate mode, reported beside `value`.
    'mixed': dict(fwd=1e-4, dx=3e-2, dparam=3e-2, mlp_dx=3e-2, mlp_dw=3e-2, mlp_db=3e-2, norm=relerr_fro),
}


# --------------------------------------------------------------------------------------------------
KNN_CASES = [(2, 64, 3, 4), (3, 200, 24, 5), (2, 256, 150, 16), (1, 130, 33, 20), (2, 2048, 3, 16),
             (1, 1024, 150, 16), (2, 70, 7, 64)]


@pytest.mark.parametrize('B,N,C,k', KNN_CASES)
def test_knn_bit_exact(gpe, B, N, C, k):
    from oracle import ref_path as O
    g = torch.Generator().manual_seed(B * 1000 + N + C + k)
    x = torch.randn(B * N, C, generator=g)
    ref = O.knn_local(x, B, k).to(torch.int32).view(B, N, k)
    got = gpe.ops.knn(x.cuda(), B, N, k).cpu()
    bad = (got != ref).any(-1).sum().item()
    assert bad == 0, '%d / %d queries differ' % (bad, B * N)


# xyz clouds (C = 3) of 128 .. 8192 points take the sorted-cloud kernel with tile pruning (gpe_knn3.hip): every shape of data that
# stresses the pruning bound or the (distance, index) order — exact ties and duplicated points, a constant axis (zero-width grid),
# all points identical, tight clusters far from the origin, a ragged last tile, k up to 64 and k close to N
K3_CASES = [(2, 128, 5), (3, 300, 16), (1, 1000, 20), (1, 4096, 20), (1, 8192, 16), (2, 200, 64), (1, 130, 64), (4, 577, 9)]


@pytest.mark.parametrize('B,N,k', K3_CASES)
@pytest.mark.parametrize('data', ['gauss', 'lattice', 'planar', 'clusters', 'same', 'line', 'surface'])
def test_knn_xyz_sorted_cloud_bit_exact(gpe, B, N, k, data):
    from oracle import ref_path as O
    if N >= 4096 and data not in ('gauss', 'lattice', 'surface'):
        pytest.skip('large clouds: three data kinds are enough')
    g = torch.Generator().manual_seed(B * 31 + N + k)
    if data == 'gauss':
        x = torch.randn(B * N, 3, generator=g)
    elif data == 'lattice':
        x = torch.randint(0, 4, (B * N, 3), generator=g).float()
    elif data == 'planar':
        x = torch.randn(B * N, 3, generator=g)
        x[:, 2] = 0.5
    elif data == 'clusters':
        x = (torch.randn(8, 3, generator=g) * 20)[torch.randint(0, 8, (B * N,), generator=g)] + 1e-2 * torch.randn(B * N, 3, generator=g)
    elif data == 'same':
        x = torch.full((B * N, 3), 0.3)
    elif data == 'line':
        x = torch.zeros(B * N, 3)
        x[:, 0] = torch.randn(B * N, generator=g)
    else:                                                   # a thin sheet wrapped around a body: what a garment scan looks like
        u = torch.rand(B * N, generator=g) * 6.2831853
        v = torch.rand(B * N, generator=g) * 1.5
        x = torch.stack([0.3 * torch.cos(u) * (1 + 0.2 * torch.sin(3 * v)), v, 0.2 * torch.sin(u)], 1)
        x = x + 0.003 * torch.randn(B * N, 3, generator=g)
    ref = O.knn_local(x.contiguous(), B, k).to(torch.int32).view(B, N, k)
    got = gpe.ops.knn(x.cuda(), B, N, k).cpu()
    bad = (got != ref).any(-1).sum().item()
    assert bad == 0, '%d / %d queries differ' % (bad, B * N)
    # padded rows (ldx = 4) through the same path
    buf = torch.zeros(B * N, 4)
    buf[:, :3] = x
    got = gpe.ops.knn(buf.cuda()[:, :3], B, N, k).cpu()
    assert torch.equal(got, ref)


# C -> fp16-pipe filter instance (blocks of 32 channels, NB = ceil(C / 32)): <2> 32, 33, 64; <5> 100 (two full steps), 150;
# <8> 200 (seven blocks: an odd last step), 256; C = 300 > 256 keeps the exact-product fp32 filter
MF_CASES = [(2, 300, 150, 152, 16), (1, 1000, 64, 64, 20), (3, 97, 32, 32, 5), (2, 513, 256, 256, 64), (1, 2048, 33, 36, 16),
            (2, 200, 100, 100, 10), (1, 700, 200, 200, 16), (1, 300, 300, 300, 8),
            (2, 130, 150, 152, 9)]


@pytest.mark.parametrize('B,N,C,ld,k', MF_CASES)
@pytest.mark.parametrize('data', ['gauss', 'offset', 'clusters', 'lattice'])
def test_knn_wide_rows_hard_data(gpe, B, N, C, ld, k, data):
    """padded wide rows (the float4 staging path) on data that is hard for a nearest-neighbour search: well-spread points,
    features with a large common offset, tight clusters far from the origin (tiny distance differences on large norms) and
    an integer lattice (exact ties).  (These cases were written for a bf16 matrix-pipe pre-filter with an exact recheck; it
    passed them but was slower than the all-exact kernel and was dropped — DESIGN.md §9.)"""
    from oracle import ref_path as O
    g = torch.Generator().manual_seed(B * 7 + N + C + k)
    if data == 'gauss':
        x = torch.randn(B * N, C, generator=g)
    elif data == 'offset':
        x = torch.randn(B * N, C, generator=g).abs() * 0.3 + 5.0 + torch.randn(1, C, generator=g)
    elif data == 'clusters':
        cen = torch.randn(8, C, generator=g) * 20
        x = cen[torch.randint(0, 8, (B * N,), generator=g)] + 1e-2 * torch.randn(B * N, C, generator=g)
    else:
        x = torch.randint(0, 3, (B * N, C), generator=g).float()
        x[:, 8:] = 0                                        # few distinct points: many exactly equal distances
    buf = torch.zeros(B * N, ld)
    buf[:, :C] = x
    ref = O.knn_local(x.contiguous(), B, k).to(torch.int32).view(B, N, k)
    got = gpe.ops.knn(buf.cuda()[:, :C], B, N, k).cpu()
    bad = (got != ref).any(-1).sum().item()
    assert bad == 0, '%d / %d queries differ' % (bad, B * N)


_KNN_ALT_WORKER = '''
import sys, torch
sys.path.insert(0, %r)
import gpe_amd
from oracle import ref_path as O
g = torch.Generator().manual_seed(5)
for (B, N, C, ld, k) in [(8, 256, 150, 152, 16), (9, 200, 33, 36, 5), (8, 130, 64, 64, 20), (8, 300, 3, 3, 16)]:
    for kind in ('gauss', 'clusters', 'lattice'):
        if kind == 'gauss':
            x = torch.randn(B * N, C, generator=g)
        elif kind == 'clusters':
            x = (torch.randn(8, C, generator=g) * 20)[torch.randint(0, 8, (B * N,), generator=g)] + 1e-2 * torch.randn(B * N, C, generator=g)
        else:
            x = torch.randint(0, 3, (B * N, C), generator=g).float(); x[:, 8:] = 0
        buf = torch.zeros(B * N, ld); buf[:, :C] = x
        ref = O.knn_local(x.contiguous(), B, k).to(torch.int32).view(B, N, k)
        got = gpe_amd.ops.knn(buf.cuda()[:, :C], B, N, k).cpu()
        assert torch.equal(got, ref), (B, N, C, k, kind, (got != ref).any(-1).sum().item())
print('alt path ok')
'''


@pytest.mark.parametrize('env', [{'GPE_KNN_EXACT': '1'}, {'GPE_KNN_SPLIT': '2'}, {'GPE_KNN_EXACT': '1', 'GPE_KNN_SPLIT': '2'},
                                 {'GPE_KNN_F32FILTER': '1'}, {'GPE_KNN_F32FILTER': '1', 'GPE_KNN_SPLIT': '2'},
                                 {'GPE_KNN_SORTED': '0'}, {'GPE_KNN_SORTED': '0', 'GPE_KNN_SPLIT': '2'}])
def test_knn_alternative_paths(gpe, env, tmp_path):
    """The paths the dispatcher no longer takes by default on wide rows — the all-exact kernel's float4 / float2 staging
    (C >= 16 goes through a matrix-pipe filter), the exact-product fp32 filter (16 <= C <= 256 now runs the fp16-pipe filter)
    and the candidate split with its list merge (forced: B >= 8 pins clouds to XCDs, which is what enables pieces), the all-pairs
    kernel on an xyz cloud (GPE_KNN_SORTED=0: 128 .. 8192 points otherwise take the sorted-cloud kernel) — stay bit-exact.  The overrides are read once per process, hence the subprocess."""
    import os, subprocess, sys
    script = tmp_path / 'w.py'
    script.write_text(_KNN_ALT_WORKER % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, GPE_DEBUG='1', **env), capture_output=True, text=True, timeout=600)   # (the switches are only read under GPE_DEBUG=1)
    assert r.returncode == 0 and 'alt path ok' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_knn_ties_lower_index_wins(gpe):
    from oracle import ref_path as O
    # integer lattice -> many exactly equal distances, plus duplicated points
    g = torch.Generator().manual_seed(7)
    x = torch.randint(0, 4, (2 * 192, 3), generator=g).float()
    ref = O.knn_local(x, 2, 9).to(torch.int32).view(2, 192, 9)
    got = gpe.ops.knn(x.cuda(), 2, 192, 9).cpu()
    assert torch.equal(got, ref)


def test_knn_candidate_split_full_size(gpe):
    """B = 32 clouds of 2048 x 150 features (the layer-2 shape of BASELINE cfg 2) through the default dispatch (matrix-pipe
    filter + exact rerank; the all-exact kernel would cut the candidate range in two here).  The first and the last cloud
    must be bit-exact against the C oracle (exact duplicates included)."""
    from oracle import ref_path as O
    B, N, C, k = 32, 2048, 150, 16
    g = torch.Generator().manual_seed(123)
    buf = torch.randn(B * N, 152, generator=g)
    buf[5 * N + 7] = buf[5 * N + 3]                       # exact duplicates: equal distances across the two halves' merge
    buf[31 * N + 1500] = buf[31 * N + 100]
    x = buf[:, :C]
    got = gpe.ops.knn(buf.cuda()[:, :C], B, N, k).cpu()
    for b in (0, 5, 31):
        ref = O.knn_local(x[b * N:(b + 1) * N].contiguous(), 1, k).to(torch.int32).view(N, k)
        assert torch.equal(got[b], ref), b
    # no duplicate neighbours anywhere
    srt = got.sort(-1).values
    assert (srt[..., 1:] != srt[..., :-1]).all()


@pytest.mark.parametrize('B,N,C,k,kind', [(32, 2048, 150, 16, 'random'), (32, 2048, 150, 16, 'curve'), (3, 700, 150, 16, 'reverse'),
                                          (9, 1000, 40, 20, 'random'), (2, 192, 64, 9, 'lattice')])
def test_knn_with_a_locality_order_is_still_exact(gpe, B, N, C, k, kind):
    """gpe_knn's order_in (round 6) is a SPEED hint of the wide-feature search: plane rows laid out in the caller's order, every query
    tile's scan started one tile before its own tile.  The answer must not depend on it: bit-exact against the C oracle and equal to
    the un-hinted search for a random permutation, for the curve order of an xyz search over correlated positions (features =
    smooth function of position + noise: the case the hint is for), for the reversed identity, and on a lattice with massive ties
    (the rerank's exact per-query fallback)."""
    from oracle import ref_path as O
    g = torch.Generator().manual_seed(B * 1000 + N)
    pos = torch.randn(B * N, 3, generator=g)
    if kind == 'lattice':
        x = torch.randint(0, 3, (B * N, C), generator=g).float()
    elif kind == 'curve':
        proj = torch.randn(3, C, generator=g)
        x = torch.tanh(pos @ proj) + 0.05 * torch.randn(B * N, C, generator=g)
    else:
        x = torch.randn(B * N, C, generator=g)
    buf = torch.zeros(B * N, (C + 3) // 4 * 4)
    buf[:, :C] = x
    xd = buf.cuda()[:, :C]
    if kind == 'curve':
        _, order = gpe.ops.knn(pos.cuda(), B, N, 8, want_order=True)        # the Morton-curve order of the xyz search
        assert torch.equal(order.long().sort(1).values.cpu(), torch.arange(N).expand(B, N))
    elif kind == 'reverse':
        order = torch.arange(N - 1, -1, -1, dtype=torch.int32).expand(B, N).contiguous().cuda()
    else:
        order = torch.stack([torch.randperm(N, generator=g) for _ in range(B)]).to(torch.int32).cuda()
    plain = gpe.ops.knn(xd, B, N, k)
    hinted, echo = gpe.ops.knn(xd, B, N, k, order=order, want_order=True)
    assert torch.equal(plain, hinted)
    assert torch.equal(echo.cpu(), torch.arange(N, dtype=torch.int32).expand(B, N))      # a filter-path search reports the identity
    for b in sorted({0, B // 2, B - 1}):
        ref = O.knn_local(x[b * N:(b + 1) * N].contiguous(), 1, k).to(torch.int32).view(N, k)
        assert torch.equal(hinted[b].cpu(), ref), b


def test_knn_strided_rows(gpe):
    from oracle import ref_path as O
    g = torch.Generator().manual_seed(11)
    buf = torch.randn(2 * 128, 152, generator=g)
    x = buf[:, :150]
    ref = O.knn_local(x.contiguous(), 2, 16).to(torch.int32).view(2, 128, 16)
    got = gpe.ops.knn(buf.cuda()[:, :150], 2, 128, 16).cpu()
    assert torch.equal(got, ref)


def test_knn_reverse(gpe):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3 * 200, 5, generator=g).cuda()
    idx = gpe.ops.knn(x, 3, 200, 6)
    off, edge = gpe.ops.knn_reverse(idx)
    idx, off, edge = idx.cpu(), off.cpu(), edge.cpu()
    for b in range(3):
        flat = idx[b].reshape(-1)
        for j in range(200):
            exp = torch.nonzero(flat == j).view(-1).to(torch.int32)
            got = edge[b, off[b, j]:off[b, j + 1]]
            assert torch.equal(got, exp), (b, j)
        assert off[b, 200].item() == 200 * 6


# --------------------------------------------------------------------------------------------------
LIN_CASES = [(64, 200, 200), (100, 8, 250), (736, 1000, 250), (32, 250, 1000), (33, 7, 3), (1000, 400, 150),
             (130, 23, 153), (65, 300, 520),
             # K <= 8 streaming-store kernel / deep reduce-GEMM with an unaligned V (the layer-1 [P|Q] projection shapes)
             (5000, 400, 3), (4100, 152, 6), (70000, 400, 3)]


@pytest.mark.parametrize('M,N,K', LIN_CASES)
def test_linear_fwd_bwd(gpe, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    gy = torch.randn(M, N, generator=g)
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    yr = xr @ wr.t() + br
    yr.backward(gy.double())
    xd, wd, bd = x.cuda().requires_grad_(), w.cuda().requires_grad_(), b.cuda().requires_grad_()
    y = gpe.ops.linear(xd, wd, bd)
    y.backward(gy.cuda())
    assert relerr(y, yr) < 2e-6
    assert relerr(xd.grad, xr.grad) < 2e-6
    assert relerr(wd.grad, wr.grad) < 2e-6
    assert relerr(bd.grad, br.grad) < 2e-6


def test_linear_strided_addend_act(gpe):
    ops = gpe.ops
    g = torch.Generator().manual_seed(5)
    Bn, T, H = 40, 5, 36
    hs = torch.randn(Bn, T + 1, H, generator=g).cuda()
    w = (torch.randn(4 * H, H, generator=g) / 6).cuda()
    add = torch.randn(Bn, T, 4 * H, generator=g).cuda()
    out = torch.zeros(T, Bn, 4 * H).cuda()
    wp = ops.pack_weight(w)
    for t in range(T):
        ops.linear_raw((hs[:, t], (T + 1) * H, 0, 0), wp, None, Bn, 4 * H, H, (out[t], 4 * H, 0, 0), 1,
                       (add[:, t], T * 4 * H, 0, 0))
    ref = torch.relu(torch.einsum('bth,gh->tbg', hs[:, :T].double().cpu(), w.double().cpu())
                     + add.double().cpu().transpose(0, 1))
    assert relerr(out, ref) < 2e-6
    # 2-level rows + transposed pack
    y = torch.empty(Bn * T, H).cuda()
    ops.linear_raw(ops._rows3d(add), ops.pack_weight(w, transpose=True), None, Bn * T, H, 4 * H, (y, H, 0, 0))
    ref2 = add.double().cpu().reshape(Bn * T, 4 * H) @ w.double().cpu()
    assert relerr(y, ref2) < 2e-6
    # K <= 8 streaming-store kernel with bias, addend and ReLU, 2-level output rows
    M, N, K = 5000, 24, 5
    a = torch.randn(M, K, generator=g).cuda()
    w3 = torch.randn(N, K, generator=g).cuda()
    b3 = torch.randn(N, generator=g).cuda()
    ad = torch.randn(M, N, generator=g).cuda()
    y3 = torch.zeros(M // 100, 101, N).cuda()                    # row r = o * 100 + i lives at y3[o, i]; slot 100 unused
    ops.linear_raw((a, K, 0, 0), ops.pack_weight(w3), b3, M, N, K, (y3, 101 * N, N, 100), 1, (ad, N, 0, 0))
    ref3 = torch.relu(a.double().cpu() @ w3.double().cpu().t() + b3.double().cpu() + ad.double().cpu())
    assert relerr(y3[:, :100].reshape(M, N), ref3) < 2e-6
    assert float(y3[:, 100].abs().max()) == 0.0


@pytest.mark.parametrize('rows,Mg,Ng', [(1000, 150, 200), (77, 8, 250), (5000, 1000, 250), (4096, 400, 3),
                                        (300, 23, 153), (65536, 400, 152), (10304, 1000, 252), (40000, 200, 7)])
def test_redgemm(gpe, rows, Mg, Ng):
    ops = gpe.ops
    g = torch.Generator().manual_seed(rows + Mg)
    u = torch.randn(rows, Mg, generator=g)
    v = torch.randn(rows, Ng, generator=g)
    G, cs = ops.redgemm_raw(ops._rows2d(u.cuda()), ops._rows2d(v.cuda()), rows, Mg, Ng)
    assert relerr(G, u.double().t() @ v.double()) < 3e-6
    assert relerr(cs, u.double().sum(0)) < 3e-6


@pytest.mark.parametrize('mode,B,N,k,Mg,Ng', [('dense', 1, 524288 + 13, 1, 200, 200), ('dense', 1, 524288 + 31, 1, 150, 200),
                                              ('gather', 3, 12001, 15, 150, 200), ('gather', 5, 6563, 16, 200, 200),
                                              ('gather', 32, 2048, 16, 200, 200)])
def test_edge_redgemm_producer_consumer_tiles(gpe, mode, B, N, k, Mg, Ng):
    """the producer/consumer reduce-GEMM (13 x 13 / 10 x 13 tile grids, >= 4 row tiles per CU) incl. a PARTIAL last row tile
    (rows % 32 != 0), padded operand pitches with non-finite pad columns, and the V shift; fp64 reference on the device."""
    ops, L = gpe.ops, gpe._lib
    g = torch.Generator().manual_seed(B * N + Mg)
    E = B * N * k
    pu = (Mg + 3) // 4 * 4 + 4
    ubuf = torch.full((E, pu), float('nan')).cuda()                    # pad columns must never reach a valid output
    ubuf[:, :Mg] = torch.randn(E, Mg, generator=g).cuda()
    shift = torch.randn(Ng, generator=g).cuda()
    G = torch.empty(Mg, Ng).cuda()
    cs = torch.empty(Mg).cuda()
    ws = torch.empty(L.query('gpe_redgemm_ws', Mg, Ng)).cuda()
    if mode == 'dense':
        pv = Ng + 8
        vbuf = torch.full((E, pv), float('nan')).cuda()
        vbuf[:, :Ng] = torch.randn(E, Ng, generator=g).cuda()
        L.call('gpe_edge_redgemm', ubuf, pu, 1, vbuf, pv, None, 0, None, shift, B, N, k, Mg, Ng, G, Ng, cs, ws, None, None, None, 0, None, 0, None, None, 0, None)
        vref = vbuf[:, :Ng].double() - shift.double()
    else:
        pq = torch.randn(B * N, 2 * Ng, generator=g).cuda()
        jg = (torch.randint(0, N, (B, N, k), generator=g) + torch.arange(B).view(B, 1, 1) * N).int().cuda()
        L.call('gpe_edge_redgemm', ubuf, pu, 0, None, 0, pq, 2 * Ng, jg, shift, B, N, k, Mg, Ng, G, Ng, cs, ws, None, None, None, 0, None, 0, None, None, 0, None)
        i = torch.arange(B * N, device='cuda').repeat_interleave(k)
        vref = torch.relu(pq[i, :Ng].double() + pq[jg.view(-1).long(), Ng:].double()) - shift.double()
    uref = ubuf[:, :Mg].double()
    assert relerr(G, uref.t() @ vref) < 3e-6
    assert relerr(cs, uref.sum(0)) < 3e-6


@pytest.mark.parametrize('rows,Mg,Ng,pitch', [(65536, 400, 3, 400), (5000, 150, 1, 152), (20001, 1000, 4, 1000),
                                             (4099, 37, 2, 40)])
def test_redgemm_thin(gpe, rows, Mg, Ng, pitch):
    """Ng <= 4 (the weight gradient of a Linear on raw positions): the streaming kernel, incl. padded U pitch with
    non-finite pad columns, a ragged row split and the V shift."""
    ops = gpe.ops
    g = torch.Generator().manual_seed(rows + Mg + Ng)
    ubuf = torch.full((rows, pitch), float('nan')).cuda()
    ubuf[:, :Mg] = torch.randn(rows, Mg, generator=g).cuda()
    u = ubuf[:, :Mg]
    v = torch.randn(rows, Ng, generator=g).cuda()
    shift = torch.randn(Ng, generator=g).cuda()
    G, cs = ops.redgemm_raw(ops._rows2d(u), ops._rows2d(v), rows, Mg, Ng, v_shift=shift)
    assert relerr(G, u.double().t() @ (v.double() - shift.double())) < 3e-6
    assert relerr(cs, u.double().sum(0)) < 3e-6


def test_redgemm_two_level_rows(gpe):
    """row-poor product over [sequence][step] descriptors (the LSTM weight gradients): the deep-reduction kernel."""
    ops = gpe.ops
    g = torch.Generator().manual_seed(3)
    Bn, T, GH, H = 150, 7, 200, 50
    dg = torch.randn(Bn, T, GH, generator=g).cuda()
    hs = torch.randn(Bn, T + 1, 52, generator=g).cuda()              # padded pitch, one extra slot per sequence
    G, cs = ops.redgemm_raw(ops._rows3d(dg), ops._rows3d(hs[:, :T, :H]), Bn * T, GH, H)
    ref = dg.double().cpu().reshape(-1, GH).t() @ hs[:, :T, :H].double().cpu().reshape(-1, H)
    assert relerr(G, ref) < 3e-6
    assert relerr(cs, dg.double().cpu().reshape(-1, GH).sum(0)) < 3e-6


def test_dense_gemms_on_the_bf16_pipe(gpe):
    """csrc/gpe_gemm_x6.hip (round 6): in f16x3 mode the row-rich dense products of gpe_linear / gpe_redgemm run on the bf16 pipe with
    three-term splits (six MFMAs per product, no scales).  Same bars as the exact kernels against fp64 — ragged M / N / K, bias +
    addend + ReLU, 16-byte and scalar output rows, two-level rows, centring shift, accumulate — and gpe_debug_set(16384) (the exact
    kernels) must give other bits: the new path really ran."""
    from gpe_amd import _lib as Lb
    ops = gpe.ops
    g = torch.Generator().manual_seed(11)
    prev = gpe.set_math('f16x3')
    try:
        # ---- NT: Y = act(A W^T + b + addend) ----
        for (M, N, K, act, addend, ldy) in [(65536, 400, 150, 0, False, 400), (65536, 150, 400, 0, False, 150), (5003, 250, 1000, 1, True, 252),
                                            (20000, 72, 100, 0, False, 72), (10304, 1000, 250, 0, True, 1000)]:
            ap = torch.randn(M, (K + 3) // 4 * 4, generator=g)              # rows padded to 16 bytes (what the package's tensors are)
            a = ap[:, :K]
            w = torch.randn(N, K, generator=g) / K ** 0.5
            b = torch.randn(N, generator=g)
            ad = torch.randn(M, N, generator=g) if addend else None
            ad_c = ad.cuda() if addend else None
            ref = a.double() @ w.double().t() + b.double() + (ad.double() if addend else 0)
            if act:
                ref = torch.relu(ref)
            ac, wp, bc = ap.cuda()[:, :K], ops.pack_weight(w.cuda()), b.cuda()
            outs = []
            for dbg in (0, 16384):
                Lb.query('gpe_debug_set', dbg)
                y = torch.zeros(M, ldy, device='cuda')
                ops.linear_raw((ac, ac.stride(0), 0, 0), wp, bc, M, N, K, (y, ldy, 0, 0), act, (ad_c, N, 0, 0) if addend else None)
                assert relerr(y[:, :N], ref) < 2e-6, (M, N, K, dbg)
                assert float(y[:, N:].abs().max()) == 0.0 if ldy > N else True
                outs.append(y)
            assert not torch.equal(outs[0], outs[1]), (M, N, K)
        # ---- TN: G = U^T (V - shift), colsum ----
        for (rows, Mg, Ng, shift, two_level) in [(10304, 1000, 250, False, True), (65536, 400, 150, True, False), (736, 1000, 250, False, True),
                                                 (20000, 77, 130, True, False), (65536, 150, 200, False, False)]:
            if two_level:
                T = 14 if rows % 14 == 0 else 23
                Bn = rows // T
                ub = torch.randn(Bn, T, Mg, generator=g).cuda()
                vb = torch.randn(Bn, T + 1, (Ng + 3) // 4 * 4, generator=g).cuda()
                ud, vd = ops._rows3d(ub), ops._rows3d(vb[:, :T, :Ng])
                u, v = ub.reshape(rows, Mg).double().cpu(), vb[:, :T, :Ng].reshape(rows, Ng).double().cpu()
            else:
                uc = torch.randn(rows, (Mg + 3) // 4 * 4, generator=g).cuda()
                vc = (torch.randn(rows, (Ng + 3) // 4 * 4, generator=g) + 3.0).cuda()
                ud, vd = (uc[:, :Mg], uc.stride(0), 0, 0), (vc[:, :Ng], vc.stride(0), 0, 0)
                u, v = uc[:, :Mg].double().cpu(), vc[:, :Ng].double().cpu()
            sh = v.mean(0).float().cuda() if shift else None
            ref = u.t() @ (v - (sh.double().cpu() if shift else 0))
            outs = []
            for dbg in (0, 16384):
                Lb.query('gpe_debug_set', dbg)
                G, cs = ops.redgemm_raw(ud, vd, rows, Mg, Ng, v_shift=sh)
                assert relerr(G, ref) < 3e-6, (rows, Mg, Ng, dbg)
                assert relerr(cs, u.sum(0)) < 3e-6
                outs.append(G)
            assert not torch.equal(outs[0], outs[1]), (rows, Mg, Ng)
            Lb.query('gpe_debug_set', 0)
            G2, cs2 = ops.redgemm_raw(ud, vd, rows, Mg, Ng, v_shift=sh, accumulate_into=(outs[0].clone(), cs.clone()))
            assert relerr(G2, 2 * ref) < 3e-6 and relerr(cs2, 2 * u.sum(0)) < 3e-6
    finally:
        Lb.query('gpe_debug_set', 0)
        gpe.set_math(prev)


# --------------------------------------------------------------------------------------------------
def _oracle_conv(C, H, Fo, k, seed):
    from oracle import ref_path as O
    torch.manual_seed(seed)
    conv = O.DynamicEdgeConv(O.MLP([2 * C, H, H, Fo]), k=k)
    # non-trivial BN affine parameters, including a negative scale (exercises the min-tracking path)
    with torch.no_grad():
        for blk in conv.nn:
            blk[2].weight.uniform_(0.5, 1.5)
            blk[2].bias.uniform_(-0.3, 0.3)
        conv.nn[2][2].weight[::5] *= -1
    return conv


def _product_conv(gpe, oconv, C, H, Fo, k):
    pconv = gpe.net_blocks.DynamicEdgeConv(gpe.net_blocks.MLP([2 * C, H, H, Fo]), k=k)
    pconv.load_state_dict(oconv.state_dict())
    return pconv.cuda()


# the (200, 150) cases run the register-stationary edge kernels: k = 16 the compile-time-slot variant, k = 5 / 8 / 10 the
# generic one (3 / 2 / 1 points per wave, ragged last tile: E is not a multiple of the tile), k = 20 / 24 / 32 the
# pseudo-point split (5 x 4, 3 x 8, 2 x 16 rows per point, folded afterwards), k = 17 (prime) the paired fallback.  (The fp64 oracle picks ONE argmax per (point, channel): a
# near-tie the build resolves the other way moves that channel's gradient to another edge — seen at (1, 61, 150, .., 17)
# with data seed 1 only, 9e-4 — so a case that fails on one seed and passes on five others is a tie, not a kernel fault.)
@pytest.mark.parametrize('B,N,C,H,Fo,k', [(2, 64, 3, 32, 24, 4), (2, 96, 24, 32, 24, 5), (2, 128, 3, 200, 150, 16),
                                          (1, 256, 150, 200, 150, 16), (3, 50, 6, 64, 30, 20),
                                          (2, 100, 3, 200, 150, 5), (1, 77, 150, 200, 150, 8), (2, 67, 3, 200, 150, 10),
                                          (1, 90, 150, 200, 150, 20), (2, 75, 150, 200, 150, 24),
                                          (1, 70, 3, 200, 150, 32), (1, 64, 150, 200, 150, 17),
                                          # k = 16 with a RAGGED last tile (130 points = 32 tiles of 4 points + 2): the straight-line
                                          # instances finish the tile's two absent points into their dummy image
                                          (1, 130, 150, 200, 150, 16), (3, 43, 3, 200, 150, 16),
                                          # k = 20 / 24 with E a multiple of 64: the straight-line four-row pseudo-point instances
                                          (2, 64, 150, 200, 150, 20), (1, 32, 3, 200, 150, 24)])
def test_edgeconv_layer_fwd_bwd(gpe, math_mode, B, N, C, H, Fo, k):
    from oracle import ref_path as O
    tol = TOL[math_mode]
    oconv = _oracle_conv(C, H, Fo, k, seed=B + N + C)
    pconv = _product_conv(gpe, oconv, C, H, Fo, k)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B * N, C, generator=g)
    wgt = torch.randn(B * N, Fo, generator=g)
    batch = torch.arange(B).repeat_interleave(N)

    xd = x.cuda().requires_grad_()
    pconv.train()
    out = pconv(xd, B, N)
    (out * wgt.cuda()).sum().backward()

    ref_idx = O.knn_local(x, B, k)
    assert torch.equal(pconv.last_knn.cpu().view(B * N, k).long(), ref_idx)

    o64 = copy.deepcopy(oconv).double().train()
    o64.knn_override = ref_idx
    xr = x.double().requires_grad_()
    out_r = o64(xr, batch)
    (out_r * wgt.double()).sum().backward()

    o32 = copy.deepcopy(oconv).train()
    o32.knn_override = ref_idx
    out_32 = o32(x.clone(), batch)
    err32 = relerr(out_32, out_r)
    err = relerr(out, out_r)
    print('edgeconv fwd relerr build=%.2e oracle-fp32=%.2e' % (err, err32))
    assert err < max(tol['fwd'], 20 * err32)
    e = tol.get('dx_norm', tol['norm'])(xd.grad, xr.grad)
    print('edgeconv %s dx err %.2e' % (math_mode, e))
    assert e < tol['dx']
    pn = dict(pconv.named_parameters())
    for n, p in o64.named_parameters():
        e = tol['norm'](pn[n].grad, p.grad)
        print('edgeconv %s %s err %.2e' % (math_mode, n, e))
        assert e < tol['dparam'], (n, e)
    # BatchNorm running statistics (momentum 0.1, unbiased variance) and the batch counter
    pb = dict(pconv.named_buffers())
    for n, bbuf in o64.named_buffers():
        if 'num_batches' in n:
            assert pb[n].item() == bbuf.item()
        else:
            assert relerr(pb[n], bbuf) < 1e-5, n


def test_edgeconv_eval_mode(gpe):
    from oracle import ref_path as O
    B, N, C, H, Fo, k = 2, 64, 3, 32, 24, 4
    oconv = _oracle_conv(C, H, Fo, k, seed=9)
    with torch.no_grad():
        for blk in oconv.nn:
            blk[2].running_mean.uniform_(0.1, 0.4)
            blk[2].running_var.uniform_(0.5, 1.5)
    pconv = _product_conv(gpe, oconv, C, H, Fo, k).eval()
    x = torch.randn(B * N, C, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        out = pconv(x.cuda(), B, N)
    o64 = copy.deepcopy(oconv).double().eval()
    o64.knn_override = pconv.last_knn.cpu().view(B * N, k).long()
    ref = o64(x.double(), torch.arange(B).repeat_interleave(N))
    assert relerr(out, ref) < 1e-5
    assert dict(pconv.named_buffers())['nn.0.2.num_batches_tracked'].item() == 0


def test_segment_mean(gpe):
    x = torch.randn(3 * 100, 37, generator=torch.Generator().manual_seed(4))
    xd = x.cuda().requires_grad_()
    y = gpe.ops.segment_mean(xd, 3, 100)
    y.sum().backward()
    assert relerr(y, x.double().view(3, 100, 37).mean(1)) < 1e-6
    assert relerr(xd.grad, torch.full((300, 37), 0.01)) < 1e-6


# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('Bn,In,Hh,T,L,Out', [(6, 40, 40, 5, 2, 40), (46, 40, 40, 14, 3, 8),
                                              (32, 250, 250, 23, 2, 250), (64, 250, 250, 14, 3, 8)])
def test_lstm_decoder_fwd_bwd(gpe, Bn, In, Hh, T, L, Out):
    from oracle import ref_path as O
    torch.manual_seed(Bn + T)
    odec = O.LSTMDecoderModule(In, Hh, Out, L, custom_init='kaiming_normal_')
    pdec = gpe.net_blocks.LSTMDecoderModule(In, Hh, Out, L, custom_init='kaiming_normal_')
    pdec.load_state_dict(odec.state_dict())
    pdec = pdec.cuda()
    enc = torch.randn(Bn, In, generator=torch.Generator().manual_seed(1))
    wgt = torch.randn(Bn, T, Out, generator=torch.Generator().manual_seed(2))
    o64 = copy.deepcopy(odec).double()
    er = enc.double().requires_grad_()
    torch.manual_seed(77)
    out_r = o64(er, T)
    (out_r * wgt.double()).sum().backward()
    ed = enc.cuda().requires_grad_()
    torch.manual_seed(77)
    out = pdec(ed, T)
    (out * wgt.cuda()).sum().backward()
    assert torch.equal(pdec.last_states[0].cpu(), o64.last_states[0].float())   # same RNG stream
    assert relerr(out, out_r) < 2e-5
    assert relerr(ed.grad, er.grad) < 1e-4
    pn = dict(pdec.named_parameters())
    for n, p in o64.named_parameters():
        e = relerr(pn[n].grad, p.grad)
        assert e < 1e-4, (n, e)


@pytest.mark.parametrize('kind,Bn,In,Hh,T,L', [('lstm', 32, 250, 250, 23, 2), ('lstm', 200, 250, 250, 14, 3), ('lstm', 46, 40, 40, 6, 2),
                                               ('gru', 70, 250, 250, 9, 2), ('lstm', 64, 100, 100, 5, 1)])
def test_recurrences_on_the_fp16_pipe(gpe, kind, Bn, In, Hh, T, L):
    """f16x3 arithmetic of the wavefront recurrences (round 4): with a PackPlan that holds the fp16 plane packs + amax words of
    the recurrent weights, the gate products of the forward run as three fp16 MFMAs per product block (state rows scaled by 2^12
    and split on the fly).  Same bars as the exact kernels against the fp64 oracle — forward, input gradient, every parameter
    gradient — and the exact path must give (nearly) the same numbers: a plan-less call in the same mode runs it."""
    from oracle import ref_path as O
    from gpe_amd import ops, net_blocks
    torch.manual_seed(Bn + T)
    rnn = (torch.nn.LSTM if kind == 'lstm' else torch.nn.GRU)(In, Hh, L, batch_first=True)
    with torch.no_grad():                                        # weights away from unit scale: the amax words must carry it
        rnn.weight_hh_l0.mul_(2.5)                               # (mild factors: a recurrence with large weights is chaotic —
        if L > 1:                                                # ANY rounding difference then grows to O(1) within a few steps)
            rnn.weight_ih_l1.mul_(1.0 / 64)
    ref = copy.deepcopy(rnn).double()
    rnn = rnn.cuda()
    G = 4 if kind == 'lstm' else 3
    g = torch.Generator().manual_seed(3)
    x = torch.randn(Bn, In, generator=g)
    h0 = torch.randn(L, Bn, Hh, generator=g) * 0.3
    c0 = torch.randn(L, Bn, Hh, generator=g) * 0.3
    wgt = torch.randn(Bn, T, Hh, generator=g)
    xr = x.double().requires_grad_()
    seq = xr[:, None, :].expand(Bn, T, In)
    out_r, _ = ref(seq, (h0.double(), c0.double())) if kind == 'lstm' else ref(seq, h0.double())
    (out_r * wgt.double()).sum().backward()
    params = net_blocks._rnn_params(rnn, L)
    plan = ops.PackPlan()
    net_blocks._register_rnn_packs(plan, rnn, L, Hh, G)
    prev = gpe.set_math('f16x3')
    try:
        outs = {}
        for with_plan in (True, False):
            for p in rnn.parameters():
                p.grad = None
            if with_plan:
                plan.refresh()
                assert ops.planned_planes(rnn.weight_hh_l0, ops.K_GATES_H3)[0] is not None
            else:
                ops.bump_weights_epoch()                          # the plan's packs are stale now: the exact kernels run
                assert ops.planned_planes(rnn.weight_hh_l0, ops.K_GATES_H3)[0] is None
            xd = x.cuda().requires_grad_()
            top, _, _ = ops.rnn_stack(xd, h0.cuda(), c0.cuda() if kind == 'lstm' else None, T, L, kind, params)
            (top * wgt.cuda()).sum().backward()
            outs[with_plan] = top.detach().clone()
            assert relerr(top, out_r) < 2e-5, with_plan
            assert relerr(xd.grad, xr.grad) < 1e-4, with_plan
            for (n, p), q in zip(rnn.named_parameters(), ref.parameters()):
                assert relerr(p.grad, q.grad) < 1e-4, (with_plan, n)
        assert relerr(outs[True], outs[False]) < 2e-5
        assert not torch.equal(outs[True], outs[False])          # ... and it really was another arithmetic
    finally:
        gpe.set_math(prev)


@pytest.mark.parametrize('Bn,In,Hh,T,L,dbg', [(32, 250, 250, 23, 2, 0), (5, 20, 20, 3, 1, 0), (33, 40, 44, 6, 3, 0), (16, 64, 256, 4, 4, 0),
                                              (100, 250, 250, 5, 2, 0), (736, 250, 250, 14, 3, 2048 | 4096), (2000, 30, 36, 4, 2, 2048 | 4096)])
def test_persistent_lstm_stack_matches_the_diagonal_launches(gpe, Bn, In, Hh, T, L, dbg):
    """csrc/gpe_rnn_persist.hip (round 6): an LSTM stack as ONE persistent launch per direction — weight slices resident in LDS, cells
    ordered by arrival counters, state rows handed on with sc1 stores / sc1 loads.  (a) Same numbers as the diagonal launches of
    gpe_rnn_wave.hip (gpe_debug_set(1024) keeps them) to fp32 rounding, in both arithmetic modes, forward and every gradient; (b) the
    hand-off is RACE-FREE: repeated runs — alone and next to a stream of unrelated kernels that loads the chip unevenly — are
    bit-identical (a stale or early read of a state row would change bits).  dbg 2048 | 4096: several row tiles per workgroup (stacks
    with more 16-row tiles than the chip has room for, e.g. the 736-row panel decoder)."""
    from gpe_amd import ops, net_blocks
    from gpe_amd import _lib as Lb
    torch.manual_seed(Bn + T)
    rnn = torch.nn.LSTM(In, Hh, L, batch_first=True).cuda()
    G = 4
    g = torch.Generator().manual_seed(3)
    x = torch.randn(Bn, In, generator=g).cuda()
    h0 = (torch.randn(L, Bn, Hh, generator=g) * 0.3).cuda()
    c0 = (torch.randn(L, Bn, Hh, generator=g) * 0.3).cuda()
    wgt = torch.randn(Bn, T, Hh, generator=g).cuda()
    params = net_blocks._rnn_params(rnn, L)
    plan = ops.PackPlan()
    net_blocks._register_rnn_packs(plan, rnn, L, Hh, G)
    # (dbg != 0: stacks with more row tiles than the chip has room for — the K-split kernels take them only under the switch; in
    # f16x3 mode their FORWARD runs in gpe_rnn_persist_mt.hip whatever the switch says: test_multi_tile_persistent_lstm_forward)
    noise = torch.randn(1 << 22, device='cuda')
    side = torch.cuda.Stream()

    def run():
        for p in rnn.parameters():
            p.grad = None
        xd = x.clone().requires_grad_()
        top, hN, cN = ops.rnn_stack(xd, h0, c0, T, L, 'lstm', params, want_state=True)
        ((top * wgt).sum() + hN.sum() * 0.5 + cN.sum() * 0.25).backward()
        return [top.detach().clone(), hN.clone(), cN.clone(), xd.grad.clone()] + [p.grad.clone() for p in rnn.parameters()]

    for mode in ('f32', 'f16x3'):
        prev = gpe.set_math(mode)
        try:
            plan.refresh()
            Lb.query('gpe_debug_set', 1024)
            ref = run()
            Lb.query('gpe_debug_set', dbg)
            assert Lb.query('gpe_rnn_seq_fwd_ws', G, L, T, Bn, Hh) > 0                # the persistent path IS taken
            first = run()
            for a, b in zip(first, ref):
                assert relerr(a, b) < 2e-5, mode
            assert not torch.equal(first[0], ref[0])                                   # ... and it was another kernel
            for rep in range(6):
                if rep % 2:
                    with torch.cuda.stream(side):                                      # uneven load from a second stream
                        for _ in range(8):
                            noise.mul_(1.0001)
                again = run()
                for a, b in zip(again, first):
                    assert torch.equal(a, b), (mode, rep)
            torch.cuda.synchronize()
        finally:
            Lb.query('gpe_debug_set', 0)
            gpe.set_math(prev)


@pytest.mark.parametrize('Bn,In,Hh,T,L,seq,reserve', [(736, 250, 250, 14, 3, False, 0), (730, 250, 250, 6, 3, False, 16), (900, 40, 96, 5, 2, True, 0),
                                                      (1500, 32, 64, 4, 1, False, 0), (2100, 64, 32, 7, 4, True, 0)])
def test_multi_tile_persistent_lstm_forward(gpe, Bn, In, Hh, T, L, seq, reserve):
    """csrc/gpe_rnn_persist_mt.hip (round 6): the forward of an LSTM stack whose 16-row tiles outnumber the chip (the 736-row panel
    decoder) as ONE persistent launch in f16x3 mode — waves own row tiles, state rows published already split into their fp16 terms,
    one 128-byte line per arrival counter.  (a) Same numbers as the diagonal launches (gpe_debug_set(131072) keeps them) to fp32
    rounding: outputs, final states and — through the saved gates / cell states the backward reads — every gradient; (b) race-free:
    repeated runs, alone and beside a second stream's load, are bit-identical; (c) ragged row counts, sequence inputs (a different
    addend per step), one to four layers, and with CUs held back for a collective."""
    from gpe_amd import ops, net_blocks
    from gpe_amd import _lib as Lb
    torch.manual_seed(Bn + T)
    rnn = torch.nn.LSTM(In, Hh, L, batch_first=True).cuda()
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(Bn, T, In, generator=g) if seq else torch.randn(Bn, In, generator=g)).cuda()
    h0 = (torch.randn(L, Bn, Hh, generator=g) * 0.3).cuda()
    c0 = (torch.randn(L, Bn, Hh, generator=g) * 0.3).cuda()
    wgt = torch.randn(Bn, T, Hh, generator=g).cuda()
    params = net_blocks._rnn_params(rnn, L)
    plan = ops.PackPlan()
    net_blocks._register_rnn_packs(plan, rnn, L, Hh, 4)
    noise = torch.randn(1 << 22, device='cuda')
    side = torch.cuda.Stream()

    def run():
        for p in rnn.parameters():
            p.grad = None
        xd = x.clone().requires_grad_()
        top, hN, cN = ops.rnn_stack(xd, h0, c0, T, L, 'lstm', params, want_state=True, h0_bounded=True)
        ((top * wgt).sum() + hN.sum() * 0.5 + cN.sum() * 0.25).backward()
        return [top.detach().clone(), hN.clone(), cN.clone(), xd.grad.clone()] + [p.grad.clone() for p in rnn.parameters()]

    prev = gpe.set_math('f16x3')
    try:
        plan.refresh()
        Lb.query('gpe_reserve_cus_set', reserve)
        Lb.query('gpe_debug_set', 131072)
        ref = run()
        Lb.query('gpe_debug_set', 0)
        assert Lb.query('gpe_rnn_seq_fwd_ws', 4, L, T, Bn, Hh) > (1 << 20)                # the state planes: this kernel IS eligible
        first = run()
        for a, b in zip(first, ref):
            assert relerr(a, b) < 2e-5
        assert not torch.equal(first[0], ref[0])                                       # ... and it was another kernel
        for rep in range(6):
            if rep % 2:
                with torch.cuda.stream(side):
                    for _ in range(8):
                        noise.mul_(1.0001)
            again = run()
            for a, b in zip(again, first):
                assert torch.equal(a, b), rep
        torch.cuda.synchronize()
    finally:
        Lb.query('gpe_debug_set', 0)
        Lb.query('gpe_reserve_cus_set', 0)
        gpe.set_math(prev)


def test_fused_cell_backward_matches_the_two_launch_diagonals(gpe):
    """gpe_debug_set(65536): the LSTM cell backward inside the split-K launch (last-arriver ticket per output block).  Measured slower
    than the two launches and off by default (csrc/gpe_rnn_wave.hip); kept correct: same gradients as the default path to rounding."""
    from gpe_amd import ops, net_blocks
    from gpe_amd import _lib as Lb
    Bn, In, Hh, T, L = 736, 250, 250, 14, 3
    torch.manual_seed(5)
    rnn = torch.nn.LSTM(In, Hh, L, batch_first=True).cuda()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(Bn, In, generator=g).cuda()
    h0 = (torch.randn(L, Bn, Hh, generator=g) * 0.3).cuda()
    c0 = (torch.randn(L, Bn, Hh, generator=g) * 0.3).cuda()
    wgt = torch.randn(Bn, T, Hh, generator=g).cuda()
    params = net_blocks._rnn_params(rnn, L)
    outs = []
    try:
        for dbg in (0, 65536):
            Lb.query('gpe_debug_set', dbg)
            for p in rnn.parameters():
                p.grad = None
            xd = x.clone().requires_grad_()
            top, hN, cN = ops.rnn_stack(xd, h0, c0, T, L, 'lstm', params, want_state=True)
            ((top * wgt).sum() + hN.sum() * 0.5 + cN.sum() * 0.25).backward()
            outs.append([xd.grad.clone()] + [p.grad.clone() for p in rnn.parameters()])
    finally:
        Lb.query('gpe_debug_set', 0)
    for a, b in zip(*outs):
        assert relerr(a, b) < 2e-6


def test_rnn_large_start_state_takes_the_exact_kernels(gpe):
    """ops.rnn_stack in f16x3 mode with a caller-supplied start state of magnitude >= 16: the fp16-pipe kernels scale the
    state rows by 2^12 and would overflow — rnn_stack reads the largest |h0| and runs the exact fp32 kernels for that call; results
    meet the fp64 oracle at the usual bars (finite, in particular)."""
    from gpe_amd import ops, net_blocks
    Bn, In, Hh, T, Lr = 32, 40, 64, 5, 2
    torch.manual_seed(4)
    rnn = torch.nn.GRU(In, Hh, Lr, batch_first=True)
    ref = copy.deepcopy(rnn).double()
    rnn = rnn.cuda()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(Bn, In, generator=g)
    h0 = torch.randn(Lr, Bn, Hh, generator=g) * 0.3
    h0[0, 3, 7] = 40.0                                          # a GRU carries a large start state through the sequence
    xr = x.double().requires_grad_()
    out_r, _ = ref(xr[:, None, :].expand(Bn, T, In), h0.double())
    out_r.sum().backward()
    plan = ops.PackPlan()
    net_blocks._register_rnn_packs(plan, rnn, Lr, Hh, 3)
    prev = gpe.set_math('f16x3')
    try:
        plan.refresh()
        assert ops.planned_planes(rnn.weight_hh_l0, ops.K_GATES_H3)[0] is not None      # the fp16-pipe path WOULD be taken
        xd = x.cuda().requires_grad_()
        top, _, _ = ops.rnn_stack(xd, h0.cuda(), None, T, Lr, 'gru', net_blocks._rnn_params(rnn, Lr))
        top.sum().backward()
        assert torch.isfinite(top).all()
        assert relerr(top, out_r) < 2e-5
        assert relerr(xd.grad, xr.grad) < 1e-4
    finally:
        gpe.set_math(prev)


# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,chans', [(300, [27, 27, 27, 23]), (1000, [153, 153, 153, 23]), (130, [16, 200, 200, 200, 1])])
def test_dense_mlp_fwd_bwd(gpe, math_mode, M, chans):
    from oracle import ref_path as O
    tol = TOL[math_mode]
    torch.manual_seed(M)
    omlp = O.MLP(chans)
    with torch.no_grad():
        for blk in omlp:
            blk[2].weight.uniform_(0.5, 1.5)
            blk[2].bias.uniform_(-0.3, 0.3)
        omlp[-1][2].weight[::3] *= -1
    pmlp = gpe.net_blocks.MLP(chans)
    pmlp.load_state_dict(omlp.state_dict())
    pmlp = pmlp.cuda().train()
    x = torch.randn(M, chans[0], generator=torch.Generator().manual_seed(1))
    wgt = torch.randn(M, chans[-1], generator=torch.Generator().manual_seed(2))
    o64 = copy.deepcopy(omlp).double().train()
    xr = x.double().requires_grad_()
    yr = o64(xr)
    (yr * wgt.double()).sum().backward()
    xd = x.cuda().requires_grad_()
    y = gpe.ops.dense_mlp(xd, pmlp, True)
    (y * wgt.cuda()).sum().backward()
    o32 = copy.deepcopy(omlp).train()
    e32 = relerr(o32(x), yr)
    assert relerr(y, yr) < max(tol['fwd'], 20 * e32)
    e = tol.get('dx_norm', tol['norm'])(xd.grad, xr.grad)
    print('dense mlp %s dx err %.2e' % (math_mode, e))
    assert e < tol['mlp_dx']
    pn = dict(pmlp.named_parameters())
    for n, p in o64.named_parameters():
        e = tol['norm'](pn[n].grad, p.grad)
        print('dense mlp %s %s err %.2e' % (math_mode, n, e))
        assert e < (tol['mlp_db'] if p.grad.dim() == 1 else tol['mlp_dw']), (n, e)
    pb = dict(pmlp.named_buffers())
    for n, b in o64.named_buffers():
        if 'num_batches' in n:
            assert pb[n].item() == b.item()
        else:
            assert relerr(pb[n], b) < 1e-5, n


def test_sparsemax_fwd_bwd(gpe):
    from oracle import ref_path as O
    g = torch.Generator().manual_seed(3)
    z = torch.randn(5000, 23, generator=g) * 2
    gy = torch.randn(5000, 23, generator=g)
    zr = z.double().requires_grad_()
    pr = O.Sparsemax(dim=1)(zr)
    pr.backward(gy.double())
    zd = z.cuda().requires_grad_()
    pd = gpe.ops.SparsemaxFn.apply(zd)
    pd.backward(gy.cuda())
    assert relerr(pd, pr) < 2e-6
    assert torch.equal(pd.cpu() > 0, pr > 0)                     # identical support
    assert relerr(zd.grad, zr.grad) < 2e-6


def test_attention_pool_fwd_bwd(gpe):
    B, N, P, C = 3, 200, 23, 27
    g = torch.Generator().manual_seed(4)
    w = torch.rand(B * N, P, generator=g)
    f = torch.randn(B * N, C, generator=g)
    gy = torch.randn(B * P, C, generator=g)
    wr, fr = w.double().requires_grad_(), f.double().requires_grad_()
    ref = torch.einsum('bnp,bnc->bpc', wr.view(B, N, P), fr.view(B, N, C)) / N
    ref.reshape(B * P, C).backward(gy.double())
    wd, fd = w.cuda().requires_grad_(), f.cuda().requires_grad_()
    out = gpe.ops.AttentionPoolFn.apply(wd, fd, B, N)
    out.backward(gy.cuda())
    assert relerr(out, ref.reshape(B * P, C)) < 3e-6
    assert relerr(wd.grad, wr.grad) < 3e-6
    assert relerr(fd.grad, fr.grad) < 3e-6


# --------------------------------------------------------------------------------------------------
# round 2: wide dense MLPs, eval-mode backward, pooling variants, loss + matching kernels, optimizer, services
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,chans', [(500, [403, 403, 403, 23]), (46, [40, 560, 560, 112]), (64, [300, 520, 7])])
def test_dense_mlp_wide_layers(gpe, M, chans):
    """Layers wider than 256 (global-attention MLP = 403, MLP decoders = hidden*out_len): K slabs + several column blocks
    of the generic row GEMM, centred reduce-GEMM in 256-column passes."""
    from oracle import ref_path as O
    torch.manual_seed(M)
    omlp = O.MLP(chans)
    with torch.no_grad():
        for blk in omlp:
            blk[2].weight.uniform_(0.5, 1.5)
            blk[2].bias.uniform_(-0.3, 0.3)
    pmlp = gpe.net_blocks.MLP(chans)
    pmlp.load_state_dict(omlp.state_dict())
    pmlp = pmlp.cuda().train()
    x = torch.randn(M, chans[0], generator=torch.Generator().manual_seed(1))
    wgt = torch.randn(M, chans[-1], generator=torch.Generator().manual_seed(2))
    o64 = copy.deepcopy(omlp).double().train()
    xr = x.double().requires_grad_()
    yr = o64(xr)
    (yr * wgt.double()).sum().backward()
    xd = x.cuda().requires_grad_()
    y = gpe.ops.dense_mlp(xd, pmlp, True)
    (y * wgt.cuda()).sum().backward()
    o32 = copy.deepcopy(omlp).train()
    x32 = x.clone().requires_grad_()
    y32 = o32(x32)
    (y32 * wgt).sum().backward()
    assert relerr(y, yr) < max(5e-5, 20 * relerr(y32, yr))
    assert relerr(xd.grad, xr.grad) < max(3e-4, 20 * relerr(x32.grad, xr.grad))
    pn, p32 = dict(pmlp.named_parameters()), dict(o32.named_parameters())
    for n, p in o64.named_parameters():
        e, e32 = relerr(pn[n].grad, p.grad), relerr(p32[n].grad, p.grad)
        assert e < max(5e-3 if p.grad.dim() == 1 else 5e-4, 20 * e32), (n, e, e32)


def test_edgeconv_eval_mode_backward(gpe):
    """Backward through a layer in eval() mode: BatchNorm statistics are constants (no mean / projection terms)."""
    from oracle import ref_path as O
    B, N, C, H, Fo, k = 2, 80, 5, 32, 24, 6
    oconv = _oracle_conv(C, H, Fo, k, seed=21)
    with torch.no_grad():
        for blk in oconv.nn:
            blk[2].running_mean.uniform_(0.1, 0.4)
            blk[2].running_var.uniform_(0.5, 1.5)
    pconv = _product_conv(gpe, oconv, C, H, Fo, k).eval()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B * N, C, generator=g)
    wgt = torch.randn(B * N, Fo, generator=g)
    xd = x.cuda().requires_grad_()
    out = pconv(xd, B, N)
    (out * wgt.cuda()).sum().backward()
    o64 = copy.deepcopy(oconv).double().eval()
    o64.knn_override = pconv.last_knn.cpu().view(B * N, k).long()
    xr = x.double().requires_grad_()
    ref = o64(xr, torch.arange(B).repeat_interleave(N))
    (ref * wgt.double()).sum().backward()
    assert relerr(out, ref) < 1e-5
    assert relerr(xd.grad, xr.grad) < 2e-4
    pn = dict(pconv.named_parameters())
    for n, p in o64.named_parameters():
        assert relerr(pn[n].grad, p.grad) < 3e-4, n
    with pytest.raises(RuntimeError, match='ran twice'):
        out2 = pconv(xd, B, N)
        s = (out2 * wgt.cuda()).sum()
        s.backward(retain_graph=True)
        s.backward()


@pytest.mark.parametrize('aggr,nblocks', [('mean', 3), ('add', 3), ('max', 2), ('max', 4), ('add', 2)])
def test_edgeconv_aggr_and_depth(gpe, aggr, nblocks):
    """EConv_aggr mean / add and EConv_hidden_depth != 2 (nn/net_blocks.py:121-135)."""
    from oracle import ref_path as O
    B, N, C, H, Fo, k = 2, 70, 6, 40, 28, 5
    torch.manual_seed(nblocks)
    oconv = O.DynamicEdgeConv(O.MLP([2 * C] + [H] * (nblocks - 1) + [Fo]), k=k, aggr=aggr)
    with torch.no_grad():
        for blk in oconv.nn:
            blk[2].weight.uniform_(0.5, 1.5)
            blk[2].bias.uniform_(-0.3, 0.3)
        oconv.nn[-1][2].weight[::4] *= -1
    pconv = gpe.net_blocks.DynamicEdgeConv(gpe.net_blocks.MLP([2 * C] + [H] * (nblocks - 1) + [Fo]), k=k, aggr=aggr)
    pconv.load_state_dict(oconv.state_dict())
    pconv = pconv.cuda().train()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B * N, C, generator=g)
    wgt = torch.randn(B * N, Fo, generator=g)
    batch = torch.arange(B).repeat_interleave(N)
    xd = x.cuda().requires_grad_()
    out = pconv(xd, B, N)
    (out * wgt.cuda()).sum().backward()
    o64 = copy.deepcopy(oconv).double().train()
    o64.knn_override = pconv.last_knn.cpu().view(B * N, k).long()
    xr = x.double().requires_grad_()
    ref = o64(xr, batch)
    (ref * wgt.double()).sum().backward()
    assert relerr(out, ref) < 5e-5
    assert relerr(xd.grad, xr.grad) < 2e-4
    pn = dict(pconv.named_parameters())
    for n, p in o64.named_parameters():
        assert relerr(pn[n].grad, p.grad) < 5e-4, (n, relerr(pn[n].grad, p.grad))


@pytest.mark.parametrize('H,Fo,aggr,C', [(30, 22, 'max', 3), (300, 150, 'max', 3), (30, 22, 'mean', 6), (258, 64, 'add', 150),
                                         (7, 5, 'max', 150)])
def test_edgeconv_general_widths(gpe, H, Fo, aggr, C):
    """EConv_hidden outside the fused path's menu (not a multiple of 4, or > 256): the explicit-message formulation
    (ops.edge_conv_general), same contract as every other width (nn/net_blocks.py:108-135)."""
    from oracle import ref_path as O
    B, N, k = 2, 60, 5
    torch.manual_seed(H)
    oconv = O.DynamicEdgeConv(O.MLP([2 * C, H, H, Fo]), k=k, aggr=aggr)
    with torch.no_grad():
        for blk in oconv.nn:
            blk[2].weight.uniform_(0.5, 1.5)
            blk[2].bias.uniform_(-0.3, 0.3)
        oconv.nn[-1][2].weight[::4] *= -1
    pconv = gpe.net_blocks.DynamicEdgeConv(gpe.net_blocks.MLP([2 * C, H, H, Fo]), k=k, aggr=aggr)
    pconv.load_state_dict(oconv.state_dict())
    pconv = pconv.cuda().train()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B * N, C, generator=g)
    wgt = torch.randn(B * N, Fo, generator=g)
    batch = torch.arange(B).repeat_interleave(N)
    xd = x.cuda().requires_grad_()
    out = pconv(xd, B, N)
    (out * wgt.cuda()).sum().backward()
    ref_idx = O.knn_local(x, B, k)
    assert torch.equal(pconv.last_knn.cpu().view(B * N, k).long(), ref_idx)
    o64 = copy.deepcopy(oconv).double().train()
    o64.knn_override = ref_idx
    xr = x.double().requires_grad_()
    ref = o64(xr, batch)
    (ref * wgt.double()).sum().backward()
    assert relerr(out, ref) < 5e-5
    assert relerr(xd.grad, xr.grad) < 2e-4
    pn = dict(pconv.named_parameters())
    for n, p in o64.named_parameters():
        assert relerr(pn[n].grad, p.grad) < 5e-4, (n, relerr(pn[n].grad, p.grad))
    pb = dict(pconv.named_buffers())
    for n, bbuf in o64.named_buffers():
        if 'num_batches' in n:
            assert pb[n].item() == bbuf.item()
        else:
            assert relerr(pb[n], bbuf) < 1e-5, n
    # eval mode (running statistics) through the same path
    pconv.eval(); o64.eval()
    with torch.no_grad():
        assert relerr(pconv(x.cuda(), B, N), o64(x.double(), batch)) < 5e-5


def _f16x3_dense_block(gpe, a_in, Wm, bias, B, N, k, amax_a=None, amax_out=None):
    """One fused dense edge block (gpe_edge_mlp_fwd a_mode 1, no statistics / aggregation) -> a fresh output tensor.  amax_a /
    amax_out: the caller-owned f16x3 scale words of include/gpe_hip.h (None: the operand is measured in-call / nothing kept)."""
    L = gpe._lib
    E, Cin = a_in.shape
    Cout = Wm.shape[0]
    out = torch.empty(E, (Cout + 3) // 4 * 4, device='cuda')
    ws, nws = gpe.ops.edge_workspace(B, N, k, 4, 'cuda')
    L.call('gpe_edge_mlp_fwd', 1, None, 0, None, a_in, a_in.stride(0), B, N, k, Cin, Cout, gpe.ops.pack_weight(Wm), bias, out,
           out.stride(0), None, 0, None, None, None, None, 0, amax_a, amax_out, ws, nws, 0)
    return out[:, :Cout]


@pytest.fixture
def f16x3_ungated(gpe):
    """f16x3 arithmetic with the size gate lifted (the kernel tests use a few thousand rows)."""
    prev = gpe.set_math('f16x3')
    gate = gpe.set_f16x3_min_rows(0)
    yield
    gpe.set_f16x3_min_rows(gate)
    gpe.set_math(prev)


@pytest.mark.parametrize('scale', [1e-25, 3e-9, 1.0, 7e6, 1e20])
def test_f16x3_extreme_magnitudes(gpe, scale):
    """The f16x3 kernels normalise every operand by a power of two taken from its largest magnitude: activations (and
    weights) far outside fp16's range give the exact mode's result."""
    B, N, k, Cin, Cout = 2, 64, 16, 200, 150
    g = torch.Generator().manual_seed(3)
    a = (torch.randn(B * N * k, Cin, generator=g).abs() * scale).cuda()
    a[5, 7] *= 300.0                                            # one outlier sets the scale; the rest keep their precision
    Wm = (torch.randn(Cout, Cin, generator=g) / 14 / max(scale, 1e-12) ** 0.5).cuda()
    ref = a.double().cpu() @ Wm.double().cpu().t()
    outs = {}
    gate = gpe.set_f16x3_min_rows(0)
    for mode in ('f32', 'f16x3'):
        prev = gpe.set_math(mode)
        try:
            outs[mode] = _f16x3_dense_block(gpe, a, Wm, None, B, N, k).double().cpu()
        finally:
            gpe.set_math(prev)
    gpe.set_f16x3_min_rows(gate)
    ref = torch.relu(ref)
    e32, e16 = relerr(outs['f32'], ref), relerr(outs['f16x3'], ref)
    print('scale %g: f32 %.2e, f16x3 %.2e' % (scale, e32, e16))
    assert torch.isfinite(outs['f16x3']).all()
    assert e16 < max(2e-6, 4 * e32)


def _word_value(w):
    return w.view(torch.float32).item()


def test_f16x3_amax_words_are_caller_owned(gpe, f16x3_ungated):
    """The f16x3 scales live in caller-owned words (include/gpe_hip.h "amax word"): the kernel that stores an activation fills
    `amax_out` with its largest magnitude, the next block takes it as `amax_a`.  The library keeps no record of tensors, so the
    CALLER re-measures (gpe_absmax) after rewriting a tensor — here the activation is multiplied by 1000 in place — or passes
    no word and lets the consumer measure the operand itself."""
    L = gpe._lib
    B, N, k, C = 2, 64, 16, 200
    g = torch.Generator().manual_seed(4)
    a0 = torch.randn(B * N * k, C, generator=g).abs().cuda()
    W1 = (torch.randn(C, C, generator=g) / 14).cuda()
    W2 = (torch.randn(150, C, generator=g) / 14).cuda()
    words = torch.zeros(2, dtype=torch.int32, device='cuda')
    a1 = _f16x3_dense_block(gpe, a0, W1, None, B, N, k, None, words[0:1])    # fills word 0 with max|a1|
    assert _word_value(words[0:1]) == a1.abs().max().item()
    ref = torch.relu(a1.double().cpu() @ W2.double().cpu().t())
    out = _f16x3_dense_block(gpe, a1, W2, None, B, N, k, words[0:1], None).double().cpu()
    assert relerr(out, ref) < 2e-6
    # a bound instead of the value is as good (tighter = more precise): 4x the maximum costs two mantissa bits of the low plane
    bound = (a1.abs().max() * 4).view(1).view(torch.int32)
    assert relerr(_f16x3_dense_block(gpe, a1, W2, None, B, N, k, bound, None).double().cpu(), ref) < 2e-6
    # the caller rewrites the tensor -> the caller re-measures
    a1 = a1.contiguous()
    L.call('gpe_scale', a1, 1000.0, a1, a1.numel())
    L.call('gpe_absmax', a1, a1.stride(0), a1.shape[0], a1.shape[1], words[1:2])
    assert _word_value(words[1:2]) == a1.abs().max().item()
    ref = torch.relu(a1.double().cpu() @ W2.double().cpu().t())
    for w in (words[1:2], None):
        out = _f16x3_dense_block(gpe, a1, W2, None, B, N, k, w, None).double().cpu()
        assert torch.isfinite(out).all()
        assert relerr(out, ref) < 2e-6
    # every mode honours amax_out (one extra streaming pass where the kernel that ran does not track it)
    prev = gpe.set_math('f32')
    try:
        words.zero_()
        a2 = _f16x3_dense_block(gpe, a0, W1, None, B, N, k, None, words[0:1])
        assert _word_value(words[0:1]) == a2.abs().max().item()
    finally:
        gpe.set_math(prev)


@pytest.mark.parametrize('v_mode', ['dense', 'gather'])
def test_f16x3_edge_redgemm_with_words(gpe, f16x3_ungated, v_mode):
    """The weight-gradient reduce-GEMM on the fp16 pipe: it runs only when the caller passes the amax words of both operands
    (U: a dz tensor; V: the stored activation, or the bound of the gathered relu(P_i + Q_j) — from gpe_edge_pq_amax, or computed
    in the call's workspace when the word is NULL); result within the exact kernel's bar of the fp64 product."""
    ops, L = gpe.ops, gpe._lib
    B, N, k, Mg, Ng = 8, 512, 16, 150, 200
    g = torch.Generator().manual_seed(17)
    E = B * N * k
    u = (torch.randn(E, 152, generator=g) * 1e-4).cuda()
    u[:, Mg:] = 0
    shift = torch.randn(Ng, generator=g).abs().cuda()
    G, cs = torch.empty(Mg, Ng).cuda(), torch.empty(Mg).cuda()
    part = torch.empty(L.query('gpe_redgemm_ws', Mg, Ng)).cuda()
    ws, nws = ops.edge_workspace(B, N, k, 2 * Ng, 'cuda')
    words = torch.zeros(3, dtype=torch.int32, device='cuda')
    L.call('gpe_absmax', u, 152, E, Mg, words[0:1])
    if v_mode == 'dense':
        v = torch.randn(E, Ng, generator=g).abs().cuda()
        L.call('gpe_absmax', v, Ng, E, Ng, words[1:2])
        vref = v.double() - shift.double()
        variants = [words[1:2]]
        run = lambda wv: L.call('gpe_edge_redgemm', u, 152, 1, v, Ng, None, 0, None, shift, B, N, k, Mg, Ng, G, Ng, cs, part,
                                words[0:1], wv, ws, nws, None, 0, None, None, 0, None)
    else:
        pq = torch.randn(B * N, 2 * Ng, generator=g).cuda()
        jg = (torch.randint(0, N, (B, N, k), generator=g) + torch.arange(B).view(B, 1, 1) * N).int().cuda()
        L.call('gpe_edge_pq_amax', pq, 2 * Ng, Ng, B * N, words[2:3], ws, nws)
        i = torch.arange(B * N, device='cuda').repeat_interleave(k)
        a0 = torch.relu(pq[i, :Ng].double() + pq[jg.view(-1).long(), Ng:].double())
        assert _word_value(words[2:3]) >= a0.max().item()
        vref = a0 - shift.double()
        variants = [words[2:3], None]
        run = lambda wv: L.call('gpe_edge_redgemm', u, 152, 0, None, 0, pq, 2 * Ng, jg, shift, B, N, k, Mg, Ng, G, Ng, cs, part,
                                words[0:1], wv, ws, nws, None, 0, None, None, 0, None)
    uref = u[:, :Mg].double()
    for wv in variants:
        G.zero_(); cs.zero_()
        run(wv)
        assert relerr(G, uref.t() @ vref) < 3e-6
        assert relerr(cs, uref.sum(0)) < 3e-6


@pytest.mark.parametrize('M,W', [(1, 23), (700, 23), (513, 5), (64, 32)])
def test_sparsemax_loss(gpe, M, W):
    """entmax.SparsemaxLoss() as composed_loss.py:323-332 calls it, on general scores (not only simplex points) and behind a
    non-unit upstream gradient."""
    from oracle import ref_path as O
    g = torch.Generator().manual_seed(M + W)
    x = torch.randn(M, W, generator=g) * 1.5
    x[::3] = torch.softmax(x[::3], -1)                          # rows that already lie on the simplex (the reference's use)
    t = torch.randint(0, W, (M,), generator=g)
    xr = x.double().requires_grad_()
    ref = O.SparsemaxLoss()(xr, t)
    (0.05 * ref).backward()
    xd = x.cuda().requires_grad_()
    out = gpe.ops.SparsemaxLossFn.apply(xd, t.cuda())
    (0.05 * out).backward()
    assert abs(out.item() - ref.item()) < 2e-6 * max(1.0, abs(ref.item()))
    assert relerr(xd.grad, xr.grad) < 2e-6
    with pytest.raises(IndexError):
        gpe.ops.SparsemaxLossFn.apply(xd, torch.full((M,), W).cuda())


@pytest.mark.parametrize('mode', ['max', 'add'])
def test_segment_pool_max_add(gpe, mode):
    x = torch.randn(3 * 130, 37, generator=torch.Generator().manual_seed(4))
    gy = torch.randn(3, 37, generator=torch.Generator().manual_seed(5))
    xr = x.double().requires_grad_()
    ref = xr.view(3, 130, 37).amax(1) if mode == 'max' else xr.view(3, 130, 37).sum(1)
    ref.backward(gy.double())
    xd = x.cuda().requires_grad_()
    y = (gpe.ops.segment_max if mode == 'max' else gpe.ops.segment_add)(xd, 3, 130)
    y.backward(gy.cuda())
    assert relerr(y, ref) < 1e-6
    assert relerr(xd.grad, xr.grad) < 1e-6


@pytest.mark.parametrize('mode', [0, 1, 2])
@pytest.mark.parametrize('B,N,P,C', [(3, 200, 23, 27), (2, 300, 23, 153)])
def test_attention_pool_modes(gpe, mode, B, N, P, C):
    g = torch.Generator().manual_seed(4 + mode)
    w = torch.rand(B * N, P, generator=g)
    f = torch.randn(B * N, C, generator=g)
    gy = torch.randn(B * P, C, generator=g)
    wr, fr = w.double().requires_grad_(), f.double().requires_grad_()
    prod = wr.view(B, N, P, 1) * fr.view(B, N, 1, C)
    ref = prod.mean(1) if mode == 0 else prod.amax(1) if mode == 1 else prod.sum(1)
    ref.reshape(B * P, C).backward(gy.double())
    wd, fd = w.cuda().requires_grad_(), f.cuda().requires_grad_()
    out = gpe.ops.AttentionPoolFn.apply(wd, fd, B, N, mode)
    out.backward(gy.cuda())
    assert relerr(out, ref.reshape(B * P, C)) < 3e-6
    assert relerr(wd.grad, wr.grad) < 3e-6
    assert relerr(fd.grad, fr.grad) < 3e-6


def _loss_inputs(B, P, Lp, seed):
    g = torch.Generator().manual_seed(seed)
    panels = torch.randn(B, P, Lp, 8, generator=g)
    place = torch.randn(B * P, 7, generator=g)
    gt = {'outlines': torch.randn(B, P, Lp, 4, generator=g), 'rotations': torch.randn(B, P, 4, generator=g),
          'translations': torch.randn(B, P, 3, generator=g), 'num_edges': torch.randint(0, Lp + 1, (B, P), generator=g),
          'empty_panels_mask': torch.zeros(B, P, dtype=torch.bool)}
    return panels, place, gt


def _views(panels, place, B, P):
    # the same kind of strided views the models hand to the loss (slices of one [B,P,L,8] and one [B*P,7] tensor)
    return {'outlines': panels[..., :4], 'rotations': place.view(B, P, 7)[..., :4],
            'translations': place.view(B, P, 7)[..., 4:]}


@pytest.mark.parametrize('origin,order', [(False, False), (True, False), (True, 'shape_translation'), (False, 'placement'),
                                          (True, 'translation')])
def test_pattern_loss_and_matching(gpe, origin, order):
    """ComposedPatternLoss on the device (HIP loss + matching kernels) vs the oracle's restatement of the reference loops
    (nn/metrics/composed_loss.py:294-334,428-703; nn/metrics/losses.py:19-51): value, loss dict, gradients, decisions."""
    from oracle import ref_path as O
    dc = gpe.configs.data_config()
    B, P, Lp = 5, dc['max_pattern_len'], dc['max_panel_len']
    cfg = dict(loss_components=['shape', 'loop', 'rotation', 'translation'], quality_components=[],
               panel_origin_invariant_loss=origin, panel_order_inariant_loss=bool(order),
               order_by=order if order else 'placement', epoch_with_order_matching=0, loop_loss_weight=0.7)
    ours = gpe.metrics.ComposedPatternLoss(dc, dict(cfg))
    theirs = O.ComposedPatternLoss(dc, dict(cfg))
    panels, place, gt = _loss_inputs(B, P, Lp, 3)
    pr, qr = panels.double().requires_grad_(), place.double().requires_grad_()
    lr_, dr, _ = theirs(_views(pr, qr, B, P), {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in gt.items()},
                        epoch=0)
    lr_.backward()
    pd, qd = panels.cuda().requires_grad_(), place.cuda().requires_grad_()
    lo, do, upd = ours(_views(pd, qd, B, P), {k: v.clone() for k, v in gt.items()}, epoch=0)
    lo.backward()
    assert set(do.keys()) == set(dr.keys())
    assert abs(lo.item() - lr_.item()) < 2e-6 * max(1, abs(lr_.item()))
    for k_ in dr:
        assert abs(do[k_].item() - dr[k_].item()) < 2e-6 * max(1, abs(dr[k_].item())), k_
    assert relerr(pd.grad, pr.grad) < 2e-6
    assert relerr(qd.grad, qr.grad) < 2e-6
    if origin:
        assert torch.equal(ours.last_leading_edges.cpu(), theirs.last_leading_edges)
    if order:
        ours.check_order_match()
        assert torch.equal(ours.last_permutation.cpu(), theirs.last_permutation)
    assert upd == bool(order)


def _stitch_gt(B, P, Lp, S, num_edges, seed, duplicate=False):
    """what the dataset hands over for the stitch terms (nn/data/datasets.py:805-819): pattern-level edge ids of the two sides
    of every stitch (zero-padded), their count, the free-edge mask, supervised tags."""
    g = torch.Generator().manual_seed(seed)
    st = torch.zeros(B, 2, S, dtype=torch.long)
    nst = torch.zeros(B, dtype=torch.long)
    free = torch.ones(B, P, Lp, dtype=torch.bool)
    for b in range(B):
        edges = [(p, e) for p in range(P) for e in range(int(num_edges[b, p]))]
        order = torch.randperm(len(edges), generator=g).tolist()
        n = min(int(torch.randint(2, S + 1, (1,), generator=g)), len(edges) // 2)
        nst[b] = n
        for i in range(n):
            for side in (0, 1):
                p_, e_ = edges[order[2 * i + side]]
                st[b, side, i] = p_ * Lp + e_
                free[b, p_, e_] = False
        if duplicate and n >= 2:
            st[b, 1, 1] = st[b, 0, 0]                           # one edge on two stitches: its gradient accumulates
    return {'stitches': st, 'num_stitches': nst, 'free_edges_mask': free,
            'stitch_tags': torch.randn(B, P, Lp, 3, generator=g)}


@pytest.mark.parametrize('hardnet,origin,order,supervised', [(False, False, False, False), (True, False, False, True),
                                                             (False, True, 'shape_translation', False),
                                                             (True, True, 'stitches', True), (False, False, 'placement', True)])
def test_stitch_losses_and_renumbering(gpe, hardnet, origin, order, supervised):
    """The stitch terms of ComposedPatternLoss at epoch >= epoch_with_stitches (HIP kernels) vs the oracle's restatement of
    nn/metrics/composed_loss.py:336-362,505-517,592-620,604-617,705-755 and nn/metrics/losses.py:54-180: value, every entry
    of the loss dict, gradients of all three prediction views, the re-numbered stitches and the shifted free-edge mask."""
    from oracle import ref_path as O
    dc = gpe.configs.data_config()
    B, P, Lp, S = 4, dc['max_pattern_len'], dc['max_panel_len'], dc['max_num_stitches']
    comps = ['shape', 'loop', 'rotation', 'translation', 'stitch', 'free_class'] + (['stitch_supervised'] if supervised else [])
    cfg = dict(loss_components=comps, quality_components=[], panel_origin_invariant_loss=origin,
               panel_order_inariant_loss=bool(order), order_by=order if order else 'placement', epoch_with_order_matching=0,
               epoch_with_stitches=40, stitch_hardnet_version=hardnet, stitch_tags_margin=0.3, stitch_supervised_weight=0.1)
    ours = gpe.metrics.ComposedPatternLoss(dc, dict(cfg))
    theirs = O.ComposedPatternLoss(dc, dict(cfg))
    panels, place, gt = _loss_inputs(B, P, Lp, 7)
    panels[..., 4:7] *= 0.4                                      # tags closer than the margin: the negative term is active
    gt.update(_stitch_gt(B, P, Lp, S, gt['num_edges'], 8, duplicate=True))

    def views(pn, pl):
        v = _views(pn, pl, B, P)
        v['stitch_tags'] = pn[..., 4:7]
        v['free_edges_mask'] = pn[..., 7]
        return v

    pr, qr = panels.double().requires_grad_(), place.double().requires_grad_()
    lr_, dr, ur = theirs(views(pr, qr), {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in gt.items()},
                         epoch=40)
    lr_.backward()
    pd, qd = panels.cuda().requires_grad_(), place.cuda().requires_grad_()
    lo, do, upd = ours(views(pd, qd), {k: v.clone() for k, v in gt.items()}, epoch=40)
    lo.backward()
    assert upd == ur and upd                                     # epoch == epoch_with_stitches: the loss structure changed
    assert set(do.keys()) == set(dr.keys()) and 'stitch_neg_loss' in do and 'free_edges_loss' in do
    assert float(dr['stitch_neg_loss'].detach()) > 0             # the test exercises the active branch
    assert abs(lo.item() - lr_.item()) < 2e-6 * max(1, abs(lr_.item()))
    for k_ in dr:
        assert abs(float(do[k_].detach()) - float(dr[k_].detach())) < 2e-6 * max(1, abs(float(dr[k_].detach()))), k_
    assert relerr(pd.grad, pr.grad) < 2e-6
    assert relerr(pd.grad[..., 4:], pr.grad[..., 4:]) < 2e-6     # the tag / free-edge columns on their own scale
    assert relerr(qd.grad, qr.grad) < 2e-6
    if order:
        ours.check_order_match()
        assert torch.equal(ours.last_permutation.cpu(), theirs.last_permutation)
    if origin:
        assert torch.equal(ours.last_leading_edges.cpu(), theirs.last_leading_edges)
    # the ground truth the stitch terms saw: re-derive it with the oracle's helpers
    with torch.no_grad():
        gt2 = {k: v.clone() for k, v in gt.items()}
        theirs.epoch = 40
        if order:
            gt2 = theirs._gt_order_match(views(pr, qr), gt2)
        if origin:
            gt2 = theirs._rotate_gt(views(pr, qr), gt2, gt2['num_edges'].int().view(-1))
    got = ours.last_matched_stitch_gt
    assert torch.equal(got['stitches'].cpu(), gt2['stitches'])
    assert torch.equal(got['free_edges_mask'].cpu().bool(), gt2['free_edges_mask'].bool())
    # before the switch-over epoch nothing of this runs and the dict has the four main keys only
    lo0, do0, upd0 = ours(views(pd.detach(), qd.detach()), {k: v.clone() for k, v in gt.items()}, epoch=39)
    assert set(do0.keys()) == {'pattern_loss', 'loop_loss', 'rotation_loss', 'translation_loss'} and upd0 == bool(order and 0 == 39)


def test_fp16_activation_store_is_clamped_not_inf(gpe):
    """gpe_edge_mlp_fwd(out_half = 1): an activation beyond the fp16 range is stored as 65504, never as inf (the lazily formed dz3
    would turn inf into NaN gradients); the per-point maxima, and the amax word, keep the fp32 value."""
    ops, L = gpe.ops, gpe._lib
    B, N, k, Cin, Cout = 8, 512, 16, 200, 150
    E, BN = B * N * k, B * N
    g = torch.Generator().manual_seed(23)
    a2 = torch.randn(E, Cin, generator=g).abs().cuda()
    W = torch.randn(Cout, Cin, generator=g).cuda() * 0.05
    W[7] = 600.0                                                # column 7: sums of 200 half-normal values x 600 ~ 1e5 > 65504
    bias = torch.zeros(Cout).cuda()
    prev = gpe.set_math('f16x3')
    try:
        assert L.query('gpe_edge_lazy_dz3_ok', B, N, k, Cout, Cin) == 1
        a3h = torch.empty(E, 152, device='cuda', dtype=torch.float16)
        mx = torch.empty(BN, 152, device='cuda'); mn = torch.empty_like(mx)
        amx = torch.empty(BN, 152, device='cuda', dtype=torch.uint8); amn = torch.empty_like(amx)
        words = torch.zeros(2, dtype=torch.int32, device='cuda')
        ws, nws = ops.edge_workspace(B, N, k, 2 * Cin, 'cuda')
        L.call('gpe_edge_mlp_fwd', 1, None, 0, None, a2, Cin, B, N, k, Cin, Cout, ops.pack_weight(W), bias, a3h, 152, None,
               1, mx, mn, amx, amn, 152, None, words[0:1], ws, nws, 1)
    finally:
        gpe.set_math(prev)
    ref = torch.relu(a2.double() @ W.double().t())
    assert ref[:, 7].max().item() > 65504
    got = a3h[:, :Cout].float()
    assert torch.isfinite(got).all()
    assert got[:, 7].max().item() == 65504.0
    small = ref < 6e4
    assert ((got.double() - ref).abs()[small] <= ref[small] * 2.0 ** -10 + 1e-6).all()   # everything in range: fp16 rounding of the fp32 value
    assert relerr(mx[:, 7], ref.view(BN, k, Cout)[:, :, 7].max(1).values) < 1e-6         # the aggregate is not clamped
    assert _word_value(words[0:1]) > 65504


def test_half_activation_guard(gpe):
    """ops.set_half_act_guard (VERDICT r4 #7-ii): the f16x3 mode keeps the aggregated block's activation in fp16 (clamped
    at 65504).  A layer whose activation outgrows fp16 — here: the last Linear scaled by 1e5 — must not train on clamped values:
    'strict' repeats the launch with fp32 rows inside the same forward (gradients = the eager fp32-storage pass, and the fp64
    oracle's at the usual bar); 'fallback' notices one step late without a host synchronisation, warns, and stores fp32 from then
    on; a layer in range keeps the fp16 storage."""
    import warnings
    ops, L = gpe.ops, gpe._lib
    B, N, C, k = 8, 512, 3, 16
    oconv = _oracle_conv(C, 200, 150, k, seed=5)
    with torch.no_grad():
        oconv.nn[2][0].weight *= 1e5
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B * N, C, generator=g)
    wgt = torch.randn(B * N, 150, generator=g)

    def run(conv):
        for p_ in conv.parameters():
            p_.grad = None
        xd = x.cuda().requires_grad_()
        y = conv(xd, B, N)
        (y * wgt.cuda()).sum().backward()
        return y.detach().clone(), xd.grad.clone(), {n: p_.grad.clone() for n, p_ in conv.named_parameters()}

    prev = gpe.set_math('f16x3')
    mode0 = ops.set_half_act_guard('strict')
    try:
        assert L.query('gpe_edge_lazy_dz3_ok', B, N, k, 150, 200) == 1
        # reference run of the build itself: fp32 storage + the eager dz3 pass (gpe_debug_set(512) closes the lazy path)
        L.query('gpe_debug_set', 512)
        eager = run(_product_conv(gpe, oconv, C, 200, 150, k).train())
        L.query('gpe_debug_set', 0)
        # strict: exact in the step that overflows
        conv = _product_conv(gpe, oconv, C, 200, 150, k).train()
        with warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter('always')
            strict = run(conv)
        assert conv.half_act_guard.disabled and conv.half_act_guard.last_amax > 65504
        assert any('fp16 storage' in str(w.message) for w in wl)
        assert torch.equal(strict[0], eager[0])
        assert relerr(strict[1], eager[1]) < 1e-5
        for n in eager[2]:
            assert relerr(strict[2][n], eager[2][n]) < 3e-5, n
        # ... and against the fp64 oracle on the build's graph (bars of test_lazy_dz3_matches_the_in_place_pass)
        o64 = copy.deepcopy(oconv).double().train()
        o64.knn_override = conv.last_knn.cpu().view(-1, k).long()
        xr = x.double().requires_grad_()
        yr = o64(xr, torch.arange(B).repeat_interleave(N))
        (yr * wgt.double()).sum().backward()
        assert relerr(strict[0], yr) < 5e-5
        assert relerr_fro(strict[1], xr.grad) < 2e-3
        for n, p_ in o64.named_parameters():
            assert relerr_fro(strict[2][n], p_.grad) < 2e-3, n
        # fallback: the first step runs on the clamped copy (not compared), the guard trips before the second
        ops.set_half_act_guard('fallback')
        conv = _product_conv(gpe, oconv, C, 200, 150, k).train()
        run(conv)
        assert not conv.half_act_guard.disabled
        torch.cuda.synchronize()
        with warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter('always')
            second = run(conv)
        assert conv.half_act_guard.disabled and any('fp16 storage' in str(w.message) for w in wl)
        assert relerr(second[1], eager[1]) < 1e-5
        # a layer in range keeps the fp16 storage
        conv = _product_conv(gpe, _oracle_conv(C, 200, 150, k, seed=5), C, 200, 150, k).train()
        run(conv)
        torch.cuda.synchronize()
        run(conv)
        assert not conv.half_act_guard.disabled and 0 < conv.half_act_guard.last_amax < 65504
    finally:
        L.query('gpe_debug_set', 0)
        ops.set_half_act_guard(mode0)
        gpe.set_math(prev)


@pytest.mark.parametrize('B,N', [(8, 512), (9, 457)])
def test_lazy_dz3_matches_the_in_place_pass(gpe, B, N):
    """f16x3, k = 16, above the size gate: the aggregated block's activation is stored in fp16 and its backward never materialises
    dz3 — the weight-gradient
    reduce-GEMM and the propagation kernel form it from the stored activation while staging it (include/gpe_hip.h "lazy dz3").
    Same gradients as with the separate in-place pass (gpe_debug_set(512) keeps it), to rounding: both run the fp16 pipe, only
    the scale word differs (a bound instead of the measured maximum) — and both meet the fp64 oracle at the layer test's bars."""
    L = gpe._lib
    # (8, 512): E = 65 536 rows, the smallest launch the lazy path takes; (9, 457): 4113 points — a ragged last tile in both
    # consumers (one point of the 64-row propagation tile, 16 of the reduce-GEMM's 32 rows) and clouds that are not pinned to XCDs
    C, k = 3, 16
    oconv = _oracle_conv(C, 200, 150, k, seed=5)                 # includes negative BatchNorm scales: the min side
    conv = _product_conv(gpe, oconv, C, 200, 150, k).train()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B * N, C, generator=g)
    wgt = torch.randn(B * N, 150, generator=g)
    prev = gpe.set_math('f16x3')
    try:
        assert L.query('gpe_edge_lazy_dz3_ok', B, N, k, 150, 200) == 1
        res = {}
        for flag in (0, 512):
            L.query('gpe_debug_set', flag)
            assert L.query('gpe_edge_lazy_dz3_ok', B, N, k, 150, 200) == (0 if flag else 1)
            for p_ in conv.parameters():
                p_.grad = None
            xd = x.cuda().requires_grad_()
            y = conv(xd, B, N)
            (y * wgt.cuda()).sum().backward()
            res[flag] = (y.detach().clone(), xd.grad.clone(), {n: p_.grad.clone() for n, p_ in conv.named_parameters()})
    finally:
        L.query('gpe_debug_set', 0)
        gpe.set_math(prev)
    assert torch.equal(res[0][0], res[512][0])                   # same forward
    # (the lazy run also keeps a3 in fp16 — include/gpe_hip.h "out_half" — which moves the gradients by 2e-6 / 5e-6 of their
    # maximum at the encoder's sizes, profiles/r04_h_row_g_probe.txt)
    assert relerr(res[0][1], res[512][1]) < 1e-5
    for n in res[0][2]:
        assert relerr(res[0][2][n], res[512][2][n]) < 3e-5, n
    # against the fp64 oracle on the build's graph
    o64 = copy.deepcopy(oconv).double().train()
    o64.knn_override = conv.last_knn.cpu().view(-1, k).long()
    xr = x.double().requires_grad_()
    yr = o64(xr, torch.arange(B).repeat_interleave(N))
    (yr * wgt.double()).sum().backward()
    # (Frobenius norms: with 4096 points x 150 channels a handful of argmax / ReLU near-ties resolve the other way in fp32 than
    # in the fp64 oracle — single elements at 1e-3 of max|grad|, see the note above test_edgeconv_layer_fwd_bwd; the lazy and the
    # eager pass agree element by element above)
    # (the oracle here does not stand on the build's decisions — tests/relu_align.py does that for the model tests — so the bar
    # only guards against gross errors: measured 8.3e-4 on dx at (9, 457), 1e-4 at (8, 512), identical for the eager pass)
    assert relerr(res[0][0], yr) < 5e-5
    assert relerr_fro(res[0][1], xr.grad) < 2e-3
    for n, p_ in o64.named_parameters():
        assert relerr_fro(res[0][2][n], p_.grad) < 2e-3, n


def test_two_streams_one_device(gpe):
    """include/gpe_hip.h: every buffer — outputs, workspaces, f16x3 scale words — is the caller's, so two streams of one device
    can run the path concurrently.  Two EdgeConv layers (kNN, gather, fused edge GEMMs, backward) on two streams give bit-identical
    results to the same layers run one after the other, in the exact and in the f16x3 arithmetic."""
    from gpe_amd import net_blocks
    B, N, C = 8, 512, 3
    g = torch.Generator().manual_seed(11)
    xs = [torch.randn(B * N, C, generator=g).cuda() for _ in range(2)]
    torch.manual_seed(3)
    conv0 = net_blocks.DynamicEdgeConv(net_blocks.MLP([2 * C, 200, 200, 150]), k=16, aggr='max').cuda().train()
    convs = [conv0, copy.deepcopy(conv0)]                       # same weights; own BatchNorm buffers and .grad per stream

    def run(i):
        xr = xs[i].clone().requires_grad_()
        y = convs[i](xr, B, N)
        y.square().sum().backward()
        return y.detach().clone(), xr.grad.clone()

    for mode in ('f32', 'f16x3'):
        prev = gpe.set_math(mode)
        gate = gpe.set_f16x3_min_rows(0)
        try:
            seq = [run(i) for i in range(2)]
            torch.cuda.synchronize()
            streams = [torch.cuda.Stream(), torch.cuda.Stream()]
            par = [None, None]
            for rep in range(3):                                   # interleave launches of the two streams
                for i, st in enumerate(streams):
                    with torch.cuda.stream(st):
                        par[i] = run(i)
            torch.cuda.synchronize()
            for (y0, g0), (y1, g1) in zip(seq, par):
                assert torch.equal(y0, y1) and torch.equal(g0, g1), mode
        finally:
            gpe.set_f16x3_min_rows(gate)
            gpe.set_math(prev)


def test_panel_loop_loss_standalone(gpe):
    """metrics.PanelLoopLoss used on its own (the reference's class, nn/metrics/losses.py:8-51) == the oracle's per-panel loop:
    value and gradient, batched [B, P, L, 4] view of the decoder output and flat [n_panels, L, 4] input."""
    from oracle import ref_path as O
    dc = gpe.configs.data_config()
    stats = {'shift': dc['standardize']['gt_shift']['outlines'], 'scale': dc['standardize']['gt_scale']['outlines']}
    ours = gpe.metrics.PanelLoopLoss(dc['max_panel_len'], data_stats=stats)
    theirs = O.PanelLoopLoss(O.eval_pad_vector(stats))
    g = torch.Generator().manual_seed(3)
    B, P, Lp = 3, 23, 14
    panels = torch.randn(B, P, Lp, 8, generator=g)
    n = torch.randint(0, Lp + 1, (B, P), generator=g)
    pr = panels.double().requires_grad_()
    lr_ = theirs(pr[..., :4], n.view(-1))
    lr_.backward()
    pd = panels.cuda().requires_grad_()
    lo = ours(pd[..., :4], n.cuda())
    lo.backward()
    assert abs(lo.item() - lr_.item()) < 2e-6 * max(1, abs(lr_.item()))
    assert relerr(pd.grad, pr.grad) < 2e-6
    flat = panels[..., :4].reshape(B * P, Lp, 4).contiguous().cuda()
    assert abs(ours(flat, n.view(-1).cuda()).item() - lr_.item()) < 2e-6 * max(1, abs(lr_.item()))
    with pytest.raises(RuntimeError, match='no CPU path'):
        ours(panels[..., :4], n)


def test_fused_adam_onecycle_vs_torch(gpe):
    """gpe_adam_step over a flat arena + the OneCycle schedule vs torch.optim.Adam + OneCycleLR in fp64
    (nn/trainer.py:162-185)."""
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(37, 50), torch.nn.Linear(50, 3)).cuda()
    ref = copy.deepcopy(net).double()
    total = 40
    sched = gpe.optim.OneCycle(2e-3, total)
    opt = gpe.optim.FusedAdam(net, lr=2e-3, weight_decay=1e-4, schedule=sched)
    ropt = torch.optim.Adam(ref.parameters(), lr=2e-3, weight_decay=1e-4)
    rs = torch.optim.lr_scheduler.OneCycleLR(ropt, max_lr=2e-3, epochs=4, steps_per_epoch=10, cycle_momentum=False)
    g = torch.Generator().manual_seed(1)
    for step in range(12):
        x = torch.randn(16, 37, generator=g)
        assert abs(opt.schedule.lr(step) - ropt.param_groups[0]['lr']) < 1e-12
        net(x.cuda()).square().mean().backward()          # plain torch modules: autograd accumulates into the arena views
        ref(x.cuda().double()).square().mean().backward()
        opt.step()
        ropt.step()
        rs.step()
        ropt.zero_grad()
        assert not opt.arena.grad.any()                    # cleared by the same launch
    for p, q in zip(net.parameters(), ref.parameters()):
        assert relerr(p, q) < 1e-5


def test_fused_adam_checkpoint_exchanges_with_torch_adam(gpe):
    """nn/trainer.py:281-285 saves optimizer.state_dict() and _restore_run loads it: FusedAdam speaks torch.optim.Adam's
    format in both directions — a run stepped by torch Adam continues under FusedAdam and vice versa, to fp32 round-off."""
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(21, 30), torch.nn.Tanh(), torch.nn.Linear(30, 5)).cuda()
    twin = copy.deepcopy(net)
    g = torch.Generator().manual_seed(2)
    xs = [torch.randn(8, 21, generator=g).cuda() for _ in range(6)]
    topt = torch.optim.Adam(twin.parameters(), lr=3e-3, weight_decay=1e-4)
    for x in xs[:3]:                                           # three steps under torch Adam ...
        topt.zero_grad()
        twin(x).square().mean().backward()
        topt.step()
    net.load_state_dict(twin.state_dict())
    opt = gpe.optim.FusedAdam(net, lr=1.0, weight_decay=0.5)   # wrong hyper-parameters on purpose: the checkpoint's win
    opt.load_state_dict(topt.state_dict())
    assert opt.t == 3 and opt.lr == 3e-3 and opt.weight_decay == 1e-4
    for x in xs[3:]:                                           # ... three more under FusedAdam and under torch
        net(x).square().mean().backward()
        opt.step()
        topt.zero_grad()
        twin(x).square().mean().backward()
        topt.step()
    for p, q in zip(net.parameters(), twin.parameters()):
        assert relerr(p, q) < 1e-5
    # and back: FusedAdam's state loads into a fresh torch Adam (same keys, shapes, step count)
    sd = opt.state_dict()
    ref_sd = topt.state_dict()
    assert set(sd['state'].keys()) == set(ref_sd['state'].keys())
    assert sd['param_groups'][0]['params'] == ref_sd['param_groups'][0]['params']
    for k, st in sd['state'].items():
        assert float(st['step']) == float(ref_sd['state'][k]['step']) == 6.0
        assert relerr(st['exp_avg'], ref_sd['state'][k]['exp_avg']) < 1e-5
        assert relerr(st['exp_avg_sq'], ref_sd['state'][k]['exp_avg_sq']) < 1e-5
    fresh = torch.optim.Adam(twin.parameters(), lr=1.0)
    fresh.load_state_dict(sd)
    assert fresh.param_groups[0]['lr'] == 3e-3
    # a checkpoint of another model is refused instead of loading misaligned moments
    other = gpe.optim.FusedAdam(torch.nn.Linear(21, 31).cuda())
    with pytest.raises(ValueError):
        other.load_state_dict(sd)
    # the schedule: torch's OneCycleLR state carries the step count
    sched = gpe.optim.OneCycle(2e-3, 40)
    rs = torch.optim.lr_scheduler.OneCycleLR(torch.optim.Adam(twin.parameters(), lr=2e-3), max_lr=2e-3, epochs=4,
                                             steps_per_epoch=10, cycle_momentum=False)
    assert gpe.optim.OneCycle(1.0, 7).load_state_dict(sched.state_dict()) is None
    assert sched.load_state_dict(rs.state_dict()) == rs.last_epoch


def test_fused_adam_onecycle_checkpoint_restores_under_torch(gpe):
    """The reference's restore flow (nn/trainer.py _restore_run): build Adam + OneCycleLR, then optimizer.load_state_dict()
    REPLACES the param groups by the checkpoint's — so a FusedAdam checkpoint written under a OneCycle schedule must carry the
    `initial_lr` / `max_lr` / `min_lr` keys torch's scheduler reads from the group on its next step() — and back: a torch
    checkpoint's rates are taken by FusedAdam + OneCycle."""
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(11, 13), torch.nn.Tanh(), torch.nn.Linear(13, 2)).cuda()
    twin = copy.deepcopy(net)
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(8, 11, generator=g).cuda() for _ in range(8)]
    opt = gpe.optim.FusedAdam(net, lr=1e-3, schedule=gpe.optim.OneCycle(4e-3, 40))
    for x in xs[:4]:
        net(x).square().mean().backward()
        opt.step()
    sd = opt.state_dict()
    assert {'initial_lr', 'max_lr', 'min_lr'} <= set(sd['param_groups'][0])
    twin.load_state_dict(net.state_dict())
    topt = torch.optim.Adam(twin.parameters(), lr=4e-3)
    tsch = torch.optim.lr_scheduler.OneCycleLR(topt, max_lr=4e-3, epochs=4, steps_per_epoch=10, cycle_momentum=False)
    topt.load_state_dict(sd)
    tsch.last_epoch = opt.t                                     # what scheduler.load_state_dict restores
    for x in xs[4:]:                                            # both continue: same rates, same parameters
        assert abs(topt.param_groups[0]['lr'] - opt.schedule.lr(opt.t)) < 1e-12
        net(x).square().mean().backward()
        opt.step()
        topt.zero_grad()
        twin(x).square().mean().backward()
        topt.step()
        tsch.step()                                             # KeyError 'max_lr' here before the group carried the rates
        assert abs(topt.param_groups[0]['lr'] - opt.schedule.lr(opt.t)) < 1e-12
    # torch -> FusedAdam: the rates of the checkpoint win over the constructor's
    other = gpe.optim.FusedAdam(copy.deepcopy(twin), lr=1.0, schedule=gpe.optim.OneCycle(123.0, 40))
    other.load_state_dict(topt.state_dict())
    assert other.schedule.max_lr == 4e-3 and abs(other.schedule.initial - 4e-3 / 25) < 1e-15
    assert abs(other.schedule.min_lr - 4e-3 / 25 / 1e4) < 1e-18


def test_fused_adam_intermittent_gradients_follow_torch(gpe):
    """torch.optim.Adam counts steps per parameter: a parameter that first receives a gradient late (or only now and then)
    is bias-corrected by ITS step count, and has no state entry before.  FusedAdam keeps the same per-parameter counts."""
    torch.manual_seed(2)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(7, 9)
            self.late = torch.nn.Linear(9, 9)
            self.b = torch.nn.Linear(9, 2)

        def forward(self, x, use_late):
            h = torch.relu(self.a(x))
            if use_late:
                h = h + self.late(h)
            return self.b(h)

    net = Net().cuda()
    twin = copy.deepcopy(net)
    opt = gpe.optim.FusedAdam(net, lr=1e-2)
    topt = torch.optim.Adam(twin.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(6)
    for step, use_late in enumerate([False, False, True, False, True, True]):
        x = torch.randn(5, 7, generator=g).cuda()
        net(x, use_late).square().mean().backward()
        opt.step()
        topt.zero_grad(set_to_none=True)
        twin(x, use_late).square().mean().backward()
        topt.step()
        sd, rsd = opt.state_dict(), topt.state_dict()
        assert set(sd['state']) == set(rsd['state'])            # no entry for a parameter that never took a step
        for k, st in sd['state'].items():
            assert float(st['step']) == float(rsd['state'][k]['step'])
    for p, q in zip(net.parameters(), twin.parameters()):
        assert relerr(p, q) < 1e-5


def test_fused_adam_skips_parameters_without_gradient(gpe):
    """torch.optim.Adam leaves a parameter whose grad is None alone (no weight decay, no moment decay); FusedAdam does the
    same for arena segments that received no gradient in the step."""
    torch.manual_seed(1)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(9, 12)
            self.unused = torch.nn.Linear(12, 4)
            self.b = torch.nn.Linear(12, 3)

        def forward(self, x):
            return self.b(torch.relu(self.a(x)))

    net = Net().cuda()
    twin = copy.deepcopy(net)
    opt = gpe.optim.FusedAdam(net, lr=1e-2, weight_decay=0.1)
    topt = torch.optim.Adam(twin.parameters(), lr=1e-2, weight_decay=0.1)
    w0 = net.unused.weight.detach().clone()
    g = torch.Generator().manual_seed(3)
    for _ in range(4):
        x = torch.randn(6, 9, generator=g).cuda()
        net(x).square().mean().backward()
        opt.step()
        topt.zero_grad()
        twin(x).square().mean().backward()
        topt.step()
    assert torch.equal(net.unused.weight, w0)                  # untouched, although weight_decay > 0
    for p, q in zip(net.parameters(), twin.parameters()):
        assert relerr(p, q) < 1e-5
    assert not opt.arena.grad.any()


def test_order_match_degenerate_input_stays_in_bounds(gpe):
    """NaN predictions: the reference raises ValueError inside the loss; the device matching reports through its flag and
    keeps the permutation a valid index (the gathers that consume it must not trip a device-side assert)."""
    from gpe_amd import ops
    B, P, D = 3, 23, 7
    g = torch.Generator().manual_seed(4)
    pf = torch.randn(B, P, D, generator=g).cuda()
    gf = torch.randn(B, P, D, generator=g).cuda()
    pf[1] = float('nan')
    perm, fail = ops.order_match(pf, gf)
    assert int(fail.item()) == 1
    assert perm.min().item() >= 0 and perm.max().item() < P
    for b in (0, 2):                                           # healthy patterns: a true permutation
        assert sorted(perm[b].tolist()) == list(range(P))
    torch.gather(gf, 1, perm[:, :, None].expand(-1, -1, D))    # in bounds
    torch.cuda.synchronize()


def test_pack_plan_and_grad_sink_are_transparent(gpe, golden_dir):
    """One pack launch per weight change + in-place gradient sink: same outputs and gradients as the plain path; stale
    packs are never used after an in-place weight change."""
    import os
    fx = torch.load(os.path.join(golden_dir, 'full3d_small.pt'), weights_only=False)

    def build():
        torch.manual_seed(fx['seed'])
        return gpe.nets.GarmentFullPattern3D(fx['data_config'], copy.deepcopy(fx['nn_config']),
                                             copy.deepcopy(fx['loss_config'])).cuda().train()

    def run(model):
        torch.manual_seed(fx['seed'] + 2)
        preds = model(fx['features'].cuda(), log_step=0, epoch=0)
        loss, _, _ = model.loss(preds, {k: v.clone() for k, v in fx['gt'].items()}, epoch=0)
        loss.backward()
        return preds, loss

    plain = build()
    p0, l0 = run(plain)
    g0 = {n: p.grad.clone() for n, p in plain.named_parameters()}
    sunk = build()
    arena = gpe.optim.FlatArena(sunk)
    assert arena.is_sink
    p1, l1 = run(sunk)
    assert torch.equal(l0, l1)
    for k_ in p0:
        assert torch.equal(p0[k_], p1[k_]), k_
    for n, p in sunk.named_parameters():
        assert p.grad.untyped_storage().data_ptr() == arena.grad.untyped_storage().data_ptr()
        assert torch.equal(p.grad, g0[n]), n
    assert len(arena.written) == len(arena.params)
    with pytest.raises(RuntimeError, match='second gradient'):
        run(sunk)                                           # no optimizer step in between: refuse to overwrite silently
    arena.zero_grad()
    arena.unregister_sink()
    # stale-pack guard: change one weight in place, the next forward must see it
    with torch.no_grad():
        plain.placement_decoder.weight.mul_(2.0)
    plain.zero_grad(set_to_none=True)
    p2, _ = run(plain)
    assert not torch.equal(p2['rotations'], p0['rotations'])
    with torch.no_grad():
        plain.placement_decoder.weight.mul_(0.5)
    plain.zero_grad(set_to_none=True)
    p3, _ = run(plain)
    assert torch.equal(p3['rotations'], p0['rotations'])
    # an update through `p.data` does not move torch's version counter (EMA weights, hand-written copies): the packs follow anyway
    v0 = plain.placement_decoder.weight._version
    plain.placement_decoder.weight.data.mul_(2.0)
    assert plain.placement_decoder.weight._version == v0
    plain.zero_grad(set_to_none=True)
    p4, _ = run(plain)
    assert torch.equal(p4['rotations'], p2['rotations'])


def test_standardize(gpe):
    x = torch.randn(1000, 3, generator=torch.Generator().manual_seed(1))
    st = gpe.configs.data_config()['standardize']
    y = gpe.ops.standardize(x.cuda(), st['f_shift'], st['f_scale'])
    ref = (x - torch.tensor(st['f_shift'])) / torch.tensor(st['f_scale'])
    assert torch.allclose(y.cpu(), ref, rtol=1e-6, atol=1e-7)


def test_stitch_model_known_answer(gpe, golden_dir):
    """The reference's only shipped trained weights (models/att/neural_tailor_stitch_model.pth) through the eval path of
    the dense-MLP kernels: a true known-answer test against the reference's own class; then a training-mode step."""
    import os
    fx = torch.load(os.path.join(golden_dir, 'stitch_pairs_known_answer.pt'), weights_only=False)
    model = gpe.nets.StitchOnEdge3DPairs(fx['data_config'], dict(fx['nn_config']), {})
    model.load_state_dict(fx['state_dict'])
    model = model.cuda().eval()
    with torch.no_grad():
        out = model(fx['pairs'].cuda())
    scale = fx['out_eval'].abs().max().item()
    assert (out.cpu() - fx['out_eval']).abs().max().item() < 1e-4 * max(1.0, scale)
    model.train()
    model.loss.with_quality_eval = False
    out_t = model(fx['pairs'].cuda())
    loss, _, _ = model.loss(out_t, fx['labels'].cuda())
    loss.backward()
    assert (out_t.detach().cpu() - fx['out_train']).abs().max().item() < 1e-4 * max(1.0, fx['out_train'].abs().max().item())
    assert abs(loss.item() - fx['loss'].item()) < 1e-5
    for n, p in model.named_parameters():
        ref = fx['grads'][n]
        assert (p.grad.cpu() - ref).abs().max().item() < 5e-3 * (ref.abs().max().item() + 1e-12), n
    for k_, v in fx['state_after_train'].items():
        if v.is_floating_point():
            assert torch.allclose(model.state_dict()[k_].cpu(), v, rtol=1e-4, atol=1e-6), k_


def test_rccl_world1_gradient_exchange(gpe, golden_dir, tmp_path):
    """The nccl (= RCCL) branch of parallel.init_distributed + one DistributedHotPath step of the real HIP model on this
    box: process-group init, in-place bucket all-reduce on the compute stream's data, stream ordering vs backward."""
    import os, socket, subprocess, sys, textwrap
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'w.py'
    script.write_text(textwrap.dedent("""
        import copy, os, sys
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        import gpe_amd
        from gpe_amd import parallel
        rank, local, world = parallel.init_distributed(backend='nccl')
        assert dist.is_initialized() and dist.get_backend() == 'nccl' and world == 1
        fx = torch.load(os.path.join(%r, 'full3d_small.pt'), weights_only=False)
        def build():
            torch.manual_seed(fx['seed'])
            return gpe_amd.nets.GarmentFullPattern3D(fx['data_config'], copy.deepcopy(fx['nn_config']),
                                                     copy.deepcopy(fx['loss_config'])).cuda().train()
        def run(m, call):
            torch.manual_seed(fx['seed'] + 2)
            preds = call(fx['features'].cuda(), log_step=0, epoch=0)
            loss, _, _ = m.loss(preds, {k: v.clone() for k, v in fx['gt'].items()}, epoch=0)
            loss.backward()
            return loss
        plain = build(); l0 = run(plain, plain)
        model = build()
        ddp = parallel.DistributedHotPath(model, device_ids=[torch.device('cuda', local)], bucket_bytes=64 << 10)
        ddp.world = 2                      # exercise the exchange code path: hooks were not armed for world 1, arm now
        ddp.arena.listeners.append(ddp._on_written)
        l1 = run(model, ddp)
        assert len(ddp._launched) > 0      # buckets left while backward was still running
        ddp.finish_gradient_sync()
        torch.cuda.synchronize()
        assert torch.equal(l0, l1)
        assert ddp._avg_op                 # RCCL averages inside the collective (ReduceOp.AVG): no pre-division launch
        for (n, p), (_, q) in zip(model.named_parameters(), plain.named_parameters()):
            # the collective averages over the REAL group (1 rank): the gradients come back unchanged.  (Until round 4 the wrapper
            # divided by `ddp.world` itself and summed: this check then saw grad / 2.)
            assert torch.allclose(p.grad, q.grad, rtol=1e-6, atol=1e-12), n
        # what bench.py prints for N > 1: the exchange on its own over RCCL (async all-reduce of every bucket on a scratch
        # arena, barrier, synchronisation) and the exposed part measured around the waits of finish_gradient_sync
        ex = ddp.measure_exchange(iters=3)
        assert ex['backend'] == 'nccl' and ex['buckets'] == len(ddp._buckets) and ex['ms_per_step'] > 0, ex
        assert ex['bytes_per_step'] == sum((hi - lo) * 4 for _, _, lo, hi in ddp._buckets)
        assert ddp.exposed_ms() is not None and ddp.exposed_ms() >= 0
        dist.destroy_process_group()
        print('rccl ok', len(ddp._buckets), 'buckets')
    """) % (repo, str(golden_dir)))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               GPE_FORCE_DIST='1')
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'rccl ok' in r.stdout


@pytest.mark.parametrize('cls', ['GRUDecoderModule', 'LSTMDoubleReverseDecoderModule'])
@pytest.mark.parametrize('Bn,In,Hh,T,L,Out', [(6, 40, 40, 5, 2, 40), (46, 40, 40, 14, 3, 8), (64, 250, 250, 14, 2, 8)])
def test_alternative_recurrent_decoders(gpe, cls, Bn, In, Hh, T, L, Out):
    """GRUDecoderModule (nn/net_blocks.py:457-497) and LSTMDoubleReverseDecoderModule (:405-454: sequence input, start state
    with gradient, final state out) on the general recurrent stack, vs torch.nn.GRU / LSTM in fp64."""
    from oracle import ref_path as O
    torch.manual_seed(Bn + T)
    odec = getattr(O, cls)(In, Hh, Out, L, custom_init='kaiming_normal_')
    pdec = getattr(gpe.net_blocks, cls)(In, Hh, Out, L, custom_init='kaiming_normal_')
    pdec.load_state_dict(odec.state_dict())
    pdec = pdec.cuda()
    enc = torch.randn(Bn, In, generator=torch.Generator().manual_seed(1))
    wgt = torch.randn(Bn, T, Out, generator=torch.Generator().manual_seed(2))
    o64 = copy.deepcopy(odec).double()
    er = enc.double().requires_grad_()
    torch.manual_seed(77)
    out_r = o64(er, T)
    (out_r * wgt.double()).sum().backward()
    ed = enc.cuda().requires_grad_()
    torch.manual_seed(77)
    out = pdec(ed, T)
    (out * wgt.cuda()).sum().backward()
    assert torch.equal(pdec.last_states[0].cpu(), o64.last_states[0].float())   # same RNG stream
    assert relerr(out, out_r) < 3e-5
    assert relerr(ed.grad, er.grad) < 2e-4
    pn = dict(pdec.named_parameters())
    for n, p in o64.named_parameters():
        e = relerr(pn[n].grad, p.grad)
        assert e < 2e-4, (n, e)


def test_lstm_encoder_module(gpe):
    """LSTMEncoderModule (nn/net_blocks.py:336-360): sequence in, last layer's final hidden state out."""
    Bn, T, El, Hh, L = 20, 9, 6, 24, 2
    torch.manual_seed(5)
    penc = gpe.net_blocks.LSTMEncoderModule(El, Hh, L, custom_init='kaiming_normal_')
    ref = torch.nn.LSTM(El, Hh, L, batch_first=True).double()
    ref.load_state_dict({k: v.double() for k, v in penc.lstm.state_dict().items()})
    penc = penc.cuda()
    x = torch.randn(Bn, T, El, generator=torch.Generator().manual_seed(1))
    wgt = torch.randn(Bn, Hh, generator=torch.Generator().manual_seed(2))
    xd = x.cuda().requires_grad_()
    torch.manual_seed(3)
    out = penc(xd)
    (out * wgt.cuda()).sum().backward()
    torch.manual_seed(3)
    h0 = gpe.net_blocks._init_tenzor(L, Bn, Hh, init_type='kaiming_normal_')
    c0 = gpe.net_blocks._init_tenzor(L, Bn, Hh, init_type='kaiming_normal_')
    xr = x.double().requires_grad_()
    _, (hN, _) = ref(xr, (h0.double(), c0.double()))
    (hN[-1] * wgt.double()).sum().backward()
    assert relerr(out, hN[-1]) < 2e-5
    assert relerr(xd.grad, xr.grad) < 1e-4
    for (n, p), (_, q) in zip(penc.lstm.named_parameters(), ref.named_parameters()):
        assert relerr(p.grad, q.grad) < 1e-4, n


def test_pointnetpp_block(gpe, golden_dir):
    """PointNetPlusPlus (nn/net_blocks.py:50-88) with PyG's conventions — fps(random_start=True) drawing from torch's generator,
    PointConv(add_self_loops=True) re-indexing the bipartite edge list: fps + ball query + the re-indexed edge list bit-exact vs
    the oracle's definitions and vs the reference run's edge list, outputs within 1e-4 of the reference-generated fixture and
    of the fp64 oracle, parameter gradients vs fp64."""
    import os
    from oracle import ref_path as O
    fx = torch.load(os.path.join(golden_dir, 'pointnetpp_small.pt'), weights_only=False)
    assert 'PyG conventions' in fx['provenance']
    torch.manual_seed(fx['seed'])
    pnet = gpe.net_blocks.PointNetPlusPlus(fx['out_size'], dict(fx['config']))
    assert [(k, tuple(v.shape)) for k, v in pnet.state_dict().items()] == [tuple(x) for x in fx['state_keys']]
    pnet.load_state_dict(fx['state_dict'])
    pnet = pnet.cuda().train()
    pos = fx['positions']
    torch.manual_seed(fx['fwd_seed'])                    # the forward draws the fps start points (one torch.rand(B))
    out = pnet(pos.cuda())
    (out * fx['wgt'].cuda()).sum().backward()
    o64 = O.PointNetPlusPlus(fx['out_size'], dict(fx['config'])).double().train()
    o64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in fx['state_dict'].items()})
    torch.manual_seed(fx['fwd_seed'])
    ref = o64(pos.double())
    (ref * fx['wgt'].double()).sum().backward()
    B, N = pos.shape[:2]
    # integer work: bit-exact (fps selection order incl. the random start, ball-query neighbour lists)
    tr = o64.sa1_module.trace
    last = pnet.sa1_module.last
    M = last['idx'].shape[1]
    gidx = (last['idx'].cpu().long() + (torch.arange(B) * N)[:, None]).view(-1)
    assert torch.equal(gidx, tr['idx'])
    assert gidx.view(B, M)[:, 0].tolist() != [0, N]      # the start points are random, not the clouds' first points
    cnt = last['cnt'].cpu().long()
    assert torch.equal(cnt, torch.bincount(tr['row'], minlength=B * M))
    nbr = last['nbr'].cpu().long()
    cols = torch.cat([nbr[s, :cnt[s]] + (s // M) * N for s in range(B * M)])
    assert torch.equal(cols, tr['col'])
    # PyG's re-indexed edge list: per centroid the kept ball neighbours in order, then the loop from flat point s
    ei = o64.sa1_module.conv.last_edge_index
    assert torch.equal(ei, fx['edge_index'])             # ... and it is the list the reference's own run used
    drop, ecnt = last['drop'].cpu().long(), last['edge_cnt'].cpu().long()
    assert int(ecnt.sum()) == ei.shape[1]
    for s in range(B * M):
        srcs = [int(nbr[s, q]) + (s // M) * N for q in range(cnt[s]) if q != drop[s]] + [s]
        assert srcs == ei[0][ei[1] == s].tolist(), s
    assert (drop >= 0).any() and (drop < 0).any()        # both cases occur in the fixture
    assert (out.detach().cpu() - fx['out']).abs().max().item() < 1e-4
    assert relerr(out, ref) < 5e-5
    pn = dict(pnet.named_parameters())
    for n, p in o64.named_parameters():
        assert relerr(pn[n].grad, p.grad) < 5e-3 if p.grad.dim() == 1 else relerr(pn[n].grad, p.grad) < 5e-4, n
    # the start points can be pinned instead (e.g. to replay a run): point 0 of every cloud = random_start=False
    pnet.sa1_module.fps_start = torch.zeros(B, dtype=torch.int32)
    pnet(pos.cuda())
    assert pnet.sa1_module.last['idx'][:, 0].tolist() == [0] * B


def test_batch_stager(gpe):
    """Pinned, double-buffered H2D staging + on-device standardisation (nn/trainer.py:93; nn/data/transforms.py:35-50)."""
    st = gpe.configs.data_config()['standardize']
    stager = gpe.staging.BatchStager('cuda:0', st['f_shift'], st['f_scale'])
    g = torch.Generator().manual_seed(9)
    for _ in range(5):                                    # more batches than slots: buffers are reused safely
        x = torch.randn(4, 300, 3, generator=g) * 20
        y = stager.stage(x)
        ref = (x - torch.tensor(st['f_shift'])) / torch.tensor(st['f_scale'])
        assert y.is_cuda and y.shape == x.shape
        assert torch.allclose(y.cpu(), ref, rtol=1e-6, atol=1e-6)


def test_two_rank_exchange_on_shared_gpu(gpe, golden_dir, tmp_path):
    """The N > 1 path with the REAL HIP model: two processes share cuda:0 (GPE_SHARE_DEVICE=1) and exchange gradients over
    gloo (RCCL refuses two ranks on one device).  Attention variant: its `feature_extractor.lin` never gets a gradient, so
    one bucket only completes in finish_gradient_sync().  Result must equal the mean of the two ranks' local gradients."""
    import os, socket, subprocess, sys, textwrap
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'w2.py'
    script.write_text(textwrap.dedent("""
        import copy, os, sys
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        import gpe_amd
        from gpe_amd import parallel
        rank, local, world = parallel.init_distributed(backend='gloo')
        assert world == 2 and torch.cuda.current_device() == 0
        fx = torch.load(os.path.join(%r, 'segment3d_small.pt'), weights_only=False)
        def build():
            torch.manual_seed(fx['seed'])
            return gpe_amd.nets.GarmentSegmentPattern3D(fx['data_config'], copy.deepcopy(fx['nn_config']),
                                                        copy.deepcopy(fx['loss_config'])).cuda().train()
        sl = slice(rank, rank + 1)                       # one garment per rank
        def run(m, call):
            torch.manual_seed(fx['seed'] + 2 + rank)
            preds = call(fx['features'][sl].cuda(), log_step=0, epoch=0)
            loss, _, _ = m.loss(preds, {k: v[sl].clone() for k, v in fx['gt'].items()}, epoch=0)
            loss.backward()
        plain = build(); run(plain, plain)
        local_g = {n: (p.grad.clone() if p.grad is not None else None) for n, p in plain.named_parameters()}
        model = build()
        from gpe_amd import ops
        ops.SIDE_MIN_EDGES = 0                           # the leaf weight-gradient launches on the side stream (ops.side_grads) although the
        assert ops.SIDE_GRADS                            # fixture is small: buckets must still leave with complete gradients
        ddp = parallel.DistributedHotPath(model, device_ids=[torch.device('cuda', 0)], bucket_bytes=16 << 10)
        assert len(ddp._buckets) > 2
        # N > 1: two CUs per XCD are kept out of the persistent launches by default (room for the collective's kernels)
        assert ddp.reserved_cus == 16 and gpe_amd._lib.lib().gpe_reserve_cus_set(16) == 16
        run(model, ddp)
        early = len(ddp._launched)
        ddp.finish_gradient_sync()
        torch.cuda.synchronize()
        ex = ddp.exposed_ms()
        assert ex is not None and ex >= 0.0               # the un-hidden part of the exchange is measured and reported (bench.py N > 1 lines)
        assert 0 < early < len(ddp._buckets)             # some buckets left during backward, the None-grad one at the end
        assert ops._SIDE_STREAMS                          # (the side stream was used)
        for n, p in model.named_parameters():
            g = local_g[n]
            if g is None:
                assert not p.grad.any(), n
                continue
            both = [torch.empty_like(g) for _ in range(2)]
            dist.all_gather(both, g)
            want = (both[0] + both[1]) / 2
            assert torch.allclose(p.grad, want, rtol=1e-6, atol=1e-12), n
        dist.barrier()
        dist.destroy_process_group()
        print('rank', rank, 'shared-gpu exchange ok', early, len(ddp._buckets))
    """) % (repo, str(golden_dir)))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), GPE_SHARE_DEVICE='1')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert 'rank %d shared-gpu exchange ok' % rank in out


# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('K,N,with_ws', [(200, 200, True), (200, 150, True), (64, 96, False), (150, 130, True), (13, 7, False)])
def test_pack_fold_is_the_three_launches(gpe, K, N, with_ws):
    """gpe_pack_fold (ABI v7) = gpe_pack_weight(col_scale = s) -> gpe_fold_bias(t) -> the packed weight's f16x3 amax pass in one launch:
    every output BIT-identical to the separate entry points, the ticket word left zero, twice in a row."""
    from gpe_amd import ops, _lib as L
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(K * 1000 + N)
    stats = torch.randn(4, K, generator=g).to(dev)
    W = torch.randn(N, K, generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    wp1 = ops.pack_weight(W, col_scale=stats[2])
    bf1 = ops.fold_bias(W, bias, stats[3])
    ews, ews_n = ops.edge_workspace(2, 64, 4, 512, dev)
    word = torch.full((1,), 77, device=dev, dtype=torch.int32)
    for rep in range(2):
        if with_ws:
            ews.zero_()
            word.fill_(77)
        wp2, bf2 = ops.pack_fold(W, bias, stats, ews if with_ws else None, ews_n, word if with_ws else None)
        torch.cuda.synchronize()
        assert torch.equal(wp1, wp2) and torch.equal(bf1, bf2)
        if with_ws:
            assert int(ops._ticket(dev)) == 0
            slots = ews.view(torch.int32)[64 * 512: 64 * 512 + 2].cpu()       # behind the dummy store image: [A amax, packed-weight amax]
            assert int(slots[0]) == 0 and int(word) == 0
            assert int(slots[1]) == int(wp1.abs().max().view(torch.int32))
