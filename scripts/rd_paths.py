"""Measurement aid (GPU): the edge weight-gradient reduce-GEMMs at mid sizes — which kernel family wins between the row-poor
("deep") kernel and the producer/consumer kernels (GPE_RD_DEEP=0 keeps the deep kernel off)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpe_amd as gpe
ops, L = gpe.ops, gpe._lib

def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

gpe.set_f16x3_min_rows(0)
for mode in ('f32', 'f16x3'):
    gpe.set_math(mode)
    for B, N, k in [(8, 1024, 5), (8, 512, 16), (8, 1024, 16), (16, 1024, 16), (8, 2048, 16), (16, 2048, 16)]:
        E, BN, H, Fo = B * N * k, B * N, 200, 150
        g = torch.Generator().manual_seed(1)
        a3 = (torch.randn(E, 152, generator=g) * 1e-3).cuda(); a3[:, Fo:] = 0
        a2 = torch.randn(E, H, generator=g).abs().cuda()
        pq = torch.randn(BN, 2 * H, generator=g).cuda()
        jg = (torch.randint(0, N, (B, N, k), generator=g) + torch.arange(B).view(B, 1, 1) * N).int().cuda()
        shift = torch.randn(H, generator=g).abs().cuda()
        G, cs = torch.empty(H, H).cuda(), torch.empty(H).cuda()
        part = torch.empty(L.query('gpe_redgemm_ws', H, H)).cuda()
        ws, nws = ops.edge_workspace(B, N, k, 2 * H, 'cuda')
        words = torch.zeros(4, dtype=torch.int32, device='cuda')
        wu = wv = wq = wd = None
        if mode == 'f16x3':
            L.call('gpe_absmax', a3, 152, E, Fo, words[0:1]); L.call('gpe_absmax', a2, H, E, H, words[1:2])
            L.call('gpe_edge_pq_amax', pq, 2 * H, H, BN, words[2:3], ws, nws)
            wu, wv, wq, wd = words[0:1], words[1:2], words[2:3], words[1:2]
        rd = lambda: L.call('gpe_edge_redgemm', a3, 152, 1, a2, H, None, 0, None, shift, B, N, k, Fo, H, G, H, cs, part, wu, wv, ws, nws, None, 0, None, None, 0, None)
        rg = lambda: L.call('gpe_edge_redgemm', a2, H, 0, None, 0, pq, 2 * H, jg, shift, B, N, k, H, H, G, H, cs, part, wd, wq, ws, nws, None, 0, None, None, 0, None)
        print('%-6s B=%-3d N=%-5d k=%-3d E=%-8d dense %7.1f us   gather %7.1f us' % (mode, B, N, k, E, timeit(rd), timeit(rg)), flush=True)
# the decoders' small weight-gradient products (rows, Mg, Ng)
gpe.set_math('f16x3')
for rows, Mg, Ng in [(32, 250, 150), (32, 1000, 250), (736, 1000, 250), (736, 250, 250), (736, 7, 250), (10304, 8, 250), (10304, 1000, 250), (65536, 400, 3), (65536, 400, 150)]:
    u = torch.randn(rows, Mg).cuda(); v = torch.randn(rows, Ng).cuda()
    t = timeit(lambda: ops.redgemm_raw(ops._rows2d(u), ops._rows2d(v), rows, Mg, Ng, want_colsum=True))
    print('redgemm rows=%-6d %4d x %-4d %7.1f us  %6.1f GFLOP/s' % (rows, Mg, Ng, t, 2.0 * rows * Mg * Ng / t * 1e-3), flush=True)
for M, K, N_ in [(32, 250, 1000), (32, 150, 250), (736, 250, 1000), (736, 1000, 250), (736, 250, 250), (736, 250, 7), (10304, 250, 8), (65536, 150, 400), (65536, 400, 150), (65536, 3, 400)]:
    x = torch.randn(M, K).cuda(); w = torch.randn(N_, K).cuda(); b = torch.randn(N_).cuda()
    wp = ops.pack_weight(w)
    y = torch.empty(M, N_).cuda()
    t = timeit(lambda: ops.linear_raw(ops._rows2d(x), wp, b, M, N_, K, ops._rows2d(y)))
    print('linear  M=%-6d K=%-4d N=%-4d %7.1f us  %6.1f GFLOP/s' % (M, K, N_, t, 2.0 * M * K * N_ / t * 1e-3), flush=True)
