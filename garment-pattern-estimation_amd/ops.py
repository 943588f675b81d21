"""Host-side orchestration of the gfx950 kernels (C ABI in include/gpe_hip.h) + the autograd glue.

torch is used here for device memory, streams, parameter storage and the CPU RNG of the reference's random
LSTM states — every arithmetic step of the path is a call into libgpe_hip.so.  Nothing in this file falls back
to torch math: a missing library or a failing launch raises.

Reference lines restated by each Function are cited in its docstring (paths relative to /root/reference).
"""
import torch

from . import _lib as L

F32 = torch.float32


def _dev_check(t):
    if not t.is_cuda:
        raise RuntimeError('gpe ops need tensors on the MI355X (got a %s tensor); there is no CPU path' % t.device)
    if t.dtype != F32:
        raise RuntimeError('gpe ops are fp32 (got %s)' % t.dtype)


def _rows2d(t):
    """(tensor, stride_outer, stride_inner, inner) descriptor of a [M, K] tensor with unit inner stride."""
    assert t.dim() == 2 and (t.shape[1] == 1 or t.stride(1) == 1), (t.shape, t.stride())
    return (t, t.stride(0), 0, 0)


def _rows3d(t):
    """descriptor of a [R, T, K] tensor flattened to R*T logical rows (row r*T+t)."""
    assert t.dim() == 3 and (t.shape[2] == 1 or t.stride(2) == 1), (t.shape, t.stride())
    return (t, t.stride(0), t.stride(1), t.shape[1])


def round_up(a, b):
    return (a + b - 1) // b * b


# -------------------------------------------------------------------------------------------------
# raw wrappers
# -------------------------------------------------------------------------------------------------
def pack_weight(w, transpose=False, col_scale=None):
    """w: [N,K] (nn.Linear layout).  transpose=True packs w^T (operand [K_src_cols][K_src_rows])."""
    _dev_check(w)
    assert w.dim() == 2 and w.stride(1) == 1
    N, K = (w.shape[1], w.shape[0]) if transpose else (w.shape[0], w.shape[1])
    wp = torch.empty(L.query('gpe_packed_size', N, K), device=w.device, dtype=F32)
    L.call('gpe_pack_weight', w, w.stride(0), N, K, int(transpose), col_scale, wp)
    return wp


def fold_bias(w, bias, t):
    out = torch.empty(w.shape[0], device=w.device, dtype=F32)
    L.call('gpe_fold_bias', w, w.stride(0), w.shape[0], w.shape[1], bias, t, out)
    return out


def linear_raw(a_desc, wp, bias, M, N, K, y_desc, act=0, addend_desc=None):
    ad = addend_desc if addend_desc is not None else (None, 0, 0, 0)
    L.call('gpe_linear', a_desc[0], a_desc[1], a_desc[2], a_desc[3], wp, bias,
           ad[0], ad[1], ad[2], ad[3], y_desc[0], y_desc[1], y_desc[2], y_desc[3], M, N, K, act)


def redgemm_raw(u_desc, v_desc, rows, Mg, Ng, want_colsum=True, accumulate_into=None):
    """G[Mg,Ng] = sum_r U[r,:]^T V[r,:], colsum[Mg] = sum_r U[r,:]."""
    dev = u_desc[0].device
    if accumulate_into is not None:
        G, cs = accumulate_into
    else:
        G = torch.empty(Mg, Ng, device=dev, dtype=F32)
        cs = torch.empty(Mg, device=dev, dtype=F32) if want_colsum else None
    acc = int(accumulate_into is not None)
    for n0 in range(0, Ng, 256):          # the kernel keeps <= 256 V columns resident per pass
        nb = min(256, Ng - n0)
        ws = torch.empty(L.query('gpe_redgemm_ws', Mg, nb), device=dev, dtype=F32)
        L.call('gpe_redgemm', u_desc[0], u_desc[1], u_desc[2], u_desc[3],
               v_desc[0][..., n0:], v_desc[1], v_desc[2], v_desc[3], None,
               rows, Mg, nb, G[:, n0:], G.stride(0), cs if n0 == 0 else None, ws, acc)
    return G, cs


def knn(x, B, N, k, want_global=False):
    """x: [B*N, C] rows (ld = x.stride(0)).  -> int32 [B, N, k] local neighbour indices (and, optionally, the same
    graph as global row numbers b*N + idx, the form the gather kernels consume).
    Replaces torch_cluster.knn under DynamicEdgeConv (nn/net_blocks.py:127-135)."""
    _dev_check(x)
    idx = torch.empty(B, N, k, device=x.device, dtype=torch.int32)
    jg = torch.empty(B, N, k, device=x.device, dtype=torch.int32) if want_global else None
    L.call('gpe_knn', x, B, N, x.shape[1], x.stride(0), k, idx, jg)
    return (idx, jg) if want_global else idx


def knn_reverse(idx):
    B, N, k = idx.shape
    off = torch.empty(B, N + 1, device=idx.device, dtype=torch.int32)
    edge = torch.empty(B, N * k, device=idx.device, dtype=torch.int32)
    L.call('gpe_knn_reverse', idx, B, N, k, off, edge)
    return off, edge


def bn_finalize(part, nblk, C, count, gamma, beta, eps, momentum, rm, rv, nbt):
    stats = torch.empty(4, C, device=gamma.device, dtype=F32)
    L.call('gpe_bn_finalize', part, nblk, C, float(count), gamma, beta, float(eps), float(momentum), rm, rv, nbt,
           stats)
    return stats


def bn_from_running(rm, rv, gamma, beta, eps):
    C = gamma.shape[0]
    stats = torch.empty(4, C, device=gamma.device, dtype=F32)
    L.call('gpe_bn_from_running', rm, rv, C, gamma, beta, float(eps), stats)
    return stats


def bn_bwd_coef(part, nblk, stats, C, count, want_param_grads=True):
    dev = stats.device
    coef = torch.empty(4, C, device=dev, dtype=F32)
    dg = torch.empty(C, device=dev, dtype=F32) if want_param_grads else None
    db = torch.empty(C, device=dev, dtype=F32) if want_param_grads else None
    L.call('gpe_bn_bwd_coef', part, nblk, stats, C, float(count), coef, dg, db)
    return coef, dg, db


# -------------------------------------------------------------------------------------------------
# nn.Linear
# -------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) on the MFMA row-GEMM; backward = row-GEMM (dx) + reduce-GEMM (dW, db).
    Replaces torch.nn.Linear at nn/net_blocks.py:158,187,397 and nn/nets.py:128-130,153,229-233."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _dev_check(x)
        M, K = x.shape
        N = weight.shape[0]
        y = torch.empty(M, N, device=x.device, dtype=F32)
        linear_raw(_rows2d(x), pack_weight(weight), bias, M, N, K, _rows2d(y))
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        M, K = x.shape
        N = weight.shape[0]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(M, K, device=x.device, dtype=F32)
            linear_raw(_rows2d(gy), pack_weight(weight, transpose=True), None, M, K, N, _rows2d(gx))
        if ctx.needs_input_grad[1] or ctx.has_bias:
            gw, gb = redgemm_raw(_rows2d(gy), _rows2d(x), M, N, K, want_colsum=True)
            if not ctx.has_bias:
                gb = None
        return gx, gw, gb


def linear(x, weight, bias=None):
    return LinearFn.apply(x, weight, bias)


# -------------------------------------------------------------------------------------------------
# global mean pool
# -------------------------------------------------------------------------------------------------
class SegmentMeanFn(torch.autograd.Function):
    """torch_geometric.nn.global_mean_pool over equal-sized clouds (nn/net_blocks.py:148,184; nn/nets.py:272)."""

    @staticmethod
    def forward(ctx, x, B, N):
        _dev_check(x)
        C = x.shape[1]
        y = torch.empty(B, C, device=x.device, dtype=F32)
        L.call('gpe_segment_mean_fwd', x, x.stride(0), B, N, C, y, C)
        ctx.dims = (B, N, C)
        return y

    @staticmethod
    def backward(ctx, gy):
        B, N, C = ctx.dims
        gy = gy.contiguous()
        gx = torch.empty(B * N, C, device=gy.device, dtype=F32)
        L.call('gpe_segment_mean_bwd', gy, C, B, N, C, gx, C, 0)
        return gx, None, None


def segment_mean(x, B, N):
    return SegmentMeanFn.apply(x, B, N)


# -------------------------------------------------------------------------------------------------
# EdgeConv layer
# -------------------------------------------------------------------------------------------------
class EdgeConvFn(torch.autograd.Function):
    """One DynamicEdgeConv(MLP([2C, H, H, F]), k, aggr='max') layer (nn/net_blocks.py:43-47,124-135,174):
    kNN graph on the input features -> per-edge [Linear->ReLU->BatchNorm]x3 -> max over the k messages.

    Pipeline (training): kNN | split W1 -> per-point P,Q GEMM | gather+stats(a1) | fold BN1->W2, fused
    gather+GEMM+ReLU(+stats) | fold BN2->W3, fused GEMM+ReLU(+stats)+max/min | BN3 after the max.
    Saved for backward: idx, PQ, a2 [E,H], a3 [E,F], max/min + arg slots, the three BN stat blocks."""

    @staticmethod
    def forward(ctx, x, B, N, k, training, eps, momentum,
                W1, b1, g1, be1, W2, b2, g2, be2, W3, b3, g3, be3,
                rm1, rv1, nb1, rm2, rv2, nb2, rm3, rv3, nb3):
        _dev_check(x)
        dev = x.device
        BN, C = x.shape
        assert BN == B * N
        H, Fo = W1.shape[0], W3.shape[0]
        assert W1.shape[1] == 2 * C and W2.shape == (H, H) and W3.shape[1] == H
        if H % 4:
            raise ValueError('EConv_hidden must be a multiple of 4 (got %d)' % H)
        E = BN * k
        ldF = round_up(Fo, 4)
        nblk = L.query('gpe_stats_blocks')

        idx, jg = knn(x, B, N, k, want_global=True)
        # per-point projection: [P|Q] = x [W1a-W1b | W1b]^T + [b1|0]
        wpq = torch.empty(2 * H, C, device=dev, dtype=F32)
        bpq = torch.empty(2 * H, device=dev, dtype=F32)
        L.call('gpe_w1_split', W1, W1.stride(0), b1, H, C, wpq, C, bpq)
        PQ = torch.empty(BN, 2 * H, device=dev, dtype=F32)
        linear_raw(_rows2d(x), pack_weight(wpq), bpq, BN, 2 * H, C, _rows2d(PQ))

        def stats_of(part, Cc, g, be, rm, rv, nb):
            if training:
                return bn_finalize(part, nblk, Cc, E, g, be, eps, momentum, rm, rv, nb)
            return bn_from_running(rm, rv, g, be, eps)

        part = torch.empty(nblk, 2, H, device=dev, dtype=torch.float64) if training else None
        if training:
            L.call('gpe_edge_gather_stats', PQ, 2 * H, H, jg, B, N, k, part)
        st1 = stats_of(part, H, g1, be1, rm1, rv1, nb1)

        a2 = torch.empty(E, H, device=dev, dtype=F32)
        L.call('gpe_edge_mlp_fwd', 0, PQ, 2 * H, jg, None, 0, B, N, k, H, H,
               pack_weight(W2, col_scale=st1[2]), fold_bias(W2, b2, st1[3]), a2, H, part,
               0, None, None, None, None, 0)
        st2 = stats_of(part, H, g2, be2, rm2, rv2, nb2)

        a3 = torch.empty(E, ldF, device=dev, dtype=F32)
        mx = torch.empty(BN, ldF, device=dev, dtype=F32)
        mn = torch.empty(BN, ldF, device=dev, dtype=F32)
        amx = torch.empty(BN, ldF, device=dev, dtype=torch.uint8)
        amn = torch.empty(BN, ldF, device=dev, dtype=torch.uint8)
        part3 = torch.empty(nblk, 2, Fo, device=dev, dtype=torch.float64) if training else None
        L.call('gpe_edge_mlp_fwd', 1, None, 0, None, a2, H, B, N, k, H, Fo,
               pack_weight(W3, col_scale=st2[2]), fold_bias(W3, b3, st2[3]), a3, ldF, part3,
               1, mx, mn, amx, amn, ldF)
        st3 = stats_of(part3, Fo, g3, be3, rm3, rv3, nb3)

        out = torch.empty(BN, ldF, device=dev, dtype=F32)[:, :Fo]
        L.call('gpe_edge_finish', mx, mn, ldF, st3, BN, Fo, out, ldF)

        ctx.dims = (B, N, k, C, H, Fo, ldF)
        ctx.save_for_backward(x, idx, jg, PQ, a2, a3, mx, mn, amx, amn, st1, st2, st3, W1, W2, W3, wpq)
        ctx.mark_non_differentiable(idx)
        return out, idx

    @staticmethod
    def backward(ctx, g_out, _g_idx):
        (x, idx, jg, PQ, a2, a3, mx, mn, amx, amn, st1, st2, st3, W1, W2, W3, wpq) = ctx.saved_tensors
        B, N, k, C, H, Fo, ldF = ctx.dims
        dev = x.device
        BN, E = B * N, B * N * k
        if g_out.stride(1) != 1:
            g_out = g_out.contiguous()
        ldg = g_out.stride(0)

        # ---- block 3: BN3 applied after the max -------------------------------------------------------
        psb = L.query('gpe_point_sums_blocks')
        part = torch.empty(psb, 2, Fo, device=dev, dtype=torch.float64)
        L.call('gpe_edge_bwd_point_sums', g_out, ldg, mx, mn, ldF, st3, BN, Fo, part)
        coef3, dg3, dbe3 = bn_bwd_coef(part, psb, st3, Fo, E)
        # dz3 in place over a3 (one coalesced pass), then everything downstream reads dense rows
        L.call('gpe_edge_dz3', a3, ldF, g_out, ldg, amx, amn, ldF, coef3, B, N, k, Fo)
        G3 = torch.empty(Fo, H, device=dev, dtype=F32)
        db3 = torch.empty(Fo, device=dev, dtype=F32)
        ws = torch.empty(max(L.query('gpe_redgemm_ws', Fo, H), L.query('gpe_redgemm_ws', H, H)), device=dev,
                         dtype=F32)
        L.call('gpe_edge_redgemm', a3, ldF, 1, a2, H, None, 0, None, st2[0], B, N, k, Fo, H, G3, H, db3, ws)
        sums = torch.empty(1, 2, H, device=dev, dtype=torch.float64)
        dW3 = torch.empty(Fo, H, device=dev, dtype=F32)
        L.call('gpe_bn_bwd_from_G', G3, H, db3, W3, W3.stride(0), Fo, H, st2, sums, dW3, H)
        coef2, dg2, dbe2 = bn_bwd_coef(sums, 1, st2, H, E)
        # dz2 = (a2>0) ? s2*(dz3 W3) - k1 - a2*k2 : 0, in place over a2
        L.call('gpe_edge_mlp_bwd', a3, ldF, 0, None, 0, None, B, N, k, Fo, H,
               pack_weight(W3, transpose=True), coef2, a2, H, None, 0)

        # ---- block 2 -----------------------------------------------------------------------------------
        G2 = torch.empty(H, H, device=dev, dtype=F32)
        db2 = torch.empty(H, device=dev, dtype=F32)
        L.call('gpe_edge_redgemm', a2, H, 0, None, 0, PQ, 2 * H, jg, st1[0], B, N, k, H, H, G2, H, db2, ws)
        sums1 = torch.empty(1, 2, H, device=dev, dtype=torch.float64)
        dW2 = torch.empty(H, H, device=dev, dtype=F32)
        L.call('gpe_bn_bwd_from_G', G2, H, db2, W2, W2.stride(0), H, H, st1, sums1, dW2, H)
        coef1, dg1, dbe1 = bn_bwd_coef(sums1, 1, st1, H, E)
        # dz1 = (a1>0) ? s1*(dz2 W2) - k1 - a1*k2 : 0 (a1 re-gathered), in place over a2; dP = sum over slots
        dPQ = torch.empty(BN, 2 * H, device=dev, dtype=F32)
        L.call('gpe_edge_mlp_bwd', a2, H, 1, PQ, 2 * H, jg, B, N, k, H, H,
               pack_weight(W2, transpose=True), coef1, a2, H, dPQ, 2 * H)

        # ---- block 1: gather backward = deterministic pull through the transposed graph -----------------
        rev_off, rev_edge = knn_reverse(idx)
        L.call('gpe_edge_pull_dq', a2, H, rev_off, rev_edge, B, N, k, H, dPQ[:, H:], 2 * H)
        dWpq, dbpq = redgemm_raw(_rows2d(dPQ), _rows2d(x), BN, 2 * H, C, want_colsum=True)
        dW1 = torch.empty(H, 2 * C, device=dev, dtype=F32)
        L.call('gpe_w1_grad_from_pq', dWpq, C, H, C, dW1, 2 * C)
        db1 = dbpq[:H].clone()
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(BN, C, device=dev, dtype=F32)
            linear_raw(_rows2d(dPQ), pack_weight(wpq, transpose=True), None, BN, C, 2 * H, _rows2d(gx))
        return (gx, None, None, None, None, None, None,
                dW1, db1, dg1, dbe1, dW2, db2, dg2, dbe2, dW3, db3, dg3, dbe3,
                None, None, None, None, None, None, None, None, None)


# -------------------------------------------------------------------------------------------------
# LSTM decoder
# -------------------------------------------------------------------------------------------------
class LSTMDecoderFn(torch.autograd.Function):
    """nn.LSTM(enc, hid, n_layers, batch_first) fed the SAME encoding at every step + the output Linear
    (LSTMDecoderModule.forward, nn/net_blocks.py:382-402).  Layer 0's input projection is computed once
    (the input is time-invariant, :388); layers > 0 project all T steps in one GEMM; the recurrence runs
    T x (row-GEMM with fused `+xproj` addend, pointwise cell).  Gate order i,f,g,o."""

    @staticmethod
    def forward(ctx, enc, h0, c0, T, n_layers, lin_w, lin_b, *lstm_params):
        _dev_check(enc)
        dev = enc.device
        Bn, In = enc.shape
        Hh = h0.shape[2]
        saved_layers = []
        x_desc, x_is_enc = None, True
        prev_hs = None
        for l in range(n_layers):
            w_ih, w_hh, b_ih, b_hh = lstm_params[4 * l: 4 * l + 4]
            bias = torch.empty(4 * Hh, device=dev, dtype=F32)
            L.call('gpe_add', b_ih, b_hh, bias, 4 * Hh)
            # row pitch padded to 16 B (zero pad) so the recurrence's A operand is staged with plain aligned loads
            Hp = round_up(Hh, 4)
            hs = torch.zeros(Bn, T + 1, Hp, device=dev, dtype=F32)[:, :, :Hh]
            hs[:, 0].copy_(h0[l])
            cs = torch.empty(T + 1, Bn, Hh, device=dev, dtype=F32)
            cs[0].copy_(c0[l])
            gates = torch.empty(T, Bn, 4 * Hh, device=dev, dtype=F32)
            if l == 0:
                xproj = torch.empty(Bn, 4 * Hh, device=dev, dtype=F32)
                linear_raw(_rows2d(enc), pack_weight(w_ih), bias, Bn, 4 * Hh, In, _rows2d(xproj))
            else:
                xproj = torch.empty(Bn, T, 4 * Hh, device=dev, dtype=F32)
                linear_raw(_rows3d(prev_hs[:, 1:]), pack_weight(w_ih), bias, Bn * T, 4 * Hh, Hh,
                           (xproj, 4 * Hh, 0, 0))
            whh_p = torch.empty(L.query('gpe_packed_gates_size', Hh, Hh), device=dev, dtype=F32)
            L.call('gpe_pack_weight_gates', w_hh, w_hh.stride(0), Hh, Hh, whh_p)
            for t in range(T):
                xp, xps = (xproj, 4 * Hh) if l == 0 else (xproj[:, t], T * 4 * Hh)
                # gates = h_{t-1} W_hh^T + xproj_t, cell update, h_t / c_t / activated gates: one launch
                L.call('gpe_lstm_step_fwd', hs[:, t], hs.stride(0), whh_p, xp, xps, cs[t], Hh,
                       gates[t], cs[t + 1], hs[:, t + 1], hs.stride(0), Bn, Hh)
            saved_layers += [hs, cs, gates]
            prev_hs = hs
        out_sz = lin_w.shape[0]
        out = torch.empty(Bn, T, out_sz, device=dev, dtype=F32)
        linear_raw(_rows3d(prev_hs[:, 1:]), pack_weight(lin_w), lin_b, Bn * T, out_sz, Hh, (out, out_sz, 0, 0))
        ctx.dims = (Bn, In, Hh, T, n_layers, out_sz)
        ctx.save_for_backward(enc, lin_w, *lstm_params, *saved_layers)
        return out

    @staticmethod
    def backward(ctx, g_out):
        Bn, In, Hh, T, n_layers, out_sz = ctx.dims
        sv = ctx.saved_tensors
        enc, lin_w = sv[0], sv[1]
        lstm_params = sv[2: 2 + 4 * n_layers]
        saved_layers = sv[2 + 4 * n_layers:]
        dev = enc.device
        g_out = g_out.contiguous()
        top_hs = saved_layers[3 * (n_layers - 1)]
        d_lin_w, d_lin_b = redgemm_raw((g_out, out_sz, 0, 0), _rows3d(top_hs[:, 1:]), Bn * T, out_sz, Hh)
        dH = torch.empty(Bn, T, Hh, device=dev, dtype=F32)
        linear_raw((g_out, out_sz, 0, 0), pack_weight(lin_w, transpose=True), None, Bn * T, Hh, out_sz,
                   (dH, Hh, 0, 0))
        grads = [None] * (4 * n_layers)
        d_enc = None
        for l in reversed(range(n_layers)):
            w_ih, w_hh = lstm_params[4 * l], lstm_params[4 * l + 1]
            hs, cs, gates = saved_layers[3 * l: 3 * l + 3]
            dG = torch.empty(Bn, T, 4 * Hh, device=dev, dtype=F32)
            nz = (4 * Hh + 255) // 256                     # split-K partials of dh_rec = dG_t . W_hh
            dh_rec = torch.empty(nz, Bn, Hh, device=dev, dtype=F32)
            dc = [torch.empty(Bn, Hh, device=dev, dtype=F32) for _ in range(2)]
            whh_t = pack_weight(w_hh, transpose=True)
            for t in reversed(range(T)):
                last = t == T - 1
                L.call('gpe_lstm_cell_bwd', dH[:, t], T * Hh, None if last else dh_rec, nz,
                       None if last else dc[(t + 1) & 1], gates[t], cs[t + 1], cs[t], Hh,
                       dG[:, t], T * 4 * Hh, dc[t & 1], Bn, Hh)
                if t > 0:
                    L.call('gpe_linear_splitk', dG[:, t], T * 4 * Hh, whh_t, dh_rec, Bn, Hh, 4 * Hh)
            dG_rows = (dG, 4 * Hh, 0, 0)
            d_whh, d_b = redgemm_raw(dG_rows, _rows3d(hs[:, :T]), Bn * T, 4 * Hh, Hh)
            if l == 0:
                d_wih, _ = redgemm_raw(dG_rows, (enc, enc.stride(0), 0, T), Bn * T, 4 * Hh, In, want_colsum=False)
                if ctx.needs_input_grad[0]:
                    dGs = torch.empty(Bn, 4 * Hh, device=dev, dtype=F32)
                    L.call('gpe_reduce_inner', dG, T * 4 * Hh, 4 * Hh, T, Bn, 4 * Hh, dGs, 4 * Hh, 0)
                    d_enc = torch.empty(Bn, In, device=dev, dtype=F32)
                    linear_raw(_rows2d(dGs), pack_weight(w_ih, transpose=True), None, Bn, In, 4 * Hh,
                               _rows2d(d_enc))
            else:
                lower_hs = saved_layers[3 * (l - 1)]
                d_wih, _ = redgemm_raw(dG_rows, _rows3d(lower_hs[:, 1:]), Bn * T, 4 * Hh, Hh, want_colsum=False)
                dH = torch.empty(Bn, T, Hh, device=dev, dtype=F32)
                linear_raw(dG_rows, pack_weight(w_ih, transpose=True), None, Bn * T, Hh, 4 * Hh, (dH, Hh, 0, 0))
            grads[4 * l: 4 * l + 4] = [d_wih, d_whh, d_b, d_b.clone()]
        return (d_enc, None, None, None, None, d_lin_w, d_lin_b, *grads)


# -------------------------------------------------------------------------------------------------
# attention variant (GarmentSegmentPattern3D)
# -------------------------------------------------------------------------------------------------
class DenseMLPFn(torch.autograd.Function):
    """MLP(channels) = [Linear -> ReLU -> BatchNorm1d] x n on dense rows (nn/net_blocks.py:43-47 as used by
    point_segment_mlp, nn/nets.py:223-226).  Same kernels as the edge MLP with one "message" per row (k = 1):
    fused Linear+ReLU(+fp64 BN statistics), every BatchNorm folded into the next Linear, the last one applied
    explicitly; backward = the edge MLP's chain (centred reduce-GEMM -> BN coefficients -> propagate in place)."""

    @staticmethod
    def forward(ctx, x, training, eps, momentum, n_blocks, *tensors):
        _dev_check(x)
        dev = x.device
        M = x.shape[0]
        nblk = L.query('gpe_stats_blocks')
        params = tensors[:4 * n_blocks]
        bufs = tensors[4 * n_blocks:]
        acts, stats = [], []
        a_in, Cin = x, x.shape[1]
        scale = tvec = None
        for l in range(n_blocks):
            W, b, g, be = params[4 * l: 4 * l + 4]
            rm, rv, nb = bufs[3 * l: 3 * l + 3]
            Cout = W.shape[0]
            ldo = round_up(Cout, 4)
            a = torch.empty(M, ldo, device=dev, dtype=F32)
            part = torch.empty(nblk, 2, Cout, device=dev, dtype=torch.float64) if training else None
            L.call('gpe_edge_mlp_fwd', 1, None, 0, None, a_in, a_in.stride(0), 1, M, 1, Cin, Cout,
                   pack_weight(W, col_scale=scale), b if tvec is None else fold_bias(W, b, tvec), a, ldo, part,
                   0, None, None, None, None, 0)
            st = bn_finalize(part, nblk, Cout, M, g, be, eps, momentum, rm, rv, nb) if training \
                else bn_from_running(rm, rv, g, be, eps)
            acts.append(a)
            stats.append(st)
            scale, tvec = st[2], st[3]
            a_in, Cin = a, Cout
        y = torch.empty(M, Cin, device=dev, dtype=F32)
        L.call('gpe_bn_apply', a_in, a_in.stride(0), stats[-1], M, Cin, y, Cin)
        ctx.n_blocks = n_blocks
        ctx.save_for_backward(x, *params, *acts, *stats)
        return y

    @staticmethod
    def backward(ctx, gy):
        n = ctx.n_blocks
        sv = ctx.saved_tensors
        x, params, acts, stats = sv[0], sv[1:1 + 4 * n], sv[1 + 4 * n:1 + 5 * n], sv[1 + 5 * n:1 + 6 * n]
        dev = x.device
        M = x.shape[0]
        gy = gy.contiguous()
        grads = [None] * (4 * n)
        # last block: BN applied explicitly -> same algebra as the edge layer's BN-after-max with one slot per row
        a, st = acts[-1], stats[-1]
        C = params[4 * (n - 1)].shape[0]
        psb = L.query('gpe_point_sums_blocks')
        part = torch.empty(psb, 2, C, device=dev, dtype=torch.float64)
        L.call('gpe_edge_bwd_point_sums', gy, C, a, a, a.stride(0), st, M, C, part)
        coef, dgam, dbet = bn_bwd_coef(part, psb, st, C, M)
        slot0 = torch.zeros(M, a.stride(0), device=dev, dtype=torch.uint8)
        L.call('gpe_edge_dz3', a, a.stride(0), gy, C, slot0, slot0, a.stride(0), coef, 1, M, 1, C)
        grads[4 * (n - 1) + 2], grads[4 * (n - 1) + 3] = dgam, dbet
        for l in reversed(range(n)):
            W = params[4 * l]
            dz = acts[l]                          # holds dz_l now
            C = W.shape[0]
            if l > 0:
                prev, stp = acts[l - 1], stats[l - 1]
                Cp = params[4 * (l - 1)].shape[0]
                G = torch.empty(C, Cp, device=dev, dtype=F32)
                db = torch.empty(C, device=dev, dtype=F32)
                ws = torch.empty(L.query('gpe_redgemm_ws', C, Cp), device=dev, dtype=F32)
                L.call('gpe_edge_redgemm', dz, dz.stride(0), 1, prev, prev.stride(0), None, 0, None, stp[0],
                       1, M, 1, C, Cp, G, Cp, db, ws)
                sums = torch.empty(1, 2, Cp, device=dev, dtype=torch.float64)
                dW = torch.empty(C, Cp, device=dev, dtype=F32)
                L.call('gpe_bn_bwd_from_G', G, Cp, db, W, W.stride(0), C, Cp, stp, sums, dW, Cp)
                coef_p, dgam_p, dbet_p = bn_bwd_coef(sums, 1, stp, Cp, M)
                L.call('gpe_edge_mlp_bwd', dz, dz.stride(0), 0, None, 0, None, 1, M, 1, C, Cp,
                       pack_weight(W, transpose=True), coef_p, prev, prev.stride(0), None, 0)
                grads[4 * l], grads[4 * l + 1] = dW, db
                grads[4 * (l - 1) + 2], grads[4 * (l - 1) + 3] = dgam_p, dbet_p
            else:
                K0 = x.shape[1]
                dW, db = redgemm_raw((dz, dz.stride(0), 0, 0), _rows2d(x), M, C, K0)
                grads[0], grads[1] = dW, db
        gx = None
        if ctx.needs_input_grad[0]:
            dz0, W0 = acts[0], params[0]
            gx = torch.empty(M, x.shape[1], device=dev, dtype=F32)
            linear_raw((dz0, dz0.stride(0), 0, 0), pack_weight(W0, transpose=True), None, M, x.shape[1],
                       W0.shape[0], _rows2d(gx))
        return (gx, None, None, None, None, *grads, *([None] * (3 * n)))


def dense_mlp(x, mlp, training):
    """mlp: the nn.Sequential parameter container built by net_blocks.MLP."""
    blocks = [mlp[i] for i in range(len(mlp))]
    params, bufs = [], []
    for blk in blocks:
        params += [blk[0].weight, blk[0].bias, blk[2].weight, blk[2].bias]
        bufs += [blk[2].running_mean, blk[2].running_var, blk[2].num_batches_tracked]
    x = x if x.stride(1) == 1 else x.contiguous()
    return DenseMLPFn.apply(x, training, blocks[0][2].eps, blocks[0][2].momentum, len(blocks), *params, *bufs)


class SparsemaxFn(torch.autograd.Function):
    """sparsemax.Sparsemax(dim=1) (nn/nets.py:225)."""

    @staticmethod
    def forward(ctx, z):
        _dev_check(z)
        z = z.contiguous()
        M, W = z.shape
        out = torch.empty_like(z)
        L.call('gpe_sparsemax_fwd', z, W, M, W, out, W)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        out, = ctx.saved_tensors
        g = g.contiguous()
        M, W = out.shape
        gz = torch.empty_like(out)
        L.call('gpe_sparsemax_bwd', out, W, g, W, M, W, gz, W)
        return gz


class AttentionPoolFn(torch.autograd.Function):
    """pooled[b, p, :] = mean_n w[b, n, p] * feat[b, n, :]  — the reference's 23-iteration loop of
    `w[:, p] * features -> global_mean_pool` (nn/nets.py:263-276) as one reduce-GEMM [P x N].[N x C] per cloud."""

    @staticmethod
    def forward(ctx, w, feat, B, N):
        _dev_check(w)
        dev = w.device
        P, C = w.shape[1], feat.shape[1]
        pooled = torch.empty(B, P, C, device=dev, dtype=F32)
        for b in range(B):
            G = pooled[b]
            ws = torch.empty(L.query('gpe_redgemm_ws', P, C), device=dev, dtype=F32)
            L.call('gpe_redgemm', w[b * N:(b + 1) * N], w.stride(0), 0, 0, feat[b * N:(b + 1) * N], feat.stride(0),
                   0, 0, None, N, P, C, G, C, None, ws, 0)
        L.call('gpe_scale', pooled, 1.0 / N, pooled, pooled.numel())
        ctx.dims = (B, N, P, C)
        ctx.save_for_backward(w, feat)
        return pooled.view(B * P, C)

    @staticmethod
    def backward(ctx, g):
        w, feat = ctx.saved_tensors
        B, N, P, C = ctx.dims
        dev = w.device
        g = g.contiguous().view(B, P, C)
        gs = torch.empty_like(g)
        L.call('gpe_scale', g, 1.0 / N, gs, g.numel())
        gw = torch.empty(B * N, P, device=dev, dtype=F32)
        gf = torch.empty(B * N, C, device=dev, dtype=F32)
        for b in range(B):
            sl = slice(b * N, (b + 1) * N)
            # dw[n, p] = sum_c feat[n, c] * gs[b, p, c]
            linear_raw(_rows2d(feat[sl]), pack_weight(gs[b]), None, N, P, C, _rows2d(gw[sl]))
            # dfeat[n, c] = sum_p w[n, p] * gs[b, p, c]
            linear_raw(_rows2d(w[sl]), pack_weight(gs[b], transpose=True), None, N, C, P, _rows2d(gf[sl]))
        return gw, gf, None, None
