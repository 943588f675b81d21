"""Drop-in for the reference's nn/net_blocks.py (classes resolved by name from the YAML config,
nn/nets.py:100,106,116).  The torch modules below only HOLD parameters/buffers under the reference's names
(`conv_layers.{i}.nn.{j}.{0,2}.*`, `lstm.weight_ih_l0`, `lin.*` ...); every forward/backward runs the HIP
kernels through ops.py."""
import torch
import torch.nn as nn

from . import ops


def MLP(channels, batch_norm=True):
    """Parameter container with the reference layout [Linear -> ReLU -> BatchNorm1d] x n, BN after ReLU and on
    the last layer too (nn/net_blocks.py:43-47; `batch_norm` is ignored there as well)."""
    return nn.Sequential(*[
        nn.Sequential(nn.Linear(channels[i - 1], channels[i]), nn.ReLU(), nn.BatchNorm1d(channels[i]))
        for i in range(1, len(channels))])


class DynamicEdgeConv(nn.Module):
    """Holder with PyG's attribute name `.nn` (state-dict compatibility); the arithmetic is ops.EdgeConvFn.
    aggr: 'max' (shipped), 'mean', 'add'; the edge MLP may have any number of blocks >= 2 (EConv_hidden_depth >= 1)."""

    def __init__(self, nn_module, k, aggr='max'):
        super().__init__()
        if aggr not in ('max', 'mean', 'add'):
            raise ValueError("unsupported aggregation '%s' (PyG MessagePassing accepts max / mean / add here)" % aggr)
        if len(nn_module) < 2:
            raise NotImplementedError('EConv_hidden_depth must be >= 1 (edge MLP of at least 2 blocks), got %d block(s)'
                                      % len(nn_module))
        self.nn = nn_module
        self.k = k
        self.aggr = aggr
        self.last_knn = None
        self.last_order = None                               # the locality order the last graph search worked in (int32 [B, N])
        self.half_act_guard = ops.HalfActGuard()             # f16x3: watches the fp16-stored activation (ops.set_half_act_guard)

    def register_packs(self, plan):
        blocks = [self.nn[i] for i in range(len(self.nn))]
        plan.add_edge_first(blocks[0][0].weight, blocks[0][0].bias)
        for b in blocks[1:]:
            plan.add_linear(b[0].weight, fwd=False, bwd=True)     # forward packs carry the folded BatchNorm scale

    def forward(self, x, n_clouds, n_points, order=None):
        """order: the previous EdgeConv layer's `last_order` (a speed hint for this layer's graph search; results do not depend on it)"""
        nb = len(self.nn)
        blocks = [self.nn[i] for i in range(nb)]
        lin = [b[0] for b in blocks]
        bn = [b[2] for b in blocks]
        eps, mom = bn[0].eps, bn[0].momentum
        args = []
        for l, b in zip(lin, bn):
            args += [l.weight, l.bias, b.weight, b.bias]
        for b in bn:
            args += [b.running_mean, b.running_var, b.num_batches_tracked]
        H0 = lin[0].weight.shape[0]
        if H0 % 4 or H0 > 256:
            # a first-block width the fused P|Q kernels do not take: the general (explicit-message) formulation
            out, idx = ops.edge_conv_general(x, n_clouds, n_points, self.k, self.training, eps, mom, nb, self.aggr, args)
            self.last_order = None
        else:
            out, idx, self.last_order = ops.EdgeConvFn.apply(x, n_clouds, n_points, self.k, self.training, eps, mom, nb, self.aggr,
                                                             self.half_act_guard, order, *args)
        self.last_knn = idx
        return out


class EdgeConvFeatures(nn.Module):
    """nn/net_blocks.py:93-191: defaults, 2x DynamicEdgeConv, optional skip concat, global pool, Linear.
    forward(positions [B,N,3], global_pool=True) -> (encoding [B,out] | None, per-point [B*N,F(+3)], batch)."""

    def __init__(self, out_size, config={}):
        super().__init__()
        self.config = {
            'conv_depth': 2, 'k_neighbors': 5, 'EConv_hidden': 200, 'EConv_hidden_depth': 2,
            'EConv_feature': 112, 'EConv_aggr': 'max', 'global_pool': 'mean',
            'skip_connections': False, 'graph_pooling': False, 'pool_ratio': 0.1}
        self.config.update(config)
        if self.config['graph_pooling']:
            raise NotImplementedError('graph_pooling (DynamicASAPool) is outside the accelerated path; '
                                      'no shipped config enables it')
        depth = self.config['conv_depth']
        feat = [self.config['EConv_feature']] * depth
        hid = [self.config['EConv_hidden']] * depth
        mlp_depth = self.config['EConv_hidden_depth']
        self.conv_layers = nn.ModuleList()
        self.conv_layers.append(DynamicEdgeConv(
            MLP([2 * 3] + [hid[0]] * mlp_depth + [feat[0]]),
            k=self.config['k_neighbors'], aggr=self.config['EConv_aggr']))
        for c in range(1, depth):
            self.conv_layers.append(DynamicEdgeConv(
                MLP([2 * feat[c - 1]] + [hid[c]] * mlp_depth + [feat[c]]),
                k=self.config['k_neighbors'], aggr=self.config['EConv_aggr']))
        if self.config['global_pool'] == 'max':
            self.global_pool = ops.segment_max
        elif self.config['global_pool'] == 'mean':
            self.global_pool = ops.segment_mean
        elif self.config['global_pool'] == 'add':
            self.global_pool = ops.segment_add
        else:
            raise ValueError('{} pooling is not supported'.format(self.config['global_pool']))
        out_features = self.config['EConv_feature'] + 3 if self.config['skip_connections'] \
            else self.config['EConv_feature']
        self.lin = nn.Linear(out_features, out_size)

    def register_packs(self, plan):
        for conv in self.conv_layers:
            conv.register_packs(plan)
        plan.add_linear(self.lin.weight)

    def forward(self, positions, global_pool=True, want_batch=True):
        B, N = positions.size(0), positions.size(1)
        pos_flat = positions.reshape(-1, positions.size(-1))
        if pos_flat.dtype != torch.float32:
            pos_flat = pos_flat.float()
        pos_flat = pos_flat.contiguous()
        # batch vector of the reference (nn/net_blocks.py:165-167); the kernels only need (B, N)
        # (want_batch=False: a caller that only takes the pooled encoding — nets.GarmentFullPattern3D — saves the two launches)
        batch = torch.arange(B, device=positions.device).repeat_interleave(N) if want_batch else None
        out = pos_flat
        order = None                                         # layer l + 1 searches its graph in layer l's locality order
        for conv in self.conv_layers:
            out = conv(out, B, N, order=order)
            order = conv.last_order
        if self.config['skip_connections']:
            out = torch.cat([out, pos_flat], dim=-1)
        if global_pool:
            pooled = self.global_pool(out.contiguous() if out.stride(1) != 1 else out, B, N)
            return ops.linear(pooled, self.lin.weight, self.lin.bias), out, batch
        return None, out, batch


class _PointConvHolder(nn.Module):
    """parameter container with PyG PointConv's attribute names (state-dict keys `conv.local_nn.*`)."""

    def __init__(self, local_nn):
        super().__init__()
        self.local_nn = local_nn
        self.global_nn = None


class _SetAbstractionModule(nn.Module):
    """nn/net_blocks.py:10-27: fps -> ball query (<= 25 neighbours) -> PointConv(MLP) with max aggregation, with PyG's defaults:
    fps(random_start=True) — the start points are drawn from torch's CPU generator (ops.fps_start; `fps_start` overrides) — and
    PointConv(add_self_loops=True), whose re-indexing of the bipartite edge list is restated in csrc/gpe_pointnet.hip."""

    def __init__(self, ratio, conv_radius, per_point_nn):
        super().__init__()
        self.ratio = ratio
        self.radius = conv_radius
        self.conv = _PointConvHolder(per_point_nn)
        self.fps_start = None              # int32 [B] tensor: fixed start points instead of the random draw
        self.add_self_loops = True         # PyG PointNetConv's default
        self.last = {}

    def forward(self, features, pos_flat, B, N):
        import math
        M = int(math.ceil(self.ratio * N))
        start = self.fps_start.to(pos_flat.device) if self.fps_start is not None else ops.fps_start(B, N, pos_flat.device)
        idx = ops.fps(pos_flat, B, N, M, start)
        nbr, cnt = ops.radius_neighbors(pos_flat, idx, B, N, self.radius, 25)
        ecnt, drop = ops.pointconv_self_loops(nbr, cnt, B, N, M) if self.add_self_loops else (cnt, None)
        off = torch.zeros(B * M + 1, device=pos_flat.device, dtype=torch.int64)
        torch.cumsum(ecnt, 0, out=off[1:])
        n_edges = int(off[-1].item())          # ragged edge list: the one host sync of this block (sizes the MLP rows)
        msg, seg = ops.ball_messages(pos_flat, features, idx, nbr, off, n_edges, B, N, drop)
        h = ops.dense_mlp(msg, self.conv.local_nn, self.training)
        out = ops.RaggedMaxFn.apply(h, off, seg, B * M)
        gidx = (idx.long() + (torch.arange(B, device=idx.device) * N)[:, None]).view(-1)
        self.last = {'idx': idx, 'nbr': nbr, 'cnt': cnt, 'edge_cnt': ecnt, 'drop': drop, 'seg': seg}
        return out, pos_flat[gidx], M


class _GlobalSetAbstractionModule(nn.Module):
    """nn/net_blocks.py:30-42: MLP on [features | pos] -> global max pool."""

    def __init__(self, per_point_net):
        super().__init__()
        self.nn = per_point_net

    def forward(self, features, pos, B, M):
        feats = torch.cat([features, pos], dim=1) if features is not None else pos
        return ops.segment_max(ops.dense_mlp(feats, self.nn, self.training), B, M)


class PointNetPlusPlus(nn.Module):
    """nn/net_blocks.py:50-88 (one set-abstraction level + the global level, as in the reference).  forward(positions [B,N,3])
    -> [B, out_size].  PyG's conventions (random fps start from torch's generator, PointConv's add_self_loops re-indexing) are
    restated in csrc/gpe_pointnet.hip; the two implementation-defined choices left (argmax ties, which neighbours survive the
    cap of 25) are fixed there for oracle and kernels alike."""

    def __init__(self, out_size, config={}):
        super().__init__()
        self.config = {'r1': 0.3, 'r2': 0.4, 'r3': 5, 'r4': 7}
        self.config.update(config)
        H, F = self.config['EConv_hidden'], self.config['EConv_feature']
        self.sa1_module = _SetAbstractionModule(0.2, self.config['r1'], MLP([3, H, H, F]))
        self.sa_last_module = _GlobalSetAbstractionModule(MLP([3 + F, H, H, F]))
        self.lin = nn.Linear(F, out_size)

    def forward(self, positions):
        B, N = positions.size(0), positions.size(1)
        pos_flat = positions.reshape(-1, positions.size(-1)).float().contiguous()
        feats, cpos, M = self.sa1_module(None, pos_flat, B, N)
        pooled = self.sa_last_module(feats, cpos, B, M)
        return ops.linear(pooled, self.lin.weight, self.lin.bias)


class _StateUploader:
    """Host-drawn tensors reach the GPU without draining the compute stream.

    The reference draws the LSTM start states on the CPU generator inside forward() (nn/net_blocks.py:391-392) and
    moves them with `.to(device)`.  From pageable memory that copy is stream-ordered AND blocks the host until it has
    run, i.e. until the whole encoder queued in front of it has finished — the GPU then idles while the host catches
    up (measured: 0.85 ms in front of each decoder, profiles/r01_e_gap_trace.md).  Here the draw lands in a pinned
    staging buffer, the copy runs on a side stream, and the compute stream only waits on its event."""
    _per_device = {}

    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device)
        self.slots = {}          # shape -> (pinned buffer, event of the last copy out of it)

    @classmethod
    def get(cls, device):
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        if key not in cls._per_device:
            cls._per_device[key] = cls(device)
        return cls._per_device[key]

    def staging(self, shape):
        if shape not in self.slots:
            self.slots[shape] = (torch.empty(shape, pin_memory=True), torch.cuda.Event())
        buf, ev = self.slots[shape]
        ev.synchronize()         # the previous copy out of this buffer is long done; makes the reuse safe
        return buf, ev

    def upload(self, buf, ev, device):
        compute = torch.cuda.current_stream(device)
        with torch.cuda.stream(self.stream):
            out = torch.empty(buf.shape, device=device)      # owned by the side stream's pool
            out.copy_(buf, non_blocking=True)
            ev.record(self.stream)
        compute.wait_event(ev)
        out.record_stream(compute)
        return out


def _init_tenzor(*shape, device='cpu', init_type=''):
    """nn/net_blocks.py:302-315 — drawn on the CPU generator, then moved (keeps the reference's RNG stream)."""
    device = torch.device(device)
    zeros = not init_type or len(shape) == 1
    if not zeros and 'kaiming_normal' not in init_type:
        raise NotImplementedError('{} tenzor initialization is not implemented'.format(init_type))
    if zeros:
        return torch.zeros(shape, device=device)
    if device.type != 'cuda':
        t = torch.empty(shape)
        nn.init.kaiming_normal_(t)
        return t.to(device)
    if ops.HOST_DRAWN is not None:
        # a step driven by graph.StepGraph: a static device buffer, refilled from the same generator in front of every replay
        return ops.HOST_DRAWN.get(tuple(shape), nn.init.kaiming_normal_)
    up = _StateUploader.get(device)
    buf, ev = up.staging(tuple(shape))
    nn.init.kaiming_normal_(buf)                             # same generator, same element order as torch.empty(shape)
    return up.upload(buf, ev, device)


def _init_weights(module, init_type=''):
    """nn/net_blocks.py:318-333."""
    if not init_type:
        return
    for name, param in module.named_parameters():
        if 'weight' in name:
            if 'kaiming_normal' in init_type:
                if len(param.shape) > 1:
                    nn.init.kaiming_normal_(param)
            else:
                raise NotImplementedError('{} weight initialization is not implemented'.format(init_type))


def _rnn_params(rnn, n_layers):
    ps = []
    for l in range(n_layers):
        ps += [getattr(rnn, 'weight_ih_l%d' % l), getattr(rnn, 'weight_hh_l%d' % l),
               getattr(rnn, 'bias_ih_l%d' % l), getattr(rnn, 'bias_hh_l%d' % l)]
    return ps


def _register_rnn_packs(plan, rnn, n_layers, hidden, gates):
    for l in range(n_layers):
        w_ih, w_hh = getattr(rnn, 'weight_ih_l%d' % l), getattr(rnn, 'weight_hh_l%d' % l)
        plan.add_linear(w_ih)
        plan.add_gates(w_hh, hidden, gates)
        plan.add_linear(w_hh, fwd=False, bwd=True)
        b_ih, b_hh = getattr(rnn, 'bias_ih_l%d' % l), getattr(rnn, 'bias_hh_l%d' % l)
        if l > 0:
            plan.add_gates(w_ih, hidden, gates)        # layers > 0 project h_{l-1,t} inside the fused cell launch
        if gates == 4:
            plan.add_bias_sum(b_ih, b_hh)
        else:
            plan.add_gru_bias(b_ih, b_hh, hidden)


class LSTMEncoderModule(nn.Module):
    """nn/net_blocks.py:336-360: sequence -> final hidden state of the last LSTM layer."""

    def __init__(self, elem_len, encoding_size, n_layers, dropout=0, custom_init='kaiming_normal'):
        super().__init__()
        self.dropout = float(dropout)
        self.custom_init = custom_init
        self.n_layers = n_layers
        self.encoding_size = encoding_size
        self.lstm = nn.LSTM(elem_len, encoding_size, n_layers, dropout=dropout, batch_first=True)
        _init_weights(self.lstm, init_type=custom_init)

    def register_packs(self, plan):
        _register_rnn_packs(plan, self.lstm, self.n_layers, self.encoding_size, 4)

    def forward(self, batch_sequence):
        device = batch_sequence.device
        bs = batch_sequence.size(0)
        h0 = _init_tenzor(self.n_layers, bs, self.encoding_size, device=device, init_type=self.custom_init)
        c0 = _init_tenzor(self.n_layers, bs, self.encoding_size, device=device, init_type=self.custom_init)
        seq = batch_sequence if batch_sequence.stride(-1) == 1 else batch_sequence.contiguous()
        _, hN, _ = ops.rnn_stack(seq, h0, c0, seq.size(1), self.n_layers, 'lstm', _rnn_params(self.lstm, self.n_layers),
                                 want_state=True, dropout=self.dropout, training=self.training, h0_bounded=True)
        return hN[-1]


class LSTMDecoderModule(nn.Module):
    """nn/net_blocks.py:363-402.  `self.lstm` is a torch.nn.LSTM used ONLY as the parameter container
    (weight_ih_l*, weight_hh_l*, bias_ih_l*, bias_hh_l*); the recurrence runs in ops.RNNStackFn."""

    def __init__(self, encoding_size, hidden_size, out_elem_size, n_layers, dropout=0,
                 custom_init='kaiming_normal', **kwargs):
        super().__init__()
        self.dropout = float(dropout)
        self.custom_init = custom_init
        self.n_layers = n_layers
        self.encoding_size = encoding_size
        self.hidden_size = hidden_size
        self.out_elem_size = out_elem_size
        self.lstm = nn.LSTM(encoding_size, hidden_size, n_layers, dropout=dropout, batch_first=True)
        self.lin = nn.Linear(hidden_size, out_elem_size)
        _init_weights(self.lstm, init_type=custom_init)
        self.last_states = None

    def register_packs(self, plan):
        _register_rnn_packs(plan, self.lstm, self.n_layers, self.hidden_size, 4)
        plan.add_linear(self.lin.weight)

    def forward(self, batch_enc, out_len):
        device = batch_enc.device
        bs = batch_enc.size(0)
        # hidden first, then cell: the reference's draw order (nn/net_blocks.py:391-392)
        h0 = _init_tenzor(self.n_layers, bs, self.hidden_size, device=device, init_type=self.custom_init)
        c0 = _init_tenzor(self.n_layers, bs, self.hidden_size, device=device, init_type=self.custom_init)
        self.last_states = (h0, c0)
        enc = batch_enc if batch_enc.is_contiguous() else batch_enc.contiguous()
        top, _, _ = ops.rnn_stack(enc, h0, c0, out_len, self.n_layers, 'lstm', _rnn_params(self.lstm, self.n_layers),
                                  dropout=self.dropout, training=self.training, h0_bounded=True)
        return ops.linear(top, self.lin.weight, self.lin.bias).view(bs, out_len, -1)


class LSTMDoubleReverseDecoderModule(nn.Module):
    """nn/net_blocks.py:405-454: decode the sequence in reverse order, then refine it with a second LSTM that reads
    [flipped first pass | encoding] and starts from the first LSTM's final state."""

    def __init__(self, encoding_size, hidden_size, out_elem_size, n_layers, dropout=0,
                 custom_init='kaiming_normal', **kwargs):
        super().__init__()
        self.dropout = float(dropout)
        self.custom_init = custom_init
        self.n_layers = n_layers
        self.encoding_size = encoding_size
        self.hidden_size = hidden_size
        self.out_elem_size = out_elem_size
        self.lstm_reverse = nn.LSTM(encoding_size, hidden_size, n_layers, dropout=dropout, batch_first=True)
        self.lstm_forward = nn.LSTM(hidden_size + encoding_size, hidden_size, n_layers, dropout=dropout,
                                    batch_first=True)
        self.lin = nn.Linear(hidden_size, out_elem_size)
        _init_weights(self.lstm_reverse, init_type=custom_init)
        _init_weights(self.lstm_forward, init_type=custom_init)
        self.last_states = None

    def register_packs(self, plan):
        _register_rnn_packs(plan, self.lstm_reverse, self.n_layers, self.hidden_size, 4)
        _register_rnn_packs(plan, self.lstm_forward, self.n_layers, self.hidden_size, 4)
        plan.add_linear(self.lin.weight)

    def forward(self, batch_enc, out_len):
        device = batch_enc.device
        bs = batch_enc.size(0)
        h0 = _init_tenzor(self.n_layers, bs, self.hidden_size, device=device, init_type=self.custom_init)
        c0 = _init_tenzor(self.n_layers, bs, self.hidden_size, device=device, init_type=self.custom_init)
        self.last_states = (h0, c0)
        enc = batch_enc if batch_enc.is_contiguous() else batch_enc.contiguous()
        out, hN, cN = ops.rnn_stack(enc, h0, c0, out_len, self.n_layers, 'lstm',
                                    _rnn_params(self.lstm_reverse, self.n_layers), want_state=True,
                                    dropout=self.dropout, training=self.training, h0_bounded=True)
        dec_input = enc.unsqueeze(1).expand(-1, out_len, -1)
        seq = torch.cat([torch.flip(out, [1]), dec_input], -1)            # skip connection with the original input
        top, _, _ = ops.rnn_stack(seq, hN, cN, out_len, self.n_layers, 'lstm',
                                  _rnn_params(self.lstm_forward, self.n_layers), dropout=self.dropout,
                                  training=self.training, h0_bounded=True)      # (states of the first LSTM: |h| < 1)
        return ops.linear(top, self.lin.weight, self.lin.bias).view(bs, out_len, -1)


class GRUDecoderModule(nn.Module):
    """nn/net_blocks.py:457-497; `self.recurrent_cell` is the parameter container, ops.RNNStackFn the arithmetic."""

    def __init__(self, encoding_size, hidden_size, out_elem_size, n_layers, dropout=0,
                 custom_init='kaiming_normal', **kwargs):
        super().__init__()
        self.dropout = float(dropout)
        self.custom_init = custom_init
        self.n_layers = n_layers
        self.encoding_size = encoding_size
        self.hidden_size = hidden_size
        self.out_elem_size = out_elem_size
        self.recurrent_cell = nn.GRU(encoding_size, hidden_size, n_layers, dropout=dropout, batch_first=True)
        self.lin = nn.Linear(hidden_size, out_elem_size)
        _init_weights(self.recurrent_cell, init_type=custom_init)
        self.last_states = None

    def register_packs(self, plan):
        _register_rnn_packs(plan, self.recurrent_cell, self.n_layers, self.hidden_size, 3)
        plan.add_linear(self.lin.weight)

    def forward(self, batch_enc, out_len):
        device = batch_enc.device
        bs = batch_enc.size(0)
        h0 = _init_tenzor(self.n_layers, bs, self.hidden_size, device=device, init_type=self.custom_init)
        self.last_states = (h0,)
        enc = batch_enc if batch_enc.is_contiguous() else batch_enc.contiguous()
        top, _, _ = ops.rnn_stack(enc, h0, None, out_len, self.n_layers, 'gru',
                                  _rnn_params(self.recurrent_cell, self.n_layers), dropout=self.dropout,
                                  training=self.training, h0_bounded=True)
        return ops.linear(top, self.lin.weight, self.lin.bias).view(bs, out_len, -1)


class MLPDecoder(nn.Module):
    """nn/net_blocks.py:273-298: latent code -> MLP([enc, hid*out_len x n_layers, out_elem*out_len]) -> [B, out_len, out_elem].
    The arithmetic is ops.DenseMLPFn (wide layers run as several column blocks of the row GEMM)."""

    def __init__(self, encoding_size, hidden_size, out_elem_size, n_layers, out_len=1, dropout=0,
                 custom_init='kaiming_normal'):
        super().__init__()
        self.out_len = out_len
        self.mlp = MLP([encoding_size] + [hidden_size * out_len for _ in range(n_layers)] + [out_elem_size * out_len])
        _init_weights(self.mlp, init_type=custom_init)

    def register_packs(self, plan):
        blocks = [self.mlp[i] for i in range(len(self.mlp))]
        plan.add_linear(blocks[0][0].weight)                      # first block: nothing folded in
        for b in blocks[1:]:
            plan.add_linear(b[0].weight, fwd=False, bwd=True)

    def forward(self, batch_enc, *args):
        batch_size = batch_enc.size(0)
        out = ops.dense_mlp(batch_enc, self.mlp, self.training)
        return out.contiguous().view(batch_size, self.out_len, -1)


def _not_accelerated(name):
    class _Missing(nn.Module):
        def __init__(self, *a, **kw):
            raise NotImplementedError(
                '%s is selectable in the reference (nn/net_blocks.py) but used by no shipped config; '
                'it is a "next" row of the scope table (SURVEY.md §8f) and has no kernels yet' % name)
    _Missing.__name__ = name
    return _Missing


EdgeConvPoolingFeatures = _not_accelerated('EdgeConvPoolingFeatures')
DynamicASAPool = _not_accelerated('DynamicASAPool')
