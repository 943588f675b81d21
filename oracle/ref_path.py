"""ORACLE — TEST INFRASTRUCTURE ONLY (CPU, pure torch + oracle/knn_ref.c).

A restatement of the reference's point-cloud-encoder -> panel-sequence-decoder
path, used ONLY as the checker in tests/, in __graft_entry__.smoke() and as the
`cpu_baseline` leg of bench.py.  The product path (the HIP library behind
`garment-pattern-estimation_amd/`) never imports this module.

What it follows (paths relative to /root/reference):
  * nn/net_blocks.py:43-47     MLP = [Linear -> ReLU -> BatchNorm1d] x n (BN after ReLU, last layer too)
  * nn/net_blocks.py:93-191    EdgeConvFeatures (defaults, 2x DynamicEdgeConv, pool, Linear)
  * nn/net_blocks.py:302-333   _init_tenzor / _init_weights
  * nn/net_blocks.py:363-402   LSTMDecoderModule
  * nn/net_blocks.py:273-298,405-497   MLPDecoder, LSTMDoubleReverseDecoderModule, GRUDecoderModule
  * nn/nets.py:11-184          BaseModule, GarmentFullPattern3D
  * nn/nets.py:187-299         GarmentSegmentPattern3D
  * nn/metrics/composed_loss.py:222-334,428-703, nn/metrics/losses.py:8-51  main loss terms (backward seed) and the
    panel-order / panel-origin matching of the ground truth in front of them
  * nn/nets.py:303-353         StitchOnEdge3DPairs (model)
  * nn/trainer.py:92-99        the fwd -> loss -> bwd step

Pinning status:
  * The torch-native parts (module construction order, config logic, init, LSTM
    decoders, output slicing, loss) are checked against the reference's OWN code
    imported in the build container (oracle/refgen/make_golden.py ->
    tests/golden/*.pt, tests/test_oracle_golden.py).
  * torch_geometric.DynamicEdgeConv / global_mean_pool, torch_cluster.knn and
    sparsemax.Sparsemax are third-party, un-vendored, version-unpinned
    (docs/Installation.md:46-48,67; requirements.txt:2).  Their arithmetic is
    restated here from the published definitions -> PARITY UNPINNED at those
    call sites (nn/net_blocks.py:127-135,145-150,184; nn/nets.py:225,272).

fp64 mode: `model.double()` + fp64 input gives a high-precision reference; the
random LSTM states are always drawn as fp32 (like the reference) and then cast.
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _oracle_lib():
    """ctypes handle on oracle/libgpe_oracle.so (built by oracle/Makefile or __graft_entry__.build())."""
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'libgpe_oracle.so')
        if not os.path.exists(path):
            raise RuntimeError(
                'oracle/libgpe_oracle.so is missing: run `make -C oracle` or __graft_entry__.build()')
        lib = ctypes.CDLL(path)
        lib.gpe_oracle_knn.restype = ctypes.c_int
        lib.gpe_oracle_knn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_void_p]
        lib.gpe_oracle_sqdist.restype = ctypes.c_int
        lib.gpe_oracle_sqdist.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _LIB = lib
    return _LIB


# ---------------------------------------------------------------------------------------------
# third-party ops restated (parity unpinned, see module docstring)
# ---------------------------------------------------------------------------------------------
# 'c'     : oracle/knn_ref.c — THE definition (fp32 fmaf chain, ties -> lower index); what every parity test uses.
# 'torch' : cdist + topk in torch — same neighbours except at fp32 near-ties; used ONLY by bench.py's cpu_baseline leg,
#           where the naive scalar C loop (O(N^2 C), one thread) would misrepresent what a CPU can do.
KNN_IMPL = 'c'


def _knn_torch(x, n_clouds, k):
    total, C = x.shape
    N = total // n_clouds
    xb = x.detach().to(torch.float32).view(n_clouds, N, C)
    d = torch.cdist(xb, xb)
    return d.topk(k, dim=-1, largest=False, sorted=True).indices.reshape(total, k)


def knn_local(x, n_clouds, k):
    """Per-cloud exact kNN incl. self; x: [B*N, C] (any float dtype, evaluated in fp32).
    Returns LongTensor [B*N, k] of indices LOCAL to each cloud, ascending distance, ties -> lower index.
    Restates torch_cluster.knn as used by DynamicEdgeConv (call sites nn/net_blocks.py:127-135)."""
    if KNN_IMPL == 'torch':
        return _knn_torch(x, n_clouds, k)
    total, C = x.shape
    N = total // n_clouds
    xf = np.ascontiguousarray(x.detach().to(torch.float32).cpu().numpy())
    out = np.empty((total, k), dtype=np.int32)
    rc = _oracle_lib().gpe_oracle_knn(xf.ctypes.data, n_clouds, N, C, k, out.ctypes.data)
    if rc != 0:
        raise ValueError('gpe_oracle_knn: bad arguments (B=%d N=%d C=%d k=%d)' % (n_clouds, N, C, k))
    return torch.from_numpy(out.astype(np.int64))


def sqdist_one_cloud(x):
    """[N,C] -> [N,N] fp32 squared distances with the oracle's arithmetic (debug aid for tie analysis)."""
    N, C = x.shape
    xf = np.ascontiguousarray(x.detach().to(torch.float32).cpu().numpy())
    d = np.empty((N, N), dtype=np.float32)
    _oracle_lib().gpe_oracle_sqdist(xf.ctypes.data, N, C, d.ctypes.data)
    return torch.from_numpy(d)


def global_pool(x, batch, size, kind):
    """torch_geometric.nn.global_{mean,max,add}_pool restated: segment reduce by `batch`."""
    C = x.shape[1]
    if kind == 'max':
        out = torch.full((size, C), float('-inf'), dtype=x.dtype)
        return out.scatter_reduce(0, batch[:, None].expand(-1, C), x, reduce='amax', include_self=True)
    out = torch.zeros((size, C), dtype=x.dtype).index_add(0, batch, x)
    if kind == 'add':
        return out
    counts = torch.zeros(size, dtype=x.dtype).index_add(0, batch, torch.ones_like(batch, dtype=x.dtype))
    return out / counts.clamp(min=1)[:, None]


def global_mean_pool(x, batch, size=None):
    return global_pool(x, batch, int(batch.max()) + 1 if size is None else size, 'mean')


def global_max_pool(x, batch, size=None):
    return global_pool(x, batch, int(batch.max()) + 1 if size is None else size, 'max')


def global_add_pool(x, batch, size=None):
    return global_pool(x, batch, int(batch.max()) + 1 if size is None else size, 'add')


class DynamicEdgeConv(nn.Module):
    """torch_geometric.nn.DynamicEdgeConv restated (DGCNN EdgeConv on a kNN graph rebuilt from the
    current features): out_i = aggr_{j in kNN(i)} nn(cat[x_i, x_j - x_i]).  `nn` attribute name kept so
    state-dict keys read `conv_layers.{i}.nn....` like the reference's."""

    def __init__(self, nn_module, k, aggr='max'):
        super().__init__()
        self.nn = nn_module
        self.k = k
        self.aggr = aggr
        self.last_knn = None      # local indices used by the last forward (for stage-wise parity)
        self.knn_override = None  # inject a graph (LongTensor [B*N,k], local) instead of searching
        # inject the winners of the max aggregation: (argmax, argmin) slots of the pre-BatchNorm activation, LongTensors
        # [B*N, F] — the same role as knn_override at the other discontinuity of the layer (tests/relu_align.py).
        # `argsel_gap` then holds the largest distance between the true maximum and the injected winner's message.
        self.argsel_override = None
        self.argsel_gap = 0.0

    def forward(self, x, batch):
        total = x.shape[0]
        n_clouds = int(batch.max()) + 1
        N = total // n_clouds
        if self.knn_override is not None:
            local = self.knn_override
        else:
            local = knn_local(x, n_clouds, self.k)
        self.last_knn = local
        glob = local + (torch.arange(total) // N * N)[:, None]
        x_i = x[:, None, :].expand(-1, self.k, -1)
        x_j = x[glob]
        msg = self.nn(torch.cat([x_i, x_j - x_i], dim=-1).reshape(total * self.k, -1))
        msg = msg.view(total, self.k, -1)
        if self.aggr == 'max':
            if self.argsel_override is not None:
                amx, amn = self.argsel_override
                gamma = self.nn[-1][2].weight.detach()           # BN after ReLU: max_j(s a_j + t) = s (s >= 0 ? max : min) a + t
                sel = torch.where(gamma >= 0, amx, amn)
                out = msg.gather(1, sel[:, None, :]).squeeze(1)
                self.argsel_gap = (msg.max(dim=1).values - out).abs().max().item()
                return out
            return msg.max(dim=1).values
        if self.aggr == 'mean':
            return msg.mean(dim=1)
        if self.aggr == 'add':
            return msg.sum(dim=1)
        raise ValueError('unsupported aggregation {}'.format(self.aggr))


class _SparsemaxFn(torch.autograd.Function):
    """sparsemax (Martins & Astudillo 2016) over the last dim: Euclidean projection onto the simplex;
    backward nz * (g - sum(g*nz)/|nz|).  Restates sparsemax.Sparsemax (nn/nets.py:3,225)."""

    @staticmethod
    def forward(ctx, z):
        zs, _ = torch.sort(z, dim=-1, descending=True)
        rng = torch.arange(1, z.shape[-1] + 1, dtype=z.dtype)
        cs = zs.cumsum(-1)
        support = (1 + rng * zs) > cs
        ksup = support.to(z.dtype).sum(-1, keepdim=True)
        tau = (torch.gather(cs, -1, ksup.long() - 1) - 1) / ksup
        out = torch.clamp(z - tau, min=0)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        out, = ctx.saved_tensors
        nz = (out > 0).to(g.dtype)
        return nz * (g - (g * nz).sum(-1, keepdim=True) / nz.sum(-1, keepdim=True))


class Sparsemax(nn.Module):
    def __init__(self, dim=-1):
        super().__init__()
        self.dim = dim

    def forward(self, z):
        z = z.transpose(self.dim, -1)
        return _SparsemaxFn.apply(z).transpose(self.dim, -1)


class _SparsemaxLossFn(torch.autograd.Function):
    """entmax.SparsemaxLoss's function (deep-spin/entmax, third-party and un-vendored; call site nn/metrics/composed_loss.py:4,
    196,329): the sparsemax Fenchel-Young loss  L(x, t) = (1 - |p|^2)/2 + <p - e_t, x>,  p = sparsemax(x);  dL/dx = p - e_t
    (Martins & Astudillo 2016, eq. 19-20; Blondel et al. 2019).  PARITY UNPINNED against entmax itself (absent here)."""

    @staticmethod
    def forward(ctx, x, target):
        p = _SparsemaxFn.forward(ctx, x).clone()
        loss = (1 - (p ** 2).sum(dim=1)) / 2
        p.scatter_add_(1, target.unsqueeze(1), torch.full_like(p, -1))
        loss = loss + (p * x).sum(dim=1)
        ctx.save_for_backward(p)
        return loss

    @staticmethod
    def backward(ctx, g):
        p, = ctx.saved_tensors
        return g.unsqueeze(1) * p, None


class SparsemaxLoss(nn.Module):
    """entmax.SparsemaxLoss() with its defaults (ignore_index=-100: nothing ignored; reduction='elementwise_mean': the sum of
    the row losses over the number of rows)."""

    def forward(self, x, target):
        return _SparsemaxLossFn.apply(x, target.long()).sum() / float(target.shape[0])


# ---- PointNet++ pieces (torch_geometric.nn.fps / radius / PointConv; call sites nn/net_blocks.py:16-24) ---------------
def fps_start(batch_sizes, random_start=True):
    """Start point of the farthest-point sampling of every cloud (LOCAL index).  torch_cluster.fps(random_start=True) — PyG's
    default, and what nn/net_blocks.py:16 calls — draws it at random; the one upstream variant driven by torch's generator is the
    device kernel's recipe `(torch.rand(batch_size) * deg).long()`, restated here on torch's default CPU generator (the stream the
    reference's CPU run consumes, like the random LSTM states): ONE torch.rand(B) draw per fps call."""
    n = torch.as_tensor(batch_sizes, dtype=torch.float32)
    if not random_start:
        return torch.zeros(len(batch_sizes), dtype=torch.long)
    start = (torch.rand(len(batch_sizes)) * n).long()
    return torch.minimum(start, torch.as_tensor(batch_sizes, dtype=torch.long) - 1)


def fps(pos, batch, ratio, random_start=True, start=None):
    """Farthest point sampling per cloud -> GLOBAL indices, clouds in order, selection order inside a cloud.
    `start` (LOCAL index per cloud) defaults to fps_start(...): PyG's random start, drawn from torch's CPU generator.  Still
    open upstream and fixed here: distances are the fp32 fma chain of (a-b)^2 over the coordinates, argmax ties -> lower index."""
    B = int(batch.max()) + 1
    out = []
    p32 = pos.detach().to(torch.float32)
    groups = [torch.nonzero(batch == b).view(-1) for b in range(B)]
    if start is None:
        start = fps_start([g.numel() for g in groups], random_start)
    for b in range(B):
        ids = groups[b]
        P = p32[ids].numpy().astype(np.float32)
        n = P.shape[0]
        m = int(np.ceil(ratio * n))
        sel = [int(start[b])]
        mind = np.full(n, np.inf, dtype=np.float32)
        for _ in range(1, m):
            d = np.zeros(n, dtype=np.float32)
            for c in range(P.shape[1]):
                diff = (P[:, c] - P[sel[-1], c]).astype(np.float32)
                d = (diff.astype(np.float64) * diff.astype(np.float64) + d.astype(np.float64)).astype(np.float32)   # fmaf
            mind = np.minimum(mind, d)
            sel.append(int(np.argmax(mind)))              # first maximum
        out.append(ids[torch.tensor(sel)])
    return torch.cat(out)


def radius(x, y, r, batch_x, batch_y, max_num_neighbors=32):
    """torch_cluster.radius: for every row of y the points of x (same batch entry) within distance r.  Returns
    (row = index into y, col = index into x).  Open upstream (which neighbours survive the cap): fixed here to the FIRST
    max_num_neighbors in ascending x index with squared distance <= r^2 (fp32 fma chain)."""
    x32, y32 = x.detach().to(torch.float32).numpy(), y.detach().to(torch.float32).numpy()
    rows, cols = [], []
    r2 = np.float32(r) * np.float32(r)
    bx, by = batch_x.numpy(), batch_y.numpy()
    for i in range(y32.shape[0]):
        cand = np.nonzero(bx == by[i])[0]
        d = np.zeros(cand.shape[0], dtype=np.float32)
        for c in range(x32.shape[1]):
            diff = (x32[cand, c] - y32[i, c]).astype(np.float32)
            d = (diff.astype(np.float64) * diff.astype(np.float64) + d.astype(np.float64)).astype(np.float32)
        keep = cand[d <= r2][:max_num_neighbors]
        rows += [i] * len(keep)
        cols += keep.tolist()
    return torch.tensor(rows, dtype=torch.long), torch.tensor(cols, dtype=torch.long)


def pointconv_edges(edge_index, n_src, n_dst):
    """What PyG's PointNetConv.forward does to a Tensor edge_index under its default add_self_loops=True:
    `remove_self_loops(edge_index)` — drop every edge whose source INDEX equals its target INDEX, although in the bipartite
    call of nn/net_blocks.py:24 the two live in different index spaces (all points vs. sampled centroids) — then
    `add_self_loops(edge_index, num_nodes=min(n_src, n_dst))` (PyG 2.x; 1.6 / 1.7: num_nodes = n_dst — the same number here,
    centroids are a subset): append i -> i for i < num_nodes at the END of the list, i.e. centroid i also receives a message
    from the point with flat index i, whichever cloud that point belongs to."""
    src, dst = edge_index[0], edge_index[1]
    keep = src != dst
    n = min(n_src, n_dst)
    loop = torch.arange(n, dtype=src.dtype)
    return torch.stack([torch.cat([src[keep], loop]), torch.cat([dst[keep], loop])], dim=0)


class PointConv(nn.Module):
    """PointNet set-abstraction convolution (Qi et al. 2017; PyG PointNetConv): out_i = max_j local_nn([x_j, pos_j - pos_i])
    over the given edges (source j -> target i), with PyG's default `add_self_loops=True` re-indexing of the edge list
    (pointconv_edges above) — a checkpoint trained under PyG saw exactly those messages."""

    def __init__(self, local_nn=None, global_nn=None, add_self_loops=True):
        super().__init__()
        self.local_nn = local_nn
        self.global_nn = global_nn
        self.add_self_loops = add_self_loops

    def forward(self, x, pos, edge_index):
        pos_src, pos_dst = pos
        if self.add_self_loops:
            edge_index = pointconv_edges(edge_index, pos_src.shape[0], pos_dst.shape[0])
        self.last_edge_index = edge_index
        src, dst = edge_index[0], edge_index[1]
        msg = pos_src[src] - pos_dst[dst]
        if x is not None:
            msg = torch.cat([x[src], msg], dim=1)
        msg = self.local_nn(msg)
        out = torch.zeros(pos_dst.shape[0], msg.shape[1], dtype=msg.dtype)
        out = out.scatter_reduce(0, dst[:, None].expand(-1, msg.shape[1]), msg, reduce='amax', include_self=False)
        return out if self.global_nn is None else self.global_nn(out)


class _SetAbstractionModule(nn.Module):
    """nn/net_blocks.py:10-27."""

    def __init__(self, ratio, conv_radius, per_point_nn):
        super().__init__()
        self.ratio = ratio
        self.radius = conv_radius
        self.conv = PointConv(per_point_nn)
        self.trace = {}

    def forward(self, features, pos, batch):
        idx = fps(pos, batch, ratio=self.ratio)
        row, col = radius(pos, pos[idx], self.radius, batch, batch[idx], max_num_neighbors=25)
        self.trace = {'idx': idx, 'row': row, 'col': col}
        edge_index = torch.stack([col, row], dim=0)
        features = self.conv(features, (pos, pos[idx]), edge_index)
        return features, pos[idx], batch[idx]


class _GlobalSetAbstractionModule(nn.Module):
    """nn/net_blocks.py:30-42."""

    def __init__(self, per_point_net):
        super().__init__()
        self.nn = per_point_net

    def forward(self, features, pos, batch):
        features = torch.cat([features, pos], dim=1) if features is not None else pos
        features = self.nn(features)
        features = global_max_pool(features, batch)
        pos = pos.new_zeros((features.size(0), 3))
        batch = torch.arange(features.size(0))
        return features, pos, batch


# ---------------------------------------------------------------------------------------------
# nn/net_blocks.py restated
# ---------------------------------------------------------------------------------------------
def MLP(channels, batch_norm=True):
    """nn/net_blocks.py:43-47 — BN sits AFTER the ReLU, also on the last layer; `batch_norm` is ignored."""
    return nn.Sequential(*[
        nn.Sequential(nn.Linear(channels[i - 1], channels[i]), nn.ReLU(), nn.BatchNorm1d(channels[i]))
        for i in range(1, len(channels))])


def _init_tenzor(*shape, device='cpu', init_type=''):
    """nn/net_blocks.py:302-315."""
    if not init_type or len(shape) == 1:
        t = torch.zeros(shape)
    elif 'kaiming_normal' in init_type:
        t = torch.empty(shape)
        nn.init.kaiming_normal_(t)
    else:
        raise NotImplementedError('{} tenzor initialization is not implemented'.format(init_type))
    return t.to(device)


def _init_weights(module, init_type=''):
    """nn/net_blocks.py:318-333 (1-D 'weight' params are left untouched: the reference rebinds a local)."""
    if not init_type:
        return
    for name, param in module.named_parameters():
        if 'weight' in name:
            if 'kaiming_normal' in init_type:
                if len(param.shape) > 1:
                    nn.init.kaiming_normal_(param)
            else:
                raise NotImplementedError('{} weight initialization is not implemented'.format(init_type))


class EdgeConvFeatures(nn.Module):
    """nn/net_blocks.py:93-191 (graph_pooling branch not restated: no shipped config enables it)."""

    def __init__(self, out_size, config={}):
        super().__init__()
        self.config = {
            'conv_depth': 2, 'k_neighbors': 5, 'EConv_hidden': 200, 'EConv_hidden_depth': 2,
            'EConv_feature': 112, 'EConv_aggr': 'max', 'global_pool': 'mean',
            'skip_connections': False, 'graph_pooling': False, 'pool_ratio': 0.1}
        self.config.update(config)
        if self.config['graph_pooling']:
            raise NotImplementedError('graph_pooling is outside the restated path')
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
            self.global_pool = global_max_pool
        elif self.config['global_pool'] == 'mean':
            self.global_pool = global_mean_pool
        elif self.config['global_pool'] == 'add':
            self.global_pool = global_add_pool
        else:
            raise ValueError('{} pooling is not supported'.format(self.config['global_pool']))
        out_features = self.config['EConv_feature'] + 3 if self.config['skip_connections'] \
            else self.config['EConv_feature']
        self.lin = nn.Linear(out_features, out_size)
        self.trace = {}  # per-layer outputs of the last forward (stage-wise parity)

    def forward(self, positions, global_pool=True):
        B, N = positions.size(0), positions.size(1)
        pos_flat = positions.reshape(-1, positions.size(-1))
        batch = torch.arange(B).repeat_interleave(N)
        out = pos_flat
        for c in range(self.config['conv_depth']):
            out = self.conv_layers[c](out, batch)
            self.trace['conv%d' % c] = out
        if self.config['skip_connections']:
            out = torch.cat([out, pos_flat], dim=-1)
        if global_pool:
            pooled = self.global_pool(out, batch, B)
            self.trace['pooled'] = pooled
            return self.lin(pooled), out, batch
        return None, out, batch


class LSTMDecoderModule(nn.Module):
    """nn/net_blocks.py:363-402."""

    def __init__(self, encoding_size, hidden_size, out_elem_size, n_layers, dropout=0,
                 custom_init='kaiming_normal', **kwargs):
        super().__init__()
        self.custom_init = custom_init
        self.n_layers = n_layers
        self.encoding_size = encoding_size
        self.hidden_size = hidden_size
        self.out_elem_size = out_elem_size
        self.lstm = nn.LSTM(encoding_size, hidden_size, n_layers, dropout=dropout, batch_first=True)
        self.lin = nn.Linear(hidden_size, out_elem_size)
        _init_weights(self.lstm, init_type=custom_init)
        self.last_states = None
        self.state_override = None

    def forward(self, batch_enc, out_len):
        bs = batch_enc.size(0)
        dec_input = batch_enc.unsqueeze(1).repeat(1, out_len, 1)
        # reference order: hidden first, then cell (nn/net_blocks.py:391-392); fp32 draw, then cast
        h0 = _init_tenzor(self.n_layers, bs, self.hidden_size, init_type=self.custom_init).to(batch_enc.dtype)
        c0 = _init_tenzor(self.n_layers, bs, self.hidden_size, init_type=self.custom_init).to(batch_enc.dtype)
        if self.state_override is not None:       # tests: replay the start states another run drew (slice checks)
            h0, c0 = (t.to(batch_enc.dtype) for t in self.state_override)
        self.last_states = (h0, c0)
        out, _ = self.lstm(dec_input, (h0, c0))
        out = self.lin(out.contiguous().view(-1, self.hidden_size))
        return out.contiguous().view(bs, out_len, -1)


class PointNetPlusPlus(nn.Module):
    """nn/net_blocks.py:50-88."""

    def __init__(self, out_size, config={}):
        super().__init__()
        self.config = {'r1': 0.3, 'r2': 0.4, 'r3': 5, 'r4': 7}
        self.config.update(config)
        H, F = self.config['EConv_hidden'], self.config['EConv_feature']
        self.sa1_module = _SetAbstractionModule(0.2, self.config['r1'], MLP([3, H, H, F]))
        self.sa_last_module = _GlobalSetAbstractionModule(MLP([3 + F, H, H, F]))
        self.lin = nn.Linear(F, out_size)

    def forward(self, positions):
        pos_flat = positions.view(-1, positions.size(-1))
        batch = torch.arange(positions.size(0)).repeat_interleave(positions.size(1))
        sa_out = self.sa1_module(None, pos_flat, batch)
        out, _, _ = self.sa_last_module(*sa_out)
        return self.lin(out)


class MLPDecoder(nn.Module):
    """nn/net_blocks.py:273-298."""

    def __init__(self, encoding_size, hidden_size, out_elem_size, n_layers, out_len=1, dropout=0,
                 custom_init='kaiming_normal'):
        super().__init__()
        self.out_len = out_len
        self.mlp = MLP([encoding_size] + [hidden_size * out_len for _ in range(n_layers)] + [out_elem_size * out_len])
        _init_weights(self.mlp, init_type=custom_init)

    def forward(self, batch_enc, *args):
        out = self.mlp(batch_enc)
        return out.contiguous().view(batch_enc.size(0), self.out_len, -1)


class GRUDecoderModule(nn.Module):
    """nn/net_blocks.py:457-497."""

    def __init__(self, encoding_size, hidden_size, out_elem_size, n_layers, dropout=0,
                 custom_init='kaiming_normal', **kwargs):
        super().__init__()
        self.custom_init = custom_init
        self.n_layers = n_layers
        self.encoding_size = encoding_size
        self.hidden_size = hidden_size
        self.out_elem_size = out_elem_size
        self.recurrent_cell = nn.GRU(encoding_size, hidden_size, n_layers, dropout=dropout, batch_first=True)
        self.lin = nn.Linear(hidden_size, out_elem_size)
        _init_weights(self.recurrent_cell, init_type=custom_init)
        self.last_states = None

    def forward(self, batch_enc, out_len):
        bs = batch_enc.size(0)
        dec_input = batch_enc.unsqueeze(1).repeat(1, out_len, 1)
        h0 = _init_tenzor(self.n_layers, bs, self.hidden_size, init_type=self.custom_init).to(batch_enc.dtype)
        self.last_states = (h0,)
        out, _ = self.recurrent_cell(dec_input, h0)
        out = self.lin(out.contiguous().view(-1, self.hidden_size))
        return out.contiguous().view(bs, out_len, -1)


class LSTMDoubleReverseDecoderModule(nn.Module):
    """nn/net_blocks.py:405-454."""

    def __init__(self, encoding_size, hidden_size, out_elem_size, n_layers, dropout=0,
                 custom_init='kaiming_normal', **kwargs):
        super().__init__()
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

    def forward(self, batch_enc, out_len):
        bs = batch_enc.size(0)
        dec_input = batch_enc.unsqueeze(1).repeat(1, out_len, 1)
        h0 = _init_tenzor(self.n_layers, bs, self.hidden_size, init_type=self.custom_init).to(batch_enc.dtype)
        c0 = _init_tenzor(self.n_layers, bs, self.hidden_size, init_type=self.custom_init).to(batch_enc.dtype)
        self.last_states = (h0, c0)
        out, state = self.lstm_reverse(dec_input, (h0, c0))
        out = torch.flip(out, [1])
        out = torch.cat([out, dec_input], -1)
        out, _ = self.lstm_forward(out, state)
        out = self.lin(out.contiguous().view(-1, self.hidden_size))
        return out.contiguous().view(bs, out_len, -1)


_BLOCKS = {'EdgeConvFeatures': EdgeConvFeatures, 'LSTMDecoderModule': LSTMDecoderModule, 'MLPDecoder': MLPDecoder,
           'GRUDecoderModule': GRUDecoderModule, 'LSTMDoubleReverseDecoderModule': LSTMDoubleReverseDecoderModule}


# ---------------------------------------------------------------------------------------------
# nn/metrics: the loss terms active in the shipped configs at epoch < epoch_with_stitches
# ---------------------------------------------------------------------------------------------
class PanelLoopLoss:
    """nn/metrics/losses.py:8-51 — per-panel loop over B*P panels, data-dependent skip for < 3 edges."""

    def __init__(self, pad_vector):
        self.pad_vector = pad_vector

    def __call__(self, predicted_panels, gt_panel_num_edges):
        panels = predicted_panels.reshape(-1, predicted_panels.shape[-2], predicted_panels.shape[-1])
        pad = self.pad_vector.to(panels.dtype)
        sums = torch.zeros((panels.shape[0], 2), dtype=panels.dtype)
        rows = []
        for el in range(panels.shape[0]):
            n = int(gt_panel_num_edges[el])
            if n < 3:
                rows.append(sums[el])
                continue
            rows.append((panels[el][:n, :2] - pad[:2]).sum(dim=0))
        sums = torch.stack(rows)
        sq = sums ** 2
        return sq.sum() / (sq.shape[0] * sq.shape[1])


class PatternStitchLoss:
    """nn/metrics/losses.py:54-180 — stitch tags of the two sides of a stitch are pulled together (mean squared distance
    per stitch, averaged per pattern, then over the batch) and tags of different stitches pushed `triplet_margin` apart,
    either against every other tag (:112-146) or only against the closest one (HardNet, :148-180).  A pattern without
    stitches divides by zero exactly as the reference does (NaN)."""

    def __init__(self, triplet_margin=0.1, use_hardnet=True):
        self.triplet_margin = triplet_margin
        self.neg_loss = self._hardnet_neg if use_hardnet else self._all_pairs_neg

    @staticmethod
    def _pattern_tags(both_sides, n):
        half = len(both_sides) // 2
        return torch.cat([both_sides[:n, :], both_sides[half:half + n, :]])

    def __call__(self, stitch_tags, gt_stitches, gt_stitches_nums):
        gt_stitches = gt_stitches.long()
        B = stitch_tags.shape[0]
        flat = stitch_tags.reshape(B, -1, stitch_tags.shape[-1])              # pattern-level edge id = panel * L + edge
        rows = torch.arange(B).unsqueeze(-1)
        left, right = flat[rows, gt_stitches[:, 0, :]], flat[rows, gt_stitches[:, 1, :]]
        both = torch.cat([left, right], dim=1)
        sq = (left - right) ** 2
        similarity = 0.
        for b in range(B):
            n = gt_stitches_nums[b]
            similarity = similarity + sq[b][:n, :].sum() / n
        similarity = similarity / B
        neg = self.neg_loss(both, gt_stitches_nums)
        return similarity + neg, dict(stitch_similarity_loss=similarity, stitch_neg_loss=neg)

    def _all_pairs_neg(self, both, nums):
        per_tag = []
        for b, sides in enumerate(both):
            n = nums[b]
            tags = self._pattern_tags(sides, n)
            for i, tag in enumerate(tags):
                gap = self.triplet_margin - ((tag - tags) ** 2).sum(dim=-1)
                gap[i] = 0
                gap[i + n if i < n else i - n] = 0                            # the other side of the same stitch
                gap = torch.max(gap, torch.zeros_like(gap))
                per_tag.append(gap.sum() / len(gap))
        return sum(per_tag) / len(per_tag)

    def _hardnet_neg(self, both, nums):
        per_tag = []
        for b, sides in enumerate(both):
            n = nums[b]
            tags = self._pattern_tags(sides, n)
            for i, tag in enumerate(tags):
                d = ((tag - tags) ** 2).sum(dim=-1)
                d[i] = float('inf')
                d[i + n if i < n else i - n] = float('inf')
                per_tag.append(max(self.triplet_margin - d.min(), 0))
        return sum(per_tag) / len(per_tag)


def eval_pad_vector(data_stats):
    """nn/metrics/eval_utils.py (pad vector = -shift/scale when GT is standardised, zeros otherwise)."""
    if data_stats:
        shift = torch.as_tensor(data_stats['shift'], dtype=torch.float32)
        scale = torch.as_tensor(data_stats['scale'], dtype=torch.float32)
        return -shift / scale
    return torch.zeros(4)


class ComposedPatternLoss:
    """nn/metrics/composed_loss.py:129-362: shape / loop / rotation / translation terms, from `epoch_with_stitches` on
    also the stitch terms (:336-362 — PatternStitchLoss, supervised stitch tags, free-edge classification), INCLUDING the
    ground-truth pre-processing in front of them: panel-order matching (:428-590) and panel-origin matching (:593-755) with
    the re-numbering of the stitched edges and the per-panel shift of the free-edge mask, restated with the reference's own
    loops; the segmentation term (:323-332) through the restated entmax.SparsemaxLoss above."""

    def __init__(self, data_config, in_config={}):
        self.config = {
            'loss_components': ['shape'], 'quality_components': [], 'loop_loss_weight': 1.,
            'segm_loss_weight': 0.05, 'stitch_tags_margin': 0.3, 'epoch_with_stitches': 40,
            'stitch_supervised_weight': 0.1, 'stitch_hardnet_version': False,
            'panel_origin_invariant_loss': True, 'panel_order_inariant_loss': True,
            'order_by': 'placement', 'epoch_with_order_matching': 0}
        self.config.update(in_config)
        self.with_quality_eval = False
        self.training = False
        self.debug_prints = False
        self.l_components = self.config['loss_components']
        self.max_panel_len = data_config['max_panel_len']
        self.max_pattern_size = data_config['max_pattern_len']
        stats = data_config.get('standardize')
        outl = {'shift': stats['gt_shift']['outlines'], 'scale': stats['gt_scale']['outlines']} if stats else {}
        self.loop_loss = PanelLoopLoss(eval_pad_vector(outl))
        if 'stitch' in self.l_components:
            self.stitch_loss = PatternStitchLoss(self.config['stitch_tags_margin'],
                                                 use_hardnet=self.config['stitch_hardnet_version'])
        self.last_permutation = None
        self.last_leading_edges = None

    # -- composed_loss.py:572-590
    @staticmethod
    def _feature_permute(feat, perm):
        ext = perm
        while ext.dim() < feat.dim():
            ext = ext.unsqueeze(-1)
        return torch.gather(feat, 1, ext.expand(feat.shape))

    # -- composed_loss.py:530-570
    def _panel_order_match(self, pred_features, gt_features):
        B, P = pred_features.shape[0], gt_features.shape[1]
        if self.epoch < self.config['epoch_with_order_matching']:
            return torch.stack([torch.randperm(P, dtype=torch.long) for _ in range(B)])
        dist = torch.cdist(pred_features.reshape(B, P, -1), gt_features.reshape(B, P, -1))
        flat = dist.view(B, -1)
        perm = torch.full((B, P), -1, dtype=torch.long)
        for _ in range(P):
            ids = flat.argmin(dim=1)
            rows, cols = ids // P, ids % P
            for i in range(B):
                perm[i, rows[i]] = cols[i]
                dist[i, rows[i], :] = float('inf')
                dist[i, :, cols[i]] = float('inf')
        if torch.isfinite(dist).any():
            raise ValueError('ComposedPatternLoss::Error::Failed to match panel order')
        return perm

    # -- composed_loss.py:428-528 (stitch re-numbering not restated: stitch terms raise)
    def _gt_order_match(self, preds, gt):
        with torch.no_grad():
            by = self.config['order_by']
            if by == 'placement':
                pf = torch.cat([preds['translations'], preds['rotations']], dim=-1)
                gf = torch.cat([gt['translations'], gt['rotations']], dim=-1)
            elif by == 'translation':
                pf, gf = preds['translations'], gt['translations']
            elif by == 'shape_translation':
                B, P = preds['outlines'].shape[:2]
                pf = torch.cat([preds['translations'], preds['outlines'].contiguous().view(B, P, -1)], dim=-1)
                gf = torch.cat([gt['translations'], gt['outlines'].contiguous().view(B, P, -1)], dim=-1)
            elif by == 'stitches':
                pf = torch.cat([preds['translations'], preds['rotations']], dim=-1)
                gf = torch.cat([gt['translations'], gt['rotations']], dim=-1)
                if self.epoch >= self.config['epoch_with_stitches']:        # :464-477: the free-edge mask joins the feature
                    B, P = preds['free_edges_mask'].shape[:2]
                    pf = torch.cat([pf, torch.round(torch.sigmoid(preds['free_edges_mask'])).view(B, P, -1)], dim=-1)
                    gf = torch.cat([gf, gt['free_edges_mask'].view(B, P, -1).to(gf.dtype)], dim=-1)
            else:
                raise NotImplementedError(by)
            perm = self._panel_order_match(pf.detach().to(gf.dtype), gf)
            self.last_permutation = perm
            out = dict(gt)
            out['outlines'] = self._feature_permute(gt['outlines'], perm)
            out['num_edges'] = self._feature_permute(gt['num_edges'], perm)
            if 'empty_panels_mask' in gt:
                out['empty_panels_mask'] = self._feature_permute(gt['empty_panels_mask'], perm)
            if 'rotation' in self.l_components:
                out['rotations'] = self._feature_permute(gt['rotations'], perm)
            if 'translation' in self.l_components:
                out['translations'] = self._feature_permute(gt['translations'], perm)
            if self._stitch_terms_active():                                    # :505-517
                out['stitches'] = self._stitch_after_permute(gt['stitches'], gt['num_stitches'], perm, self.max_panel_len)
                out['free_edges_mask'] = self._feature_permute(gt['free_edges_mask'], perm)
                if 'stitch_supervised' in self.l_components:
                    out['stitch_tags'] = self._feature_permute(gt['stitch_tags'], perm)
        return out

    def _stitch_terms_active(self):
        return self.epoch >= self.config['epoch_with_stitches'] and any(
            c in self.l_components for c in ('stitch', 'stitch_supervised', 'free_class'))

    # -- composed_loss.py:592-620: where did each panel go? (a panel named twice by the permutation keeps its LAST slot)
    @staticmethod
    def _stitch_after_permute(stitches, nums, perm, L):
        out = stitches.clone()
        for b in range(len(stitches)):
            slot_of = [-1] * perm.shape[1]
            for slot in range(perm.shape[1]):
                slot_of[int(perm[b][slot])] = slot
            for side in (0, 1):
                for i in range(int(nums[b])):
                    e = int(stitches[b][side][i])
                    panel = e // L
                    out[b][side][i] = slot_of[panel] * L + (e - panel * L)
        return out

    # -- composed_loss.py:705-725: the loop of a panel now starts at `lead`; padding stays where it is
    @staticmethod
    def _per_panel_shift(feat, leads, num_edges):
        out = feat.clone()
        P = feat.shape[1]
        for b in range(len(feat)):
            for p in range(P):
                lead, n = int(leads[b * P + p]), int(num_edges[b * P + p])
                if n < 3 or not lead:
                    continue
                cur = feat[b][p]
                out[b][p] = torch.cat((cur[lead:n], cur[:lead], cur[n:]))
        return out

    # -- composed_loss.py:727-755: the same shift applied to the edge numbers the stitches refer to
    @staticmethod
    def _gt_stitches_shift(stitches, nums, leads, num_edges, P, L):
        out = stitches.clone()
        for b in range(len(stitches)):
            for side in (0, 1):
                for i in range(int(nums[b])):
                    e = int(stitches[b][side][i])
                    panel = e // L
                    g = b * P + panel
                    lead, n = int(leads[g]), int(num_edges[g])
                    inner = e - panel * L
                    out[b][side][i] = panel * L + (inner - lead if inner >= lead else n - (lead - inner))
        return out

    # -- composed_loss.py:656-703,757-765: per panel, try every edge-loop origin, keep the first best
    def _rotate_gt(self, preds, gt, gt_num_edges):
        with torch.no_grad():
            pr = preds['outlines'].detach()
            gto = gt['outlines'].to(pr.dtype)
            B = pr.shape[0]
            pr = pr.reshape(-1, pr.shape[-2], pr.shape[-1])
            g = gto.reshape(-1, gto.shape[-2], gto.shape[-1])
            chosen, leads = [], []
            for el in range(pr.shape[0]):
                n = int(gt_num_edges[el])
                shifted = g[el]
                best = ((pr[el] - shifted) ** 2).sum()
                keep, lead = shifted, 0
                for i in range(1, n):
                    shifted = torch.cat((shifted[1:n], shifted[0:1, :], shifted[n:]))
                    d = ((pr[el] - shifted) ** 2).sum()
                    if d < best:
                        best, keep, lead = d, shifted, i
                chosen.append(keep)
                leads.append(lead)
            out = dict(gt)
            out['outlines'] = torch.stack(chosen).view(B, -1, g.shape[-2], g.shape[-1])
            self.last_leading_edges = torch.tensor(leads, dtype=torch.int32)
            if self._stitch_terms_active():                                    # :604-617
                out['stitches'] = self._gt_stitches_shift(gt['stitches'], gt['num_stitches'], leads, gt_num_edges,
                                                          self.max_pattern_size, self.max_panel_len)
                out['free_edges_mask'] = self._per_panel_shift(gt['free_edges_mask'], leads, gt_num_edges)
                if 'stitch_supervised' in self.l_components:
                    out['stitch_tags'] = self._per_panel_shift(gt['stitch_tags'], leads, gt_num_edges)
        return out

    def __call__(self, preds, ground_truth, names=None, epoch=1000):
        self.epoch = epoch
        gt = ground_truth
        if self.config['panel_order_inariant_loss']:
            if 'segmentation' in self.l_components:                            # composed_loss.py:242-243
                raise NotImplementedError('Order matching not supported for training with segmentation losses')
            gt = self._gt_order_match(preds, gt)
        dt = preds['outlines'].dtype
        num_edges = gt['num_edges'].int().view(-1)
        if self.config['panel_origin_invariant_loss']:
            gt = self._rotate_gt(preds, gt, num_edges)
        loss, d = 0., {}
        mse = nn.functional.mse_loss
        if 'shape' in self.l_components:
            d['pattern_loss'] = mse(preds['outlines'], gt['outlines'].to(dt))
            loss = loss + d['pattern_loss']
        if 'loop' in self.l_components:
            d['loop_loss'] = self.loop_loss(preds['outlines'], num_edges)
            loss = loss + self.config['loop_loss_weight'] * d['loop_loss']
        if 'rotation' in self.l_components:
            d['rotation_loss'] = mse(preds['rotations'], gt['rotations'].to(dt))
            loss = loss + d['rotation_loss']
        if 'translation' in self.l_components:
            d['translation_loss'] = mse(preds['translations'], gt['translations'].to(dt))
            loss = loss + d['translation_loss']
        if 'segmentation' in self.l_components:                                # composed_loss.py:323-332
            att = preds['att_weights']
            d['segm_loss'] = SparsemaxLoss()(att.reshape(-1, att.shape[-1]), gt['segmentation'].reshape(-1))
            loss = loss + self.config['segm_loss_weight'] * d['segm_loss']
        if self._stitch_terms_active():                                        # :259-266, 336-362
            extra = 0.
            if 'stitch' in self.l_components:
                st, parts = self.stitch_loss(preds['stitch_tags'], gt['stitches'], gt['num_stitches'])
                d.update(parts)
                extra = extra + st
            if 'stitch_supervised' in self.l_components:
                d['stitch_supervised_loss'] = mse(preds['stitch_tags'], gt['stitch_tags'].to(dt))
                extra = extra + self.config['stitch_supervised_weight'] * d['stitch_supervised_loss']
            if 'free_class' in self.l_components:
                d['free_edges_loss'] = nn.functional.binary_cross_entropy_with_logits(
                    preds['free_edges_mask'], gt['free_edges_mask'].to(dt))
                extra = extra + d['free_edges_loss']
            loss = loss + extra
        update = (epoch == self.config['epoch_with_stitches'] and any(
            c in self.l_components for c in ('stitch', 'stitch_supervised', 'free_class'))
            or epoch == self.config['epoch_with_order_matching'] and self.config['panel_order_inariant_loss'])
        return loss, d, update

    def train(self, mode=True):
        self.training = mode

    def eval(self):
        self.training = False


# ---------------------------------------------------------------------------------------------
# nn/nets.py restated
# ---------------------------------------------------------------------------------------------
class BaseModule(nn.Module):
    """nn/nets.py:11-37."""

    def __init__(self):
        super().__init__()
        self.config = {'loss': 'MSELoss', 'model': self.__class__.__name__}

    def train(self, mode=True):
        super().train(mode)
        if isinstance(self.loss, object):
            self.loss.train(mode)
        return self

    def eval(self):
        super().eval()
        if isinstance(self.loss, object):
            self.loss.eval()
        return self


class GarmentFullPattern3D(BaseModule):
    """nn/nets.py:41-184: encoder -> pattern LSTM -> panel LSTM + placement Linear."""

    def __init__(self, data_config, config={}, in_loss_config={}):
        super().__init__()
        self.panel_elem_len = data_config['element_size']
        self.max_panel_len = data_config['max_panel_len']
        self.max_pattern_size = data_config['max_pattern_len']
        self.rotation_size = data_config['rotation_size']
        self.translation_size = data_config['translation_size']
        self.config.update({
            'panel_encoding_size': 250, 'panel_hidden_size': 250, 'panel_n_layers': 3,
            'pattern_encoding_size': 250, 'pattern_hidden_size': 250, 'pattern_n_layers': 2,
            'dropout': 0, 'lstm_init': 'kaiming_normal_', 'feature_extractor': 'EdgeConvFeatures',
            'panel_decoder': 'LSTMDecoderModule', 'pattern_decoder': 'LSTMDecoderModule',
            'stitch_tag_dim': 3})
        if 'panel_hidden_size' not in config:          # nn/nets.py:75-78 mutate the CALLER's dict
            config['panel_hidden_size'] = config['panel_encoding_size']
        if 'pattern_hidden_size' not in config:
            config['pattern_hidden_size'] = config['pattern_encoding_size']
        self.config.update(config)
        self.config['loss'] = {
            'loss_components': ['shape', 'loop', 'rotation', 'translation'],
            'quality_components': ['shape', 'discrete', 'rotation', 'translation'],
            'loop_loss_weight': 1., 'stitch_tags_margin': 0.3, 'epoch_with_stitches': 40,
            'stitch_supervised_weight': 0.1, 'stitch_hardnet_version': False,
            'panel_origin_invariant_loss': True}
        self.config['loss'].update(in_loss_config)
        self.loss = ComposedPatternLoss(data_config, self.config['loss'])
        self.config['loss'] = self.loss.config

        self.feature_extractor = _BLOCKS[self.config['feature_extractor']](
            self.config['pattern_encoding_size'], self.config)
        self.config.update(self.feature_extractor.config)
        self.panel_decoder = _BLOCKS[self.config['panel_decoder']](
            encoding_size=self.config['panel_encoding_size'], hidden_size=self.config['panel_hidden_size'],
            out_elem_size=self.panel_elem_len + self.config['stitch_tag_dim'] + 1,
            n_layers=self.config['panel_n_layers'], out_len=self.max_panel_len,
            dropout=self.config['dropout'], custom_init=self.config['lstm_init'])
        self.pattern_decoder = _BLOCKS[self.config['pattern_decoder']](
            encoding_size=self.config['pattern_encoding_size'], hidden_size=self.config['pattern_hidden_size'],
            out_elem_size=self.config['panel_encoding_size'], n_layers=self.config['pattern_n_layers'],
            out_len=self.max_pattern_size, dropout=self.config['dropout'],
            custom_init=self.config['lstm_init'])
        self.placement_decoder = nn.Linear(self.config['panel_encoding_size'],
                                           self.rotation_size + self.translation_size)
        self.trace = {}

    def forward_encode(self, positions_batch):
        return self.feature_extractor(positions_batch)[0]

    def forward_pattern_decode(self, garment_encodings):
        enc = self.pattern_decoder(garment_encodings, self.max_pattern_size)
        return enc.contiguous().view(-1, enc.shape[-1])

    def forward_panel_decode(self, flat_panel_encodings, batch_size):
        self.trace['panel_encodings'] = flat_panel_encodings
        flat_panels = self.panel_decoder(flat_panel_encodings, self.max_panel_len)
        flat_placement = self.placement_decoder(flat_panel_encodings)
        flat_rot = flat_placement[:, :self.rotation_size]
        flat_tr = flat_placement[:, self.rotation_size:]
        pp = flat_panels.contiguous().view(batch_size, self.max_pattern_size, self.max_panel_len, -1)
        return {
            'outlines': pp[:, :, :, :self.panel_elem_len],
            'rotations': flat_rot.contiguous().view(batch_size, self.max_pattern_size, -1),
            'translations': flat_tr.contiguous().view(batch_size, self.max_pattern_size, -1),
            'stitch_tags': pp[:, :, :, self.panel_elem_len:-1],
            'free_edges_mask': pp[:, :, :, -1]}

    def forward_decode(self, garment_encodings):
        self.trace['encoding'] = garment_encodings
        return self.forward_panel_decode(self.forward_pattern_decode(garment_encodings),
                                         garment_encodings.size(0))

    def forward(self, positions_batch, **kwargs):
        return self.forward_decode(self.forward_encode(positions_batch))


class GarmentSegmentPattern3D(GarmentFullPattern3D):
    """nn/nets.py:187-299: per-point sparsemax attention -> per-panel pooled encodings -> panel LSTM."""

    def __init__(self, data_config, config={}, in_loss_config={}):
        if 'loss_components' not in in_loss_config:
            in_loss_config.update(loss_components=['shape', 'loop', 'rotation', 'translation'],
                                  quality_components=['shape', 'discrete', 'rotation', 'translation'])
        super().__init__(data_config, config, in_loss_config)
        self.save_att_weights = 'segmentation' in self.loss.config['loss_components']
        if 'local_attention' not in self.config:
            self.config['local_attention'] = False
        att_in = self.feature_extractor.config['EConv_feature']
        if not self.config['local_attention']:
            att_in += self.config['pattern_encoding_size']
        if self.config['skip_connections']:
            att_in += 3
        self.point_segment_mlp = nn.Sequential(
            MLP([att_in, att_in, att_in, self.max_pattern_size]), Sparsemax(dim=1))
        panel_att_out = self.feature_extractor.config['EConv_feature']
        if self.config['skip_connections']:
            panel_att_out += 3
        self.panel_dec_lin = nn.Linear(panel_att_out, self.feature_extractor.config['panel_encoding_size'])
        del self.pattern_decoder

    def forward_panel_enc_from_3d(self, positions_batch):
        B = positions_batch.shape[0]
        init_enc, feats, batch = self.feature_extractor(positions_batch, not self.config['local_attention'])
        n_pts = feats.shape[0] // B
        if self.config['local_attention']:
            w = self.point_segment_mlp(feats)
        else:
            glob = init_enc.unsqueeze(1).repeat(1, n_pts, 1).view([-1, init_enc.shape[-1]])
            w = self.point_segment_mlp(torch.cat([glob, feats], dim=-1))
        self.trace['point_features'] = feats
        self.trace['att_weights'] = w
        per_panel = []
        for p in range(w.shape[-1]):                          # nn/nets.py:263-276
            pf = self.feature_extractor.global_pool(w[:, p].unsqueeze(-1) * feats, batch, B)
            pf = self.panel_dec_lin(pf)
            per_panel.append(pf.view(B, -1, pf.shape[-1]))
        enc = torch.cat(per_panel, dim=1)
        enc = enc.view(B, -1, enc.shape[-1])
        w = w.view(B, -1, w.shape[-1]) if self.save_att_weights else []
        return enc, w

    def forward(self, positions_batch, **kwargs):
        B = positions_batch.shape[0]
        enc, att = self.forward_panel_enc_from_3d(positions_batch)
        panels = self.forward_panel_decode(enc.view(-1, enc.shape[-1]), B)
        if len(att) > 0:
            panels.update(att_weights=att)
        return panels


class StitchOnEdge3DPairs(nn.Module):
    """nn/nets.py:303-353 (model only: MLP([element_size, 200 x 3, 1]) on every pair row)."""

    def __init__(self, data_config, config={}, in_loss_config={}):
        super().__init__()
        self.config = {'stitch_hidden_size': 200, 'stitch_mlp_n_layers': 3}
        self.config.update(config)
        mid = [self.config['stitch_hidden_size']] * self.config['stitch_mlp_n_layers']
        self.mlp = MLP([data_config['element_size']] + mid + [1])

    def forward(self, pairs_batch, **kwargs):
        shape = list(pairs_batch.shape)
        shape.pop(-1)
        return self.mlp(pairs_batch.contiguous().view(-1, pairs_batch.shape[-1])).view(shape)


# ---------------------------------------------------------------------------------------------
# the timed unit: nn/trainer.py:92-99
# ---------------------------------------------------------------------------------------------
def train_step(model, features, gt, epoch=0, seed=None):
    """One fwd -> loss -> bwd pass (no optimizer step).  `seed` is set right before the forward so the
    random LSTM states are reproducible."""
    if seed is not None:
        torch.manual_seed(seed)
    preds = model(features, log_step=0, epoch=epoch)
    loss, loss_dict, _ = model.loss(preds, gt, epoch=epoch)
    loss.backward()
    return preds, loss, loss_dict


def synthetic_batch(B, N, data_config, seed=0, dtype=torch.float32):
    """SURVEY.md §8(d) synthetic inputs: randn cloud, randn targets, num_edges in [3, max_panel_len]."""
    g = torch.Generator().manual_seed(seed)
    P, L = data_config['max_pattern_len'], data_config['max_panel_len']
    feats = torch.randn(B, N, 3, generator=g).to(dtype)
    gt = {
        'outlines': torch.randn(B, P, L, data_config['element_size'], generator=g),
        'rotations': torch.randn(B, P, data_config['rotation_size'], generator=g),
        'translations': torch.randn(B, P, data_config['translation_size'], generator=g),
        'num_edges': torch.randint(3, L + 1, (B, P), generator=g),
    }
    return feats, gt
