"""Host-side orchestration of the gfx950 kernels (C ABI in include/gpe_hip.h) + the autograd glue.

torch is used here for device memory, streams, parameter storage and the CPU RNG of the reference's random
LSTM states — every arithmetic step of the path is a call into libgpe_hip.so.  Nothing in this file falls back
to torch math: a missing library or a failing launch raises.

Reference lines restated by each Function are cited in its docstring (paths relative to /root/reference).

Two per-model services live here as well (both optional; without them every Function behaves like a plain autograd op):
  * PackPlan  — all weight-derived kernel operands of a model (MFMA-packed / transposed / gate-interleaved weights, the
                P|Q split of the first edge Linear, b_ih + b_hh) are rebuilt by ONE launch when the weights changed,
                instead of ~45 small launches per training step;
  * grad sink — when a model's parameters live in a flat arena (optim.FlatArena), backward kernels write weight gradients
                straight into the arena's gradient buffer (no per-parameter tensors, no cat/copy for the all-reduce
                buckets, one fused Adam launch).
"""
import os
import warnings
import weakref

import numpy as np
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


_WS = {}                   # (device index, stream handle) -> uint8 tensor: the scratch of that stream's C-ABI calls


def _workspace(nbytes, device):
    """caller-owned scratch of a C-ABI call (include/gpe_hip.h: `ws`).  One grow-only buffer per (device, stream): launches of one
    stream are ordered, so consecutive calls may share it, and two streams never do (the library's only rule for workspaces).
    Re-allocation goes through torch's caching allocator, which keeps the old block alive until the stream has passed it."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), device=torch.device('cuda', idx), dtype=torch.uint8)
        _WS[key] = buf
    return buf


_EDGE_WS_BYTES = {}


def edge_workspace(B, N, k, ldmax, device):
    """-> (ws, bytes) for gpe_edge_mlp_fwd / _bwd / gpe_edge_redgemm / gpe_edge_pq_amax calls on B clouds of N points, k
    neighbours, per-point output pitches <= ldmax floats."""
    key = (B, N, k, ldmax)
    n = _EDGE_WS_BYTES.get(key)
    if n is None:
        n = L.query('gpe_edge_ws_bytes', B, N, k, ldmax)
        if n < 0:
            raise RuntimeError('gpe_edge_ws_bytes failed with code %d' % n)
        _EDGE_WS_BYTES[key] = n
    return _workspace(n, device), n


def f16x3_words(n, rows, device):
    """n caller-owned "amax words" (include/gpe_hip.h) when the f16x3 arithmetic will run on `rows`-row edge launches, else None:
    one uint32 per tensor whose largest magnitude a kernel measures while storing it and a later kernel scales by."""
    if L.get_math() != 'f16x3' or rows < L.query('gpe_f16x3_min_rows'):
        return None
    return torch.zeros(n, device=device, dtype=torch.int32)


def _word(words, i):
    return None if words is None else words[i:i + 1]


_TICKETS = {}


def _ticket(device):
    """the last-arriver ticket of gpe_pack_fold: one zeroed uint32 per (device, stream), left zero by every launch."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    t = _TICKETS.get(key)
    if t is None:
        t = _TICKETS[key] = torch.zeros(1, device=torch.device('cuda', idx), dtype=torch.int32)
    return t


def pack_fold(W, bias, stats, ews, ews_n, clear_word):
    """-> (packed W diag(s), b + W t) of the BatchNorm fold in one launch; ews given (f16x3): the packed weight's amax lands in the
    edge workspace and the coming gpe_edge_mlp_fwd is told so (bit 1 of its out_half)."""
    N, K = W.shape
    wp = torch.empty(L.query('gpe_packed_size', N, K), device=W.device, dtype=F32)
    bf = torch.empty(N, device=W.device, dtype=F32)
    L.call('gpe_pack_fold', W, W.stride(0), N, K, stats[2], stats[3], bias, wp, bf, ews, ews_n if ews is not None else 0, clear_word,
           _ticket(W.device) if ews is not None else None)
    return wp, bf


# -------------------------------------------------------------------------------------------------
# Leaf weight-gradient launches on a side stream
# -------------------------------------------------------------------------------------------------
# The weight-gradient reduce-GEMMs of a recurrent stack are LEAVES of the backward pass (nothing downstream reads them before the
# optimizer), and what follows the panel decoder's backward on the critical path is the pattern decoder's persistent backward
# launch, which occupies a few CUs for 0.2 ms.  `side_grads(...)` moves the launches issued inside it to one side stream per device
# (forked behind everything the current stream has queued, joined back by an autograd callback when the backward pass ends), so
# that the two overlap.  Only when every gradient involved lands in a flat arena (optim.FlatArena: nothing is returned to autograd),
# and outside stream captures (measured: no gain inside a captured step).  With more than one rank a gradient bucket leaves for the
# all-reduce when the arena is told its last gradient is written = queued: parallel.DistributedHotPath joins the side stream in front
# of a collective issued from the main stream.  GPE_DEBUG=1 GPE_SIDE_GRADS=0 keeps everything on one stream (A/B).
# Only for steps the GPU bounds: the fork / join / record_stream bookkeeping costs the host ~0.3 ms per step, which a host-bound shape
# (BASELINE cfg 1 with eager launches: 3.6 -> 4.0 ms) cannot hide.  The proxy is the size of the step's last EdgeConv graph (edges).
SIDE_GRADS = not (os.environ.get('GPE_DEBUG') == '1' and os.environ.get('GPE_SIDE_GRADS') == '0')
SIDE_MIN_EDGES = 1 << 17
SIDE_PQ = not (os.environ.get('GPE_DEBUG') == '1' and os.environ.get('GPE_SIDE_PQ') == '0')     # the forward fork of EdgeConvFn (A/B)
_LAST_EDGES = [0]
_SIDE_STREAMS = {}
_SIDE_JOIN_QUEUED = [False]
_SIDE_DIRTY = [False]                 # the side stream holds launches the current stream has not waited for


def _side_stream():
    idx = torch.cuda.current_device()
    st = _SIDE_STREAMS.get(idx)
    if st is None:
        st = _SIDE_STREAMS[idx] = torch.cuda.Stream(device=idx)
    return st


def join_side():
    """The current stream waits for everything on the side stream.  Runs as an autograd callback at the end of every backward pass
    that used the side stream, and once more in front of every optimizer step (optim.FusedAdam) — the second call is what a pass
    that died with an exception half-way leaves to."""
    _SIDE_JOIN_QUEUED[0] = False
    if _SIDE_DIRTY[0]:
        _SIDE_DIRTY[0] = False
        torch.cuda.current_stream().wait_stream(_side_stream())


# Work that only a LATER phase of the step needs and that has its inputs early — the transposed graphs of the gather backward need
# the forward's graphs only — waits in _IDLE_JOBS for a stretch of the step that leaves the chip idle: the first recurrent stack of the
# forward (the pattern decoder's persistent launch sits on the CUs of one XCD) launches it on the side stream (run_idle_jobs).
_IDLE_JOBS = []


def run_idle_jobs():
    if not _IDLE_JOBS:
        return
    if torch.cuda.is_current_stream_capturing():
        del _IDLE_JOBS[:]
        return
    main, side = torch.cuda.current_stream(), _side_stream()
    side.wait_stream(main)
    _SIDE_DIRTY[0] = True
    with torch.cuda.stream(side):
        for job in _IDLE_JOBS:
            job['rev'] = knn_reverse(job['idx'])
            job['idx'].record_stream(side)
            job['event'] = side.record_event()
    del _IDLE_JOBS[:]


class side_grads:
    """with side_grads(params, tensors) as on_side: ...  — the launches inside run on the side stream when that is legal (see above);
    `tensors` = every main-stream tensor the launches read (kept away from the allocator until the side stream has passed them)."""

    def __init__(self, params, tensors):
        self.on = bool(SIDE_GRADS and _LAST_EDGES[0] >= SIDE_MIN_EDGES and params and all(_SINK.get(p.data_ptr()) is not None and _SINK[p.data_ptr()][0]() is not None
                                                     for p in params) and not torch.cuda.is_current_stream_capturing())
        self.tensors = tensors
        self.ctx = None

    def __enter__(self):
        if not self.on:
            return False
        main, side = torch.cuda.current_stream(), _side_stream()
        side.wait_stream(main)
        for t in self.tensors:
            if t is not None:
                t.record_stream(side)
        _SIDE_DIRTY[0] = True
        if not _SIDE_JOIN_QUEUED[0]:
            _SIDE_JOIN_QUEUED[0] = True
            torch.autograd.Variable._execution_engine.queue_callback(join_side)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return True

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


# -------------------------------------------------------------------------------------------------
# PackPlan: weight-derived operands refreshed by one launch
# -------------------------------------------------------------------------------------------------
WEIGHTS_EPOCH = 0          # bumped by optimizers that update parameters through raw pointers (optim.FusedAdam)
_PACKS = {}                # (param data_ptr, kind) -> (plan, output tensor); a plan's entries die with the plan (weakref.finalize)

K_PLAIN, K_TRANS, K_GATES, K_PQ, K_PQT, K_VADD, K_BPQ, K_GRUB = 0, 1, 2, 3, 4, 5, 6, 7
K_GATES_H3, K_AMAX, K_TRANS_H3 = 8, 9, 10      # f16x3 operands of the recurrences: fp16 plane packs + the amax job they depend on
_JOB = np.dtype([('w', '<u8'), ('w2', '<u8'), ('out', '<u8'), ('total', '<i8'), ('first_block', '<i8'),
                 ('ldw', '<i4'), ('N', '<i4'), ('K', '<i4'), ('kind', '<i4'), ('Npad', '<i4'), ('aux', '<i4')])
assert _JOB.itemsize == 64


# (PackPlan.refresh on a side stream under the layer-1 kNN search was built in round 5 and MEASURED SLOWER — cfg 2, one session, twice
# each: 10.19 / 10.17 ms per step without, 10.34 / 10.34 with: the stream fork / join costs more than the 0.09 ms of pack launches it
# hides — and removed in round 6: profiles/r05_c_reverted_experiments.md.)


def bump_weights_epoch():
    global WEIGHTS_EPOCH
    WEIGHTS_EPOCH += 1


class PackPlan:
    """Collects the static pack jobs of a model (net_blocks modules call `add_*` from their `register_packs`) and re-runs them
    all with ONE gpe_pack_multi launch at the start of every model forward (40 us at the shipped sizes).  Outputs are
    persistent buffers, so backward reads what forward used.

    Why every forward and not "when a parameter changed": torch's version counters do not see an update made through
    `p.data` (an EMA of the weights, a hand-written `p.data.copy_()`, a foreign optimizer working on raw storage), and a stale
    pack gives silently wrong numbers.  In training the weights change every step anyway.  A caller whose weights are truly
    constant (inference loops) may set `plan.frozen = True` after the first forward: the launch is then skipped while the
    version counters and WEIGHTS_EPOCH stand still — that caller vouches for not touching `p.data`.

    A plan — and therefore the module that owns it — is SINGLE-STREAM: its outputs and amax words are rewritten by every refresh,
    so two streams driving the same module at the same time would race (the C library itself is stream-safe: all state is the
    caller's).  Use one module copy per stream (tests/test_gpu_kernels.py::test_two_streams_one_device).  A module that moves from
    one stream to another between forwards is handled: refresh() makes the new stream wait for the previous one."""

    def __init__(self):
        self.specs = []        # (param, param2, kind, N, K, aux)
        self.outs = []
        self.table = None
        self.blocks = 0
        self.vers = None
        self.epoch = -1
        self.home = None
        self.keys = []
        self.frozen = False
        self.h3_current = False
        self._stream = None
        self._keys_box = box = []              # shared with the finalizer: the keys this plan currently owns in _PACKS
        me = weakref.ref(self)
        # pop only entries that are still THIS plan's: a newer plan over the same parameters may have re-registered the keys
        weakref.finalize(self, lambda: [_PACKS.pop(k, None) for k in box if (_PACKS.get(k) or (None,))[0] is me])
        self._me = me

    def _add(self, p, kind, N, K, aux=0, p2=None, out_numel=None):
        self.specs.append((p, p2, kind, N, K, aux, out_numel))

    def add_linear(self, w, fwd=True, bwd=True):
        if fwd:
            self._add(w, K_PLAIN, w.shape[0], w.shape[1])
        if bwd:
            self._add(w, K_TRANS, w.shape[1], w.shape[0])

    def add_gates(self, w, H, G=4):
        """gate-interleaved pack of a recurrent weight [G*H, K] (LSTM G = 4, GRU G = 3) — and, for the f16x3 arithmetic, its two
        fp16 planes + the amax word they are normalised by (a kind-9 job of the plan's FIRST launch)."""
        self._add(w, K_GATES, G * H, w.shape[1], aux=H)
        self._add(w, K_GATES_H3, G * H, w.shape[1], aux=H)
        if G == 4:
            # the persistent backward launch (csrc/gpe_rnn_persist.hip) reads W^T through the fp16 pipe
            self._add(w, K_TRANS_H3, w.shape[1], G * H)

    def add_edge_first(self, w1, b1):
        H, C2 = w1.shape
        C = C2 // 2
        self._add(w1, K_PQ, 2 * H, C, aux=H)
        self._add(w1, K_PQT, C, 2 * H, aux=H)
        self._add(b1, K_BPQ, 2 * H, 0, aux=H, out_numel=2 * H)

    def add_bias_sum(self, b_ih, b_hh):
        self._add(b_ih, K_VADD, b_ih.numel(), 0, p2=b_hh, out_numel=b_ih.numel())

    def add_gru_bias(self, b_ih, b_hh, H):
        """b_ih + [b_hr | b_hz | 0]: nn.GRU keeps b_hn inside the r-gated term."""
        self._add(b_ih, K_GRUB, b_ih.numel(), 0, aux=2 * H, p2=b_hh, out_numel=b_ih.numel())

    def _versions(self):
        return [p._version + (p2._version if p2 is not None else 0) for p, p2, *_ in self.specs]

    def _build(self):
        for k in self.keys:
            if (_PACKS.get(k) or (None,))[0] is self._me:
                _PACKS.pop(k, None)
        self.keys, self.outs = [], []
        del self._keys_box[:]
        dev = self.specs[0][0].device
        tab = np.zeros(len(self.specs), dtype=_JOB)
        blk = 0
        # f16x3 plane packs are normalised by their tensor's largest magnitude: one amax word per planes job, filled by a kind-9
        # job of an EARLIER launch (self.pre_table) into self.words (zeroed before every refresh)
        h3 = [i for i, sp in enumerate(self.specs) if sp[2] in (K_GATES_H3, K_TRANS_H3)]
        # ... one word and one amax job per MATRIX (its gate-interleaved and its transposed planes share them)
        mats = []
        for i in h3:
            if all(self.specs[i][0] is not self.specs[j][0] for j in mats):
                mats.append(i)
        self.words = torch.zeros(max(len(mats), 1), device=dev, dtype=torch.int32)
        pre = np.zeros(len(mats), dtype=_JOB)
        pblk = 0
        for wi, i in enumerate(mats):
            p = self.specs[i][0]
            n_el = p.numel()
            pre[wi] = (p.data_ptr(), 0, self.words[wi:wi + 1].data_ptr(), n_el, pblk, p.stride(0), p.shape[0], p.shape[1], K_AMAX, 0, 0)
            pblk += L.query('gpe_pack_job_blocks', K_AMAX, n_el, 0, 0)
        self.pre_blocks = pblk
        self.pre_table = torch.from_numpy(pre.view(np.uint8).copy()).to(dev) if h3 else None
        self.word_of = {}
        for i, (p, p2, kind, N, K, aux, out_numel) in enumerate(self.specs):
            w2 = p2.data_ptr() if p2 is not None else 0
            if kind in (K_VADD, K_BPQ, K_GRUB):
                total, npad = out_numel, 0
            elif kind == K_GATES:
                npad = 16 * (N // aux) * ((aux + 15) // 16)
                total = npad * round_up(K, 16)
            elif kind in (K_GATES_H3, K_TRANS_H3):
                npad = 16 * (N // aux) * ((aux + 15) // 16) if kind == K_GATES_H3 else round_up(N, 16)
                total = L.query('gpe_packed_planes_size', npad, K)
                wi = [j for j, m_ in enumerate(mats) if self.specs[m_][0] is p][0]
                word = self.words[wi:wi + 1]
                w2 = word.data_ptr()
                self.word_of[(p.data_ptr(), kind)] = word
            else:
                npad = round_up(N, 16)
                total = L.query('gpe_packed_size', N, K)      # K filled up to the edge kernels' resident chunk count
            out = torch.empty(total, device=dev, dtype=F32)
            self.outs.append(out)
            tab[i] = (p.data_ptr(), w2, out.data_ptr(), total, blk,
                      p.stride(0) if p.dim() == 2 else 0, N, K, kind, npad, aux)
            blk += L.query('gpe_pack_job_blocks', kind, total, npad, K)
            key = (p.data_ptr(), kind)
            _PACKS[key] = (self._me, out, i)
            self.keys.append(key)
            self._keys_box.append(key)
        self.blocks = blk
        self.table = torch.from_numpy(tab.view(np.uint8).copy()).to(dev)
        # the same table without the fp16-plane jobs, for every arithmetic mode but f16x3 (their outputs and amax words are only
        # read by the f16x3 recurrences: ADVICE r4 — two launches and the plane arithmetic per step for nothing)
        keep = [i for i in range(len(self.specs)) if i not in h3]
        tab2 = tab[keep].copy()
        blk2 = 0
        for j in range(len(keep)):
            tab2[j]['first_block'] = blk2
            blk2 += L.query('gpe_pack_job_blocks', int(tab2[j]['kind']), int(tab2[j]['total']), int(tab2[j]['Npad']), int(tab2[j]['K']))
        self.table_noh3 = torch.from_numpy(tab2.view(np.uint8).copy()).to(dev) if h3 else self.table
        self.blocks_noh3, self.n_noh3 = (blk2, len(keep)) if h3 else (blk, len(self.specs))
        self.home = self._home()

    def _home(self):
        """where every source parameter lives right now: the job table holds raw pointers, so ANY parameter that moved
        (a partial re-homing, `p.data = ...` on a subset) must rebuild it"""
        return tuple((p.data_ptr(), p2.data_ptr() if p2 is not None else 0, p.device) for p, p2, *_ in self.specs)

    def refresh(self):
        """Called at the start of a model forward: one gpe_pack_multi launch (skipped only by a `frozen` plan whose parameters'
        version counters and WEIGHTS_EPOCH did not move)."""
        if not self.specs:
            return
        vers = self._versions()
        if self.specs[0][0].is_cuda:
            st = torch.cuda.current_stream(self.specs[0][0].device)
            if self._stream is not None and st != self._stream:
                st.wait_stream(self._stream)     # the packs / words the previous stream's launches still read are about to change
            self._stream = st
        if self.table is None or self.home != self._home():
            self._build()                      # first use, or a parameter moved (.to(device), arena re-homing)
        elif (self.frozen and self.vers == vers and self.epoch == WEIGHTS_EPOCH and
              self.h3_current == (self.pre_table is not None and L.get_math() == 'f16x3')):   # (a mode change re-runs the packs)
            return
        self.h3_current = self.pre_table is not None and L.get_math() == 'f16x3'

        def launches():
            if self.h3_current:
                self.words.zero_()
                L.call('gpe_pack_multi', self.pre_table, self.pre_table.numel() // 64, self.pre_blocks)
                L.call('gpe_pack_multi', self.table, len(self.specs), self.blocks)
            else:
                L.call('gpe_pack_multi', self.table_noh3, self.n_noh3, self.blocks_noh3)

        launches()
        self.vers, self.epoch = vers, WEIGHTS_EPOCH


def _planned(t, kind, t2=None):
    """The persistent packed operand of parameter `t` if a PackPlan owns one that is still current, else None."""
    hit = _PACKS.get((t.data_ptr(), kind))
    if hit is None:
        return None
    plan, out, i = hit
    plan = plan()
    if plan is None or plan.vers is None or plan.epoch != WEIGHTS_EPOCH:
        return None
    if plan.vers[i] != t._version + (t2._version if t2 is not None else 0):
        return None
    return out


# -------------------------------------------------------------------------------------------------
# gradient sink (optim.FlatArena): weight gradients are written straight into the arena
# -------------------------------------------------------------------------------------------------
_SINK = {}                 # param data_ptr -> (weakref to the arena, flat gradient view shaped like the parameter)


def _gbuf(param, shape=None):
    """Output buffer for the gradient of `param`: the arena view if the parameter is registered, else a new tensor."""
    hit = _SINK.get(param.data_ptr())
    if hit is not None and hit[0]() is not None:
        return hit[1]
    return torch.empty(param.shape if shape is None else shape, device=param.device, dtype=F32)


def _gret(param, buf):
    """What backward returns for `param`: None when the gradient already sits in the arena (and the arena is told, so that
    a gradient bucket can leave for the all-reduce), else the tensor itself."""
    hit = _SINK.get(param.data_ptr())
    arena = hit[0]() if hit is not None else None
    if arena is None:
        return buf
    arena.mark_written(param)
    return None


# -------------------------------------------------------------------------------------------------
# raw wrappers
# -------------------------------------------------------------------------------------------------
def pack_weight(w, transpose=False, col_scale=None):
    """w: [N,K] (nn.Linear layout).  transpose=True packs w^T (operand [K_src_cols][K_src_rows])."""
    _dev_check(w)
    assert w.dim() == 2 and w.stride(1) == 1
    if col_scale is None:
        hit = _planned(w, K_TRANS if transpose else K_PLAIN)
        if hit is not None:
            return hit
    N, K = (w.shape[1], w.shape[0]) if transpose else (w.shape[0], w.shape[1])
    wp = torch.empty(L.query('gpe_packed_size', N, K), device=w.device, dtype=F32)
    L.call('gpe_pack_weight', w, w.stride(0), N, K, int(transpose), col_scale, wp)
    return wp


def pack_gates(w, H, G=4):
    """gate-interleaved pack of a recurrent weight [G*H, K]: one column block = the G gates of 16 units."""
    hit = _planned(w, K_GATES)
    if hit is not None:
        return hit
    wp = torch.empty(L.query('gpe_packed_ngates_size', H, G, w.shape[1]), device=w.device, dtype=F32)
    L.call('gpe_pack_weight_ngates', w, w.stride(0), H, G, w.shape[1], wp)
    return wp


def planned_planes(w, kind):
    """(fp16 plane pack, amax word) of a recurrent weight when a PackPlan owns them (f16x3 arithmetic of the recurrences), else
    (None, None): the wavefront kernels then run the exact fp32 instruction."""
    hit = _PACKS.get((w.data_ptr(), kind))
    if hit is None:
        return None, None
    plan = hit[0]()
    out = _planned(w, kind)
    if plan is None or out is None or not plan.h3_current:   # (planes are only rebuilt by a refresh that ran in f16x3 mode)
        return None, None
    return out, plan.word_of.get((w.data_ptr(), kind))


def bias_sum(b_ih, b_hh):
    hit = _planned(b_ih, K_VADD, b_hh)
    if hit is not None:
        return hit
    out = torch.empty_like(b_ih)
    L.call('gpe_add', b_ih, b_hh, out, b_ih.numel())
    return out


def gru_bias(b_ih, b_hh, H):
    hit = _planned(b_ih, K_GRUB, b_hh)
    if hit is not None:
        return hit
    out = b_ih.detach().clone()
    L.call('gpe_add', b_ih, b_hh, out, 2 * H)
    return out


def _ptr_array(tensors):
    """host array of device pointers (the `const void* const*` arguments of the wavefront recurrences)."""
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])


def edge_first_operands(W1, b1):
    """-> (packed [W1a-W1b ; W1b], its packed transpose, bias [b1 | 0]) for the per-point P|Q projection."""
    H, C = W1.shape[0], W1.shape[1] // 2
    a, b, c = _planned(W1, K_PQ), _planned(W1, K_PQT), _planned(b1, K_BPQ)
    if a is not None and b is not None and c is not None:
        return a, b, c
    wpq = torch.empty(2 * H, C, device=W1.device, dtype=F32)
    bpq = torch.empty(2 * H, device=W1.device, dtype=F32)
    L.call('gpe_w1_split', W1, W1.stride(0), b1, H, C, wpq, C, bpq)
    return pack_weight(wpq), pack_weight(wpq, transpose=True), bpq


def fold_bias(w, bias, t):
    out = torch.empty(w.shape[0], device=w.device, dtype=F32)
    L.call('gpe_fold_bias', w, w.stride(0), w.shape[0], w.shape[1], bias, t, out)
    return out


def linear_raw(a_desc, wp, bias, M, N, K, y_desc, act=0, addend_desc=None):
    ad = addend_desc if addend_desc is not None else (None, 0, 0, 0)
    L.call('gpe_linear', a_desc[0], a_desc[1], a_desc[2], a_desc[3], wp, bias,
           ad[0], ad[1], ad[2], ad[3], y_desc[0], y_desc[1], y_desc[2], y_desc[3], M, N, K, act)


def redgemm_raw(u_desc, v_desc, rows, Mg, Ng, want_colsum=True, accumulate_into=None, v_shift=None, out=None):
    """G[Mg,Ng] = sum_r U[r,:]^T (V[r,:] - v_shift), colsum[Mg] = sum_r U[r,:].  `out` = (G, colsum) buffers to fill."""
    dev = u_desc[0].device
    if accumulate_into is not None:
        G, cs = accumulate_into
    elif out is not None:
        G, cs = out
    else:
        G = torch.empty(Mg, Ng, device=dev, dtype=F32)
        cs = torch.empty(Mg, device=dev, dtype=F32) if want_colsum else None
    acc = int(accumulate_into is not None)
    for n0 in range(0, Ng, 256):          # the kernel keeps <= 256 V columns resident per pass
        nb = min(256, Ng - n0)
        ws = torch.empty(L.query('gpe_redgemm_ws', Mg, nb), device=dev, dtype=F32)
        L.call('gpe_redgemm', u_desc[0], u_desc[1], u_desc[2], u_desc[3],
               v_desc[0][..., n0:], v_desc[1], v_desc[2], v_desc[3], None if v_shift is None else v_shift[n0:],
               rows, Mg, nb, G[:, n0:], G.stride(0), cs if n0 == 0 else None, ws, acc)
    return G, cs


def knn(x, B, N, k, want_global=False, order=None, want_order=False):
    """x: [B*N, C] rows (ld = x.stride(0)).  -> int32 [B, N, k] local neighbour indices (and, optionally, the same
    graph as global row numbers b*N + idx, the form the gather kernels consume).
    Replaces torch_cluster.knn under DynamicEdgeConv (nn/net_blocks.py:127-135,174).
    order (int32 [B, N], optional): a locality order of the points, a speed hint for the wide-feature search (include/gpe_hip.h
    gpe_knn); want_order: also return the order this search worked in (the xyz search's Morton-curve order; else the identity)."""
    _dev_check(x)
    idx = torch.empty(B, N, k, device=x.device, dtype=torch.int32)
    jg = torch.empty(B, N, k, device=x.device, dtype=torch.int32) if want_global else None
    oo = torch.empty(B, N, device=x.device, dtype=torch.int32) if want_order else None
    if order is not None:
        assert order.dtype == torch.int32 and order.shape == (B, N) and order.is_contiguous() and order.device == x.device
    nws = L.query('gpe_knn_ws_bytes', B, N, x.shape[1], k)
    L.call('gpe_knn', x, B, N, x.shape[1], x.stride(0), k, idx, jg, order, oo, _workspace(nws, x.device), nws)
    out = (idx, jg) if want_global else (idx,)
    if want_order:
        out = out + (oo,)
    return out if len(out) > 1 else out[0]


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


def bn_bwd_coef(part, nblk, stats, C, count, gamma=None, beta=None, training=True):
    """-> coef [4][C] = {s, c1, k2, mean}, dgamma, dbeta (written into the parameters' gradient buffers).  In eval mode the
    statistics are constants, so the projection terms c1 / k2 vanish (dz = s * dy under the ReLU mask)."""
    dev = stats.device
    coef = torch.empty(4, C, device=dev, dtype=F32)
    dg = _gbuf(gamma) if gamma is not None else torch.empty(C, device=dev, dtype=F32)
    db = _gbuf(beta) if beta is not None else torch.empty(C, device=dev, dtype=F32)
    L.call('gpe_bn_bwd_coef', part, nblk, stats, C, float(count), coef, dg, db)
    if not training:
        coef[1:3].zero_()
    return coef, dg, db


# -------------------------------------------------------------------------------------------------
# nn.Linear
# -------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) on the MFMA row-GEMM; backward = row-GEMM (dx) + reduce-GEMM (dW, db).
    Replaces torch.nn.Linear at nn/net_blocks.py:158,187,397 and nn/nets.py:128-130,153,229-233.
    x: [M, K] rows, or a strided [R, T, K] view (e.g. the top-layer h history of a recurrent stack) -> y [R*T, N]."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _dev_check(x)
        K = x.shape[-1]
        M = x.numel() // K
        N = weight.shape[0]
        y = torch.empty(M, N, device=x.device, dtype=F32)
        linear_raw(_rows3d(x) if x.dim() == 3 else _rows2d(x), pack_weight(weight), bias, M, N, K, _rows2d(y))
        ctx.save_for_backward(x, weight, bias)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, bias = ctx.saved_tensors
        gy = gy.contiguous()
        K = x.shape[-1]
        M = x.numel() // K
        N = weight.shape[0]
        xd = _rows3d(x) if x.dim() == 3 else _rows2d(x)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(*x.shape, device=x.device, dtype=F32)
            linear_raw(_rows2d(gy), pack_weight(weight, transpose=True), None, M, K, N, (gx, K, 0, 0))
        if ctx.needs_input_grad[1] or bias is not None:
            # (a leaf of the backward pass: on the side stream when weight and bias gradients both land in the arena)
            with side_grads([weight, bias] if bias is not None else [], [gy, x]):
                gw = _gbuf(weight)
                gb = _gbuf(bias) if bias is not None else torch.empty(N, device=x.device, dtype=F32)
                redgemm_raw(_rows2d(gy), xd, M, N, K, out=(gw, gb))
                gw = _gret(weight, gw)
                gb = _gret(bias, gb) if bias is not None else None
        return gx, gw, gb


def linear(x, weight, bias=None):
    return LinearFn.apply(x, weight, bias)


# -------------------------------------------------------------------------------------------------
# global pooling
# -------------------------------------------------------------------------------------------------
POOL_MODES = {'mean': 0, 'max': 1, 'add': 2}


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


class SegmentPoolFn(torch.autograd.Function):
    """torch_geometric.nn.global_max_pool / global_add_pool over equal-sized clouds (nn/net_blocks.py:145-150)."""

    @staticmethod
    def forward(ctx, x, B, N, mode):
        _dev_check(x)
        C = x.shape[1]
        y = torch.empty(B, C, device=x.device, dtype=F32)
        arg = torch.empty(B, C, device=x.device, dtype=torch.int32) if mode == 1 else None
        L.call('gpe_segment_pool_fwd', x, x.stride(0), B, N, C, mode, y, C, arg)
        ctx.dims = (B, N, C, mode)
        ctx.arg = arg
        return y

    @staticmethod
    def backward(ctx, gy):
        B, N, C, mode = ctx.dims
        gy = gy.contiguous()
        gx = torch.empty(B * N, C, device=gy.device, dtype=F32)
        L.call('gpe_segment_pool_bwd', gy, C, ctx.arg, B, N, C, mode, gx, C)
        return gx, None, None, None


def segment_mean(x, B, N):
    return SegmentMeanFn.apply(x, B, N)


def segment_max(x, B, N):
    return SegmentPoolFn.apply(x, B, N, 1)


def segment_add(x, B, N):
    return SegmentPoolFn.apply(x, B, N, 2)


segment_mean.pool_mode, segment_max.pool_mode, segment_add.pool_mode = 0, 1, 2


# -------------------------------------------------------------------------------------------------
# EdgeConv layer
# -------------------------------------------------------------------------------------------------
_HALF_ACT_GUARD = 'fallback'
# graph.StepGraph: CAPTURE is the StepGraph whose hipGraph capture is in progress (host reads are impossible then), HOST_DRAWN the
# provider of host-drawn device tensors (start states, dropout masks) while a StepGraph drives the step — else None
CAPTURE = None
HOST_DRAWN = None


def set_half_act_guard(mode):
    """What watches the fp16 storage of the aggregated block's activation (f16x3 mode, lazy dz3; DESIGN.md 8 row g).  The stored
    copy is clamped at 65504 and flushes positives below 6e-8; the forward measures the activation's true largest magnitude
    (an amax word on the device) — this guard reads it:
      'fallback' (default)  asynchronously, one step late and without a host synchronisation: a layer whose activation reached
                            the fp16 limit switches to fp32 storage + the eager dz3 pass for every later step, with a RuntimeWarning
                            (the step that tripped the guard ran with clamped values: its gradient elements of the clamped
                            activations are off);
      'strict'              synchronously inside the forward (one host read per EdgeConv layer and step): the launch is repeated
                            with fp32 storage before anything consumes the clamped rows — exact at every step;
      'off'                 no check (rounds 3 - 4 behaviour).
    Returns the previous mode."""
    global _HALF_ACT_GUARD
    if mode not in ('fallback', 'strict', 'off'):
        raise ValueError("half-activation guard mode must be 'fallback', 'strict' or 'off'")
    prev, _HALF_ACT_GUARD = _HALF_ACT_GUARD, mode
    return prev


if os.environ.get('GPE_HALF_ACT_GUARD'):                 # (validated: a typo raises instead of silently meaning 'fallback')
    set_half_act_guard(os.environ['GPE_HALF_ACT_GUARD'])


class HalfActGuard:
    """Per-layer state of set_half_act_guard: owned by the module that owns the layer (net_blocks.DynamicEdgeConv)."""
    LIMIT = 65504.0

    def __init__(self):
        self.disabled = False
        self.last_amax = None
        self._pending = None

    def __getstate__(self):                              # copies / pickles of the owning module carry the decision, not the
        return {'disabled': self.disabled, 'last_amax': self.last_amax, '_pending': None}    # in-flight read (pinned buffer + event)

    def __deepcopy__(self, memo):
        g = HalfActGuard()
        g.disabled, g.last_amax = self.disabled, self.last_amax
        return g

    @staticmethod
    def _value(host):
        return float(np.array([int(host[0])], dtype=np.int32).view(np.float32)[0])

    def _trip(self, v):
        self.disabled = True
        warnings.warn('f16x3: the aggregated EdgeConv activation reached %.3g (fp16 storage clamps at 65504): this layer keeps it '
                      'in fp32 and forms dz3 eagerly from now on (gpe_amd.ops.set_half_act_guard)' % v, RuntimeWarning, stacklevel=3)

    def allow(self):
        """May the coming forward store the activation in fp16?  Polls the previous step's word without blocking."""
        if _HALF_ACT_GUARD == 'off':
            return True
        if CAPTURE is not None:                          # no polling inside a capture: the decision as it stands
            return not self.disabled
        if self._pending is not None and self._pending[1].query():
            v = self._value(self._pending[0])
            self._pending = None
            self.last_amax = v
            if not v < self.LIMIT:                      # (also true for NaN)
                self._trip(v)
        return not self.disabled

    def watch(self, word):
        """After the forward launch that filled `word` (int32[1], the activation's amax bits).  'strict': returns False when the
        launch must be repeated with fp32 storage."""
        if _HALF_ACT_GUARD == 'off':
            return True
        if CAPTURE is not None:                          # the StepGraph reads this word after every replay
            CAPTURE.guards.append((self, word))
            return True
        if _HALF_ACT_GUARD == 'strict':
            v = self._value(word.cpu())
            self.last_amax = v
            if not v < self.LIMIT:
                self._trip(v)
                return False
            return True
        if self._pending is not None:
            # an earlier step's read has not been consumed: look at it now if it has landed, and otherwise KEEP it — the oldest unread
            # word is the one that says when the limit was first reached (a newer read must not overwrite it: ADVICE r5)
            self.allow()
            if self._pending is not None:
                return True
        host = torch.empty(1, dtype=torch.int32, pin_memory=True)
        host.copy_(word, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pending = (host, ev)
        return True


class EdgeConvFn(torch.autograd.Function):
    """One DynamicEdgeConv(MLP([2C, H, .., H, F]), k, aggr) layer (nn/net_blocks.py:43-47,124-135,174):
    kNN graph on the input features -> per-edge [Linear->ReLU->BatchNorm] x nb -> max / mean / add over the k messages.

    Pipeline (training): kNN | per-point P,Q GEMM (block 0 split: W1.[x_i, x_j-x_i] = P_i + Q_j) | gather+stats(a_0) |
    for every further block: fold the previous BatchNorm into this Linear, fused (gather|dense)+GEMM+ReLU(+stats)
    (+max/min over the point's messages in the last block) | last BatchNorm applied AFTER the aggregation.
    Saved for backward: the graph, PQ, the post-ReLU activations a_1..a_{nb-1} [E, .], the aggregates, the BN stat blocks.
    nb = EConv_hidden_depth + 1 >= 2; the shipped configs use nb = 3, aggr = 'max'."""

    @staticmethod
    def forward(ctx, x, B, N, k, training, eps, momentum, nb, aggr, guard, order, *tensors):
        # order (int32 [B, N] or None): a locality order of the points — the previous layer's curve order — handed to the graph
        # search as a speed hint (include/gpe_hip.h gpe_knn); third output: the order this layer's search worked in
        _dev_check(x)
        ctx.set_materialize_grads(False)                   # (no zero tensors for the gradients of idx / order_out)
        dev = x.device
        BN, C = x.shape
        assert BN == B * N
        params, bufs = tensors[:4 * nb], tensors[4 * nb:]
        Ws = [params[4 * l] for l in range(nb)]
        widths = [w.shape[0] for w in Ws]
        H0 = widths[0]
        assert Ws[0].shape[1] == 2 * C
        if H0 % 4 or H0 > 256:
            raise ValueError('EdgeConvFn (the fused P|Q path) needs a first-block width that is a multiple of 4 and <= 256 '
                             '(got %d): use ops.edge_conv_general' % H0)
        E = BN * k
        _LAST_EDGES[0] = E                                 # (what side_grads takes for the size of the step)
        nblk = L.query('gpe_stats_blocks')
        wpq_p, _, bpq = edge_first_operands(Ws[0], params[1])
        ews, ews_n = edge_workspace(B, N, k, max(round_up(widths[-1], 4), 2 * H0), dev)
        words = f16x3_words(2 * nb + 1, E, dev)           # (+ 1: max |s g| of the layer-output gradient, lazy dz3)
        # The [P|Q] projection (and its f16x3 bound) needs the layer input only — not the graph: on a GPU-bound step it runs on the
        # side stream beside the graph search, whose kernels leave most of the chip's issue slots empty
        fork = bool(SIDE_GRADS and SIDE_PQ and E >= SIDE_MIN_EDGES and not torch.cuda.is_current_stream_capturing())
        if fork:
            main, side = torch.cuda.current_stream(), _side_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                PQ = torch.empty(BN, 2 * H0, device=dev, dtype=F32)
                linear_raw(_rows2d(x), wpq_p, bpq, BN, 2 * H0, C, _rows2d(PQ))
                if words is not None:
                    sws, sws_n = edge_workspace(B, N, k, max(round_up(widths[-1], 4), 2 * H0), dev)      # (the side stream's own workspace)
                    L.call('gpe_edge_pq_amax', PQ, 2 * H0, H0, BN, _word(words, 0), sws, sws_n)
            x.record_stream(side)
            if words is not None:
                words.record_stream(side)
            idx, jg, order_out = knn(x, B, N, k, want_global=True, order=order, want_order=True)
            main.wait_stream(side)
            PQ.record_stream(main)
        else:
            idx, jg, order_out = knn(x, B, N, k, want_global=True, order=order, want_order=True)
            PQ = torch.empty(BN, 2 * H0, device=dev, dtype=F32)
            linear_raw(_rows2d(x), wpq_p, bpq, BN, 2 * H0, C, _rows2d(PQ))
            if words is not None:
                L.call('gpe_edge_pq_amax', PQ, 2 * H0, H0, BN, _word(words, 0), ews, ews_n)

        def stats_of(part, l):
            g, be = params[4 * l + 2], params[4 * l + 3]
            rm, rv, nbt = bufs[3 * l: 3 * l + 3]
            if training:
                return bn_finalize(part, nblk, widths[l], E, g, be, eps, momentum, rm, rv, nbt)
            return bn_from_running(rm, rv, g, be, eps)

        # caller-owned state of the edge calls: one workspace, and in f16x3 mode the amax words of this layer's tensors —
        # [0] the bound of relu(P_i + Q_j), [l] activation a_l, [nb + l] dz_l (backward)
        part = torch.empty(nblk, 2, H0, device=dev, dtype=torch.float64) if training else None
        if training:
            L.call('gpe_edge_gather_stats', PQ, 2 * H0, H0, jg, B, N, k, part)
        stats = [stats_of(part, 0)]
        acts = [None]
        mx = mn = amx = amn = abar = None
        for l in range(1, nb):
            Cin, Cout = widths[l - 1], widths[l]
            ldo = round_up(Cout, 4)
            last = l == nb - 1
            # row g: the aggregated block's activation is kept in fp16 when its backward will form dz3 lazily — its
            # only readers then (include/gpe_hip.h "out_half"; gradients move by 2e-6 / 5e-6 of their maximum)
            half = bool(last and training and aggr == 'max' and words is not None and nb >= 3 and
                        L.query('gpe_edge_lazy_dz3_ok', B, N, k, Cout, Cin) == 1 and (guard is None or guard.allow()))
            a = torch.empty(E, ldo, device=dev, dtype=torch.float16 if half else F32)
            part = torch.empty(nblk, 2, Cout, device=dev, dtype=torch.float64) if training else None
            agg = int(last and aggr == 'max')
            if agg:
                mx = torch.empty(BN, ldo, device=dev, dtype=F32)
                mn = torch.empty(BN, ldo, device=dev, dtype=F32)
                amx = torch.empty(BN, ldo, device=dev, dtype=torch.uint8)
                amn = torch.empty(BN, ldo, device=dev, dtype=torch.uint8)
            w_out = _word(words, l)                        # (the last activation's word feeds the bound of a lazily formed dz)
            # the BatchNorm fold of this block's Linear: pack + folded bias (+ the packed weight's f16x3 amax) in one launch
            wp, bf = pack_fold(Ws[l], params[4 * l + 1], stats[l - 1], ews if words is not None else None, ews_n, w_out)
            wr = 2 if words is not None else 0             # bit 1 of out_half: the weight's scale is in the workspace
            if l == 1:
                L.call('gpe_edge_mlp_fwd', 0, PQ, 2 * H0, jg, None, 0, B, N, k, Cin, Cout, wp, bf, a, ldo, part,
                       agg, mx, mn, amx, amn, ldo, _word(words, 0), w_out, ews, ews_n, wr)
            else:
                prev = acts[l - 1]
                L.call('gpe_edge_mlp_fwd', 1, None, 0, None, prev, prev.stride(0), B, N, k, Cin, Cout, wp, bf, a, ldo,
                       part, agg, mx, mn, amx, amn, ldo, _word(words, l - 1), w_out, ews, ews_n, int(half) | wr)
                if half and guard is not None and not guard.watch(w_out):
                    # 'strict' guard: the activation does not fit fp16 — the same launch once more with fp32 rows (nothing has
                    # consumed the clamped copy; statistics, maxima and the amax word are rewritten with the same values)
                    a = torch.empty(E, ldo, device=dev, dtype=F32)
                    L.call('gpe_edge_mlp_fwd', 1, None, 0, None, prev, prev.stride(0), B, N, k, Cin, Cout, wp, bf, a, ldo,
                           part, agg, mx, mn, amx, amn, ldo, _word(words, l - 1), w_out, ews, ews_n, 0)
            acts.append(a)
            stats.append(stats_of(part, l))
        Fo = widths[-1]
        ldF = round_up(Fo, 4)
        out = torch.empty(BN, ldF, device=dev, dtype=F32)[:, :Fo]
        if aggr == 'max':
            L.call('gpe_edge_finish', mx, mn, ldF, stats[-1], BN, Fo, out, ldF)
        else:
            # mean / add: BatchNorm is affine, so it commutes with the sum over the k messages
            abar = torch.empty(BN, ldF, device=dev, dtype=F32)
            L.call('gpe_edge_sum_k', acts[-1], ldF, BN, k, Fo, abar, ldF)
            L.call('gpe_scale', abar, 1.0 / k, abar, abar.numel())
            a_s, t_s = (1.0, 1.0) if aggr == 'mean' else (float(k), float(k))
            L.call('gpe_bn_apply_scaled', abar, ldF, stats[-1], BN, Fo, a_s, t_s, out, ldF)
        ctx.dims = (B, N, k, C, nb, aggr, training)
        ctx.widths = widths
        ctx.done = False
        ctx.words = words
        ctx.rev_job = None
        if fork and training:
            if order is None:
                del _IDLE_JOBS[:]                            # (a new forward pass: whatever an abandoned one left behind is dropped)
            ctx.rev_job = {'idx': idx, 'rev': None, 'event': None}
            _IDLE_JOBS.append(ctx.rev_job)
        ctx.save_for_backward(x, idx, jg, PQ, *params, *acts[1:], *stats,
                              *([mx, mn, amx, amn] if aggr == 'max' else [abar]))
        ctx.mark_non_differentiable(idx, order_out)
        return out, idx, order_out

    @staticmethod
    def backward(ctx, g_out, _g_idx, _g_order=None):
        if ctx.done:
            raise RuntimeError('EdgeConvFn.backward ran twice on the same graph: the stored activations are overwritten '
                               'in place by the first pass (retain_graph is not supported)')
        B, N, k, C, nb, aggr, training = ctx.dims
        if g_out is None:                                  # only the graph / the order were used downstream
            return (None,) * (11 + 4 * nb + 3 * nb)
        ctx.done = True
        widths = ctx.widths
        sv = ctx.saved_tensors
        x, idx, jg, PQ = sv[:4]
        params = sv[4: 4 + 4 * nb]
        acts = [None] + list(sv[4 + 4 * nb: 4 + 4 * nb + (nb - 1)])
        stats = list(sv[4 + 5 * nb - 1: 4 + 6 * nb - 1])
        tail = sv[4 + 6 * nb - 1:]
        Ws = [params[4 * l] for l in range(nb)]
        dev = x.device
        BN, E = B * N, B * N * k
        H0, Fo = widths[0], widths[-1]
        ldF = round_up(Fo, 4)
        if g_out.stride(1) != 1:
            g_out = g_out.contiguous()
        ldg = g_out.stride(0)
        grads = [None] * (4 * nb)
        words = ctx.words                  # f16x3 amax words of this layer (None in the other modes): see forward
        ews, ews_n = edge_workspace(B, N, k, max(ldF, 2 * H0), dev)
        # the transposed graph of the gather backward needs the forward's graph only: on a GPU-bound step it is built on the side stream
        # now and finds its CUs in the gaps between the edge kernels (main waits for it in front of gpe_edge_pull_dq)
        rev, rev_event = None, None
        job = getattr(ctx, 'rev_job', None)
        if job is not None and job['rev'] is not None:
            rev, rev_event = job['rev'], job['event']       # built during the forward's idle stretch (run_idle_jobs)
        elif SIDE_GRADS and SIDE_PQ and E >= SIDE_MIN_EDGES and not torch.cuda.is_current_stream_capturing():
            if job is not None and job in _IDLE_JOBS:
                _IDLE_JOBS.remove(job)
            main, side = torch.cuda.current_stream(), _side_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                rev = knn_reverse(idx)
            idx.record_stream(side)

        # ---- last block: BatchNorm applied after the aggregation -----------------------------------------
        psb = L.query('gpe_point_sums_blocks')
        part = torch.empty(psb, 2, Fo, device=dev, dtype=torch.float64)
        g_last, be_last = params[4 * (nb - 1) + 2], params[4 * (nb - 1) + 3]
        a_last = acts[-1]
        lz = (None, 0, None, None, 0, None)                # lazy dz3 arguments of the two consumers (off)
        if aggr == 'max':
            mx, mn, amx, amn = tail
            # (the forward stored a3 in fp16 exactly when it found the lazy path open; should the mode have changed since — a debug
            # flag, another arithmetic — the eager pass below runs on an fp32 copy)
            lazy = a_last.dtype == torch.float16 and L.query('gpe_edge_lazy_dz3_ok', B, N, k, Fo, widths[-2]) == 1
            if a_last.dtype == torch.float16 and not lazy:
                a_last = a_last.float()
            L.call('gpe_edge_bwd_point_sums', g_out, ldg, mx, mn, ldF, stats[-1], BN, Fo, part,
                   _word(words, 2 * nb) if lazy else None)
            coef, dg, dbe = bn_bwd_coef(part, psb, stats[-1], Fo, E, g_last, be_last, training)
            if lazy:
                # f16x3, k = 16: dz of the aggregated block is never materialised — the weight-gradient reduce-GEMM and the
                # propagation below form it from the stored activation while staging it (the in-place pass is 1.3 GB at cfg 2).
                # Its fp16 scale comes from a bound of |dz|: max |s g| (measured by the point sums above) + the coefficient terms.
                L.call('gpe_edge_dz3_bound', _word(words, 2 * nb), coef, Fo, _word(words, nb - 1), _word(words, 2 * nb - 1))
                lz = (g_out, ldg, amx, amn, ldF, coef)
            else:
                # dz in place over the stored activation (one coalesced pass)
                L.call('gpe_edge_dz3', a_last, ldF, g_out, ldg, amx, amn, ldF, coef, B, N, k, Fo, _word(words, 2 * nb - 1))
        else:
            # every message carries dy_e = w * g_i (w = 1/k mean, 1 add): sums over edges = (w*k) * per-point sums at
            # the mean activation of the point
            abar, = tail
            gs = g_out
            if aggr == 'add':
                gs = torch.empty(BN, Fo, device=dev, dtype=F32)
                L.call('gpe_scale', g_out.contiguous(), float(k), gs, gs.numel())
            L.call('gpe_edge_bwd_point_sums', gs, gs.stride(0), abar, abar, ldF, stats[-1], BN, Fo, part, None)
            coef, dg, dbe = bn_bwd_coef(part, psb, stats[-1], Fo, E, g_last, be_last, training)
            L.call('gpe_edge_dz3_all', a_last, ldF, g_out, ldg, 1.0 / k if aggr == 'mean' else 1.0, coef, B, N, k, Fo,
                   _word(words, 2 * nb - 1))
        grads[4 * (nb - 1) + 2], grads[4 * (nb - 1) + 3] = _gret(g_last, dg), _gret(be_last, dbe)

        # ---- blocks nb-1 .. 1: weight gradient (centred reduce-GEMM) -> previous BN coefficients -> propagate ----
        ws = torch.empty(max(L.query('gpe_redgemm_ws', widths[l], widths[l - 1]) for l in range(1, nb)), device=dev,
                         dtype=F32)
        dPQ = torch.empty(BN, 2 * H0, device=dev, dtype=F32)
        dz = a_last
        for l in range(nb - 1, 0, -1):
            Cl, Cp = widths[l], widths[l - 1]
            W, b = Ws[l], params[4 * l + 1]
            G = torch.empty(Cl, Cp, device=dev, dtype=F32)
            db = _gbuf(b)
            w_dz = _word(words, nb + l)                    # dz_l: written by dz3 (l = nb - 1) or by the propagation below
            if l == 1:
                L.call('gpe_edge_redgemm', dz, dz.stride(0), 0, None, 0, PQ, 2 * H0, jg, stats[0][0], B, N, k, Cl, Cp, G,
                       Cp, db, ws, w_dz, _word(words, 0), ews, ews_n, None, 0, None, None, 0, None)
            else:
                prev = acts[l - 1]
                L.call('gpe_edge_redgemm', dz, dz.stride(0), 1, prev, prev.stride(0), None, 0, None, stats[l - 1][0],
                       B, N, k, Cl, Cp, G, Cp, db, ws, w_dz, _word(words, l - 1), ews, ews_n,
                       *(lz if l == nb - 1 else (None, 0, None, None, 0, None)))
            sums = torch.empty(1, 2, Cp, device=dev, dtype=torch.float64)
            dW = _gbuf(W)
            L.call('gpe_bn_bwd_from_G', G, Cp, db, W, W.stride(0), Cl, Cp, stats[l - 1], sums, dW, Cp)
            gp, bp = params[4 * (l - 1) + 2], params[4 * (l - 1) + 3]
            coef_p, dgp, dbp = bn_bwd_coef(sums, 1, stats[l - 1], Cp, E, gp, bp, training)
            grads[4 * l], grads[4 * l + 1] = _gret(W, dW), _gret(b, db)
            grads[4 * (l - 1) + 2], grads[4 * (l - 1) + 3] = _gret(gp, dgp), _gret(bp, dbp)
            wt = pack_weight(W, transpose=True)
            if l == 1:
                # dz_0 = (a_0>0) ? s_0*(dz_1 W_1) - c1 - (a_0-mean)*k2 : 0 with a_0 re-gathered; dP = sum over slots.
                # In place over dz_1's buffer when the row pitch fits, else a fresh [E, H0]
                dst = dz if dz.stride(0) == H0 else torch.empty(E, H0, device=dev, dtype=F32)
                L.call('gpe_edge_mlp_bwd', dz, dz.stride(0), 1, PQ, 2 * H0, jg, B, N, k, Cl, Cp, wt, coef_p, dst, H0,
                       dPQ, 2 * H0, w_dz, None, ews, ews_n, None, 0, None, None, 0, None)
                dz = dst
            else:
                prev = acts[l - 1]
                L.call('gpe_edge_mlp_bwd', dz, dz.stride(0), 0, None, 0, None, B, N, k, Cl, Cp, wt, coef_p, prev,
                       prev.stride(0), None, 0, w_dz, _word(words, nb + l - 1), ews, ews_n,
                       *(lz if l == nb - 1 else (None, 0, None, None, 0, None)))
                dz = prev

        # ---- block 0: gather backward = deterministic pull through the transposed graph -----------------
        if rev is not None:
            if rev_event is not None:
                torch.cuda.current_stream().wait_event(rev_event)
            else:
                torch.cuda.current_stream().wait_stream(_side_stream())
            rev_off, rev_edge = rev
            rev_off.record_stream(torch.cuda.current_stream()); rev_edge.record_stream(torch.cuda.current_stream())
        else:
            rev_off, rev_edge = knn_reverse(idx)
        L.call('gpe_edge_pull_dq', dz, dz.stride(0), rev_off, rev_edge, B, N, k, H0, dPQ[:, H0:], 2 * H0)
        dWpq, dbpq = redgemm_raw(_rows2d(dPQ), _rows2d(x), BN, 2 * H0, C, want_colsum=True)
        W1, b1 = Ws[0], params[1]
        dW1 = _gbuf(W1)
        L.call('gpe_w1_grad_from_pq', dWpq, C, H0, C, dW1, 2 * C)
        db1 = _gbuf(b1)
        db1.copy_(dbpq[:H0])
        grads[0], grads[1] = _gret(W1, dW1), _gret(b1, db1)
        gx = None
        if ctx.needs_input_grad[0]:
            _, wpq_t, _ = edge_first_operands(W1, b1)
            gx = torch.empty(BN, C, device=dev, dtype=F32)
            linear_raw(_rows2d(dPQ), wpq_t, None, BN, C, 2 * H0, _rows2d(gx))
        return (gx, None, None, None, None, None, None, None, None, None, None, *grads, *([None] * (3 * nb)))


class EdgeInputsFn(torch.autograd.Function):
    """cat[x_i, x_j - x_i] per edge of the kNN graph, materialised [E, 2C] — the general formulation of the DynamicEdgeConv
    message input (nn/net_blocks.py:124-135), used by edge_conv_general for first-block widths the fused P|Q path does not
    take."""

    @staticmethod
    def forward(ctx, x, idx, jg, B, N, k):
        _dev_check(x)
        C = x.shape[1]
        ld = round_up(2 * C, 4)
        out = torch.empty(B * N * k, ld, device=x.device, dtype=F32)
        L.call('gpe_edge_inputs_fwd', x, x.stride(0), C, jg, B * N, k, out, ld)
        ctx.save_for_backward(idx)
        ctx.dims = (B, N, k, C)
        return out[:, :2 * C]

    @staticmethod
    def backward(ctx, g):
        idx, = ctx.saved_tensors
        B, N, k, C = ctx.dims
        if g.stride(1) != 1:
            g = g.contiguous()
        rev_off, rev_edge = knn_reverse(idx)
        gx = torch.empty(B * N, C, device=g.device, dtype=F32)
        L.call('gpe_edge_inputs_bwd', g, g.stride(0), C, rev_off, rev_edge, B, N, k, gx, C)
        return gx, None, None, None, None, None


def edge_conv_general(x, B, N, k, training, eps, momentum, nb, aggr, tensors):
    """DynamicEdgeConv for first-block widths outside the fused path's menu (EConv_hidden % 4 != 0 or > 256): kNN graph ->
    explicit [x_i, x_j - x_i] rows -> the edge MLP as a dense MLP over the E message rows (DenseMLPFn: the same kernels, every
    BatchNorm incl. the last applied per message) -> max / mean / add over each point's k messages.  Same arithmetic as the
    reference formulation; several times slower than EdgeConvFn (no P|Q split, E-row first block, stored per-edge outputs)."""
    idx, jg = knn(x, B, N, k, want_global=True)
    msg_in = EdgeInputsFn.apply(x, idx, jg, B, N, k)
    msg = DenseMLPFn.apply(msg_in, training, eps, momentum, nb, *tensors)
    if aggr == 'max':
        out = SegmentPoolFn.apply(msg, B * N, k, 1)
    elif aggr == 'add':
        out = SegmentPoolFn.apply(msg, B * N, k, 2)
    else:
        out = SegmentMeanFn.apply(msg, B * N, k)
    return out, idx


# -------------------------------------------------------------------------------------------------
# recurrent stacks (LSTM / GRU)
# -------------------------------------------------------------------------------------------------
RNN_WS_HOOK = None         # measurement aid (scripts/rnn_trace.py): called with ('fwd' | 'bwd', workspace tensor) of every recurrence


class RNNStackFn(torch.autograd.Function):
    """torch.nn.LSTM / torch.nn.GRU (batch_first, n_layers, no dropout) as used by the reference's decoders / encoder
    (nn/net_blocks.py:336-497): returns (top-layer outputs [Bn, T, H] as a view of the h history, h_T [L, Bn, H],
    c_T [L, Bn, H] or None).

    x is either [Bn, In] — the SAME input at every step (decoders feed the encoding T times, :388,430,483): layer 0's input
    projection is then computed once — or a sequence [Bn, T, In] (one GEMM over all steps).  The recurrence itself runs in
    WAVEFRONT order (csrc/gpe_rnn_wave.hip): all cells (layer l, step t) of an anti-diagonal l + t = d are independent, so a
    diagonal is one fused launch forward (gate GEMMs on h_{l,t-1} and h_{l-1,t} + cell update) and two launches backward
    (split-K dh products of all cells, pointwise cell backward): T + L - 1 dependent steps instead of T * L.  Weight
    gradients are reduce-GEMMs over all Bn*T rows afterwards.  Gate orders follow torch: LSTM i,f,g,o; GRU r,z,n.
    Start states may carry gradient (LSTMDoubleReverseDecoderModule threads the first LSTM's final state into the
    second), final states too."""

    @staticmethod
    def forward(ctx, x, h0, c0, T, n_layers, kind, want_state, h0_ok, *params):
        run_idle_jobs()                                    # (the chip is about to idle: ops._IDLE_JOBS)
        _dev_check(x)
        ctx.set_materialize_grads(False)
        dev = x.device
        lstm = kind == 'lstm'
        G = 4 if lstm else 3
        seq = x.dim() == 3
        Bn, In = x.shape[0], x.shape[-1]
        Hh = h0.shape[2]
        Hp = round_up(Hh, 4)
        Lr = n_layers
        # h history with a 16-B row pitch (zero pad): slot 0 = h0, slot t+1 = h_t
        hs = torch.zeros(Lr, Bn, T + 1, Hp, device=dev, dtype=F32)
        hs[:, :, 0, :Hh].copy_(h0)
        cs = None
        if lstm:
            cs = torch.empty(Lr, T + 1, Bn, Hh, device=dev, dtype=F32)
            cs[:, 0].copy_(c0)
        saved = torch.empty(Lr, T, Bn, 4 * Hh, device=dev, dtype=F32)
        w_ih0, _, b_ih0, b_hh0 = params[0:4]
        bias0 = bias_sum(b_ih0, b_hh0) if lstm else gru_bias(b_ih0, b_hh0, Hh)
        if seq:
            xproj = torch.empty(Bn, T, G * Hh, device=dev, dtype=F32)
            linear_raw(_rows3d(x), pack_weight(w_ih0), bias0, Bn * T, G * Hh, In, (xproj, G * Hh, 0, 0))
            xp_sb, xp_st = T * G * Hh, G * Hh
        else:
            xproj = torch.empty(Bn, G * Hh, device=dev, dtype=F32)
            linear_raw(_rows2d(x), pack_weight(w_ih0), bias0, Bn, G * Hh, In, _rows2d(xproj))
            xp_sb, xp_st = G * Hh, 0
        whh, wih, biases, bhns = [], [None], [None], []
        for l in range(Lr):
            w_ih, w_hh, b_ih, b_hh = params[4 * l: 4 * l + 4]
            whh.append(pack_gates(w_hh, Hh, G))
            if l > 0:
                wih.append(pack_gates(w_ih, Hh, G))
                biases.append(bias_sum(b_ih, b_hh) if lstm else gru_bias(b_ih, b_hh, Hh))
            bhns.append(None if lstm else b_hh[2 * Hh:])
        # f16x3 arithmetic: the plan's fp16 plane packs + amax words of every recurrent weight (absent -> exact fp32 kernels)
        pl_hh, am_hh, pl_ih, am_ih = [], [], [None], [None]
        for l in range(Lr):
            w_ih, w_hh = params[4 * l], params[4 * l + 1]
            a, b = planned_planes(w_hh, K_GATES_H3)
            pl_hh.append(a); am_hh.append(b)
            if l > 0:
                a, b = planned_planes(w_ih, K_GATES_H3)
                pl_ih.append(a); am_ih.append(b)
        # (h0_ok: the fp16-pipe kernels scale the state rows by 2^12 before the fp16 split — a start state of magnitude >= 16 would
        # overflow to inf; rnn_stack decides)
        h3 = bool(h0_ok) and all(t is not None for t in pl_hh + am_hh + pl_ih[1:] + am_ih[1:])
        keep = (whh, wih, biases, pl_hh, am_hh, pl_ih, am_ih)   # operands stay referenced until the launches are queued
        ws_n = L.query('gpe_rnn_seq_fwd_ws', G, Lr, T, Bn, Hh)    # arrival counters of the persistent launch (0: diagonal launches)
        ws = torch.empty(ws_n, device=dev, dtype=torch.uint8) if ws_n > 0 else None
        if RNN_WS_HOOK is not None:
            RNN_WS_HOOK('fwd', ws)
        L.call('gpe_rnn_seq_fwd', G, Lr, T, Bn, Hh, xproj, xp_sb, xp_st, _ptr_array(whh), _ptr_array(wih),
               _ptr_array(biases), None if lstm else _ptr_array(bhns), hs, hs.stride(0), hs.stride(1), hs.stride(2),
               cs, cs.stride(0) if lstm else 0, cs.stride(1) if lstm else 0, saved, saved.stride(0), saved.stride(1),
               _ptr_array(pl_hh) if h3 else None, _ptr_array(pl_ih) if h3 else None,
               _ptr_array(am_hh) if h3 else None, _ptr_array(am_ih) if h3 else None, ws, max(ws_n, 0))
        del keep
        hN = cN = None
        if want_state:
            hN = hs[:, :, T, :Hh].contiguous()
            cN = cs[:, T].contiguous() if lstm else None
        ctx.dims = (Bn, In, Hh, T, Lr, kind, seq)
        ctx.save_for_backward(x, hs, cs, saved, *params)
        ctx.needs_state_grad = (h0.requires_grad, c0 is not None and c0.requires_grad)
        top = hs[Lr - 1, :, 1:, :Hh]
        if lstm:
            return top, hN, cN
        return top, hN

    @staticmethod
    def backward(ctx, g_top, g_hN, g_cN=None):
        Bn, In, Hh, T, Lr, kind, seq = ctx.dims
        lstm = kind == 'lstm'
        G = 4 if lstm else 3
        sv = ctx.saved_tensors
        x, hs, cs, saved = sv[:4]
        params = sv[4:]
        dev = x.device
        want_h0, want_c0 = ctx.needs_state_grad
        GH = G * Hh
        GHp = round_up(GH, 4)
        dgx = torch.empty(Lr, Bn, T, GHp, device=dev, dtype=F32)
        dgh = dgx if lstm else torch.empty(Lr, Bn, T, GHp, device=dev, dtype=F32)
        part = torch.empty(L.query('gpe_rnn_seq_bwd_ws', G, Lr, T, Bn, Hh), device=dev, dtype=F32)
        if RNN_WS_HOOK is not None:
            RNN_WS_HOOK('bwd', part)
        carry = torch.empty(2, Lr, Bn, Hh, device=dev, dtype=F32)
        whh_t = [pack_weight(params[4 * l + 1], transpose=True) for l in range(Lr)]
        wih_t = [None] + [pack_weight(params[4 * l], transpose=True) for l in range(1, Lr)]
        if g_top is not None and g_top.stride(2) != 1:
            g_top = g_top.contiguous()
        ghN = g_hN.contiguous() if g_hN is not None else None
        gcN = g_cN.contiguous() if (lstm and g_cN is not None) else None
        # f16x3: transposed fp16 plane packs + amax words for the persistent backward launch (absent -> exact fp32 there)
        tp_hh, ta_hh, tp_ih, ta_ih = [], [], [None], [None]
        if lstm:
            for l in range(Lr):
                a, b = planned_planes(params[4 * l + 1], K_TRANS_H3)
                tp_hh.append(a); ta_hh.append(b)
                if l > 0:
                    a, b = planned_planes(params[4 * l], K_TRANS_H3)
                    tp_ih.append(a); ta_ih.append(b)
        h3b = lstm and all(t is not None for t in tp_hh + ta_hh + tp_ih[1:] + ta_ih[1:])
        L.call('gpe_rnn_seq_bwd', G, Lr, T, Bn, Hh, g_top, g_top.stride(0) if g_top is not None else 0,
               g_top.stride(1) if g_top is not None else 0, ghN, gcN, _ptr_array(whh_t), _ptr_array(wih_t),
               hs, hs.stride(0), hs.stride(1), hs.stride(2), cs, cs.stride(0) if lstm else 0, cs.stride(1) if lstm else 0,
               saved, saved.stride(0), saved.stride(1), dgx, dgh, dgx.stride(0), dgx.stride(1), dgx.stride(2), part, carry,
               _ptr_array(tp_hh) if h3b else None, _ptr_array(tp_ih) if h3b else None,
               _ptr_array(ta_hh) if h3b else None, _ptr_array(ta_ih) if h3b else None)
        grads = [None] * (4 * Lr)
        d_x = None
        d_h0 = torch.empty(Lr, Bn, Hh, device=dev, dtype=F32) if want_h0 else None
        d_c0 = carry[0].clone() if want_c0 else None
        nz = (GH + 255) // 256
        # ---- what the rest of the backward pass waits for: the input gradient and the start-state gradients ----
        dGs = None
        gx0 = _rows3d(dgx[0][:, :, :GH])
        if not seq:
            # the same input row feeds every step: sum the gate gradients over T first, then ONE Bn-row product
            # (T times fewer rows than dG^T x over the repeated input)
            dGs = torch.empty(Bn, GH, device=dev, dtype=F32)
            L.call('gpe_reduce_inner', dgx[0], T * GHp, GHp, T, Bn, GH, dGs, GH, 0)
            if ctx.needs_input_grad[0]:
                d_x = torch.empty(Bn, In, device=dev, dtype=F32)
                linear_raw(_rows2d(dGs), pack_weight(params[0], transpose=True), None, Bn, In, GH, _rows2d(d_x))
        elif ctx.needs_input_grad[0]:
            d_x = torch.empty(Bn, T, In, device=dev, dtype=F32)
            linear_raw(gx0, pack_weight(params[0], transpose=True), None, Bn * T, In, GH, (d_x, In, 0, 0))
        if want_h0:
            for l in range(Lr):
                # dh_0 = dG_{l,0} . W_hh (+ the z-gated direct path of a GRU)
                rec = torch.empty(nz, Bn, Hh, device=dev, dtype=F32)
                L.call('gpe_linear_splitk', dgh[l][:, 0], T * GHp, whh_t[l], rec, Bn, Hh, GH)
                L.call('gpe_reduce_inner', rec, Hh, Bn * Hh, nz, Bn, Hh, d_h0[l], Hh, 0)
                if not lstm:
                    L.call('gpe_add', d_h0[l], carry[0][l], d_h0[l], Bn * Hh)
        # ---- the weight gradients: leaves of the backward pass (on the side stream when every one of them lands in the arena) ----
        with side_grads(list(params), [dgx, dgh, hs, x, dGs]):
            for l in range(Lr):
                w_ih, w_hh, b_ih, b_hh = params[4 * l: 4 * l + 4]
                gx_rows, gh_rows = _rows3d(dgx[l][:, :, :GH]), _rows3d(dgh[l][:, :, :GH])
                d_whh, d_bhh = _gbuf(w_hh), _gbuf(b_hh)
                redgemm_raw(gh_rows, _rows3d(hs[l][:, :T, :Hh]), Bn * T, GH, Hh, out=(d_whh, d_bhh))
                d_wih, d_bih = _gbuf(w_ih), _gbuf(b_ih)
                if l == 0 and not seq:
                    redgemm_raw(_rows2d(dGs), _rows2d(x), Bn, GH, In, out=(d_wih, d_bih))
                else:
                    Kin = In if l == 0 else Hh
                    src = _rows3d(x) if l == 0 else _rows3d(hs[l - 1][:, 1:, :Hh])
                    redgemm_raw(gx_rows, src, Bn * T, GH, Kin, out=(d_wih, d_bih))
                grads[4 * l: 4 * l + 4] = [_gret(w_ih, d_wih), _gret(w_hh, d_whh), _gret(b_ih, d_bih),
                                           _gret(b_hh, d_bhh)]
        return (d_x, d_h0, d_c0, None, None, None, None, None, *grads)


class DropoutMulFn(torch.autograd.Function):
    """y = x * mask on a (strided) [Bn, T, H] sequence: the dropout torch's recurrent modules apply between layers."""

    @staticmethod
    def forward(ctx, x, mask):
        _dev_check(x)
        assert x.dim() == 3 and x.stride(2) == 1 and mask.is_contiguous() and mask.shape == x.shape
        Bn, T, H = x.shape
        out = torch.empty(Bn, T, H, device=x.device, dtype=F32)
        L.call('gpe_mul_rows', x, x.stride(0), x.stride(1), mask, Bn, T, H, out)
        ctx.save_for_backward(mask)
        return out

    @staticmethod
    def backward(ctx, g):
        mask, = ctx.saved_tensors
        if g.stride(2) != 1:
            g = g.contiguous()
        Bn, T, H = mask.shape
        gx = torch.empty(Bn, T, H, device=g.device, dtype=F32)
        L.call('gpe_mul_rows', g, g.stride(0), g.stride(1), mask, Bn, T, H, gx)
        return gx, None


def _dropout_mask(T, Bn, H, p, device):
    """The mask at::dropout draws for a recurrent layer's output: torch's nn.LSTM / nn.GRU work time-major, so the noise tensor
    is [T, Bn, H] (ATen RNN.cpp apply_layer_stack -> dropout -> empty_like(input).bernoulli_(1 - p).div_(1 - p)).  Drawn on the
    CPU generator in that shape — the stream the reference's CPU run consumes, like the random start states — and handed over
    batch-major."""
    if HOST_DRAWN is not None:
        def fill(pin):
            pin.copy_(torch.empty(T, Bn, H).bernoulli_(1 - p).div_(1 - p).transpose(0, 1))
        return HOST_DRAWN.get((Bn, T, H), fill)
    noise = torch.empty(T, Bn, H).bernoulli_(1 - p).div_(1 - p)
    return noise.transpose(0, 1).contiguous().to(device, non_blocking=True)


def rnn_stack(x, h0, c0, T, n_layers, kind, params, want_state=False, dropout=0.0, training=False, h0_bounded=False):
    """nn.LSTM / nn.GRU(batch_first=True, num_layers, dropout).  Without dropout (or in eval mode, or with one layer) the whole
    stack runs in wavefront order (RNNStackFn).  With dropout the layers run one after the other — the mask sits between them —
    each as a one-layer RNNStackFn over the masked output sequence of the layer below.

    f16x3 arithmetic (gpe_amd.set_math): the forward gate products run on the fp16 pipe with the state rows scaled by 2^12, which
    is exact for every state a cell produces (|h| < 1) and for start states below 16.  h0_bounded=True promises |h0| < 16 (the
    package's own modules: the reference's zero / kaiming start states, nn/net_blocks.py:308-315, and chained decoder states);
    otherwise the largest |h0| is READ BACK here (one host synchronisation per call, f16x3 mode only) and a start state of
    magnitude >= 16 takes the exact fp32 kernels for the whole sequence."""
    h0_ok = True
    if not h0_bounded and L.get_math() == 'f16x3' and h0 is not None:
        h0_ok = bool(h0.abs().max().item() < 16.0)
    if not (dropout > 0 and training and n_layers > 1):
        out = RNNStackFn.apply(x, h0, c0, T, n_layers, kind, want_state, h0_ok, *params)
        return out if kind == 'lstm' else (out[0], out[1], None)
    lstm = kind == 'lstm'
    inp, hs, cs = x, [], []
    for l in range(n_layers):
        out = RNNStackFn.apply(inp, h0[l:l + 1], c0[l:l + 1] if lstm else None, T, 1, kind, want_state, h0_ok, *params[4 * l:4 * l + 4])
        top = out[0]
        if want_state:
            hs.append(out[1])
            if lstm:
                cs.append(out[2])
        if l < n_layers - 1:
            inp = DropoutMulFn.apply(top, _dropout_mask(T, top.shape[0], top.shape[2], float(dropout), top.device))
    hN = torch.cat(hs, 0) if want_state else None
    cN = torch.cat(cs, 0) if (want_state and lstm) else None
    return top, hN, cN


# -------------------------------------------------------------------------------------------------
# dense MLP (attention variant, MLP decoder, stitch model)
# -------------------------------------------------------------------------------------------------
class DenseMLPFn(torch.autograd.Function):
    """MLP(channels) = [Linear -> ReLU -> BatchNorm1d] x n on dense rows (nn/net_blocks.py:43-47 as used by
    point_segment_mlp, nn/nets.py:223-226; MLPDecoder, nn/net_blocks.py:273-298; StitchOnEdge3DPairs, nn/nets.py:342).
    Same kernels as the edge MLP with one "message" per row (k = 1): fused Linear+ReLU(+fp64 BN statistics), every
    BatchNorm folded into the next Linear, the last one applied explicitly; backward = the edge MLP's chain (centred
    reduce-GEMM -> BN coefficients -> propagate in place).  Layers wider than 256 (the global-attention MLP is 403 wide,
    an MLP decoder thousands) run as K slabs / several column blocks of the generic row GEMM."""

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
        ews, ews_n = edge_workspace(1, M, 1, 4, dev)
        words = f16x3_words(2 * n_blocks, M, dev)          # [l] activation a_l, [n + l] dz_l (backward)
        for l in range(n_blocks):
            W, b, g, be = params[4 * l: 4 * l + 4]
            rm, rv, nb = bufs[3 * l: 3 * l + 3]
            Cout = W.shape[0]
            ldo = round_up(Cout, 4)
            a = torch.empty(M, ldo, device=dev, dtype=F32)
            part = torch.empty(nblk, 2, Cout, device=dev, dtype=torch.float64) if training else None
            L.call('gpe_edge_mlp_fwd', 1, None, 0, None, a_in, a_in.stride(0), 1, M, 1, Cin, Cout,
                   pack_weight(W, col_scale=scale), b if tvec is None else fold_bias(W, b, tvec), a, ldo, part,
                   0, None, None, None, None, 0, _word(words, l - 1) if l > 0 else None,
                   _word(words, l) if l < n_blocks - 1 else None, ews, ews_n, 0)
            st = bn_finalize(part, nblk, Cout, M, g, be, eps, momentum, rm, rv, nb) if training \
                else bn_from_running(rm, rv, g, be, eps)
            acts.append(a)
            stats.append(st)
            scale, tvec = st[2], st[3]
            a_in, Cin = a, Cout
        y = torch.empty(M, Cin, device=dev, dtype=F32)
        L.call('gpe_bn_apply', a_in, a_in.stride(0), stats[-1], M, Cin, y, Cin)
        ctx.n_blocks = n_blocks
        ctx.training = training
        ctx.done = False
        ctx.words = words
        ctx.save_for_backward(x, *params, *acts, *stats)
        return y

    @staticmethod
    def backward(ctx, gy):
        if ctx.done:
            raise RuntimeError('DenseMLPFn.backward ran twice on the same graph: the stored activations are overwritten '
                               'in place by the first pass (retain_graph is not supported)')
        ctx.done = True
        n, training = ctx.n_blocks, ctx.training
        sv = ctx.saved_tensors
        x, params, acts, stats = sv[0], sv[1:1 + 4 * n], sv[1 + 4 * n:1 + 5 * n], sv[1 + 5 * n:1 + 6 * n]
        dev = x.device
        M = x.shape[0]
        gy = gy.contiguous()
        grads = [None] * (4 * n)
        # last block: BN applied explicitly -> same algebra as the edge layer's BN-after-max with one slot per row
        a, st = acts[-1], stats[-1]
        C = params[4 * (n - 1)].shape[0]
        g_l, be_l = params[4 * (n - 1) + 2], params[4 * (n - 1) + 3]
        psb = L.query('gpe_point_sums_blocks')
        part = torch.empty(psb, 2, C, device=dev, dtype=torch.float64)
        L.call('gpe_edge_bwd_point_sums', gy, C, a, a, a.stride(0), st, M, C, part, None)
        coef, dgam, dbet = bn_bwd_coef(part, psb, st, C, M, g_l, be_l, training)
        words = ctx.words
        ews, ews_n = edge_workspace(1, M, 1, 4, dev)
        L.call('gpe_edge_dz3_all', a, a.stride(0), gy, C, 1.0, coef, 1, M, 1, C, _word(words, 2 * n - 1))
        grads[4 * (n - 1) + 2], grads[4 * (n - 1) + 3] = _gret(g_l, dgam), _gret(be_l, dbet)
        for l in reversed(range(n)):
            W, b = params[4 * l], params[4 * l + 1]
            dz = acts[l]                          # holds dz_l now
            C = W.shape[0]
            if l > 0:
                prev, stp = acts[l - 1], stats[l - 1]
                Cp = params[4 * (l - 1)].shape[0]
                G = torch.empty(C, Cp, device=dev, dtype=F32)
                db = _gbuf(b)
                if Cp > 256:
                    redgemm_raw((dz, dz.stride(0), 0, 0), (prev, prev.stride(0), 0, 0), M, C, Cp, v_shift=stp[0],
                                out=(G, db))
                else:
                    ws = torch.empty(L.query('gpe_redgemm_ws', C, Cp), device=dev, dtype=F32)
                    L.call('gpe_edge_redgemm', dz, dz.stride(0), 1, prev, prev.stride(0), None, 0, None, stp[0],
                           1, M, 1, C, Cp, G, Cp, db, ws, _word(words, n + l), _word(words, l - 1), ews, ews_n,
                           None, 0, None, None, 0, None)
                sums = torch.empty(1, 2, Cp, device=dev, dtype=torch.float64)
                dW = _gbuf(W)
                L.call('gpe_bn_bwd_from_G', G, Cp, db, W, W.stride(0), C, Cp, stp, sums, dW, Cp)
                gp, bp = params[4 * (l - 1) + 2], params[4 * (l - 1) + 3]
                coef_p, dgam_p, dbet_p = bn_bwd_coef(sums, 1, stp, Cp, M, gp, bp, training)
                L.call('gpe_edge_mlp_bwd', dz, dz.stride(0), 0, None, 0, None, 1, M, 1, C, Cp,
                       pack_weight(W, transpose=True), coef_p, prev, prev.stride(0), None, 0,
                       _word(words, n + l), _word(words, n + l - 1), ews, ews_n, None, 0, None, None, 0, None)
                grads[4 * l], grads[4 * l + 1] = _gret(W, dW), _gret(b, db)
                grads[4 * (l - 1) + 2], grads[4 * (l - 1) + 3] = _gret(gp, dgam_p), _gret(bp, dbet_p)
            else:
                K0 = x.shape[1]
                dW, db = _gbuf(W), _gbuf(b)
                redgemm_raw((dz, dz.stride(0), 0, 0), _rows2d(x), M, C, K0, out=(dW, db))
                grads[0], grads[1] = _gret(W, dW), _gret(b, db)
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


class SparsemaxLossFn(torch.autograd.Function):
    """entmax.SparsemaxLoss()(x, target) as nn/metrics/composed_loss.py:323-332 calls it: mean over rows of the sparsemax
    Fenchel-Young loss; gradient (sparsemax(x) - onehot(target)) / rows."""

    @staticmethod
    def forward(ctx, x, target):
        _dev_check(x)
        M, W = x.shape
        if x.stride(1) != 1:
            x = x.contiguous()
        tgt = target.to(torch.int32).contiguous()
        if tgt.numel() != M:
            raise ValueError('SparsemaxLoss: %d targets for %d rows' % (tgt.numel(), M))
        gx = torch.empty(M, W, device=x.device, dtype=F32)
        part = torch.empty((M + 255) // 256, device=x.device, dtype=torch.float64)
        loss = torch.empty(1, device=x.device, dtype=F32)
        bad = torch.zeros(1, device=x.device, dtype=torch.int32)
        L.call('gpe_sparsemax_loss', x, x.stride(0), tgt, M, W, gx, W, part, loss, bad)
        if bad.item():
            raise IndexError('SparsemaxLoss: a segmentation label lies outside [0, %d)' % W)
        ctx.save_for_backward(gx)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        gx, = ctx.saved_tensors
        out = torch.empty_like(gx)
        L.call('gpe_scale_dev', gx, g.reshape(1).to(F32).contiguous(), out, gx.numel())
        return out, None


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
    """pooled[b, p, :] = pool_n w[b, n, p] * feat[b, n, :]  — the reference's 23-iteration loop of
    `w[:, p] * features -> global_pool` (nn/nets.py:263-276) as one batched launch for all clouds and panels.
    mode: 0 mean (shipped), 1 max, 2 add — the encoder's `global_pool` setting."""

    @staticmethod
    def forward(ctx, w, feat, B, N, mode=0):
        _dev_check(w)
        dev = w.device
        P, C = w.shape[1], feat.shape[1]
        pooled = torch.empty(B, P, C, device=dev, dtype=F32)
        nws = L.query('gpe_attn_pool_ws', B, N, P, C)
        part = torch.empty(nws, device=dev, dtype=F32)
        arg = torch.empty(B, P, C, device=dev, dtype=torch.int32) if mode == 1 else None
        part_arg = torch.empty(nws, device=dev, dtype=torch.int32) if mode == 1 else None
        L.call('gpe_attn_pool_fwd', w, w.stride(0), feat, feat.stride(0), B, N, P, C, mode, pooled, arg, part, part_arg)
        ctx.dims = (B, N, P, C, mode)
        ctx.arg = arg
        ctx.save_for_backward(w, feat)
        return pooled.view(B * P, C)

    @staticmethod
    def backward(ctx, g):
        w, feat = ctx.saved_tensors
        B, N, P, C, mode = ctx.dims
        dev = w.device
        g = g.contiguous()
        gw = torch.empty(B * N, P, device=dev, dtype=F32)
        gf = torch.empty(B * N, C, device=dev, dtype=F32)
        L.call('gpe_attn_pool_bwd', w, w.stride(0), feat, feat.stride(0), g, ctx.arg, B, N, P, C, mode, gw, P, gf, C)
        return gw, gf, None, None, None


# -------------------------------------------------------------------------------------------------
# loss (the caller-side step right after the path) and its ground-truth matching
# -------------------------------------------------------------------------------------------------
LOSS_SHAPE, LOSS_LOOP, LOSS_ROT, LOSS_TR = 1, 2, 4, 8


def _view_strides(ol):
    """(sb, sp, sl) of an outlines view [B,P,L,4] whose last dim is unit-stride (a slice of the [B,P,L,8] output)."""
    assert ol.dim() == 4 and ol.stride(3) == 1, (ol.shape, ol.stride())
    return ol.stride(0), ol.stride(1), ol.stride(2)


def _row_stride(t, P):
    """row pitch of a [B,P,c] view of a [B*P, ld] tensor."""
    assert t.dim() == 3 and t.stride(2) == 1 and t.stride(0) == P * t.stride(1), (t.shape, t.stride())
    return t.stride(1)


class PatternLossFn(torch.autograd.Function):
    """ComposedPatternLoss._main_losses (nn/metrics/composed_loss.py:294-321) + PanelLoopLoss (nn/metrics/losses.py:19-51)
    in one forward and one backward launch.  Returns a [5] tensor {total, shape, loop, rotation, translation}; only
    element 0 carries gradient."""

    @staticmethod
    def forward(ctx, outlines, rotations, translations, gt_ol, gt_rot, gt_tr, num_edges, flags, pad0, pad1, loop_w):
        _dev_check(outlines)
        dev = outlines.device
        B, P, Lp = outlines.shape[:3]
        sb, sp, sl = _view_strides(outlines)
        R = rotations.shape[-1] if rotations is not None else 0
        T = translations.shape[-1] if translations is not None else 0
        rs = _row_stride(rotations, P) if rotations is not None else 0
        ts = _row_stride(translations, P) if translations is not None else 0
        part = torch.empty(B, 4, device=dev, dtype=torch.float64)
        loop_sums = torch.empty(B * P, 2, device=dev, dtype=F32)
        out = torch.empty(5, device=dev, dtype=F32)
        args = (outlines, sb, sp, sl, rotations, rs, translations, ts, gt_ol, gt_rot, gt_tr, num_edges, B, P, Lp, R, T,
                flags, float(pad0), float(pad1), float(loop_w))
        L.call('gpe_pattern_loss_fwd', *args, part, loop_sums, out)
        ctx.args = args
        ctx.loop_sums = loop_sums
        ctx.shapes = (B, P, Lp, R, T)
        ctx.need = (rotations is not None and rotations.requires_grad,
                    translations is not None and translations.requires_grad)
        return out

    @staticmethod
    def backward(ctx, g):
        B, P, Lp, R, T = ctx.shapes
        dev = g.device
        g = g.contiguous()
        g_ol = torch.empty(B, P, Lp, 4, device=dev, dtype=F32)
        g_rot = torch.empty(B, P, R, device=dev, dtype=F32) if ctx.need[0] else None
        g_tr = torch.empty(B, P, T, device=dev, dtype=F32) if ctx.need[1] else None
        # g[0] is the upstream gradient of `total` (a device scalar: no host sync)
        L.call('gpe_pattern_loss_bwd', *ctx.args, ctx.loop_sums, g, g_ol, g_rot, g_tr)
        return g_ol, g_rot, g_tr, None, None, None, None, None, None, None, None


STITCH_MAIN, STITCH_HARDNET, STITCH_FREE, STITCH_SUP = 1, 2, 4, 8


class StitchLossFn(torch.autograd.Function):
    """ComposedPatternLoss._stitch_losses (nn/metrics/composed_loss.py:336-362): PatternStitchLoss (nn/metrics/losses.py:54-180,
    both negative-term variants), the supervised stitch-tag MSE and the free-edge BCE-with-logits in one forward and one
    backward launch.  Returns a [5] tensor {total, similarity, negative, supervised, free}; only element 0 carries gradient.
    stitch_tags [B,P,L,D] and free_logits [B,P,L] are the strided views of the panel decoder's output."""

    @staticmethod
    def forward(ctx, stitch_tags, free_logits, stitches, nums, gt_mask, gt_tags, flags, margin, sup_w):
        ref = stitch_tags if stitch_tags is not None else free_logits
        _dev_check(ref)
        dev = ref.device
        B, P, Lp = ref.shape[:3]
        D = stitch_tags.shape[3] if stitch_tags is not None else 0
        ts = _view_strides(stitch_tags) if stitch_tags is not None else (0, 0, 0)
        ms = (free_logits.stride(0), free_logits.stride(1), free_logits.stride(2)) if free_logits is not None else (0, 0, 0)
        S = stitches.shape[2] if stitches is not None else 0
        part = torch.empty(B, 6, device=dev, dtype=torch.float64)
        out = torch.empty(5, device=dev, dtype=F32)
        args = (stitch_tags, ts[0], ts[1], ts[2], D, free_logits, ms[0], ms[1], ms[2], stitches, nums, S, gt_mask, gt_tags,
                B, P, Lp, flags, float(margin), float(sup_w))
        L.call('gpe_stitch_loss_fwd', *args, part, out)
        ctx.args, ctx.part, ctx.shape = args, part, (B, P, Lp, D)
        ctx.need = (stitch_tags is not None and bool(flags & (STITCH_MAIN | STITCH_SUP)),
                    free_logits is not None and bool(flags & STITCH_FREE))
        return out

    @staticmethod
    def backward(ctx, g):
        B, P, Lp, D = ctx.shape
        dev = g.device
        g = g.contiguous()
        g_tags = torch.empty(B, P, Lp, D, device=dev, dtype=F32) if ctx.need[0] else None
        g_mask = torch.empty(B, P, Lp, device=dev, dtype=F32) if ctx.need[1] else None
        L.call('gpe_stitch_loss_bwd', *ctx.args, ctx.part, g, g_tags, g_mask)
        return g_tags, g_mask, None, None, None, None, None, None, None


def stitch_renumber(stitches, nums, P, Lp, perm=None, lead=None, num_edges=None):
    """composed_loss.py:592-620 (`perm`: the panel-order permutation) and :727-755 (`lead` / `num_edges`: the panel-origin
    shift) applied to the ground-truth stitches [B,2,S] int64 -> a new tensor."""
    B, _, S = stitches.shape
    out = torch.empty_like(stitches)
    L.call('gpe_stitch_renumber', stitches, nums, B, S, P, Lp, perm, lead, num_edges, out)
    return out


def panel_shift(feat, lead, num_edges):
    """composed_loss.py:705-725 `_per_panel_shift` on per-edge ground truth [B,P,L] or [B,P,L,D] (fp32) -> a new tensor."""
    _dev_check(feat)
    feat = feat.contiguous()
    Lp = feat.shape[2]
    D = feat.shape[3] if feat.dim() == 4 else 1
    out = torch.empty_like(feat)
    L.call('gpe_panel_shift', feat, D, lead, num_edges, feat.shape[0] * feat.shape[1], Lp, out)
    return out


def origin_match(outlines, gt_ol, num_edges):
    """composed_loss.py:656-703 `_batch_edge_order_match`: -> (gt outlines with every panel's edge loop shifted to the
    origin that best matches the prediction, leading edge per panel int32 [B*P])."""
    _dev_check(outlines)
    B, P, Lp, D = gt_ol.shape
    sb, sp, sl = _view_strides(outlines)
    out = torch.empty_like(gt_ol)
    lead = torch.empty(B * P, device=gt_ol.device, dtype=torch.int32)
    L.call('gpe_origin_match', outlines.detach(), sb, sp, sl, gt_ol, D, num_edges, B, P, Lp, out, lead)
    return out, lead


def order_match(pred_feat, gt_feat):
    """composed_loss.py:530-570 `_panel_order_match` (the greedy assignment): -> int64 [B, P] permutation of GT panels,
    and a device flag that is non-zero if the matching left a finite distance behind (the reference raises then)."""
    _dev_check(pred_feat)
    B, P, D = pred_feat.shape
    perm = torch.empty(B, P, device=pred_feat.device, dtype=torch.int64)
    fail = torch.zeros(1, device=pred_feat.device, dtype=torch.int32)
    L.call('gpe_order_match', pred_feat.contiguous(), gt_feat.contiguous(), B, P, D, perm, fail)
    return perm, fail


def standardize(x, shift, scale):
    """nn/data/transforms.py:35-50 FeatureStandartization on the device: (x - shift) / scale per column."""
    import ctypes
    _dev_check(x)
    x = x.contiguous()
    C = x.shape[-1]
    sh = (ctypes.c_float * C)(*[float(v) for v in shift])
    sc = (ctypes.c_float * C)(*[float(v) for v in scale])
    out = torch.empty_like(x)
    L.call('gpe_standardize', x, x.numel() // C, C, sh, sc, out)
    return out


# -------------------------------------------------------------------------------------------------
# PointNet++ set abstraction (PointNetPlusPlus, nn/net_blocks.py:10-88)
# -------------------------------------------------------------------------------------------------
def fps_start(B, N, device):
    """PyG's fps(random_start=True) (what nn/net_blocks.py:16 calls): the start point of every cloud, drawn as
    `(torch.rand(B) * N).long()` on torch's default CPU generator — the stream the reference's CPU run consumes, like the
    random LSTM states — ONE draw per fps call (oracle/ref_path.py fps_start)."""
    start = (torch.rand(B) * float(N)).long().clamp_(max=N - 1)
    return start.to(torch.int32).to(device, non_blocking=True)


def fps(pos, B, N, M, start=None):
    """torch_geometric.nn.fps over equal-sized clouds: -> int32 [B, M] LOCAL indices in selection order, starting at
    start[b] (int32 [B] on the device; None = the cloud's first point, i.e. random_start=False); ties -> lower index."""
    _dev_check(pos)
    idx = torch.empty(B, M, device=pos.device, dtype=torch.int32)
    L.call('gpe_fps', pos, pos.stride(0), B, N, pos.shape[1], M, start, idx)
    return idx


def radius_neighbors(pos, cidx, B, N, r, max_num_neighbors):
    """torch_geometric.nn.radius(pos, pos[idx], r, ..., max_num_neighbors): -> (nbr int32 [B*M, maxn] local point indices,
    first maxn in ascending order, cnt int32 [B*M])."""
    _dev_check(pos)
    M = cidx.shape[1]
    nbr = torch.empty(B * M, max_num_neighbors, device=pos.device, dtype=torch.int32)
    cnt = torch.empty(B * M, device=pos.device, dtype=torch.int32)
    L.call('gpe_radius', pos, pos.stride(0), cidx, B, N, pos.shape[1], M, float(r), max_num_neighbors, nbr, cnt)
    return nbr, cnt


def pointconv_self_loops(nbr, cnt, B, N, M):
    """PyG PointNetConv's default add_self_loops=True on the ball-query edge list: -> (edge count per centroid after the
    re-indexing, slot of the removed neighbour or -1).  See csrc/gpe_pointnet.hip."""
    cnt2 = torch.empty_like(cnt)
    drop = torch.empty_like(cnt)
    L.call('gpe_pointconv_self_loops', nbr, cnt, B, N, M, nbr.shape[1], cnt2, drop)
    return cnt2, drop


def ball_messages(pos, x, cidx, nbr, off, n_edges, B, N, drop=None):
    """PointConv message inputs [E, Cx + 3] over the compact edge list + the centroid (segment) id of every edge row.
    drop: from pointconv_self_loops (then `off` is the scan of ITS counts)."""
    M, maxn = cidx.shape[1], nbr.shape[1]
    C = pos.shape[1]
    Cx = 0 if x is None else x.shape[1]
    msg = torch.empty(n_edges, Cx + C, device=pos.device, dtype=F32)
    seg = torch.empty(n_edges, device=pos.device, dtype=torch.int32)
    L.call('gpe_ball_messages', pos, pos.stride(0), x, 0 if x is None else x.stride(0), Cx, cidx, nbr, off, drop, B, N, C, M,
           maxn, msg, Cx + C, seg)
    return msg, seg


class RaggedMaxFn(torch.autograd.Function):
    """max over each centroid's (ragged) run of edge rows: PointConv's aggr='max'."""

    @staticmethod
    def forward(ctx, x, off, seg, S):
        _dev_check(x)
        E, C = x.shape
        y = torch.empty(S, C, device=x.device, dtype=F32)
        arg = torch.empty(S, C, device=x.device, dtype=torch.int64)
        L.call('gpe_ragged_max_fwd', x, x.stride(0), off, S, C, y, C, arg)
        ctx.save_for_backward(off, arg, seg)
        ctx.dims = (E, C)
        return y

    @staticmethod
    def backward(ctx, gy):
        off, arg, seg = ctx.saved_tensors
        E, C = ctx.dims
        gy = gy.contiguous()
        gx = torch.empty(E, C, device=gy.device, dtype=F32)
        L.call('gpe_ragged_max_bwd', gy, C, off, arg, seg, E, C, gx, C)
        return gx, None, None, None
