#!/usr/bin/env python3
"""Item timeline of the multi-tile persistent LSTM forward (csrc/gpe_rnn_persist_mt.hip): gpe_debug_set(8192) makes lane 0 of every
wave stamp the phases of every item it walks.   python scripts/rnn_mt_trace.py [Bn In H T L]"""
import sys
import numpy as np
import torch
import gpe_amd
from gpe_amd import ops, net_blocks, _lib as Lb

Bn, In, H, T, L = [int(a) for a in sys.argv[1:6]] if len(sys.argv) > 5 else (736, 250, 250, 14, 3)
torch.manual_seed(0)
rnn = torch.nn.LSTM(In, H, L, batch_first=True).cuda()
params = net_blocks._rnn_params(rnn, L)
plan = ops.PackPlan()
net_blocks._register_rnn_packs(plan, rnn, L, H, 4)
x = torch.randn(Bn, In).cuda()
h0 = (torch.randn(L, Bn, H) * 0.3).cuda()
c0 = (torch.randn(L, Bn, H) * 0.3).cuda()
gpe_amd.set_math('f16x3')
plan.refresh()
Lb.query('gpe_debug_set', 8192)
got = {}
ops.RNN_WS_HOOK = lambda kind, ws: got.__setitem__(kind, ws)
for rep in range(3):
    with torch.no_grad():
        ops.rnn_stack(x, h0, c0, T, L, 'lstm', params, h0_bounded=True)
    torch.cuda.synchronize()
ws = got['fwd'].view(torch.uint8).cpu().numpy()
NW = int(__import__('os').environ.get('PM_NW', '8'))
MAXQ = 16 // NW
nwave_items = T * MAXQ
NB, NRT = (H + 15) // 16, (Bn + 15) // 16
tot = ws.size
# the trace is the tail of the workspace: [grid][4][T * MAXQ][8] u64
for RG in range(1, 17):
    grid = L * NB * RG
    nb = grid * NW * nwave_items * 64
    flag = L * (T + 1) * NRT * 128
    split = L * (T + 1) * 16 * NRT * (((H + 31) // 32) * 32) * 4
    if flag + split + nb == tot:
        break
else:
    sys.exit('cannot find the trace in a %d-byte workspace' % tot)
tr = ws[flag + split:].view(np.uint64).reshape(grid, NW, nwave_items, 8)
st = tr[..., :7].astype(np.float64) * 0.01
info = tr[..., 7]
valid = st[..., 0] > 0
t0 = st[..., 0][valid].min()
print('grid %d (RG %d): span %.1f us' % (grid, RG, st[..., 6][valid].max() - t0))
for l in range(L):
    sl = slice(l * RG * NB, (l + 1) * RG * NB)
    s, v, inf = st[sl], valid[sl], info[sl]
    d = lambda a, b: np.median((s[..., b] - s[..., a])[v])
    okfrac = (inf[v] & 1).mean()
    per = np.median((s[..., 1:, 0] - s[..., :-1, 0])[v[..., 1:] & v[..., :-1]])
    print(' layer %d: item period %.2f us | products A %.2f | deferred publish %.2f | products B %.2f | z %.2f | issue next %.2f | cell + stores %.2f |'
          ' slow path (publish, wait, issue) %.2f | fast path taken %.2f; first item at %.1f, last end %.1f'
          % (l, per, d(0, 1), d(1, 2), d(2, 3), 0.0, d(3, 4), d(4, 5), d(5, 6), okfrac, s[..., 0][v].min() - t0, s[..., 6][v].max() - t0))
# one wave's full timeline
w = tr[0, 0]
print('workgroup 0 wave 0:')
for it in range(min(nwave_items, 12)):
    if w[it, 0] == 0:
        break
    r = w[it, :7].astype(np.float64) * 0.01 - t0
    print('  it %2d t %2d rt %2d ok %d : ' % (it, (int(w[it, 7]) >> 8) & 255, int(w[it, 7]) >> 16, int(w[it, 7]) & 1) + ' '.join('%7.2f' % v for v in r))
