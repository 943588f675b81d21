#!/usr/bin/env python3
"""HIP-event time of PackPlan.refresh() (the per-step re-pack of every weight-derived operand) for the cfg-2 model."""
import torch
import gpe_amd
from gpe_amd import configs, nets, ops
dev = torch.device('cuda', 0)
gpe_amd.set_math('f16x3')
dc = configs.data_config()
cfg = configs.lstm_model_config(k_neighbors=16)
torch.manual_seed(0)
m = nets.GarmentFullPattern3D(dc, dict(cfg), dict(cfg['loss'])).to(dev)
pl = ops.PackPlan()
for c in m.children():
    if hasattr(c, 'register_packs'):
        c.register_packs(pl)
pl.refresh()
ref = [o.clone() for o in pl.outs]
words = pl.words.clone()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
tot = 0.0
for r in range(23):
    ops.bump_weights_epoch()
    torch.cuda.synchronize(); e0.record(); pl.refresh(); e1.record(); torch.cuda.synchronize()
    if r >= 3: tot += e0.elapsed_time(e1)
print('refresh: %.1f us; %d jobs, %d + %d blocks' % (tot / 20 * 1e3, len(pl.specs), pl.pre_blocks, pl.blocks))
same = all(torch.equal(a, b) for a, b in zip(ref, pl.outs)) and torch.equal(words, pl.words)
print('outputs unchanged:', same)
torch.save({'outs': [o.cpu() for o in pl.outs], 'words': pl.words.cpu()}, '/tmp/pack_ref.pt')
# the tiled path against independent references: gpe_pack_weight (kind 0), a numpy element map (kinds 2, 8)
import numpy as np
bad0 = bad2 = bad8 = n0 = n2 = n8 = 0
for i, ((p, p2, kind, N, K, aux, on), out) in enumerate(zip(pl.specs, pl.outs)):
    if kind == 0:
        ref1 = torch.empty_like(out)
        ops.L.call('gpe_pack_weight', p, p.stride(0), N, K, 0, None, ref1)
        bad0 += int(not torch.equal(ref1, out)); n0 += 1
    elif kind in (2, 8):
        H, G = aux, N // aux
        w = p.detach().cpu().numpy()
        npad = 16 * G * ((H + 15) // 16)
        n = np.arange(npad)
        b, gate, u = n // (16 * G), (n // 16) % G, (n // (16 * G)) * 16 + (n % 16)
        ok = u < H
        rows = np.where(ok, gate * H + np.minimum(u, H - 1), 0)
        if kind == 2:
            kpad = out.numel() // npad
            full = np.zeros((npad, kpad), np.float32)
            full[:, :K] = np.where(ok[:, None], w[rows], 0)
            ref2 = full.reshape(npad, kpad // 4, 4).transpose(1, 0, 2).reshape(-1)
            bad2 += int(not np.array_equal(ref2, out.cpu().numpy())); n2 += 1
        else:
            KP = (K + 31) // 32 * 32
            full = np.zeros((npad, KP), np.float32)
            full[:, :K] = np.where(ok[:, None], w[rows], 0)
            word = int(pl.word_of[(p.data_ptr(), kind)].item())
            e = (word >> 23) & 0xff
            sh = 0 if (word == 0 or e == 255) else max(-100, min(100, 141 - e))
            xs = full * np.float32(2.0 ** sh)
            h = xs.astype(np.float16)
            lo = (xs - h.astype(np.float32)).astype(np.float16)
            planes = np.stack([h, lo]).reshape(2, npad, KP // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
            got = out.view(torch.float16).cpu().numpy()
            bad8 += int(not np.array_equal(planes.view(np.uint16), got.view(np.uint16))); n8 += 1
print('jobs that differ from their reference: kind 0 %d of %d, kind 2 %d of %d, kind 8 %d of %d' % (bad0, n0, bad2, n2, bad8, n8))
def tm(fn, reps=20):
    tot = 0.0
    for r in range(reps + 3):
        torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if r >= 3: tot += e0.elapsed_time(e1)
    return tot / reps * 1e3
print('words.zero_ %.1f us | amax launch %.1f us | pack launch %.1f us' % (
    tm(lambda: pl.words.zero_()), tm(lambda: ops.L.call('gpe_pack_multi', pl.pre_table, pl.pre_table.numel() // 64, pl.pre_blocks)),
    tm(lambda: ops.L.call('gpe_pack_multi', pl.table, len(pl.specs), pl.blocks))))
# per kind: a table with only that kind's jobs
tab = pl.table.cpu().numpy().view(ops._JOB)
for kind in sorted(set(int(j['kind']) for j in tab)):
    sub = tab[[int(j['kind']) == kind for j in tab]].copy()
    blk = 0
    for j in range(len(sub)):
        sub[j]['first_block'] = blk
        blk += ops.L.query('gpe_pack_job_blocks', kind, int(sub[j]['total']), int(sub[j]['Npad']), int(sub[j]['K']))
    t = torch.from_numpy(sub.view(np.uint8).copy()).cuda()
    print(' kind %2d: %2d jobs, %5d blocks, %.1f MB out: %.1f us' % (kind, len(sub), blk, sum(int(j['total']) for j in sub) * 4 / 1e6,
                                                                  tm(lambda: ops.L.call('gpe_pack_multi', t, len(sub), blk))))
