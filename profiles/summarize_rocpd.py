#!/usr/bin/env python3
"""Turns a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into the per-kernel stats table kept under
profiles/.   usage: summarize_rocpd.py <results.db> <steps-in-trace> > profiles/<name>.md"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = db.execute('select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, '
                  'max(end-start)/1e3, max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels '
                  'group by name order by 3 desc').fetchall()
tot = sum(r[2] for r in rows)
print('total kernel time %.1f us over %g steps = %.2f ms/step\n' % (tot, steps, tot / steps / 1e3))
print('| kernel | launches/step | total us | avg us | min us | max us | % | vgpr | agpr | lds |')
print('|---|---|---|---|---|---|---|---|---|---|')
for r in rows:
    name = re.sub(r'\(.*', '', r[0]).replace('void ', '')[:80]
    print('| %s | %.1f | %.0f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s |' % (
        name, r[1] / steps, r[2], r[3], r[4], r[5], 100 * r[2] / tot, r[6], r[7], r[8]))

# ---- machine-readable averages (bench.py quotes the dominant kernel's next to its HIP-event duration): --json <file> ------------
if '--json' in sys.argv:
    import hashlib
    import json
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'garment-pattern-estimation_amd', 'csrc')
    h = hashlib.sha1()
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), 'rb').read())
    out = {'csrc_sha': h.hexdigest()[:16], 'steps': steps, 'source': 'rocprofv3 --kernel-trace (rocpd), per kernel over the whole trace',
           'kernels': {re.sub(r'\(.*', '', r[0]).replace('void ', ''): {'launches': r[1], 'total_us': r[2], 'avg_us': r[3]} for r in rows}}
    json.dump(out, open(sys.argv[sys.argv.index('--json') + 1], 'w'), indent=1)

# ---- idle gaps between consecutive kernels (where the GPU waits for the host) ------------------------------------
if len(sys.argv) > 3 and sys.argv[3] == 'gaps':
    ks = db.execute('select name, start, end from kernels order by start').fetchall()
    gaps = {}
    tot_gap = 0.0
    for (n0, s0, e0), (n1, s1, e1) in zip(ks[:-1], ks[1:]):
        g = (s1 - e0) / 1e3
        if g <= 0 or g > 5000:          # overlap, or a step boundary / sync
            continue
        key = re.sub(r'\(.*', '', n1).replace('void ', '')[:60]
        d = gaps.setdefault(key, [0, 0.0])
        d[0] += 1
        d[1] += g
        tot_gap += g
    print('\nidle time in front of each kernel (gaps < 5 ms): %.2f ms/step\n' % (tot_gap / steps / 1e3))
    print('| next kernel | gaps/step | idle us/step | avg us |')
    print('|---|---|---|---|')
    for k, (n, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
        print('| %s | %.1f | %.0f | %.1f |' % (k, n / steps, g / steps, g / n))
