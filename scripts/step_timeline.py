#!/usr/bin/env python3
"""One training step as an ordered kernel timeline (start offset, duration, idle time in front) out of a rocprofv3 kernel trace
(rocpd sqlite).  usage: step_timeline.py <results.db> <step index from the end, default 2> > profiles/<name>.md
A step boundary is the gpe_adam_kernel launch."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cols = [r[1] for r in db.execute('pragma table_info(kernels)').fetchall()]
qcol = 'stream_id' if 'stream_id' in cols else ('queue_id' if 'queue_id' in cols else None)       # which HIP stream / HSA queue (round 6: the side stream)
ks = db.execute('select name, start, end%s from kernels order by start' % (', ' + qcol if qcol else ', 0')).fetchall()
ends = [i for i, k in enumerate(ks) if 'gpe_adam_kernel' in k[0]]
if len(ends) < back + 1:
    sys.exit('not enough steps in the trace')
lo, hi = ends[-back - 1] + 1, ends[-back] + 1
step = ks[lo:hi]
t0 = ks[lo - 1][2]
busy = sum(e - s for _, s, e, _q in step) / 1e3
span = (step[-1][2] - t0) / 1e3
print('step of %d kernels: span %.1f us, kernel time %.1f us, idle %.1f us (%.1f %%)\n' % (len(step), span, busy, span - busy,
                                                                                          100 * (span - busy) / span))
qs = sorted(set(q for _, _, _, q in step), key=lambda q: -sum(1 for k in step if k[3] == q))
print('| # | kernel | start us | dur us | idle before us (negative: it runs beside the previous ones) | stream |')
print('|---|---|---|---|---|---|')
prev = t0
for i, (n, s, e, q) in enumerate(step):
    name = re.sub(r'\(.*', '', n).replace('void ', '')[:70]
    print('| %d | %s | %.1f | %.1f | %.1f | %s |' % (i, name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, 'main' if q == qs[0] else 'side'))
    prev = max(prev, e)
