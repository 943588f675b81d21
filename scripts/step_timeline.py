#!/usr/bin/env python3
"""One training step as an ordered kernel timeline (start offset, duration, idle time in front) out of a rocprofv3 kernel trace
(rocpd sqlite).  usage: step_timeline.py <results.db> <step index from the end, default 2> > profiles/<name>.md
A step boundary is the gpe_adam_kernel launch."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ks = db.execute('select name, start, end from kernels order by start').fetchall()
ends = [i for i, k in enumerate(ks) if 'gpe_adam_kernel' in k[0]]
if len(ends) < back + 1:
    sys.exit('not enough steps in the trace')
lo, hi = ends[-back - 1] + 1, ends[-back] + 1
step = ks[lo:hi]
t0 = ks[lo - 1][2]
busy = sum(e - s for _, s, e in step) / 1e3
span = (step[-1][2] - t0) / 1e3
print('step of %d kernels: span %.1f us, kernel time %.1f us, idle %.1f us (%.1f %%)\n' % (len(step), span, busy, span - busy,
                                                                                          100 * (span - busy) / span))
print('| # | kernel | start us | dur us | idle before us |')
print('|---|---|---|---|---|')
prev = t0
for i, (n, s, e) in enumerate(step):
    name = re.sub(r'\(.*', '', n).replace('void ', '')[:70]
    print('| %d | %s | %.1f | %.1f | %.1f |' % (i, name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3))
    prev = max(prev, e)
