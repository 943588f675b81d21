#!/usr/bin/env python3
"""Per-kernel SQ wave-state counters from one rocprofv3 PMC pass:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
        --kernel-trace --output-format csv -d <dir> -- <cmd>
usage: summarize_sq.py <dir> > table.md        (counter semantics: MI355X_MICROARCH.md, "Per-pass counter slots")
WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~= WAVE_CYCLES; the three are given
as fractions of WAVE_CYCLES.  GRBM_GUI_ACTIVE comes back SUMMED over the 8 XCDs (printed per launch as `gui/8` cycles: it matches
kernel duration x shader clock), so MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)."""
import csv, glob, os, re, sys
acc = {}
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        name = re.sub(r'\(.*', '', row['Kernel_Name']).replace('void ', '')
        d = acc.setdefault(name, {})
        c = d.setdefault(row['Counter_Name'], [0, 0.0])
        c[0] += 1
        c[1] += float(row['Counter_Value'])
print('| kernel | launches | gui/8 cycles per launch | MFMA busy | active issue | issue stall | parked (waitcnt / barrier) |')
print('|---|---|---|---|---|---|---|')
rows = []
for name, d in acc.items():
    g = lambda k: d.get(k, [0, 0.0])[1]
    wc, gui = g('SQ_WAVE_CYCLES'), g('GRBM_GUI_ACTIVE')
    if wc <= 0 or gui <= 0:
        continue
    rows.append((gui, name, d['SQ_WAVE_CYCLES'][0], g('SQ_VALU_MFMA_BUSY_CYCLES') / (gui / 8.0 * 1024.0), g('SQ_ACTIVE_INST_ANY') / wc,
                 g('SQ_WAIT_INST_ANY') / wc, g('SQ_WAIT_ANY') / wc))
for gui, name, n, mf, a, wi, wa in sorted(rows, reverse=True)[:14]:
    print('| %s | %d | %.0f | %.2f | %.2f | %.2f | %.2f |' % (name, n, gui / 8.0 / n, mf, a, wi, wa))
