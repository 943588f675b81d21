#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are collected in separate runs:
together they exceed the TCC counter slots).   usage: summarize_pmc.py <fetch-dir> <write-dir> [training steps profiled] > traffic.json

Units/corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports half the
bytes of wide coalesced reads (128-B requests tallied at 64 B), so reads are doubled; WRITE_SIZE is taken as reported
(uncalibrated per the guide).  Output: {kernel name: {launches, fetch_bytes, write_bytes, hbm_bytes}} per launch."""
import csv
import glob
import json
import os
import re
import sys


def collect(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] != counter:
                continue
            name = re.sub(r'\(.*', '', row['Kernel_Name']).replace('void ', '')
            a = acc.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += float(row['Counter_Value'])
    return acc


fetch = collect(sys.argv[1], 'FETCH_SIZE')
write = collect(sys.argv[2], 'WRITE_SIZE')
out = {}
for name in sorted(set(fetch) | set(write)):
    nf, kf = fetch.get(name, [0, 0.0])
    nw, kw = write.get(name, [0, 0.0])
    fb = 2.0 * 1024.0 * kf / max(nf, 1)
    wb = 1024.0 * kw / max(nw, 1)
    out[name] = {'launches': max(nf, nw), 'fetch_bytes': fb, 'write_bytes': wb, 'hbm_bytes': fb + wb}
def csrc_sha():
    """sha1 over the kernel sources of this tree (same definition as bench.py csrc_sha): bench.py quotes a traffic file only
    for the code that produced it."""
    import hashlib
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'garment-pattern-estimation_amd', 'csrc')
    h = hashlib.sha1()
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


json.dump({'csrc_sha': csrc_sha(), 'steps_profiled': int(sys.argv[3]) if len(sys.argv) > 3 else 3, 'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), reads x2 (gfx950 correction), per launch',
           'kernels': out}, sys.stdout, indent=1)
