#!/usr/bin/env python3
"""A/B build of several translation units with extra -D macros (their per-file scheduler flags kept), linked against the current
objects of the others:  scripts/ab_build_macro.py TAG "-DX=1 .." file1.hip file2.hip ...   -> build/ab/lib_TAG.so"""
import concurrent.futures
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'garment-pattern-estimation_amd'))
import build as B
tag, macros, files = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
out = os.path.join(ROOT, 'build', 'ab')
os.makedirs(out, exist_ok=True)


def comp(f):
    o = os.path.join(out, '%s_%s.o' % (f[:-4], tag))
    cmd = [B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(f, []) + macros + ['-c', os.path.join(B.CSRC, f), '-o', o]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-3000:])
    return o


with concurrent.futures.ThreadPoolExecutor(8) as ex:
    objs = list(ex.map(comp, files))
others = [os.path.join(B.CSRC, f[:-4] + '.o') for f in B._sources() if f not in files]
lib = os.path.join(out, 'lib_%s.so' % tag)
r = subprocess.run([B.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + others + objs, capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr[-3000:])
print('built', lib)
