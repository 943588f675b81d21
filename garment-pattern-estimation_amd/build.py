"""Builds libgpe_hip.so (gfx950) in-tree with hipcc.  No torch extension machinery: the library is a plain
C-ABI shared object (include/gpe_hip.h) loaded through ctypes by _lib.py."""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libgpe_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         '-Wno-unused-variable', '-Wno-unused-but-set-variable']
# per-file additions: the dense-A single-role edge kernels are faster under LLVM's max-ILP scheduling strategy (measured A/B,
# csrc/gpe_edgegemm_sr_kernel.h); every other file keeps the default scheduler
EXTRA_FLAGS = {'gpe_edgegemm_sr_dense.hip': ['-mllvm', '-amdgpu-sched-strategy=max-ilp'],
               # whole-step A/B in one session (scripts/ab_bench.sh): edge weight-gradient reduce-GEMMs 3.01 -> 2.94 ms per step
               'gpe_redgemm.hip': ['-mllvm', '-amdgpu-sched-strategy=max-memory-clause'],
               # kNN 1.19 -> 1.16 ms per step
               'gpe_knn.hip': ['-mllvm', '-amdgpu-sched-strategy=max-ilp'],
               # (KNN_FT_TIMING=1: per-section cycle counters instead of the first list of every wave, scripts/knn_ft_sections.py)
               'gpe_knn_ft.hip': (['-DKNN_FT_TIMING'] if os.environ.get('KNN_FT_TIMING') else []) + os.environ.get('KNN_FT_FLAGS', '').split(),
               # the bf16x6 mode's forward edge kernels 2.94 -> 2.89 ms per step
               'gpe_edgegemm_x6.hip': ['-mllvm', '-amdgpu-sched-strategy=max-memory-clause'],
               # the f16x3 mode's edge kernels: forward 2.50 -> 2.43, backward 2.24 -> 2.05 ms per step (with the slot fences off)
               'gpe_edgegemm_h3.hip': ['-mllvm', '-amdgpu-sched-strategy=max-ilp'],
               # the two-waves-per-SIMD instances of the same kernels: F2 / B2 (gpe_edgegemm_w8.hip) under the default scheduler,
               # F3 and B3 under max-ilp (without / with the slot fence: set in the files) — table in gpe_edgegemm_w8.hip
               'gpe_edgegemm_w8_f3.hip': ['-mllvm', '-amdgpu-sched-strategy=max-ilp'],
               'gpe_edgegemm_w8_b3.hip': ['-mllvm', '-amdgpu-sched-strategy=max-ilp']}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    # this file is a dependency too: the compile flags (FLAGS / EXTRA_FLAGS) live here
    deps = [os.path.join(CSRC, src), os.path.abspath(__file__)] + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith('.h')]
    return any(os.path.getmtime(d) > t for d in deps)


COMPILED, REUSED = [], []


def _compile(src):
    obj = os.path.join(CSRC, src[:-4] + '.o')
    if not _stale(obj, src):
        REUSED.append(src)
    else:
        COMPILED.append(src)
        cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + ['-c', os.path.join(CSRC, src), '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (src, r.stderr[-6000:]))
        if r.stderr.strip():
            sys.stderr.write(r.stderr[-3000:])
    return obj


def build(force=False, verbose=True):
    force = force or os.environ.get('GPE_FORCE_BUILD') == '1'
    del COMPILED[:], REUSED[:]
    srcs = _sources()
    if force:
        for s in srcs:
            o = os.path.join(CSRC, s[:-4] + '.o')
            if os.path.exists(o):
                os.remove(o)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    if (not os.path.exists(OUT)) or any(os.path.getmtime(o) > os.path.getmtime(OUT) for o in objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stderr[-4000:])
        if verbose:
            print('built', OUT)
    if verbose:
        # say what this call actually did: a driver-side "does it build" check must not mistake reuse for a compile
        print('build: compiled %d source(s) %s; reused %d up-to-date object(s) %s (force=%s; GPE_FORCE_BUILD=1 or '
              'build(force=True) recompiles everything)' % (len(COMPILED), COMPILED, len(REUSED), REUSED, force))
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
