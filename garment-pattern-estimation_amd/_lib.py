"""ctypes binding of libgpe_hip.so (C ABI: include/gpe_hip.h).  Signatures are parsed from the header itself, so
the binding cannot drift from the declared ABI.  There is NO fallback: if the HIP library is missing or a call
fails, this raises — the product path never routes around the kernels."""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPE_HIP_LIB selects another build of the same ABI (A/B measurements of kernel variants: scripts/ab_probe.sh)
LIB_PATH = os.environ.get('GPE_HIP_LIB') or os.path.join(_HERE, 'libgpe_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'gpe_hip.h')

_CT = {'p': ctypes.c_void_p, 'i': ctypes.c_int, 'l': ctypes.c_long, 'f': ctypes.c_float, 'd': ctypes.c_double}


def _code(ctype):
    t = ctype.strip()
    if '*' in t:
        return 'p'
    t = t.replace('const', '').strip()
    return {'int': 'i', 'long': 'l', 'float': 'f', 'double': 'd'}[t]


def parse_header(path=HEADER_PATH):
    """-> {symbol: (restype code, [arg codes])} for every function declared in include/gpe_hip.h"""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    src = re.sub(r'//[^\n]*', ' ', src)
    sigs = {}
    for m in re.finditer(r'\b(int|long)\s+(gpe_\w+)\s*\(([^)]*)\)\s*;', src):
        res, name, args = m.group(1), m.group(2), m.group(3).strip()
        codes = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                ctype = a[:a.rfind(' ')] if '*' not in a else a[:a.rfind('*') + 1]
                codes.append(_code(ctype))
        sigs[name] = (_code(res), codes)
    return sigs


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libgpe_hip.so not built: run `python -c "import __graft_entry__ as g; g.build()"` '
                '(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.')
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in parse_header().items():
            fn = getattr(l, name)          # AttributeError here = header/library mismatch: fail loudly
            fn.restype = _CT[res]
            fn.argtypes = [_CT[a] for a in args]
        _lib = l
        env = os.environ.get('GPE_MATH')
        if env:
            set_math(env)
        dbg = os.environ.get('GPE_DEBUG_SET')       # measurement switches of include/gpe_hip.h gpe_debug_set (A/B runs of bench.py)
        if dbg:
            l.gpe_debug_set(int(dbg, 0))
    return _lib


MATH_MODES = {'f32': 0, 'bf16x3': 1, 'mixed': 2, 'bf16x6': 3, 'f16x3': 4}


def set_math(mode):
    """Arithmetic of the fused per-edge GEMMs (include/gpe_hip.h gpe_math_set): 'f32' exact fp32 MFMA (library default);
    'f16x3' two-term fp16 split on tensor-normalised operands, three fp16 MFMAs per product, fp32 accumulate — parity-grade, the
    mode bench.py times; 'bf16x6' three-term bf16 split (parity-grade, superseded); 'bf16x3' / 'mixed' two-term bf16 splits
    (approximate).  Returns the previous mode's name.  The environment variable GPE_MATH selects the mode at library load."""
    if mode not in MATH_MODES:
        raise ValueError('unknown math mode %r (choose from %s)' % (mode, sorted(MATH_MODES)))
    prev = lib().gpe_math_set(MATH_MODES[mode])
    if prev < 0:
        raise RuntimeError('gpe_math_set failed with code %d' % prev)
    return {v: k for k, v in MATH_MODES.items()}[prev]


def set_f16x3_min_rows(rows):
    """Size gate of the f16x3 mode: edge launches with fewer rows run the exact fp32 kernels (default 32768; 0 = always use the
    fp16 pipe, which is what the parity tests of small fixtures set).  Returns the previous value."""
    prev = lib().gpe_f16x3_min_rows_set(int(rows))
    if prev < 0:
        raise RuntimeError('gpe_f16x3_min_rows_set failed with code %d' % prev)
    return prev


def set_reserved_cus(n):
    """Leave `n` compute units out of every persistent launch (include/gpe_hip.h gpe_reserve_cus_set): room for RCCL's kernels while
    the fused edge kernels run.  Returns the previous reservation."""
    prev = lib().gpe_reserve_cus_set(int(n))
    if prev < 0:
        raise ValueError('reserved CUs must lie in 0 .. 192')
    return prev


def get_math():
    return {v: k for k, v in MATH_MODES.items()}[lib().gpe_math_get()]


def _conv(a):
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        return a.data_ptr()
    return a


# When a list, every call is bracketed by HIP events recorded on the stream the kernel is launched on (torch's
# current stream): entries are (name, args, start_event, end_event).  Used by bench.py for per-kernel durations.
TIMING = None


# the raw handle of torch's current stream without building a torch.cuda.Stream object per call (a small step — cfg 1, the att
# k = 5 shape — issues ~100 launches in 4 - 7 ms: the host side of a call is part of the step there)
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None) or torch.cuda.current_device
_FN = {}


def call(name, *args):
    """Invoke a C-ABI entry point on torch's current HIP stream (appended as the trailing `stream` argument)."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(lib(), name)
    # kernels run on the CURRENT device's stream: refuse tensors that live elsewhere (a model on cuda:1 while cuda:0 is
    # current would otherwise hand foreign pointers to the wrong device's queue)
    cur = _cur_device()
    for a in args:
        if isinstance(a, torch.Tensor):
            if not a.is_cuda or a.device.index != cur:
                raise RuntimeError('%s: tensor on %s but the current device is cuda:%d — wrap the call in '
                                   'torch.cuda.device(...) (one process per GPU is the supported layout)'
                                   % (name, a.device, cur))
            break
    if TIMING is not None or _raw_stream is None:
        stream = torch.cuda.current_stream()
        handle = stream.cuda_stream
    else:
        handle = _raw_stream(cur)
    if TIMING is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
    rc = fn(*[a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args], handle)
    if TIMING is not None:
        e1.record(stream)
        TIMING.append((name, tuple(a for a in args if isinstance(a, (int, float))), e0, e1))
    if rc != 0:
        raise RuntimeError('%s failed with code %d' % (name, rc))


def query(name, *args):
    """Host-only entry points (sizes / constants)."""
    return getattr(lib(), name)(*args)
