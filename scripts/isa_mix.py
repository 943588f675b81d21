#!/usr/bin/env python3
"""Static instruction mix of a kernel's main loop (works on the build box, no GPU):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 [per-file flags of build.py] -S --cuda-device-only -o x.s csrc/<file>.hip
    python scripts/isa_mix.py x.s '<substring of the mangled kernel name>' [...]
Buckets the kernel's instructions by the depth-1 loop LLVM's block comments assign them to (the persistent tile loop of the
edge kernels; the consumer and the producer loop of the reduce-GEMMs) and counts each loop's instructions by class.  A lone wave pays ~4 cycles per VALU instruction that does not overlap its own MFMAs (profiles/r02_c_ubench.md), so
VALU count x 4 is the issue budget a change to the epilogue / staging code moves."""
import collections
import re
import sys


def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfma'):
        return 'MFMA'
    if op.startswith('v_accvgpr'):
        return 'ACC_MOV'
    if op.startswith('v_pk_'):
        return 'VALU_PK'
    if op.startswith('v_cvt'):
        return 'VALU_CVT'
    if op.startswith('v_cndmask'):
        return 'VALU_SEL'
    if op.startswith('v_cmp'):
        return 'VALU_CMP'
    if op.startswith('v_readlane') or op.startswith('v_readfirstlane') or op.startswith('v_writelane'):
        return 'VALU_LANE'
    if op.startswith('v_'):
        return 'VALU'
    if op in ('s_waitcnt', 's_barrier', 's_nop', 's_sleep'):
        return op
    if op.startswith('s_cbranch') or op == 's_branch':
        return 'BRANCH'
    if op.startswith('s_'):
        return 'SALU'
    if op.startswith('ds_'):
        return 'LDS'
    if op.startswith(('global_', 'buffer_', 'flat_')):
        return 'VMEM'
    if op.startswith('scratch_'):
        return 'SCRATCH'
    return 'OTHER'


def kernel_lines(text, needle):
    lines = text.splitlines()
    start = None
    for i, l in enumerate(lines):
        if l.startswith('_Z') and l.rstrip().split(':')[0].find(needle) >= 0 and ':' in l:
            start = i
            break
    if start is None:
        raise SystemExit('kernel %r not found' % needle)
    end = start + 1
    while end < len(lines) and not lines[end].startswith('.Lfunc_end'):
        end += 1
    return lines[start].split(':')[0], lines[start + 1:end]


def main():
    text = open(sys.argv[1]).read()
    for needle in sys.argv[2:]:
        name, body = kernel_lines(text, needle)
        # LLVM annotates every basic block with the loop it belongs to: "; =>This Loop Header: Depth=1" on the header's label,
        # ";   in Loop: Header=BB7_67 Depth=1" on its members, and the function's first block has none
        insts, cur = [], None
        for l in body:
            t = l.strip()
            m = re.match(r'^\.?L?(BB\d+_\d+):(.*)$', t)
            if m:
                c = m.group(2)
                h = re.search(r'Header=(BB\d+_\d+) Depth=1', c)
                cur = m.group(1) if 'Loop Header: Depth=1' in c else (h.group(1) if h else None)
                continue
            m = re.match(r'^; %bb\.\d+:(.*)$', t)
            if m:
                h = re.search(r'Header=(BB\d+_\d+) Depth=1', m.group(1))
                cur = h.group(1) if h else (cur if 'in Loop' in m.group(1) else None)
                continue
            if not t or t.startswith((';', '.', '//')):
                continue
            insts.append((t.split()[0], cur))
        print('%s\n  whole kernel: %d instructions' % (name, len(insts)))
        loops = collections.defaultdict(list)
        for op, h in insts:
            if h:
                loops[h].append(op)
        for h, ops_ in sorted(loops.items(), key=lambda kv: -len(kv[1]))[:3]:
            if len(ops_) < 100:
                continue
            loop = collections.Counter(classify(op) for op in ops_)
            valu = sum(v for k, v in loop.items() if k.startswith('VALU'))
            print('  loop %s: %d instructions; MFMA %d, VALU %d (%s), ACC_MOV %d, SALU %d, LDS %d, VMEM %d, SCRATCH %d, waitcnt %d, '
                  'barrier %d, branch %d, nop %d' % (h, len(ops_), loop['MFMA'], valu,
                                                     ', '.join('%s %d' % (k[5:] or 'plain', v) for k, v in sorted(loop.items()) if k.startswith('VALU')),
                                                     loop['ACC_MOV'], loop['SALU'], loop['LDS'], loop['VMEM'], loop['SCRATCH'], loop['s_waitcnt'],
                                                     loop['s_barrier'], loop['BRANCH'], loop['s_nop']))
            top = collections.Counter(op for op in ops_ if classify(op).startswith('VALU'))
            print('    top VALU: ' + ', '.join('%s %d' % kv for kv in top.most_common(12)))


if __name__ == '__main__':
    main()
