#!/usr/bin/env python3
"""Order of memory events in a kernel's loops (works on the GPU-less build box):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 [per-file flags of build.py] -S --cuda-device-only -o x.s csrc/<file>.hip
    python scripts/isa_events.py x.s '<substring of the mangled kernel name>' [min MFMAs per loop, default 0]
For every depth-1 loop: L = global / buffer load, S = store, wN = s_waitcnt vmcnt(N), X = scratch access, | = s_barrier, z = s_sleep,
D = LDS read / d = LDS write are omitted unless --lds.  What to look for (round 5, profiles/r05_a_w8_schedules.md):
  * `L w0` pairs — a load followed at once by a full wait: the compiler sank it into the conditional block that consumes it;
  * `w0 S w0 S` — every store behind a full wait (an `asm volatile` store, or a store whose data registers are reloaded);
  * X inside a loop — spill traffic per iteration."""
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 0
    lds = '--lds' in sys.argv
    lines = open(path).read().split('\n')
    starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
    starts.append(len(lines))
    for k in range(len(starts) - 1):
        name = lines[starts[k]].split(':')[0]
        if pat not in name:
            continue
        seg = lines[starts[k]:starts[k + 1]]
        end_fn = next((i for i, l in enumerate(seg) if l.startswith('.Lfunc_end')), len(seg))
        seg = seg[:end_fn]
        heads = [i for i, l in enumerate(seg) if 'Loop Header: Depth=1' in l]
        print(name)
        for hi, h in enumerate(heads):
            end = heads[hi + 1] if hi + 1 < len(heads) else len(seg)
            body = seg[h:end]
            n_mfma = sum('v_mfma' in l for l in body)
            if n_mfma < min_mfma:
                continue
            ev = []
            for l in body:
                t = l.strip()
                if t.startswith(('global_load', 'buffer_load', 'flat_load')):
                    ev.append('L')
                elif t.startswith(('global_store', 'buffer_store', 'flat_store', 'global_atomic')):
                    ev.append('S')
                elif t.startswith('s_waitcnt') and 'vmcnt' in t:
                    ev.append('w' + re.search(r'vmcnt\((\d+)\)', t).group(1))
                elif t.startswith('scratch_'):
                    ev.append('X')
                elif t.startswith('s_barrier'):
                    ev.append('|')
                elif t.startswith('s_sleep'):
                    ev.append('z')
                elif lds and t.startswith('ds_read'):
                    ev.append('D')
                elif lds and t.startswith('ds_write'):
                    ev.append('d')
            n_valu = sum(bool(re.match(r'\s+v_(?!mfma)', l)) for l in body)
            print('  loop@%d: %d lines, MFMA %d, VALU %d: %s' % (h, len(body), n_mfma, n_valu, ' '.join(ev)))


if __name__ == '__main__':
    main()
