#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage).
    python scripts/kernel_resources.py gpe_edgegemm_sr.hip [filter]
Used on the CPU build box to check that a kernel change did not introduce scratch spills."""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'garment-pattern-estimation_amd', 'csrc')


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c',
                        os.path.join(CSRC, src), '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'],
                       capture_output=True, text=True)
    cur = None
    rows = []
    for line in r.stderr.splitlines():
        m = re.search(r'remark:\s+(.*?) \[-Rpass', line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith('Function Name:'):
            cur = {'name': t.split(':', 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ':' in t:
            k, v = t.split(':', 1)
            cur[k.strip()] = v.strip()
    for c in rows:
        name = subprocess.run(['/usr/bin/c++filt', c['name']], capture_output=True, text=True).stdout.strip()
        if flt and flt not in name:
            continue
        print('%-90s sgpr %-4s vgpr %-4s agpr %-4s scratch %-6s spill s%s/v%s occ %-2s lds %s' % (
            name[:90], c.get('TotalSGPRs'), c.get('VGPRs'), c.get('AGPRs'), c.get('ScratchSize [bytes/lane]'), c.get('SGPRs Spill'), c.get('VGPRs Spill'),
            c.get('Occupancy [waves/SIMD]'), c.get('LDS Size [bytes/block]')))
    if r.returncode:
        sys.stderr.write(r.stderr[-3000:])
        sys.exit(1)


if __name__ == '__main__':
    main()
