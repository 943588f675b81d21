#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/diagprof -o diag -- python scripts/lazy_diag.py 32 > gpurun_out/r04g_diag.log 2>&1
grep -v simple_timer gpurun_out/r04g_diag.log | tail -22; timeout 300 python scripts/lazy_diag.py 8 | tail -12
find gpurun_out/diagprof -name "*kernel_stats*" | head -3
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/diagprof/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Name']
        if 'redgemm' in n or 'edgegemm' in n or 'dz3' in n:
            print(r['Calls'], r['AverageNs'], n[:160])
PY
