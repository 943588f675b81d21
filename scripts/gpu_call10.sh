#!/bin/bash
# round 4, call 10: lazy-operand loads ahead of the row loads (B3 lazy) + current call-shape table
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "lazy_dz3 or (f16x3 and cfg2_shape)" 2>&1 | tail -3
timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fast-math-line --call-shapes gpurun_out/r04h_shapes.txt > gpurun_out/r04h_bench.log 2>&1
grep '^{' gpurun_out/r04h_bench.log | tail -1 > gpurun_out/r04h_bench.json
python - <<PY
import json
d=json.load(open('gpurun_out/r04h_bench.json'))
k=d['kernel_ms_per_step']
print(round(d['value'],1), round(d['ms_per_step'],3))
for n, r in list(d.get('roofline_per_kernel', {}).items())[:16]:
    print('   ', n, r.get('bound'), round(r.get('frac') or 0,3), round(r.get('avg_launch_ms') or 0,4))
PY
sort -t'x' -k2 gpurun_out/r04h_shapes.txt | head -0
grep -E "gpe_redgemm |gpe_linear |gpe_knn|gpe_rnn" gpurun_out/r04h_shapes.txt | head -40
