#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python scripts/row_g_probe.py 4 2048 16 > gpurun_out/r04h_row_g.txt 2>&1; tail -8 gpurun_out/r04h_row_g.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "redgemm or edgeconv or cfg1 or cfg2_shape or dense_mlp" 2>&1 | tail -4
timeout 300 python bench.py --points 1024 --batch 8 --k 5 --steps 100 --no-cpu-baseline --no-fast-math-line > gpurun_out/r04h_cfg1.log 2>&1
grep '^{' gpurun_out/r04h_cfg1.log | tail -1 > gpurun_out/r04h_cfg1.json
python - <<PY
import json
d=json.load(open('gpurun_out/r04h_cfg1.json'))
print('cfg1', round(d['value'],1), round(d['ms_per_step'],3), 'exact', d.get('exact_f32'))
k=d['kernel_ms_per_step']
print(sorted(((round(v['ms_per_step'],3),n) for n,v in k.items()),reverse=True)[:10])
PY
