#!/bin/bash
# round 4: batch scaling on one GPU (N = 2048, k = 16): what a larger per-GPU shard buys
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for B in 16 64 128 256; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 4 --no-cpu-baseline --no-fast-math-line > gpurun_out/r04k_b$B.log 2>&1
  grep '^{' gpurun_out/r04k_b$B.log | tail -1 > gpurun_out/r04_k_batch${B}_bench.json
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r04_k_batch${B}_bench.json'))
    print('B=$B', round(d['value'],1), 'garments/s', round(d['ms_per_step'],3), 'ms/step', 'exact', d['exact_f32'] and round(d['exact_f32']['value'],1))
except Exception as e:
    print('B=$B FAILED', e); print(open('gpurun_out/r04k_b$B.log').read()[-800:])
PY
done
