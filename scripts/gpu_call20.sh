#!/bin/bash
# round 4, call 20: kNN with a seeded insertion bound (the previous layer's graph) — bit-exactness + A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "knn or (f16x3 and (cfg2_shape or full3d)) or cfg2_full" > gpurun_out/r04m_tests.log 2>&1
tail -5 gpurun_out/r04m_tests.log
for V in 0 1 0 1; do
  GPE_KNN_NOSEED=$V timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fast-math-line > gpurun_out/r04m_noseed_$V.log 2>&1
  grep '^{' gpurun_out/r04m_noseed_$V.log | tail -1 > gpurun_out/r04m_noseed_$V.json
  python - <<PY
import json
V='$V'
try:
    d=json.load(open('gpurun_out/r04m_noseed_%s.json'%V))
    pk=d['roofline_per_kernel']
    print('GPE_KNN_NOSEED=%s'%V, round(d['value'],1), round(d['ms_per_step'],3), 'knn filter us', round(pk['gpe_knn:filter']['avg_launch_ms']*1e3,1), 'exact us', round(pk['gpe_knn:exact']['avg_launch_ms']*1e3,1), 'loss', d['config'].get('final_loss'))
except Exception as e:
    print(V, 'FAILED', e); print(open('gpurun_out/r04m_noseed_%s.log'%V).read()[-1500:])
PY
done
