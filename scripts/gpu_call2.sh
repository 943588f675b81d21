#!/bin/bash
# round 4, call 2: fp16-pipe kNN filter + branch-free left-over tiles of the two-plane edge kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=5 > gpurun_out/r04b_tests.log 2>&1
tail -5 gpurun_out/r04b_tests.log
grep -E "^(FAILED|ERROR)" gpurun_out/r04b_tests.log | head -20
( timeout 200 python scripts/run_knn.py; GPE_KNN_F32FILTER=1 timeout 200 python scripts/run_knn.py; timeout 200 python scripts/run_knn.py 8192 64; GPE_KNN_F32FILTER=1 timeout 200 python scripts/run_knn.py 8192 64 ) > gpurun_out/r04b_knn.txt 2>&1
cat gpurun_out/r04b_knn.txt
timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r04b_bench.log 2>&1
grep '^{' gpurun_out/r04b_bench.log | tail -1 > gpurun_out/r04b_bench.json
python - <<PY
import json
d=json.load(open('gpurun_out/r04b_bench.json'))
print('value', round(d['value'],1), round(d['ms_per_step'],3), 'exact', d.get('exact_f32') and round(d['exact_f32']['value'],1))
for k,v in d['roofline_per_kernel'].items(): print('%-28s %5.2f/step x%.0f %7.1f us  %s frac %.2f' % (k, v['ms_per_step'], v['launches_per_step'], v['avg_launch_ms']*1e3, v['bound'], v['frac']))
PY
