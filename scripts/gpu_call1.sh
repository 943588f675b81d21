#!/bin/bash
# round 4, call 1: full -m gpu suite on the ABI-v4 tree, the default bench line, the cfg-1 gate check and the shipped-YAML shape
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x --durations=15 > gpurun_out/r04a_tests.log 2>&1
tail -5 gpurun_out/r04a_tests.log
timeout 600 python bench.py --steps 100 --warmup 5 > gpurun_out/r04a_bench.log 2>&1
grep '^{' gpurun_out/r04a_bench.log | tail -1 > gpurun_out/r04a_bench.json
timeout 300 python bench.py --points 1024 --batch 8 --k 5 --steps 50 --no-cpu-baseline > gpurun_out/r04a_cfg1.log 2>&1
grep '^{' gpurun_out/r04a_cfg1.log | tail -1 > gpurun_out/r04a_cfg1_bench.json
timeout 300 python bench.py --model att --points 2000 --batch 30 --k 5 --steps 50 --no-cpu-baseline > gpurun_out/r04a_attk5.log 2>&1
grep '^{' gpurun_out/r04a_attk5.log | tail -1 > gpurun_out/r04a_attk5_bench.json
python - <<PY
import json
for f in ['r04a_bench','r04a_cfg1_bench','r04a_attk5_bench']:
    try:
        d=json.load(open('gpurun_out/%s.json'%f))
        print(f, round(d['value'],1), round(d['ms_per_step'],3), 'exact', d.get('exact_f32') and round(d['exact_f32']['value'],1), d['roofline']['kernel'], d['roofline']['bound'], round(d['roofline']['frac'],3))
    except Exception as e:
        print(f, 'FAILED', e)
PY
