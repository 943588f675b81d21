#!/bin/bash
# round 4: the non-default configurations again (bench.py quotes counter traffic only for the default workload)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python bench.py --points 1024 --batch 8 --k 5 --steps 100 --no-cpu-baseline > gpurun_out/r04j_cfg1.log 2>&1
grep '^{' gpurun_out/r04j_cfg1.log | tail -1 > gpurun_out/r04_j_cfg1_bench.json
timeout 300 python bench.py --model att --points 4096 --k 20 --steps 30 --no-cpu-baseline > gpurun_out/r04j_cfg4.log 2>&1
grep '^{' gpurun_out/r04j_cfg4.log | tail -1 > gpurun_out/r04_j_cfg4_bench.json
timeout 400 python bench.py --points 8192 --batch 64 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r04j_cfg5.log 2>&1
grep '^{' gpurun_out/r04j_cfg5.log | tail -1 > gpurun_out/r04_j_cfg5_share_bench.json
timeout 300 python bench.py --model att --points 2000 --batch 30 --k 5 --steps 100 --no-cpu-baseline > gpurun_out/r04j_attk5.log 2>&1
grep '^{' gpurun_out/r04j_attk5.log | tail -1 > gpurun_out/r04_j_att_k5_bench.json
timeout 300 python bench.py --epoch 40 --steps 50 --no-cpu-baseline --no-fast-math-line > gpurun_out/r04j_e40.log 2>&1
grep '^{' gpurun_out/r04j_e40.log | tail -1 > gpurun_out/r04_j_epoch40_bench.json
python - <<PY
import json
for f in ['r04_j_cfg1_bench','r04_j_cfg4_bench','r04_j_cfg5_share_bench','r04_j_att_k5_bench','r04_j_epoch40_bench']:
    try:
        d=json.load(open('gpurun_out/%s.json'%f))
        print(f, round(d['value'],1), round(d['ms_per_step'],3), 'exact', d.get('exact_f32') and round(d['exact_f32']['value'],1), d['roofline']['kernel'], d['roofline']['bound'], round(d['roofline']['frac'],3), d['roofline']['traffic_source'], d['roofline_step'] and d['roofline_step'].get('hbm_frac'))
    except Exception as e:
        print(f, 'FAILED', e)
PY
