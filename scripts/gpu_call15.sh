#!/bin/bash
# round 4, call 15: backward epilogues as two fmas per element on the raw accumulator — parity tests + bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "lazy_dz3 or ((f16x3 or bf16x6) and (cfg2 or cfg1 or cfg4 or att_k20 or edgeconv or redgemm or amax or full3d or segment3d or dense_mlp)) or two_streams or trajectory" > gpurun_out/r04k_tests.log 2>&1
tail -4 gpurun_out/r04k_tests.log
timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fast-math-line > gpurun_out/r04k_bench.log 2>&1
grep '^{' gpurun_out/r04k_bench.log | tail -1 > gpurun_out/r04k_bench.json
python - <<PY
import json
d=json.load(open('gpurun_out/r04k_bench.json'))
print(round(d['value'],1), round(d['ms_per_step'],3), 'loss', d['config'].get('final_loss'), 'exact', d['exact_f32'] and round(d['exact_f32']['value'],1))
for n, r in list(d.get('roofline_per_kernel', {}).items())[:12]:
    print('   ', n, r.get('bound'), round(r.get('frac') or 0,3), round(r.get('avg_launch_ms') or 0,4))
PY
