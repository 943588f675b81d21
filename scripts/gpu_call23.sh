#!/bin/bash
# round 4, call 23: eight producer waves in the fp16 reduce-GEMMs — parity + A/B (GPE_RD_PW=4/8)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python scripts/lazy_diag.py 8 2>&1 | grep -E "redgemm" | head -4
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "lazy_dz3 or redgemm or (f16x3 and (cfg2_shape or edgeconv or cfg1 or full3d))" > gpurun_out/r04p_tests.log 2>&1
tail -4 gpurun_out/r04p_tests.log
for V in 8 4 8 4; do
GPE_RD_PW=$V timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fast-math-line > gpurun_out/r04p_bench.log 2>&1
grep '^{' gpurun_out/r04p_bench.log | tail -1 > gpurun_out/r04p_bench.json
python - <<PY
import json
d=json.load(open('gpurun_out/r04p_bench.json'))
pk=d['roofline_per_kernel']
g=lambda n: round(pk[n]['avg_launch_ms']*1e3,1) if n in pk else None
print('PW=$V', round(d['value'],1), round(d['ms_per_step'],3), 'loss', d['config'].get('final_loss'), 'RDd', g('gpe_edge_redgemm:dense'), 'RDg', g('gpe_edge_redgemm:gather'))
PY
done
