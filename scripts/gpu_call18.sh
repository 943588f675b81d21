#!/bin/bash
# round 4, call 18: consecutive edge kernels walk their tiles in opposite directions (memory-side cache reuse) — parity + A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "lazy_dz3 or ((f16x3 or bf16x6) and (cfg2_shape or edgeconv or redgemm)) or (f16x3 and cfg2) or two_streams" > gpurun_out/r04l_tests.log 2>&1
tail -4 gpurun_out/r04l_tests.log
for V in 1 0 1 0; do
  GPE_REV=$V timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fast-math-line > gpurun_out/r04l_rev_$V.log 2>&1
  grep '^{' gpurun_out/r04l_rev_$V.log | tail -1 > gpurun_out/r04l_rev_$V.json
  python - <<PY
import json
V='$V'
try:
    d=json.load(open('gpurun_out/r04l_rev_%s.json'%V))
    pk=d['roofline_per_kernel']
    g=lambda n: round(pk[n]['avg_launch_ms']*1e3,1) if n in pk else None
    print('GPE_REV=%s'%V, round(d['value'],1), round(d['ms_per_step'],3), 'F2', g('gpe_edge_mlp_fwd:gather'), 'F3', g('gpe_edge_mlp_fwd:dense'), 'B3', g('gpe_edge_mlp_bwd:inplace'), 'B2', g('gpe_edge_mlp_bwd:gather'), 'RDg', g('gpe_edge_redgemm:gather'), 'RDd', g('gpe_edge_redgemm:dense'), 'pull', g('gpe_edge_pull_dq'))
except Exception as e:
    print(V, 'FAILED', e); print(open('gpurun_out/r04l_rev_%s.log'%V).read()[-1500:])
PY
done
