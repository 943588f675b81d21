#!/bin/bash
# round 4, call 24: low plane by v_fma_mixlo/hi_f16, scales folded in the reduce-GEMM producers — parity + bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python scripts/lazy_diag.py 8 2>&1 | grep -E "redgemm|bwd" | head -4
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "lazy_dz3 or redgemm or two_streams or ((f16x3 or mixed or bf16x3) and (cfg2_shape or edgeconv or dense_mlp)) or (f16x3 and (cfg2 or cfg1 or cfg4 or att_k20 or full3d or segment3d or amax))" > gpurun_out/r04q_tests.log 2>&1
tail -4 gpurun_out/r04q_tests.log
for i in 1 2; do
timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fast-math-line > gpurun_out/r04q_bench.log 2>&1
grep '^{' gpurun_out/r04q_bench.log | tail -1 > gpurun_out/r04q_bench.json
python - <<PY
import json
d=json.load(open('gpurun_out/r04q_bench.json'))
pk=d['roofline_per_kernel']
g=lambda n: round(pk[n]['avg_launch_ms']*1e3,1) if n in pk else None
print(round(d['value'],1), round(d['ms_per_step'],3), 'loss', d['config'].get('final_loss'), 'F2', g('gpe_edge_mlp_fwd:gather'), 'F3', g('gpe_edge_mlp_fwd:dense'), 'B3', g('gpe_edge_mlp_bwd:inplace'), 'B2', g('gpe_edge_mlp_bwd:gather'), 'RDd', g('gpe_edge_redgemm:dense'), 'RDg', g('gpe_edge_redgemm:gather'))
PY
done
