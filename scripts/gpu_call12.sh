#!/bin/bash
# round 4, call 12: fp16 storage of the aggregated block's activation (row g) — diag, parity tests, A/B against fp32 storage
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python scripts/lazy_diag.py 8 2>&1 | grep -E "eager|lazy|bound|bad" | head -12
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "lazy_dz3 or (f16x3 and (cfg2 or cfg1 or cfg4 or att_k20 or edgeconv or redgemm or amax or full3d or segment3d)) or two_streams or trajectory" > gpurun_out/r04i_tests.log 2>&1
tail -6 gpurun_out/r04i_tests.log
for V in 1; do
  GPE_LAZY_DZ3=$V timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fast-math-line > gpurun_out/r04i_lazy_$V.log 2>&1
  grep '^{' gpurun_out/r04i_lazy_$V.log | tail -1 > gpurun_out/r04i_lazy_$V.json
  python - <<PY
import json
V='$V'
try:
    d=json.load(open('gpurun_out/r04i_lazy_%s.json'%V))
    k=d['kernel_ms_per_step']
    g=lambda n: k.get(n,{}).get('ms_per_step',0.0)
    print('GPE_LAZY_DZ3=%s'%V, round(d['value'],1), round(d['ms_per_step'],3), 'loss', d['config'].get('final_loss'), 'dz3 %.3f bwd %.3f fwd %.3f redgemm %.3f' % (g('gpe_edge_dz3'), g('gpe_edge_mlp_bwd'), g('gpe_edge_mlp_fwd'), g('gpe_edge_redgemm')))
    for n, r in list(d.get('roofline_per_kernel', {}).items())[:10]:
        print('   ', n, r.get('bound'), round(r.get('frac') or 0,3), round(r.get('avg_launch_ms') or 0,4))
except Exception as e:
    print(V, 'FAILED', e); print(open('gpurun_out/r04i_lazy_%s.log'%V).read()[-1500:])
PY
done
