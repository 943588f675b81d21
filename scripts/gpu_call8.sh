#!/bin/bash
# round 4, call 8: lazy dz3 (the in-place dz3 pass folded into its two consumers) — parity tests, A/B against the eager pass
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "lazy_dz3 or (f16x3 and (cfg2 or cfg1 or cfg4 or att_k20 or edgeconv or redgemm or amax or full3d)) or two_streams or trajectory" > gpurun_out/r04g_tests.log 2>&1
tail -15 gpurun_out/r04g_tests.log
for V in 1 0; do
  GPE_LAZY_DZ3=$V timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fast-math-line > gpurun_out/r04g_lazy_$V.log 2>&1
  grep '^{' gpurun_out/r04g_lazy_$V.log | tail -1 > gpurun_out/r04g_lazy_$V.json
  python - <<PY
import json
V='$V'
try:
    d=json.load(open('gpurun_out/r04g_lazy_%s.json'%V))
    k=d['kernel_ms_per_step']
    g=lambda n: k.get(n,{}).get('ms_per_step',0.0)
    print('GPE_LAZY_DZ3=%s'%V, round(d['value'],1), round(d['ms_per_step'],3), 'dz3 %.3f bwd %.3f redgemm %.3f bound %.3f' % (g('gpe_edge_dz3'), g('gpe_edge_mlp_bwd'), g('gpe_edge_redgemm'), g('gpe_edge_dz3_bound')))
    for n, r in list(d.get('roofline_per_kernel', {}).items())[:12]:
        print('   ', n, r.get('bound'), r.get('frac'), r.get('avg_launch_ms'))
except Exception as e:
    print(V, 'FAILED', e); print(open('gpurun_out/r04g_lazy_%s.log'%V).read()[-1500:])
PY
done
