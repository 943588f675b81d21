#!/bin/bash
# round 4, call 21: weight gradients of the decoders on a side stream, under the recurrent chains — parity + A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "fused_adam or arena or rccl or trajectory or pack_plan or lstm or gru or linear or two_streams or cfg1_whole" > gpurun_out/r04n_tests.log 2>&1
tail -5 gpurun_out/r04n_tests.log
for V in 1 0 1 0; do
  GPE_WGRAD_OVERLAP=$V timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fast-math-line > gpurun_out/r04n_ov_$V.log 2>&1
  grep '^{' gpurun_out/r04n_ov_$V.log | tail -1 > gpurun_out/r04n_ov_$V.json
  python - <<PY
import json
V='$V'
try:
    d=json.load(open('gpurun_out/r04n_ov_%s.json'%V))
    k=d['kernel_ms_per_step']
    print('GPE_WGRAD_OVERLAP=%s'%V, round(d['value'],1), round(d['ms_per_step'],3), 'loss', d['config'].get('final_loss'), 'redgemm', round(k['gpe_redgemm']['ms_per_step'],3), 'rnn bwd', round(k['gpe_rnn_seq_bwd']['ms_per_step'],3), 'linear', round(k['gpe_linear']['ms_per_step'],3))
except Exception as e:
    print(V, 'FAILED', e); print(open('gpurun_out/r04n_ov_%s.log'%V).read()[-2000:])
PY
done
for V in 1 0; do
  GPE_WGRAD_OVERLAP=$V timeout 300 python bench.py --points 1024 --batch 8 --k 5 --steps 100 --no-cpu-baseline --no-fast-math-line 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg1 overlap=$V', round(d['value'],1), round(d['ms_per_step'],3))"
done
