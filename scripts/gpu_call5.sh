#!/bin/bash
# round 4, call 5: forward recurrences on the fp16 pipe — parity tests, A/B against the exact kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "recurrences_on_the_fp16 or lstm or gru or rnn or (f16x3 and (full3d or segment3d)) or trajectory or pack_plan or cfg2_shape" > gpurun_out/r04e_tests.log 2>&1
tail -15 gpurun_out/r04e_tests.log
for V in ; do
  GPE_RNN_F32=$V timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fast-math-line > gpurun_out/r04e_rnn_$V.log 2>&1
  grep '^{' gpurun_out/r04e_rnn_$V.log | tail -1 > gpurun_out/r04e_rnn_$V.json
done
python - <<PY
import json
for V in '01':
    try:
        d=json.load(open('gpurun_out/r04e_rnn_%s.json'%V))
        k=d['kernel_ms_per_step']
        print('GPE_RNN_F32=%s'%V, round(d['value'],1), round(d['ms_per_step'],3), 'rnn fwd %.3f bwd %.3f pack %.3f' % (k['gpe_rnn_seq_fwd']['ms_per_step'], k['gpe_rnn_seq_bwd']['ms_per_step'], k['gpe_pack_multi']['ms_per_step']))
    except Exception as e:
        print(V, 'FAILED', e)
PY
