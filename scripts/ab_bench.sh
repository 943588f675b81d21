#!/bin/bash
# whole-step A/B of library builds in ONE session: scripts/ab_bench.sh TAG1 TAG2 ...   (build/ab/lib_TAG.so, twice each, interleaved)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2; do for v in "$@"; do
  GPE_HIP_LIB=$ROOT/build/ab/lib_$v.so python $ROOT/bench.py --math ${ABMATH:-f32} --steps 20 --warmup 5 --no-cpu-baseline --no-fast-math-line 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('$v', round(d['value'],1), round(d['ms_per_step'],3), ' '.join('%s %.3f'%(n[4:],k[n]['ms_per_step']) for n in ['gpe_edge_mlp_fwd','gpe_edge_mlp_bwd','gpe_edge_redgemm','gpe_knn','gpe_rnn_seq_fwd','gpe_rnn_seq_bwd','gpe_redgemm','gpe_linear']))"
done; done
