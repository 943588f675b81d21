#!/bin/bash
# per-call durations of the recurrence entry points under several gpe_debug_set values (GPE_DEBUG_SET): scripts/rnn_ab.sh TAG 'v1 v2 ..'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT
TAG=$1; shift
for V in $1; do
  GPE_DEBUG_SET=$V timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-fast-math-line --call-shapes $OUT/${TAG}_shapes_$V.txt > $OUT/${TAG}_rnn_$V.log 2>&1
  grep '^{' $OUT/${TAG}_rnn_$V.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('dbg=$V', round(d['value'],1), round(d['ms_per_step'],3))"
  grep -E "gpe_rnn_seq|gpe_redgemm |gpe_linear " $OUT/${TAG}_shapes_$V.txt | head -40
done
