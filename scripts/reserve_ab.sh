#!/bin/bash
# cost of leaving CUs out of the persistent launches at N = 1 (cfg 2): scripts/reserve_ab.sh TAG -> TAG_reserved_cus.md
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT
{ echo "| reserved CUs | garments/s | ms/step |"; echo "|---|---|---|"; } > $OUT/$1_reserved_cus.md
for rep in 1 2; do for R in 0 8 16 32; do
  timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fast-math-line --no-kernel-timing --reserve-cus $R 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('| $R | %.1f | %.3f |' % (d['value'], d['ms_per_step']))" | tee -a $OUT/$1_reserved_cus.md
done; done
