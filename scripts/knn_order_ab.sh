#!/bin/bash
# whole-step A/B of the layer-2 search's locality order (GPE_KNN_NOORDER=1 ignores it): scripts/knn_order_ab.sh TAG
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
bash scripts/gpu_session.sh gate $1 "knn"
bash scripts/gpu_session.sh ab $1 GPE_KNN_NOORDER "1 0"
for a in "--points 8192 --batch 64 --steps 8 --warmup 2"; do
  for V in 1 0; do
    GPE_DEBUG=1 GPE_KNN_NOORDER=$V timeout 400 python bench.py --no-cpu-baseline --no-fast-math-line $a 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg5 share NOORDER=$V', round(d['value'],1), round(d['ms_per_step'],2), 'knn', round(d['kernel_ms_per_step']['gpe_knn']['ms_per_step'],2))"
  done
done
