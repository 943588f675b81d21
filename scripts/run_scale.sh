#!/bin/bash
# The 1 / 2 / 4 / 8-GPU weak-scaling runs of bench.py on ONE node (32 garments per GPU, RCCL gradient all-reduce), exactly as
# the driver launches them.  Not runnable on the 1-GPU gpurun box; on an 8-GPU MI355X node:
#     scripts/run_scale.sh [steps] [warmup]      -> gpurun_out/scale_N{1,2,4,8}.json + a summary table
# Each line carries dist_world (what the process group reported), allreduce_bytes, allreduce_ms_per_step (the exchange on its
# own, all buckets back to back) and exchange.exposed_ms_per_step (what backward did not hide), so the table can be checked
# against RCCL having really connected N ranks.  No such curve has been measured yet.
# BATCH (env, default 32 = BASELINE's per-GPU share) sets the garments per GPU.  A data-parallel job that may choose its global batch
# should give each GPU 128 - 256 garments: measured on one MI355X (profiles/r05_z_batch_scaling.md) 3236 / 3574 / 3770 / 3873
# garments/s per GPU at 32 / 64 / 128 / 256 — the ~2 ms of latency-bound launches per step are constant, the edge kernels linear —
# and the gradient exchange stays 11 MB per step whatever the batch (RCCL ReduceOp.AVG on in-place arena slices).
STEPS=${1:-20}
WARMUP=${2:-5}
BATCH=${BATCH:-32}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for N in 1 2 4 8; do
  if [ "$N" = 1 ]; then
    python bench.py --gpus 1 --batch "$BATCH" --steps "$STEPS" --warmup "$WARMUP" --no-cpu-baseline --no-fast-math-line > "$OUT/scale_N1.log" 2>&1
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29500 + N)) \
        bench.py --gpus "$N" --batch "$BATCH" --steps "$STEPS" --warmup "$WARMUP" > "$OUT/scale_N$N.log" 2>&1
  fi
  grep '^{' "$OUT/scale_N$N.log" | tail -1 > "$OUT/scale_N$N.json"
done
python - "$OUT" <<'PY'
import json, sys
base = None
print('%-3s %-12s %-10s %-10s %-12s %-12s %s' % ('N', 'garments/s', 'ms/step', 'efficiency', 'dist_world', 'allreduce_ms', 'exposed_ms'))
for n in (1, 2, 4, 8):
    try:
        d = json.load(open('%s/scale_N%d.json' % (sys.argv[1], n)))
    except Exception as e:
        print(n, 'no line:', e)
        continue
    base = base or d['value']
    ex = d.get('exchange') or {}
    print('%-3d %-12.1f %-10.2f %-10.3f %-12s %-12s %s' % (n, d['value'], d['ms_per_step'], d['value'] / (n * base),
          d.get('dist_world'), d.get('allreduce_ms_per_step'), ex.get('exposed_ms_per_step')))
PY
