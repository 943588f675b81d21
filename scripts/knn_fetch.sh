#!/bin/bash
# HBM fetch of the kNN kernel per launch under measurement overrides (run on the GPU box): scripts/knn_fetch.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in ${KNN_FETCH_CFGS:-base GPE_KNN_PIN=0 GPE_KNN_VEC=1}; do
  D=$OUT/knnf_$(echo $cfg | tr -c 'A-Za-z0-9' '_')
  rm -rf $D
  env $( [ "$cfg" = base ] && echo X=1 || echo $cfg ) timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D -- python $ROOT/scripts/knn_probe.py > /dev/null 2>&1
  F=$(find $D -name '*counter_collection.csv' | head -1)
  python - "$F" "$cfg" <<'PY'
import csv, sys, collections
f, cfg = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if 'knn_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
        a = agg[r['Kernel_Name'][:40]]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, v in agg.items():
    print('%-16s %-42s launches %3d  FETCH_SIZE/launch %.0f KiB  (x2 gfx950 correction = %.1f MB)' % (cfg, k, v[0], v[1] / v[0], v[1] / v[0] * 1024 * 2 / 1e6))
PY
  rm -rf $D
done
