#!/bin/bash
# Per-kernel times of the layer-2 search at cfg 2 (threshold scan vs ordered lists) + the probe floors.  Run through gpurun.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ft in ${FTS:-1 0}; do
  rm -rf $OUT/prof_knn
  GPE_DEBUG=1 GPE_KNN_FT=$ft timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_knn -- python $ROOT/scripts/knn_l2_probe.py $ARGS > $OUT/knn_ft_trace.log 2>&1
  echo "== GPE_KNN_FT=$ft"; tail -1 $OUT/knn_ft_trace.log
  f=$(find $OUT/prof_knn -name '*kernel_stats.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if 'knn' in r['Name']:
        print('   %-60s calls %4s avg %8.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
rm -rf $OUT/prof_knn
cd $ROOT
for p in ${PROBES:-1 4 5}; do echo "probe $p: $(GPE_DEBUG=1 GPE_KNN_FT=1 GPE_KNN_PROBE=$p timeout 200 python scripts/knn_l2_probe.py $ARGS 2>&1 | tail -1)"; done
