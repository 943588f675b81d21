#!/bin/bash
# PMC passes over the kNN kernel alone (scripts/run_knn.py): where do its wave cycles go?
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  T=$(echo $SET | cut -d' ' -f1)
  rm -rf $OUT/pk
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pk -- python $ROOT/scripts/run_knn.py > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'knn_kernel' in r['Kernel_Name']:
            acc[r['Grid_Size'] + '/' + r.get('LDS_Block_Size', '')][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, 'n=%d' % len(next(iter(d.values()))))
PY
done
rm -rf $OUT/pk
