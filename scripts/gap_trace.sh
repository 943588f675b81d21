#!/bin/bash
# kernel trace + idle-gap table of one bench run (GPU box):  scripts/gap_trace.sh [bench args]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_gap
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof_gap -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline \
    --no-kernel-timing --no-fast-math-line "$@" > $OUT/gap_kt.log 2>&1
DB=$(find $OUT/prof_gap -name '*.db' | head -1)
python $ROOT/profiles/summarize_rocpd.py $DB 13 gaps > $OUT/gap_trace.md
rm -rf $OUT/prof_gap
tail -32 $OUT/gap_trace.md
