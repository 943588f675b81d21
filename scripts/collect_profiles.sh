#!/bin/bash
# Round-end measurement artefacts (run on the GPU box through gpurun):  scripts/collect_profiles.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the bench workload    -> <tag>_kernel_trace.md
#   2. two PMC passes (FETCH_SIZE, WRITE_SIZE; own runs)         -> <tag>_hbm_traffic.json (records the csrc hash)
#   3. the default bench line (with cpu_baseline), run LAST so that it quotes the traffic file just made (bench.py only
#      accepts a traffic file whose csrc hash equals the tree's)  -> <tag>_bench.json
# Everything lands in gpurun_out/ (merged back); copy the three files into profiles/ and commit them.
TAG=${1:-r01_x}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=10
rm -rf $OUT/prof_kt $OUT/prof_fetch $OUT/prof_write
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt -- python $ROOT/bench.py --steps $STEPS --warmup 3 \
    --no-cpu-baseline --no-kernel-timing --no-fast-math-line > $OUT/${TAG}_kt.log 2>&1
DB=$(find $OUT/prof_kt -name '*.db' | head -1)
python $ROOT/profiles/summarize_rocpd.py $DB $((STEPS + 3)) --json $OUT/${TAG}_kernel_avg.json > $OUT/${TAG}_kernel_trace.md
for C in FETCH_SIZE WRITE_SIZE; do
  D=$OUT/prof_$(echo $C | tr A-Z a-z | sed s/_size//)
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python $ROOT/bench.py --steps 2 --warmup 1 \
      --no-cpu-baseline --no-kernel-timing --no-fast-math-line > $OUT/${TAG}_pmc_$C.log 2>&1
done
python $ROOT/profiles/summarize_pmc.py $OUT/prof_fetch $OUT/prof_write 3 > $OUT/${TAG}_hbm_traffic.json
rm -rf $OUT/prof_kt $OUT/prof_fetch $OUT/prof_write          # keep the merge-back small: raw traces stay on the box
cd $ROOT
if [ -z "$SKIP_BENCH" ]; then
  cp $OUT/${TAG}_hbm_traffic.json $ROOT/profiles/${TAG}_hbm_traffic.json      # on the box: what bench.py will quote
  cp $OUT/${TAG}_kernel_avg.json $ROOT/profiles/${TAG}_kernel_avg.json
  timeout 900 python bench.py --call-shapes $OUT/${TAG}_call_shapes.txt > $OUT/${TAG}_bench.log 2>&1
  grep '^{' $OUT/${TAG}_bench.log | tail -1 > $OUT/${TAG}_bench.json
fi
head -c 600 $OUT/${TAG}_bench.json; echo; head -12 $OUT/${TAG}_kernel_trace.md; head -c 400 $OUT/${TAG}_hbm_traffic.json
