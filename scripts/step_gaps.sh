ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_kt
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof_kt -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-fast-math-line > $OUT/${TAG:-r06_d}_kt.log 2>&1
DB=$(find $OUT/prof_kt -name '*.db' | head -1)
python $ROOT/profiles/summarize_rocpd.py $DB 13 gaps > $OUT/${TAG:-r06_d}_gaps.md
python $ROOT/scripts/step_timeline.py $DB 2 > $OUT/${TAG:-r06_d}_step_timeline.md
rm -rf $OUT/prof_kt
grep -A30 "idle time" $OUT/${TAG:-r06_d}_gaps.md | head -40
head -3 $OUT/${TAG:-r06_d}_step_timeline.md
