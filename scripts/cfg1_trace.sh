ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_kt
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof_kt -- python $ROOT/bench.py --points 1024 --batch 8 --k 5 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-fast-math-line > $OUT/r06i_kt.log 2>&1
DB=$(find $OUT/prof_kt -name '*.db' | head -1)
python $ROOT/profiles/summarize_rocpd.py $DB 13 > $OUT/r06i_cfg1_kernel_trace.md
python $ROOT/scripts/step_timeline.py $DB 2 > $OUT/r06i_cfg1_step_timeline.md
rm -rf $OUT/prof_kt
head -50 $OUT/r06i_cfg1_kernel_trace.md
