#!/bin/bash
# One parameterised GPU-box session script (replaces the per-call scripts of rounds 1-4).  Run through gpurun from the repo root:
#     gpurun --timeout 1500 -- 'bash scripts/gpu_session.sh <step> [args] [-- <step> [args]] ...'
# Steps (each bounded by its own `timeout`; everything lands in gpurun_out/, which gpurun merges back):
#   gate  TAG 'EXPR'              pytest -m gpu -k EXPR (no -x: all failures listed)        -> TAG_gate.log
#   gatelibs TAG 'EXPR' 'libs'    the same subset against several library builds (build/ab/lib_X.so)
#   tests TAG                     the whole -m gpu suite                                   -> TAG_gpu_tests.txt
#   ab    TAG VAR 'v1 v2 ..' [bench args]   whole-step A/B of an environment switch in ONE session, twice each, interleaved;
#                                 prints the step time and the per-launch time of every edge family    -> TAG_ab_VAR.txt
#   collect TAG                   scripts/collect_profiles.sh: kernel trace + both PMC passes + the default bench line
#   sq    TAG                     SQ wave-state counters of one EdgeConv layer at the cfg-2 shape -> TAG_sq_wave_states.md
#   configs TAG                   the other configurations' bench lines (cfg 1, cfg 4, cfg 5 share, att k = 5, epoch 40)
#   batch TAG 'b1 b2 ..'          bench lines at other per-GPU batches                     -> TAG_batch_scaling.md
#   bench TAG [bench args]        one bench line                                            -> TAG_bench.json
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
OUT=$ROOT/gpurun_out

line() {   # one summary line of a bench JSON file: value, ms, per-launch microseconds of the edge families
python - "$1" "$2" <<'PY'
import json, sys
f, tag = sys.argv[1], sys.argv[2]
try:
    d = json.load(open(f))
    pk = d.get('roofline_per_kernel') or {}
    g = lambda n: round(pk[n]['avg_launch_ms'] * 1e3, 1) if n in pk else None
    km = d.get('kernel_ms_per_step') or {}
    h = lambda n: round(km[n]['ms_per_step'], 3) if n in km else None
    print(tag, round(d['value'], 1), round(d['ms_per_step'], 3), 'F2', g('gpe_edge_mlp_fwd:gather'), 'F3', g('gpe_edge_mlp_fwd:dense'),
          'B3', g('gpe_edge_mlp_bwd:inplace'), 'B2', g('gpe_edge_mlp_bwd:gather'), 'RDg', g('gpe_edge_redgemm:gather'),
          'RDd', g('gpe_edge_redgemm:dense'), 'pull', g('gpe_edge_pull_dq'), '| knn', h('gpe_knn'), 'rnn', h('gpe_rnn_seq_fwd'),
          h('gpe_rnn_seq_bwd'), 'redgemm', h('gpe_redgemm'), 'linear', h('gpe_linear'))
except Exception as e:
    print(tag, 'FAILED', e)
PY
}

while [ $# -gt 0 ]; do
  STEP=$1; shift
  case $STEP in
    gate)
      TAG=$1; EXPR=$2; shift 2
      timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "$EXPR" > $OUT/${TAG}_gate.log 2>&1
      tail -3 $OUT/${TAG}_gate.log; grep -E "^(FAILED|ERROR)" $OUT/${TAG}_gate.log | head -20 ;;
    gatelibs)   # gatelibs TAG 'EXPR' 'lib1 lib2 ..': the same pytest subset against several library builds (GPE_HIP_LIB)
      TAG=$1; EXPR=$2; LIBS=$3; shift 3
      for Lb in $LIBS; do
        GPE_HIP_LIB=$Lb timeout 600 python -m pytest tests -m gpu -q --tb=line -p no:cacheprovider -k "$EXPR" > $OUT/${TAG}_gatelib.log 2>&1
        echo "== $Lb: $(tail -1 $OUT/${TAG}_gatelib.log)"; grep -E "^(FAILED|ERROR)" $OUT/${TAG}_gatelib.log | head -8
      done ;;
    tests)
      TAG=$1; shift
      # (-rP: the captured output of passing tests — the ReLU / max-winner decision counts of tests/relu_align.py and the measured
      # gradient errors are printed there; the summary keeps those lines, the full log stays in gpurun_out/)
      timeout 1700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -rPs > $OUT/${TAG}_gpu_tests_full.txt 2>&1
      ( grep -E "ReLU decisions|binding output bar|the 1e-4 bar alone|un-aligned|self-difference|worst parameter-gradient|full size:|edgeconv .* err|^SKIPPED|^FAILED|^ERROR| passed| failed" $OUT/${TAG}_gpu_tests_full.txt | sort -u | head -900 ) > $OUT/${TAG}_gpu_tests.txt
      tail -3 $OUT/${TAG}_gpu_tests_full.txt; grep -E "^(FAILED|ERROR)" $OUT/${TAG}_gpu_tests_full.txt | head -20 ;;
    ab)
      TAG=$1; VAR=$2; VALS=$3; shift 3
      ARGS=""; while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS="$ARGS $1"; shift; done
      : > $OUT/${TAG}_ab_$VAR.txt
      for rep in $(seq ${REPS:-2}); do for V in $VALS; do
        env GPE_DEBUG=1 $VAR=$V timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fast-math-line $ARGS > $OUT/${TAG}_ab.log 2>&1
        grep '^{' $OUT/${TAG}_ab.log | tail -1 > $OUT/${TAG}_ab.json
        if [ ! -s $OUT/${TAG}_ab.json ]; then echo "$VAR=$V FAILED"; tail -5 $OUT/${TAG}_ab.log; fi
        line $OUT/${TAG}_ab.json "$VAR=$V" | tee -a $OUT/${TAG}_ab_$VAR.txt
      done; done ;;
    collect)
      TAG=$1; shift
      bash scripts/collect_profiles.sh $TAG > $OUT/${TAG}_collect.log 2>&1
      tail -4 $OUT/${TAG}_collect.log; line $OUT/${TAG}_bench.json $TAG ;;
    sq)
      TAG=$1; shift
      ( cd /tmp && export TMPDIR=/tmp && rm -rf $OUT/prof_sq &&
        GPE_MATH=f16x3 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
            --kernel-trace --output-format csv -d $OUT/prof_sq -- python $ROOT/scripts/pmc_f2.py > $OUT/${TAG}_sq.log 2>&1
        python $ROOT/profiles/summarize_sq.py $OUT/prof_sq > $OUT/${TAG}_sq_wave_states.md; rm -rf $OUT/prof_sq )
      cat $OUT/${TAG}_sq_wave_states.md ;;
    configs)
      TAG=$1; shift
      run() { N=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" > $OUT/${TAG}_$N.log 2>&1
              grep '^{' $OUT/${TAG}_$N.log | tail -1 > $OUT/${TAG}_${N}_bench.json; line $OUT/${TAG}_${N}_bench.json $N; }
      run cfg1 --points 1024 --batch 8 --k 5 --steps 100 --graph          # host-bound shape: the step replayed from one hipGraph
      run cfg1_eager --points 1024 --batch 8 --k 5 --steps 100
      run cfg1_f32 --points 1024 --batch 8 --k 5 --steps 100 --graph --math f32 --no-fast-math-line
      run cfg4 --model att --points 4096 --k 20 --steps 30
      run cfg5_share --points 8192 --batch 64 --steps 10 --warmup 2
      run att_k5 --model att --points 2000 --batch 30 --k 5 --steps 100
      run epoch40 --epoch 40 --steps 50 --no-fast-math-line ;;
    batch)      # batch TAG 'b1 b2 ..': garments per GPU at N = 2048, k = 16 (the recommendation in scripts/run_scale.sh) -> TAG_batch_scaling.md
      TAG=$1; BS=$2; shift 2
      { echo "| batch per GPU | garments/s | ms/step | ms per garment |"; echo "|---|---|---|---|"; } > $OUT/${TAG}_batch_scaling.md
      for Bv in $BS; do
        timeout 300 python bench.py --batch $Bv --steps 30 --warmup 4 --no-cpu-baseline --no-fast-math-line --no-kernel-timing > $OUT/${TAG}_batch.log 2>&1
        grep '^{' $OUT/${TAG}_batch.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=$Bv
print('| %d | %.1f | %.2f | %.3f |' % (b, d['value'], d['ms_per_step'], d['ms_per_step']/b))" | tee -a $OUT/${TAG}_batch_scaling.md
      done ;;
    bench)
      TAG=$1; shift
      ARGS=""; while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS="$ARGS $1"; shift; done
      timeout 600 python bench.py $ARGS > $OUT/${TAG}_bench.log 2>&1
      grep '^{' $OUT/${TAG}_bench.log | tail -1 > $OUT/${TAG}_bench.json; line $OUT/${TAG}_bench.json $TAG ;;
    --) ;;
    *) echo "unknown step $STEP"; exit 2 ;;
  esac
done
