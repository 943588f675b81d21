OUT=gpurun_out; mkdir -p $OUT
run() { N=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-fast-math-line "$@" > $OUT/r06g_$N.log 2>&1; grep '^{' $OUT/r06g_$N.log | tail -1 > $OUT/r06g_${N}.json; python -c "
import json; d=json.load(open('$OUT/r06g_$N.json')); print('$N', round(d['value'],1), round(d['ms_per_step'],3), d['config'].get('launch','')[:40], 'kernel sum', round(sum(v['ms_per_step'] for v in d['kernel_ms_per_step'].values()),2))" || tail -5 $OUT/r06g_$N.log; }
run cfg1_eager --points 1024 --batch 8 --k 5 --steps 100
run cfg1_graph --points 1024 --batch 8 --k 5 --steps 100 --graph
run att5_eager --model att --points 2000 --batch 30 --k 5 --steps 60
run att5_graph --model att --points 2000 --batch 30 --k 5 --steps 60 --graph
run cfg2_eager --steps 60
run cfg2_graph --steps 60 --graph
run cfg4_graph --model att --points 4096 --k 20 --steps 30 --graph
