ROOT=$GRAFT_REPO_ROOT
for rep in 1 2; do for v in X0 X1 X2; do
  GPE_HIP_LIB=$ROOT/build/ab/lib_$v.so python $ROOT/bench.py --math bf16x6 --steps 20 --warmup 5 --no-cpu-baseline --no-fast-math-line 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('$v', round(d['value'],1), round(d['ms_per_step'],3), ' '.join('%s %.3f'%(n[4:],k[n]['ms_per_step']) for n in ['gpe_edge_mlp_fwd','gpe_edge_mlp_bwd','gpe_edge_redgemm']))"
done; done
