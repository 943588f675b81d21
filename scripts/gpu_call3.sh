#!/bin/bash
# round 4, call 3: A/B of the left-over schemes of the two-plane edge kernels (GPE_H3_LEFT = F2 F3 B3 B2 digits), parity tests of the
# rotated-slab instances, phase probes of the fp16-pipe kNN filter
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for L in 0000 1111 2222; do
  GPE_H3_LEFT=$L timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-fast-math-line > gpurun_out/r04c_left_$L.log 2>&1
  grep '^{' gpurun_out/r04c_left_$L.log | tail -1 > gpurun_out/r04c_left_$L.json
done
python - <<PY
import json
for L in ['0000','1111','2222']:
    try:
        d=json.load(open('gpurun_out/r04c_left_%s.json'%L))
        pk=d['roofline_per_kernel']
        print(L, round(d['value'],1), round(d['ms_per_step'],3), ' '.join('%s %.1f'%(k.split('_')[-1], pk[k]['avg_launch_ms']*1e3) for k in ['gpe_edge_mlp_fwd:gather','gpe_edge_mlp_fwd:dense','gpe_edge_mlp_bwd:inplace','gpe_edge_mlp_bwd:gather','gpe_edge_redgemm:gather','gpe_edge_redgemm:dense']))
    except Exception as e:
        print(L, 'FAILED', e)
PY
GPE_H3_LEFT=2222 timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "f16x3 or two_streams or edgeconv or dense_mlp" > gpurun_out/r04c_tests_rot.log 2>&1
tail -3 gpurun_out/r04c_tests_rot.log; grep -E "^(FAILED|ERROR)" gpurun_out/r04c_tests_rot.log | head
GPE_H3_LEFT=1111 timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "cfg2_shape_batch4 or cfg2_full" > gpurun_out/r04c_tests_tail.log 2>&1
tail -2 gpurun_out/r04c_tests_tail.log
( for P in 0 1 4 5; do echo "GPE_KNN_PROBE=$P"; GPE_KNN_PROBE=$P timeout 120 python scripts/run_knn.py | grep C=150; done ) > gpurun_out/r04c_knn_probe.txt 2>&1
cat gpurun_out/r04c_knn_probe.txt
