cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rd
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "redgemm or edgeconv or layer" 2>&1 | tail -6 > gpurun_out/rd/tests.txt
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fast-math-line > gpurun_out/rd/b.json 2> gpurun_out/rd/b.err
