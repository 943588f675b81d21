#!/bin/bash
# round 4, final evidence set r04_m (lazy dz3 + fp16 a3 + fma epilogues + opposite walk directions): full -m gpu suite, kernel trace + PMC traffic + default bench
# line (scripts/collect_profiles.sh), SQ wave-state counters of one EdgeConv layer, the other configurations
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "lazy_dz3 or clamped or (f16x3 and (cfg2_shape or edgeconv)) or (bf16x6 and edgeconv)" > gpurun_out/r04m_gate.log 2>&1 || { tail -20 gpurun_out/r04m_gate.log; echo GATE FAILED; exit 1; }
tail -2 gpurun_out/r04m_gate.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r04m_tests.log 2>&1
tail -3 gpurun_out/r04m_tests.log; grep -E "^(FAILED|ERROR)" gpurun_out/r04m_tests.log | head
bash scripts/collect_profiles.sh r04_m > gpurun_out/r04m_collect.log 2>&1
tail -4 gpurun_out/r04m_collect.log
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_sq
GPE_MATH=f16x3 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $R/gpurun_out/prof_sq -- python $R/scripts/pmc_f2.py > $R/gpurun_out/r04m_sq.log 2>&1
python $R/profiles/summarize_sq.py $R/gpurun_out/prof_sq > $R/gpurun_out/r04_m_sq_wave_states_table.md
rm -rf $R/gpurun_out/prof_sq
cat $R/gpurun_out/r04_m_sq_wave_states_table.md
cd $R
timeout 300 python bench.py --points 1024 --batch 8 --k 5 --steps 100 --no-cpu-baseline > gpurun_out/r04m_cfg1.log 2>&1
grep '^{' gpurun_out/r04m_cfg1.log | tail -1 > gpurun_out/r04_m_cfg1_bench.json
timeout 300 python bench.py --model att --points 4096 --k 20 --steps 30 --no-cpu-baseline > gpurun_out/r04m_cfg4.log 2>&1
grep '^{' gpurun_out/r04m_cfg4.log | tail -1 > gpurun_out/r04_m_cfg4_bench.json
timeout 400 python bench.py --points 8192 --batch 64 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r04m_cfg5.log 2>&1
grep '^{' gpurun_out/r04m_cfg5.log | tail -1 > gpurun_out/r04_m_cfg5_share_bench.json
timeout 300 python bench.py --model att --points 2000 --batch 30 --k 5 --steps 100 --no-cpu-baseline > gpurun_out/r04m_attk5.log 2>&1
grep '^{' gpurun_out/r04m_attk5.log | tail -1 > gpurun_out/r04_m_att_k5_bench.json
timeout 300 python bench.py --epoch 40 --steps 50 --no-cpu-baseline --no-fast-math-line > gpurun_out/r04m_e40.log 2>&1
grep '^{' gpurun_out/r04m_e40.log | tail -1 > gpurun_out/r04_m_epoch40_bench.json
python - <<PY
import json
for f in ['r04_m_bench','r04_m_cfg1_bench','r04_m_cfg4_bench','r04_m_cfg5_share_bench','r04_m_att_k5_bench','r04_m_epoch40_bench']:
    try:
        d=json.load(open('gpurun_out/%s.json'%f))
        print(f, round(d['value'],1), round(d['ms_per_step'],3), 'exact', d.get('exact_f32') and round(d['exact_f32']['value'],1), d['roofline']['kernel'], d['roofline']['bound'], round(d['roofline']['frac'],3), d['roofline_step'] and d['roofline_step'].get('hbm_frac'))
    except Exception as e:
        print(f, 'FAILED', e)
PY
