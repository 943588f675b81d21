ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
python $ROOT/scripts/pmc_gemm.py 0 t; python $ROOT/scripts/pmc_gemm.py 16384 t
rm -rf $OUT/prof_sq; timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/prof_sq -- python $ROOT/scripts/pmc_gemm.py > $OUT/sq.log 2>&1
python $ROOT/profiles/summarize_sq.py $OUT/prof_sq
rm -rf $OUT/prof_sq; timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/prof_sq -- python $ROOT/scripts/pmc_gemm.py > $OUT/sq2.log 2>&1
python - <<'PY'
import csv,glob,os,re
acc={}
for f in glob.glob(os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'gpurun_out/prof_sq','**','*counter_collection.csv'),recursive=True):
    for row in csv.DictReader(open(f)):
        name=re.sub(r'\(.*','',row['Kernel_Name']).replace('void ','')
        if 'gemm_x6' not in name: continue
        d=acc.setdefault(name,{}); c=d.setdefault(row['Counter_Name'],[0,0.0]); c[0]+=1; c[1]+=float(row['Counter_Value'])
for n,d in acc.items():
    print(n, {k:(v[0], round(v[1]/v[0])) for k,v in d.items()})
PY
tail -3 $OUT/sq2.log; rm -rf $OUT/prof_sq
