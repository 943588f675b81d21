#!/bin/bash
# HBM fetch of the edge kernels for library builds, one EdgeConv layer at the cfg-2 shape: scripts/ab_fetch.sh TAG1 TAG2 ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/pf_$v
  GPE_HIP_LIB=$ROOT/build/ab/lib_$v.so rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_$v -- python $ROOT/scripts/pmc_f2.py > /dev/null 2>&1
  GPE_HIP_LIB=$ROOT/build/ab/lib_$v.so rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw_$v -- python $ROOT/scripts/pmc_f2.py > /dev/null 2>&1
  echo "== $v"; python $ROOT/profiles/summarize_pmc.py /tmp/pf_$v /tmp/pw_$v | python -c "
import json,sys
d=json.load(sys.stdin)['kernels']
for k,v in d.items():
    if 'edgegemm_sr' in k or 'redgemm_pc' in k: print('  %-58s fetch %7.1f MB  write %7.1f MB' % (k[:58], v['fetch_bytes']/1e6, v['write_bytes']/1e6))"
done
