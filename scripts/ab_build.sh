#!/bin/bash
# A/B builds of ONE kernel source with different compile flags, linked against the other (current) objects:
#   scripts/ab_build.sh <source.hip> <tag1> "<flags1>" [<tag2> "<flags2>" ...]   -> build/ab/lib_<tag>.so
# Run them on the GPU box in ONE session (same clocks), e.g.
#   for v in A B; do GPE_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/lib_$v.so python scripts/sr_probe.py; done
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/garment-pattern-estimation_amd/csrc
SRC=$1; shift
mkdir -p $ROOT/build/ab
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
pids=()
while [ $# -ge 2 ]; do
  TAG=$1; FLAGS=$2; shift 2
  ( /opt/rocm/bin/hipcc $BASE $FLAGS -c $CS/$SRC -o $ROOT/build/ab/${SRC%.hip}_$TAG.o > $ROOT/build/ab/$TAG.log 2>&1 && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/ab/lib_$TAG.so \
      $(ls $CS/*.o | grep -v "/${SRC%.hip}.o") $ROOT/build/ab/${SRC%.hip}_$TAG.o >> $ROOT/build/ab/$TAG.log 2>&1 && echo "built $TAG" || echo "FAILED $TAG" ) &
  pids+=($!)
done
wait
