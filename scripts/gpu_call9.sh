#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python scripts/lazy_diag.py 8 2>&1 | tail -14
