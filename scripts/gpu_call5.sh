#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu2.log 2>&1; echo "pytest rc=$?"
tail -25 $O/pytest_gpu2.log | cut -c1-300
