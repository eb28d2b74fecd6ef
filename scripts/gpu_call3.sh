#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_dist_gmres.py tests/test_dist.py tests/test_gpu_irregular.py -m gpu -x -q > $O/pytest_dist.log 2>&1; echo "pytest rc=$?"
tail -30 $O/pytest_dist.log
timeout 600 python scripts/gmres_part_bench.py > $O/gmres_part_bench.json 2> $O/gmres_part_bench.err; echo "part bench rc=$?"
tail -3 $O/gmres_part_bench.err
cat $O/gmres_part_bench.json
