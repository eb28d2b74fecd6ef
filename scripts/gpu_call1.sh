#!/bin/bash
# round 5, GPU call 1: tests, the driver's bench line, the 2-ranks-on-one-GPU bench line, profiles of the new legs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?"
MIK_FORCE_DEVICE=0 MIK_NATIVE_TRANSPORTS=mailbox timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 50 --warmup 5 --cpu-iters 5 > $O/bench_2ranks_1gpu.json 2> $O/bench_2ranks_1gpu.err; echo "bench2 rc=$?"
tail -3 $O/bench_2ranks_1gpu.err
timeout 900 scripts/prof_r05.sh gmres_large spmv_large > $O/prof1.log 2>&1; echo "prof rc=$?"
PMC=0 timeout 400 scripts/prof_r05.sh bench > $O/prof2.log 2>&1; echo "prof2 rc=$?"
ls $O $O/summary 2>/dev/null | head -60
