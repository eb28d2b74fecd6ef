#!/bin/bash
# what a round-end validation on the GPU box runs: the GPU suite, the driver bench line, the 2-ranks-on-one-GPU line (gpurun -- bash scripts/gpu_validate.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_final.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" $O/pytest_gpu_final.log | tail -3
( time timeout 600 python bench.py > $O/bench_final.json 2> $O/bench_final.err ) 2>&1 | grep real; echo "bench rc=$?"
tail -2 $O/bench_final.err
MIK_FORCE_DEVICE=0 MIK_NATIVE_TRANSPORTS=mailbox timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 50 --warmup 5 --cpu-iters 5 > $O/bench_2ranks_1gpu_final.json 2> $O/bench_2ranks_1gpu_final.err; echo "bench2 rc=$?"
timeout 200 python scripts/adjoint_solver_bench.py > $O/adjoint_solver_bench.json 2> $O/adjoint_solver_bench.err; echo "adjoint bench rc=$?"; cat $O/adjoint_solver_bench.json
python - <<'PY'
import json
j=json.loads(open('/root/repo/gpurun_out/r05/bench_final.json').read().strip().splitlines()[-1])
print('value',j['value'],'frac',j['roofline']['frac'],'default',j['default_layout_iters_per_sec'])
print('gmres_hbm',{k:(v['us_per_inner_iteration'],v['frac']) for k,v in j['gmres_hbm_bound'].items() if isinstance(v,dict) and 'us_per_inner_iteration' in v})
print('f_solvers',{lay:{k:(round(v['us_per_iteration'],1),round(v['frac'],3)) for k,v in rec.items() if isinstance(v,dict)} for lay,rec in j['f_solvers'].items()})
j2=json.loads(open('/root/repo/gpurun_out/r05/bench_2ranks_1gpu_final.json').read().strip().splitlines()[-1])
print('2ranks',j2['value'],j2['roofline']['kernel'],j2['roofline']['frac'],j2['parity_vs_oracle']['bit_identical'])
PY
