#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
MIK_FORCE_DEVICE=0 MIK_NATIVE_TRANSPORTS=mailbox timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 50 --warmup 5 --cpu-iters 5 > $O/bench_2ranks_1gpu_final.json 2> $O/bench_2ranks_1gpu_final.err; echo "bench2 rc=$?"
tail -3 $O/bench_2ranks_1gpu_final.err | cut -c1-300
python - <<'PY'
import json
j2=json.loads(open('/root/repo/gpurun_out/r05/bench_2ranks_1gpu_final.json').read().strip().splitlines()[-1])
print('2ranks',j2['value'],j2['roofline']['kernel'],j2['roofline']['frac'],j2['parity_vs_oracle']['bit_identical'], j2['contract_csr_loop']['first_residuals_equal_the_default_layout_bit_for_bit'])
PY
MIK_DIST_SELF_HALO=1 MIK_DIST_NZ=64 timeout 300 python bench.py --gpus 1 --force-dist --grid 512 --steps 100 --no-cpu-baseline > $O/bench_selfhalo.json 2> $O/bench_selfhalo.err; echo "selfhalo rc=$?"
python - <<'PY'
import json
j=json.loads(open('/root/repo/gpurun_out/r05/bench_selfhalo.json').read().strip().splitlines()[-1])
print('selfhalo', j['value'], j['ms_per_step'], j['default_layout_ms_per_step'], {k:v.get('ms_per_step') for k,v in j['config']['transports_measured'].items()})
PY
