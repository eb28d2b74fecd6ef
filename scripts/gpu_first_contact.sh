#!/bin/bash
# First-contact hardening of `bench.py --gpus N`, with what ONE GPU allows: every rank on device 0 (MIK_FORCE_DEVICE=0).
#   a  two ranks: transport_selftest (mailbox + landing buffers pass, RCCL "needs distinct devices"), mailbox measured, contract loop, parity
#   b  the same with the mailbox self-test told to fail: no transport between processes is left -> the in-process group leg gives the line
#   c  the ranks cannot be started at all (MIK_SPAWN_FAIL=1): the launcher process measures through the group itself
#   d  3 and 4 ranks (wall-time extrapolation for --gpus 8)
#   e  the gated multi-device test file with its workers as processes on one GPU
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIK_FORCE_DEVICE=0
run() { # name, env..., -- args
  name=$1; shift
  ( time timeout 900 env "$@" ) > $O/$name.json 2> $O/$name.err; echo "$name rc=$? $(grep real $O/$name.err)"
}
run fc_a_2ranks python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 50 --warmup 5
run fc_b_2ranks_mailbox_fails MIK_SELFTEST_FAIL=mailbox python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 2 --steps 50 --warmup 5 --cpu-iters 40
run fc_c_spawn_fails MIK_SPAWN_FAIL=1 python bench.py --gpus 2 --steps 50 --warmup 5 --cpu-iters 40
run fc_d_3ranks python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 3 --steps 50 --warmup 5 --cpu-iters 40
run fc_d_4ranks python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 4 --steps 50 --warmup 5 --cpu-iters 40
MIK_TEST_WORLD=2 timeout 900 python -m pytest tests/test_gpu_multidevice.py -q -m gpu 2>&1 | tail -4 > $O/fc_e_multidevice_world2.log; cat $O/fc_e_multidevice_world2.log
python - <<'PY'
import json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "r06")
for f in sorted(os.listdir(O)):
    if f.startswith("fc_") and f.endswith(".json"):
        lines = [l for l in open(os.path.join(O, f)).read().splitlines() if l.startswith("{")]
        if not lines:
            print(f, "NO LINE"); continue
        j = json.loads(lines[-1])
        st = j.get("transport_selftest") or {}
        print(f, "value", round(j["value"], 1), "transport", j["config"].get("transport_chosen"), "contract", j.get("value_is_contract"),
              "parity", (j.get("parity_vs_oracle") or {}).get("bit_identical"), "roofline", j["roofline"].get("kernel"), round(j["roofline"]["frac"], 3),
              "selftest", {k: (v.get("pass"), v.get("summary") or v.get("failure")) for k, v in st.items() if isinstance(v, dict) and "pass" in v}, "usable", st.get("usable"),
              "wall", round(j["wall"]["seconds_so_far"], 1), "cpu_iters", (j.get("cpu_baseline") or {}).get("sample", "")[:20])
PY
