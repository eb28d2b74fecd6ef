#!/bin/bash
# Steady-state kernel timeline of ONE step of the row-partitioned CG on a single GPU (z-periodic slab: the rank is its own neighbour;
# MIK_DIST_SELF_HALO=1), one file per transport -> gpurun_out/r05/dist_selfhalo_timeline_<transport>.txt
#   TRANSPORTS="rccl mailbox" bash scripts/dist_timeline.sh        (on the GPU box, through gpurun)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for TR in ${TRANSPORTS:-rccl rccl+mailbox mailbox}; do
rm -rf /tmp/dist_tl
MIK_NATIVE_TRANSPORTS=$TR MIK_DIST_SELF_HALO=1 MIK_DIST_NZ=64 rocprofv3 --kernel-trace --output-format csv -d /tmp/dist_tl -o run -- python $R/bench.py --gpus 1 --force-dist --grid 512 --steps 200 --warmup 5 --no-cpu-baseline > /tmp/dist_tl.log 2>&1
T=$(find /tmp/dist_tl -name "*kernel_trace.csv" | head -1)
python - "$T" "$TR" > $OUT/dist_selfhalo_timeline_$TR.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the shortest step (update sweep to update sweep) of the last third of the trace: inside a batch of 25 steps with one host wait
marks = [i for i, r in enumerate(rows) if "OpCgUpdateR" in r["Kernel_Name"]]
cand = [(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]), a) for a, b in zip(marks, marks[1:]) if a > 2 * len(rows) // 3]
mid = min(cand)[1]
t0 = int(rows[mid]["Start_Timestamp"])
print(f"one steady-state step of mik_cgd_iterate_many, one configs[3] slab (512 x 512 x 64 rows) on one MI355X, halo (2 x 512^2 doubles) exchanged with the rank itself; transport {sys.argv[2]}")
print("(rocprofv3 --kernel-trace; start offset and duration in us; queue = HIP stream: the halo transfer runs on the library's side stream)\n")
print(f"{'kernel':58s} {'queue':>5s} {'start':>8s} {'dur':>7s}")
n = 0
for r in rows[mid:]:
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:58]
    print(f"{name:58s} {r.get('Queue_Id', '?'):>5s} {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}")
    n += 1
    if n > 1 and "OpCgUpdateR" in r["Kernel_Name"]:
        break
PY
cat $OUT/dist_selfhalo_timeline_$TR.txt
done
