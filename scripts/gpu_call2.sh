#!/bin/bash
# MGS chain cache-hint A/B at the HBM-bound size (MIK_MGS_HINTS bits: 1 v nt, 2 z nt, 4 w load nt, 8 w store nt)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
for h in 1 3 13 9 5 0 15 11; do
  echo "== MIK_MGS_HINTS=$h" >> $O/mgs_hints.log
  MIK_MGS_HINTS=$h ORTH=mgs REPS=2 timeout 200 python scripts/gmres_large_bench.py 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])['mgs']
print(j['us_per_inner_iteration'], j['frac'], j['final_residual'])" >> $O/mgs_hints.log
done
cat $O/mgs_hints.log
