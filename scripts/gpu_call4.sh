#!/bin/bash
# the `random` irregular stand-in -- two A/Bs of the operator-stream cache policy, then the TA / TCP counters
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
for k in "" "0=1" "0=2"; do
  echo "== MIK_KNOBS=$k" >> $O/random_ab.log
  MIK_KNOBS=$k KINDS=random CSR=0 GMRES=0 timeout 300 python scripts/config5_bench.py 2>/dev/null | grep -v "^==" >> $O/random_ab.log
done
cat $O/random_ab.log
C5_KINDS=random C5_SQ=1 timeout 900 scripts/prof_r05.sh c5 > $O/prof_c5.log 2>&1; echo "prof rc=$?"
cat $O/summary/c5_random_pmc_summary.txt | head -60
