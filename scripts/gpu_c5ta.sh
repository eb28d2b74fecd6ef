#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
PMC_TIMEOUT=150 C5_KINDS=random C5_TA_ONLY=1 timeout 700 scripts/prof_r05.sh c5 > $O/prof_c5ta.log 2>&1; echo "prof c5 rc=$?"
tail -3 $O/prof_c5ta.log
