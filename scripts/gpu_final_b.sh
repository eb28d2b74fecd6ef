#!/bin/bash
# final-binary profiles: the driver's command under rocprofv3 (+ PMC passes); the texture-addresser / L1 passes of the `random` stand-in
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 700 scripts/prof_r05.sh bench > $O/prof_bench.log 2>&1; echo "prof bench rc=$?"
C5_KINDS=random C5_TA_ONLY=1 timeout 500 scripts/prof_r05.sh c5 > $O/prof_c5ta.log 2>&1; echo "prof c5 rc=$?"
ls $O/summary | head -40
