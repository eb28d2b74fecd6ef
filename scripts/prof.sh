#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun):  scripts/prof.sh <tag> [bench args...]
#   1) rocprofv3 --kernel-trace --stats           -> gpurun_out/<tag>/trace/
#   2) separate --pmc passes (never combined with other trace domains) -> gpurun_out/<tag>/pmc_*/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-prof}; shift
ARGS=${@:---steps 100 --warmup 5 --no-cpu-baseline}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py $ARGS > $OUT/bench_trace.log 2>&1
tail -1 $OUT/bench_trace.log | cut -c1-300
PMCARGS="--steps 6 --warmup 2 --no-cpu-baseline"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$i -o bench -- python $R/bench.py $PMCARGS > $OUT/pmc_$i.log 2>&1 || echo "pmc pass $i ($C) failed"
done
python $R/scripts/pmc_summary.py $OUT $OUT/traffic.json | tee $OUT/pmc_summary.txt
