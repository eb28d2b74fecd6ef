#!/bin/bash
# Round-6 profiling recipe (run on the GPU box through gpurun):  scripts/prof_r06.sh [what...]   (ROUND=r06: output directory and file prefix)
#   what = bench | gmres_large | spmv_large | c5 | gmres
#   rocprofv3 --kernel-trace --stats           -> gpurun_out/r06/<what>/trace
#   separate --pmc passes (never combined with other trace domains; PMC=0 skips them)
# scripts/prof_collect.py then condenses everything into the small CSV / txt / json files that are committed under profiles/.
#   bench        the driver's command: the CSR contract loop (k_spmv_rowgather = `value` and `roofline`) and the default-layout loop
#   gmres_large  gmres!(restart = 30) on the 256^3 operator: one directory per orth_meth, so that the traffic of a
#                directory divided by its inner iterations is that method's HBM traffic per inner iteration
#   spmv_large   CSR SpMV with x = 134 MB and x = 537 MB
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${ROUND:-r06}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WHAT=${@:-bench gmres_large spmv_large}
export MIK_BENCH_MIN_SECONDS=0            # profiled runs: one timed region is enough
pmc() {   # pmc <dir> <counters...> -- <command...>
  local d=$1; shift; local C=(); while [ "$1" != "--" ]; do C+=("$1"); shift; done; shift
  # (a counter set the hardware cannot collect in one pass makes rocprofv3 abort and then hang: bounded)
  timeout -k 5 ${PMC_TIMEOUT:-300} rocprofv3 --kernel-trace --pmc "${C[@]}" --output-format csv -d $d -o run -- "$@" > $d.log 2>&1 || echo "pmc pass $d (${C[*]}) failed"
}
for w in $WHAT; do
 case $w in
 bench)
  D=$OUT/bench; mkdir -p $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o run -- python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-gmres --no-gmres-large --no-f-solvers --no-config5 > $D/trace.log 2>&1
  grep "^{" $D/trace.log | tail -1 > $D/bench_under_rocprof.json
  B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-gmres --no-gmres-large --no-f-solvers --no-config5"
  [ "${PMC:-1}" = "0" ] && continue            # PMC=0: the kernel trace only
  pmc $D/pmc_fetch FETCH_SIZE -- $B
  pmc $D/pmc_write WRITE_SIZE -- $B
  pmc $D/pmc_l2 TCC_HIT_sum TCC_MISS_sum -- $B
  ;;
 gmres_large)
  for m in ${ORTHS:-mgs cgs}; do
   D=$OUT/gmres_large_$m; mkdir -p $D
   G="python $R/scripts/gmres_large_bench.py"
   export ORTH=$m REPS=2
   rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o run -- $G > $D/trace.log 2>&1
   grep "^{" $D/trace.log | tail -1 > $D/gmres_large_under_rocprof.json
   [ "${PMC:-1}" = "0" ] && continue
   export REPS=1
   pmc $D/pmc_fetch FETCH_SIZE -- $G
   pmc $D/pmc_write WRITE_SIZE -- $G
   pmc $D/pmc_l2 TCC_HIT_sum TCC_MISS_sum -- $G
   unset ORTH REPS
  done
  ;;
 spmv_large)
  for nz in ${NZS_LIST:-64 256}; do
   D=$OUT/spmv_large_$nz; mkdir -p $D
   S="python $R/scripts/spmv_large_bench.py"
   export NZS=$nz REPS=10
   rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o run -- $S > $D/trace.log 2>&1
   grep "^{" $D/trace.log | tail -1 > $D/spmv_large_under_rocprof.json
   [ "${PMC:-1}" = "0" ] && continue
   export REPS=4
   pmc $D/pmc_fetch FETCH_SIZE -- $S
   pmc $D/pmc_write WRITE_SIZE -- $S
   pmc $D/pmc_l2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -- $S
   unset NZS REPS
  done
  ;;
 c5)
  for k in ${C5_KINDS:-random}; do
   D=$OUT/c5_$k; mkdir -p $D
   C5="python $R/scripts/config5_bench.py"
   export KINDS=$k CSR=0 C5_CACHE=/tmp/c5cache
   if [ "${C5_TA_ONLY:-0}" != "1" ]; then
    GMRES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o run -- $C5 > $D/trace.log 2>&1
    grep "==\|SpMV\|gmres" $D/trace.log > $D/config5_under_rocprof.txt
    export GMRES=0
    pmc $D/pmc_fetch FETCH_SIZE -- $C5
    pmc $D/pmc_write WRITE_SIZE -- $C5
    pmc $D/pmc_l2 TCC_HIT_sum TCC_MISS_sum -- $C5
   fi
   export GMRES=0
   if [ "${C5_SQ:-0}" = "1" ] || [ "${C5_TA_ONLY:-0}" = "1" ]; then      # texture addresser / L1: what closes the `random` item
    pmc $D/pmc_ta TA_BUSY_avr GRBM_GUI_ACTIVE -- $C5
    pmc $D/pmc_ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum -- $C5
    pmc $D/pmc_tcp TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum -- $C5
    pmc $D/pmc_tcp2 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -- $C5
   fi
   unset GMRES KINDS CSR
  done
  ;;
 gmres)
  D=$OUT/gmres; mkdir -p $D
  CPU=0 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o run -- python $R/scripts/gmres_bench.py > $D/trace.log 2>&1
  grep -v "^W2\|^E2\|rocprofv3" $D/trace.log > $D/gmres_c3_under_rocprof.txt
  ;;
 esac
done
python $R/scripts/prof_collect.py $OUT
