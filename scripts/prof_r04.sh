#!/bin/bash
# Round-4 profiling recipe (run on the GPU box through gpurun):  scripts/prof_r04.sh [what...]   what = bench c5 gmres s27
#   rocprofv3 --kernel-trace --stats           -> gpurun_out/r04/<what>/trace
#   separate --pmc passes (never combined with other trace domains; <= 8 SQ / 4 TCC counters per pass; PMC=0 skips them for `bench`)
# scripts/prof_collect.py then condenses everything into the small CSV / txt / json files that are committed under profiles/.
# bench: the driver's command, so the CSR loop (k_spmv_rowgather, the contract's roofline) and the default loop (k_spmv_sdiab2)
# are both in the trace with the shipped cache hints.  c5: one directory per configs[4] stand-in, so that the traffic of a kernel
# is that of ONE matrix.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WHAT=${@:-bench c5 gmres}
export MIK_BENCH_MIN_SECONDS=0            # profiled runs: one timed region is enough
pmc() {   # pmc <dir> <counters...> -- <command...>
  local d=$1; shift; local C=(); while [ "$1" != "--" ]; do C+=("$1"); shift; done; shift
  rocprofv3 --kernel-trace --pmc "${C[@]}" --output-format csv -d $d -o run -- "$@" > $d.log 2>&1 || echo "pmc pass $d (${C[*]}) failed"
}
for w in $WHAT; do
 case $w in
 bench)
  D=$OUT/bench; mkdir -p $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o run -- python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-gmres --no-config5 > $D/trace.log 2>&1
  grep "^{" $D/trace.log | tail -1 > $D/bench_under_rocprof.json
  B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-gmres --no-config5"
  [ "${PMC:-1}" = "0" ] && continue            # PMC=0: the kernel trace only
  pmc $D/pmc_fetch FETCH_SIZE -- $B
  pmc $D/pmc_write WRITE_SIZE -- $B
  pmc $D/pmc_l2 TCC_HIT_sum TCC_MISS_sum -- $B
  pmc $D/pmc_sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU -- $B
  ;;
 c5)
  for k in ${C5_KINDS:-fe_shell fe_hex banded random}; do
   D=$OUT/c5_$k; mkdir -p $D
   C5="python $R/scripts/config5_bench.py"
   export KINDS=$k CSR=0
   GMRES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o run -- $C5 > $D/trace.log 2>&1
   grep "==\|SpMV\|gmres" $D/trace.log > $D/config5_under_rocprof.txt
   export GMRES=0
   pmc $D/pmc_fetch FETCH_SIZE -- $C5
   pmc $D/pmc_write WRITE_SIZE -- $C5
   pmc $D/pmc_l2 TCC_HIT_sum TCC_MISS_sum -- $C5
   if [ "${C5_SQ:-0}" = "1" ]; then        # where the time of the irregular kernels goes (development)
    pmc $D/pmc_sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU -- $C5
    pmc $D/pmc_issue SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $C5
    pmc $D/pmc_level SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -- $C5
    pmc $D/pmc_ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE -- $C5
    pmc $D/pmc_tcp TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -- $C5
   fi
   unset GMRES KINDS CSR
  done
  ;;
 s27)
  # the 27-point box stencil in the wide slice-constant layout (k_spmv_sdiaw2): where its time goes.  256 x 256 x 64 nodes: the
  # per-row time of the 256^3 fixture (the kernel is not HBM-bound) at a sixth of the generation time.
  D=$OUT/s27; mkdir -p $D
  B="python $R/scripts/box_stencil_bench.py --no-csr --shape 256,256,64 --steps 30"
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o run -- $B > $D/trace.log 2>&1
  grep "^{" $D/trace.log | tail -1 > $D/bench_under_rocprof.json
  pmc $D/pmc_fetch FETCH_SIZE -- $B
  pmc $D/pmc_write WRITE_SIZE -- $B
  pmc $D/pmc_l2 TCC_HIT_sum TCC_MISS_sum -- $B
  pmc $D/pmc_sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VALU -- $B
  pmc $D/pmc_issue SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $B
  pmc $D/pmc_level SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL GRBM_GUI_ACTIVE -- $B
  pmc $D/pmc_ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE -- $B
  pmc $D/pmc_tcp TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -- $B
  ;;
 gmres)
  D=$OUT/gmres; mkdir -p $D
  CPU=0 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o run -- python $R/scripts/gmres_bench.py > $D/trace.log 2>&1
  grep -v "^W2\|^E2\|rocprofv3" $D/trace.log > $D/gmres_c3_under_rocprof.txt
  ;;
 esac
done
python $R/scripts/prof_collect.py $OUT
