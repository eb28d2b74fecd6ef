#!/bin/bash
# Round-2 profiling recipe (run on the GPU box through gpurun):  scripts/prof_r02.sh [what...]   what = bench c5 gmres
#   rocprofv3 --kernel-trace --stats           -> gpurun_out/r02/<what>/trace
#   separate --pmc passes (never combined with other trace domains; <= 8 SQ / 4 TCC counters per pass); PMC_SET=lite skips the
#   texture-addresser / L1 passes
# scripts/prof_collect.py then condenses everything into the small CSV / txt files that are committed under profiles/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r02
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
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o run -- python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline > $D/trace.log 2>&1
  grep "^{" $D/trace.log | tail -1 | cut -c1-400 > $D/bench_under_rocprof.json
  B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity"
  pmc $D/pmc_fetch FETCH_SIZE -- $B
  pmc $D/pmc_write WRITE_SIZE -- $B
  pmc $D/pmc_l2 TCC_HIT_sum TCC_MISS_sum -- $B
  if [ "${PMC_SET:-full}" = full ]; then
  pmc $D/pmc_ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum -- $B
  fi
  pmc $D/pmc_sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -- $B
  pmc $D/pmc_insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM SQ_INSTS_FLAT -- $B
  if [ "${PMC_SET:-full}" = full ]; then
  pmc $D/pmc_ta TA_BUSY_avr TA_TA_BUSY_sum -- $B
  pmc $D/pmc_ta2 TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum -- $B
  pmc $D/pmc_ta3 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum -- $B
  pmc $D/pmc_tcp TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum -- $B
  pmc $D/pmc_tcp2 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -- $B
  pmc $D/pmc_grbm GRBM_GUI_ACTIVE -- $B
  fi
  pmc $D/pmc_issue SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_BUSY_CYCLES -- $B
  pmc $D/pmc_level SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL -- $B
  ;;
 c5)
  D=$OUT/c5; mkdir -p $D
  C5="python $R/scripts/config5_bench.py"
  GMRES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o run -- $C5 > $D/trace.log 2>&1
  cp $D/trace.log $D/config5_under_rocprof.txt
  export GMRES=0
  pmc $D/pmc_fetch FETCH_SIZE -- $C5
  pmc $D/pmc_write WRITE_SIZE -- $C5
  pmc $D/pmc_l2 TCC_HIT_sum TCC_MISS_sum -- $C5
  pmc $D/pmc_tcp2 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -- $C5
  pmc $D/pmc_sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS -- $C5
  unset GMRES
  ;;
 gmres)
  D=$OUT/gmres; mkdir -p $D
  CPU=0 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o run -- python $R/scripts/gmres_bench.py > $D/trace.log 2>&1
  cp $D/trace.log $D/gmres_c3_under_rocprof.txt
  ;;
 esac
done
python $R/scripts/prof_collect.py $OUT
