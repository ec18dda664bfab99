#!/bin/bash
# round 5, first lease: the suite on today's box, the refresh tables of the round-4 binary (the baseline this round is
# measured against) and PMC passes over the refresh kernels (the 0.07-0.21 had no counters behind it)
R=${GRAFT_REPO_ROOT:-.}
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_first_tests.log 2>&1; tail -3 gpurun_out/r05_first_tests.log
cd /tmp && export TMPDIR=/tmp
for KIND in elast poisson; do
  for K in 1 6; do
    D=$R/gpurun_out/r05_prof_refresh_${KIND}_$K; rm -rf $D
    KIND=$KIND K=$K timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r -- python $R/scripts/r4/refresh_prof.py > $R/gpurun_out/r05_prof_refresh_${KIND}_$K.log 2>&1
    grep -E "^\{" $R/gpurun_out/r05_prof_refresh_${KIND}_$K.log | cut -c1-300
  done
  A=$(find $R/gpurun_out/r05_prof_refresh_${KIND}_1 -name "*kernel_stats*" | head -1); B=$(find $R/gpurun_out/r05_prof_refresh_${KIND}_6 -name "*kernel_stats*" | head -1)
  python $R/scripts/r4/refresh_table.py $A $B 1 6 $R/gpurun_out/r05_base_refresh_${KIND}_by_kernel.csv 24 | tee $R/gpurun_out/r05_base_refresh_${KIND}_by_kernel.txt
  rm -rf $R/gpurun_out/r05_prof_refresh_${KIND}_1 $R/gpurun_out/r05_prof_refresh_${KIND}_6
done
# PMC passes (separate --pmc runs with --kernel-trace only), K = 2 refreshes
i=0
for KIND in elast poisson; do
  DIRS=""
  while IFS= read -r C; do
    i=$((i+1)); D=$R/gpurun_out/r05_pmc_refresh_$i
    KIND=$KIND K=2 timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o b -- python $R/scripts/r4/refresh_prof.py > $R/gpurun_out/r05_pmc_refresh_$i.log 2>&1
    DIRS="$DIRS $D"
  done <<'SETS'
FETCH_SIZE
WRITE_SIZE
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
SETS
  python $R/scripts/r5/pmc_by_kernel.py $R/gpurun_out/r05_base_pmc_refresh_$KIND.json $DIRS --min-grid 65536 > $R/gpurun_out/r05_base_pmc_refresh_$KIND.txt 2>&1
  rm -rf $DIRS
done
cd $R
timeout 600 python bench.py > gpurun_out/r05_base_bench.json 2> gpurun_out/r05_base_bench.err; tail -c 600 gpurun_out/r05_base_bench.json
