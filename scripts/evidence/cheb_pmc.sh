#!/bin/bash
# PMC passes over the level-0 cycle operations of configs[2] (no counters existed for any spmv_bsr3_dma)
R=${GRAFT_REPO_ROOT:-.}
cd /tmp && export TMPDIR=/tmp
i=0; DIRS=""
while IFS= read -r C; do
  i=$((i+1)); D=$R/gpurun_out/${RND:-r06}_pmc_cheb_$i
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o b -- python $R/scripts/evidence/cheb_pmc.py > $R/gpurun_out/${RND:-r06}_pmc_cheb_$i.log 2>&1
  DIRS="$DIRS $D"
done <<'SETS'
FETCH_SIZE
WRITE_SIZE
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum
SETS
python $R/scripts/evidence/pmc_by_kernel.py $R/gpurun_out/${RND:-r06}_pmc_cheb_ops.json $DIRS --min-grid 65536 > $R/gpurun_out/${RND:-r06}_pmc_cheb_ops.txt 2>&1
rm -rf $DIRS
grep -c "" $R/gpurun_out/${RND:-r06}_pmc_cheb_ops.txt
