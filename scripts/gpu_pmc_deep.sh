#!/bin/bash
# Deeper PMC passes on the bench command: where does the SpMV wait?
# SLOW: ~8 min per pass (1600 dispatches are replayed per counter set) -- give gpurun >= 90 min or trim SETS.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
while IFS= read -r C; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/deep_$i -o b -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/deep_$i.log 2>&1
done <<'SETS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_WAVES
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum
TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_BUSY_avr TCC_REQ_sum
GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_TC_BUSY GRBM_EA_BUSY
SETS
cd $R
python3 - <<'PY'
import csv, glob, collections, os
want = ("spmv_csr_pipe<256, 1>", "pcg_update_xp_kernel", "pcg_update_r_kernel")
for d in sorted(glob.glob('gpurun_out/deep_*')):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.OrderedDict()
        for row in csv.DictReader(open(f)):
            kn = row['Kernel_Name']
            k = next((w for w in want if w in kn), None)
            if not k: continue
            agg.setdefault((k, row['Counter_Name']), []).append(float(row['Counter_Value']))
        for (k, c), v in agg.items():
            big = [x for x in v if x > 0.5 * max(v)] if max(v) > 0 else v
            print(f"{k:28s} {c:38s} live={len(big):4d} mean={sum(big)/max(len(big),1):.5g}")
PY
find gpurun_out/deep_* -name "*.csv" -size +2M -delete
