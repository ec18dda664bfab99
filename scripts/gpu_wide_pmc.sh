#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
while IFS= read -r C; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/wide_$i -o b -- python $R/scripts/gpu_wide_one.py > $R/gpurun_out/wide_$i.log 2>&1
done <<'SETS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAVES
SETS
cd $R
python3 - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/wide_*')):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.OrderedDict()
        for row in csv.DictReader(open(f)):
            kn = row['Kernel_Name']
            if 'spmv_csr' not in kn or ', 0, double' not in kn: continue
            agg.setdefault((kn.split('(')[0][-50:], row['Counter_Name']), []).append(float(row['Counter_Value']))
        for (k, c), v in agg.items():
            v = v[-5:]
            print(k, c, len(v), sum(v) / len(v))
PY
tail -2 gpurun_out/wide_1.log
