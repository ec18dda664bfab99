#!/bin/bash
# PMC passes over the 256^3 AMG-PCG solve, reported for the level-1 Chebyshev product (spmv_csr_dma<32, SPMV_CHEB>) and,
# for comparison, the level-0 dictionary product: L1 / L2 hit rates, requests per gather instruction, stall reasons.
# (separate --pmc passes with --kernel-trace only, as the guide prescribes)
R=${GRAFT_REPO_ROOT:-.}
cd /tmp && export TMPDIR=/tmp
i=0
while IFS= read -r C; do
  i=$((i+1))
  AMG="${AMGJ:-{\}}" rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/r04_pmc_l1_$i -o b -- python $R/scripts/r4/poisson_prof.py > $R/gpurun_out/r04_pmc_l1_$i.log 2>&1
done <<'SETS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAVES
TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
SETS
cd $R
python3 - <<'PY'
import csv, glob, collections, os, json
out = {}
for d in sorted(glob.glob('gpurun_out/r04_pmc_l1_*')):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.OrderedDict()
        for row in csv.DictReader(open(f)):
            kn = row['Kernel_Name']
            key = None
            if 'spmv_csr_dma<32, (psolve::SpmvMode)4' in kn or 'spmv_csr_dma<32, 4' in kn: key = 'L1_cheb_step'
            elif 'spmv_csr_pat<256, (psolve::SpmvMode)4' in kn or 'spmv_csr_pat<256, 4' in kn: key = 'L0_cheb_step'
            if not key: continue
            if int(row['Grid_Size']) < 256 * 1000: continue   # (the small levels run the same template)
            agg.setdefault((key, row['Counter_Name']), []).append(float(row['Counter_Value']))
        for (k, c), v in agg.items():
            v = sorted(v)[len(v) // 4: len(v) - len(v) // 4] or v   # (drop no-op launches and outliers: middle half)
            out.setdefault(k, {})[c] = sum(v) / len(v)
json.dump(out, open('gpurun_out/r04_pmc_level1.json', 'w'), indent=1)
for k, d in out.items():
    print(k)
    for c, v in d.items(): print(f"   {c:40s} {v:16.1f}")
PY
rm -rf gpurun_out/r04_pmc_l1_*/
