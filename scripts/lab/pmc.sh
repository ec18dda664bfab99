#!/bin/bash
# PMC passes over the lab binary (separate passes: TCC slots).  Output CSVs -> gpurun_out/pmc_*/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" ; do
  tag=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -o lab -- $R/scripts/lab/spmv_lab 256 2 > $R/gpurun_out/pmc_$tag.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/pmc_*')):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.OrderedDict()
        for row in csv.DictReader(open(f)):
            k = (row['Kernel_Name'][:70], row['Counter_Name'])
            agg.setdefault(k, []).append(float(row['Counter_Value']))
        print('==', f)
        for (k, c), v in agg.items():
            print(f'{k:70s} {c:24s} n={len(v):3d} mean={sum(v)/len(v):.4g}')
PY
