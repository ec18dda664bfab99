#!/bin/bash
# Round 3: rocprofv3 evidence for "reorder" on the bench's random-numbering leg (256^3): kernel stats with the search,
# and HBM traffic of the product (FETCH_SIZE / WRITE_SIZE passes) in the caller's numbering and renumbered.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ro in 0 2; do
  rm -rf $R/gpurun_out/prof3_ro$ro
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3_ro$ro -o r -- python $R/scripts/gpu_r3_reorder_one.py $ro > $R/gpurun_out/prof3_ro$ro.log 2>&1
  cp $(find $R/gpurun_out/prof3_ro$ro -name "*kernel_stats*" | head -1) $R/gpurun_out/r03_reorder_kernel_stats_reorder$ro.csv
  find $R/gpurun_out/prof3_ro$ro -name "*kernel_trace*" -size +20M -delete
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc3_ro${ro}_$C
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc3_ro${ro}_$C -o p -- python $R/scripts/gpu_r3_reorder_one.py $ro 48 > $R/gpurun_out/pmc3_ro${ro}_$C.log 2>&1
  done
done
cd $R
for ro in 0 2; do tail -1 gpurun_out/prof3_ro$ro.log; python3 scripts/top_kernels.py gpurun_out/r03_reorder_kernel_stats_reorder$ro.csv 12; done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for ro in (0, 2):
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f'gpurun_out/pmc3_ro{ro}_{C}/**/*counter_collection.csv', recursive=True):
            agg = collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                agg[row['Kernel_Name'].split('(')[0][-70:]].append(float(row['Counter_Value']))
            for k, v in agg.items():
                if 'spmv' not in k: continue
                big = [x for x in v if x > 0.5 * max(v)]
                out.setdefault(f"reorder{ro}", {}).setdefault(k, {})[C] = {"n_live": len(big), "mean_live": sum(big) / len(big)}
json.dump(out, open('gpurun_out/r03_reorder_pmc_summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
find gpurun_out/pmc3_ro* -name "*.csv" -size +5M -delete
