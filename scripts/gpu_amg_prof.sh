#!/bin/bash
# kernel-trace stats of one 256^3 AMG-PCG bench run (setup + 3 solves)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/profamg
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profamg -o amg -- python $R/bench.py --precond amg --steps 3 --warmup 1 --no-cpu-baseline --no-north-star > $R/gpurun_out/profamg_bench.log 2>&1
cd $R
tail -c 600 gpurun_out/profamg_bench.log
f=$(find gpurun_out/profamg -name "*kernel_stats*" | head -1)
cp $f gpurun_out/r02_amg_kernel_stats.csv
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/r02_amg_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {100*float(r['TotalDurationNs'])/tot:5.1f}% calls={r['Calls']:>6} avg={float(r['AverageNs'])/1e3:8.1f} us max={float(r['MaxNs'])/1e3:8.1f}  {r['Name'][:110]}")
PY
find gpurun_out/profamg -name "*kernel_trace*" -size +20M -delete
