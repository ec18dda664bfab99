#!/bin/bash
# PMC passes (separate, as the guide prescribes: TCC slots) over the bench command; per-kernel means.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/benchpmc_$tag -o b -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-north-star > $R/gpurun_out/benchpmc_$tag.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os, json
out = collections.OrderedDict()
for d in sorted(glob.glob('gpurun_out/benchpmc_*')):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.OrderedDict()
        for row in csv.DictReader(open(f)):
            k = (row['Kernel_Name'].split('(')[0][-60:], row['Counter_Name'])
            agg.setdefault(k, []).append(float(row['Counter_Value']))
        for (k, c), v in agg.items():
            big = [x for x in v if x > 0.5 * max(v)] if max(v) > 0 else v   # drop the post-convergence no-op launches
            out.setdefault(k, {})[c] = {"n": len(v), "n_live": len(big), "mean_live": sum(big) / max(len(big), 1)}
for k, cs in out.items():
    if any(c["mean_live"] > 1e3 for c in cs.values()):
        print(k, {c: (v["n_live"], round(v["mean_live"], 1)) for c, v in cs.items()})
json.dump(out, open('gpurun_out/bench_pmc_summary.json', 'w'), indent=1)
PY
find gpurun_out/benchpmc_* -name "*.csv" -size +5M -delete
