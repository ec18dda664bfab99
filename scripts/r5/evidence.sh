#!/bin/bash
# Round-5 rocprofv3 evidence with the round's final binary (every profiler run under its own timeout):
#   * kernel stats + PMC traffic of BOTH products of the bench system (pat: the dictionary kernel; csr: --spmv-kernel 1)
#   * configs[2] and 256^3 / 216^3 Poisson AMG-PCG per-level tables (scripts/amg_by_level.py)
#   * per-kernel tables of one numeric refresh (configs[2], 256^3 Poisson)
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-.}
B="--steps 2 --warmup 1 --no-cpu-baseline --no-north-star --no-extra --no-live-traffic"
cd /tmp && export TMPDIR=/tmp
for tag in pat csr; do
  extra=""; [ $tag = csr ] && extra="--spmv-kernel 1"
  rm -rf $R/gpurun_out/prof5_$tag
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof5_$tag -o bench -- python $R/bench.py $B $extra > $R/gpurun_out/prof5_${tag}_bench.log 2>&1
  f=$(find $R/gpurun_out/prof5_$tag -name "*kernel_stats*" | head -1)
  cp $f $R/gpurun_out/r05_bench_kernel_stats_$tag.csv
  grep '^{' $R/gpurun_out/prof5_${tag}_bench.log | tail -1 > $R/gpurun_out/r05_bench_under_rocprof_$tag.json
  T=$(find $R/gpurun_out/prof5_$tag -name "*kernel_trace*" | head -1)
  python $R/scripts/amg_by_level.py $T --groups $R/gpurun_out/r05_bench_groups_$tag.csv --top 5 > $R/gpurun_out/r05_bench_groups_$tag.txt 2>&1
  find $R/gpurun_out/prof5_$tag -name "*kernel_trace*" -size +20M -delete
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    ctag=$(echo $C | tr ' ' '_')
    rm -rf $R/gpurun_out/benchpmc5_${tag}_$ctag
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/benchpmc5_${tag}_$ctag -o b -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-north-star --no-extra --no-live-traffic $extra > $R/gpurun_out/benchpmc5_${tag}_$ctag.log 2>&1
  done
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os, json
for tag in ("pat", "csr"):
    out = collections.OrderedDict()
    for d in sorted(glob.glob(f'gpurun_out/benchpmc5_{tag}_*')):
        if not os.path.isdir(d): continue
        for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
            agg = collections.OrderedDict()
            for row in csv.DictReader(open(f)):
                k = (row['Kernel_Name'].split('(')[0][-60:], row['Counter_Name'])
                agg.setdefault(k, []).append(float(row['Counter_Value']))
            for (k, c), v in agg.items():
                big = [x for x in v if x > 0.5 * max(v)] if max(v) > 0 else v   # drop the post-convergence no-op launches
                out.setdefault(k, {})[c] = {"n": len(v), "n_live": len(big), "mean_live": sum(big) / max(len(big), 1)}
    json.dump(out, open(f'gpurun_out/r05_bench_pmc_summary_{tag}.json', 'w'), indent=1)
PY
find gpurun_out/benchpmc5_* -name "*.csv" -size +5M -delete
for t in pat csr; do python3 scripts/top_kernels.py gpurun_out/r05_bench_kernel_stats_$t.csv 4; head -8 gpurun_out/r05_bench_groups_$t.txt | cut -c1-160; done
BLS=1 bash scripts/r5/prof_elast.sh 2>&1 | cut -c1-170 | grep -vE "simple_timer"
bash scripts/r5/prof_poisson.sh final '{}' 2>&1 | cut -c1-170
N=216 bash scripts/r5/prof_poisson.sh final216 '{}' 2>&1 | cut -c1-170
KINDS="elast poisson" bash scripts/r5/prof_refresh.sh 2>&1 | cut -c1-150
