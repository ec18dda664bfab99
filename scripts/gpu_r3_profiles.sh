#!/bin/bash
# Round-3 rocprofv3 evidence (kernel stats + PMC traffic), for BOTH products of the bench system:
#   pat : the dictionary kernel the bench line times (spmv_csr_pat),   bench.py
#   csr : the plain CSR stream the north_star names (spmv_csr_dma),    bench.py --spmv-kernel 1
# plus kernel stats of configs[2] (elasticity, block-3 AMG-PCG) and of the 256^3 AMG-PCG bench.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
B="--steps 2 --warmup 1 --no-cpu-baseline --no-north-star --no-extra"
cd /tmp && export TMPDIR=/tmp
for tag in pat csr; do
  extra=""; [ $tag = csr ] && extra="--spmv-kernel 1"
  rm -rf $R/gpurun_out/prof3_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3_$tag -o bench -- python $R/bench.py $B $extra > $R/gpurun_out/prof3_${tag}_bench.log 2>&1
  f=$(find $R/gpurun_out/prof3_$tag -name "*kernel_stats*" | head -1)
  cp $f $R/gpurun_out/r03_bench_kernel_stats_$tag.csv
  grep '^{' $R/gpurun_out/prof3_${tag}_bench.log | tail -1 > $R/gpurun_out/r03_bench_under_rocprof_$tag.json
  find $R/gpurun_out/prof3_$tag -name "*kernel_trace*" -size +20M -delete
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    ctag=$(echo $C | tr ' ' '_')
    rm -rf $R/gpurun_out/benchpmc3_${tag}_$ctag
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/benchpmc3_${tag}_$ctag -o b -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-north-star --no-extra $extra > $R/gpurun_out/benchpmc3_${tag}_$ctag.log 2>&1
  done
done
# elasticity + AMG kernel stats
rm -rf $R/gpurun_out/prof3_elast $R/gpurun_out/prof3_amg
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3_elast -o e -- python $R/scripts/gpu_elast_solve_prof.py > $R/gpurun_out/prof3_elast.log 2>&1
cp $(find $R/gpurun_out/prof3_elast -name "*kernel_stats*" | head -1) $R/gpurun_out/r03_elasticity_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3_amg -o a -- python $R/bench.py --precond amg --steps 3 --warmup 1 --no-cpu-baseline --no-north-star --no-extra > $R/gpurun_out/prof3_amg.log 2>&1
cp $(find $R/gpurun_out/prof3_amg -name "*kernel_stats*" | head -1) $R/gpurun_out/r03_amg_kernel_stats.csv
grep '^{' $R/gpurun_out/prof3_amg.log | tail -1 > $R/gpurun_out/r03_bench_amg.json
find $R/gpurun_out/prof3_elast $R/gpurun_out/prof3_amg -name "*kernel_trace*" -size +20M -delete
cd $R
tail -2 gpurun_out/prof3_elast.log
python3 - <<'PY'
import csv, glob, collections, os, json
for tag in ("pat", "csr"):
    out = collections.OrderedDict()
    for d in sorted(glob.glob(f'gpurun_out/benchpmc3_{tag}_*')):
        if not os.path.isdir(d): continue
        for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
            agg = collections.OrderedDict()
            for row in csv.DictReader(open(f)):
                k = (row['Kernel_Name'].split('(')[0][-60:], row['Counter_Name'])
                agg.setdefault(k, []).append(float(row['Counter_Value']))
            for (k, c), v in agg.items():
                big = [x for x in v if x > 0.5 * max(v)] if max(v) > 0 else v   # drop the post-convergence no-op launches
                out.setdefault(k, {})[c] = {"n": len(v), "n_live": len(big), "mean_live": sum(big) / max(len(big), 1)}
    json.dump(out, open(f'gpurun_out/r03_bench_pmc_summary_{tag}.json', 'w'), indent=1)
    for k, cs in out.items():
        if any(c["mean_live"] > 1e5 for c in cs.values()):
            print(tag, k, {c: (v["n_live"], round(v["mean_live"], 1)) for c, v in cs.items()})
PY
find gpurun_out/benchpmc3_* -name "*.csv" -size +5M -delete
for t in pat csr; do python3 scripts/top_kernels.py gpurun_out/r03_bench_kernel_stats_$t.csv 4; done
python3 scripts/top_kernels.py gpurun_out/r03_elasticity_kernel_stats.csv 16
python3 scripts/top_kernels.py gpurun_out/r03_amg_kernel_stats.csv 14
