#!/bin/bash
# rocprofv3 evidence of the bench line's own command (round 6; replaces the per-round evidence.sh / evidence_kinds.sh):
#   kernel stats (--kernel-trace --stats) and, in SEPARATE passes as MI355X_MICROARCH.md prescribes, the PMC counters
#   FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ | TCC_HIT/MISS over one solve, for the storage given:
#     bash scripts/evidence/pmc_bench.sh csr|auto [tag]      -> gpurun_out/${RND}_bench_kernel_stats_<tag>.csv,
#                                                               gpurun_out/${RND}_bench_pmc_summary_<tag>.json
#   then: python scripts/evidence/make_pmc_traffic.py $RND <tag>  -> profiles/${RND}_pmc_traffic_<tag>.json
STORAGE=${1:-csr}; TAG=${2:-$STORAGE}; RND=${RND:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-detail --storage $STORAGE"
D=$R/gpurun_out/${RND}_prof_$TAG; rm -rf $D
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o bench -- python $R/bench.py --steps 2 --warmup 1 $B > $R/gpurun_out/${RND}_prof_${TAG}_bench.log 2>&1
cp "$(find $D -name '*kernel_stats*' | head -1)" $R/gpurun_out/${RND}_bench_kernel_stats_$TAG.csv
grep '^{' $R/gpurun_out/${RND}_prof_${TAG}_bench.log | tail -1 > $R/gpurun_out/${RND}_bench_under_rocprof_$TAG.json
find $D -name "*kernel_trace*" -size +20M -delete
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  ctag=$(echo $C | tr ' ' '_'); P=$R/gpurun_out/${RND}_benchpmc_${TAG}_$ctag; rm -rf $P
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P -o b -- python $R/bench.py --steps 1 --warmup 0 $B > $P.log 2>&1
done
cd $R
python3 - "$RND" "$TAG" <<'PY'
import csv, glob, collections, os, json, sys
rnd, tag = sys.argv[1], sys.argv[2]
out = collections.OrderedDict()
for d in sorted(glob.glob(f'gpurun_out/{rnd}_benchpmc_{tag}_*')):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.OrderedDict()
        for row in csv.DictReader(open(f)):
            k = (row['Kernel_Name'].split('(')[0][-60:], row['Counter_Name'])
            agg.setdefault(k, []).append(float(row['Counter_Value']))
        for (k, c), v in agg.items():
            big = [x for x in v if x > 0.5 * max(v)] if max(v) > 0 else v   # drop the post-convergence no-op launches
            out.setdefault(k, {})[c] = {"n": len(v), "n_live": len(big), "mean_live": sum(big) / max(len(big), 1)}
json.dump(out, open(f'gpurun_out/{rnd}_bench_pmc_summary_{tag}.json', 'w'), indent=1)
PY
find gpurun_out/${RND}_benchpmc_* -name "*.csv" -size +5M -delete
python3 scripts/evidence/top_kernels.py gpurun_out/${RND}_bench_kernel_stats_$TAG.csv 5
