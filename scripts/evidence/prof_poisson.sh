#!/bin/bash
# 256^3 Poisson AMG-PCG per-level kernel tables: rocprofv3 kernel trace of three solves + the iteration
# plan, one pass per variant.  usage: [HIPJ='top-level /HIP json'] prof_poisson.sh name 'AMG-json' [name 'AMG-json' ...]
R=${GRAFT_REPO_ROOT:-.}
cd /tmp && export TMPDIR=/tmp
while [ $# -ge 2 ]; do
  V=$1; AMGJ=$2; shift 2
  D=$R/gpurun_out/${RND:-r06}_prof_poisson_$V
  rm -rf $D
  AMG="$AMGJ" HIP="${HIPJ:-{\}}" N=${N:-256} PLAN=$R/gpurun_out/${RND:-r06}_poisson_plan_$V.json timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python $R/scripts/evidence/poisson_prof.py > $R/gpurun_out/${RND:-r06}_prof_poisson_$V.log 2>&1
  echo "== $V $AMGJ: $(grep -E '^solve' $R/gpurun_out/${RND:-r06}_prof_poisson_$V.log | tail -1)"
  T=$(find $D -name "*kernel_trace*" | head -1)
  python $R/scripts/evidence/amg_by_level.py $T --plan $R/gpurun_out/${RND:-r06}_poisson_plan_$V.json --out $R/gpurun_out/${RND:-r06}_poisson_${V}_by_level.csv --groups $R/gpurun_out/${RND:-r06}_poisson_${V}_groups.csv --top 4 > $R/gpurun_out/${RND:-r06}_poisson_${V}_by_level.txt 2>&1
  grep -E "^ +(L0|L1|pcg) " $R/gpurun_out/${RND:-r06}_poisson_${V}_by_level.txt | cut -c1-150
  grep -E "per live iteration" $R/gpurun_out/${RND:-r06}_poisson_${V}_by_level.txt
  rm -rf $D/*/  # (keep the csv files at the top only)
done
python - <<P
import json,glob
for f in sorted(glob.glob("$R/gpurun_out/${RND:-r06}_poisson_plan_*.json")):
    print(f.split("plan_")[-1], json.load(open(f))["box"].get("probe"))
P
