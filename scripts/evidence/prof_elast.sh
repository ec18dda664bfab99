#!/bin/bash
# configs[2] per-level kernel tables: rocprofv3 kernel trace of three solves + the iteration plan,
# for "amg.block_levels" 1 (round 4) and 0 (round 3's cycle); then the A/B of the two on the same box.
R=${GRAFT_REPO_ROOT:-.}
cd /tmp && export TMPDIR=/tmp
for BL in ${BLS:-1 0}; do
  D=$R/gpurun_out/${RND:-r06}_prof_elast_bl$BL
  rm -rf $D
  BL=$BL PLAN=$R/gpurun_out/${RND:-r06}_elast_plan_bl$BL.json timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o e -- python $R/scripts/evidence/elast_prof.py > $R/gpurun_out/${RND:-r06}_prof_elast_bl$BL.log 2>&1
  tail -2 $R/gpurun_out/${RND:-r06}_prof_elast_bl$BL.log
  T=$(find $D -name "*kernel_trace*" | head -1)
  python $R/scripts/evidence/amg_by_level.py $T --plan $R/gpurun_out/${RND:-r06}_elast_plan_bl$BL.json --out $R/gpurun_out/${RND:-r06}_elast_bl${BL}_by_level.csv --groups $R/gpurun_out/${RND:-r06}_elast_bl${BL}_groups.csv --top 6 > $R/gpurun_out/${RND:-r06}_elast_bl${BL}_by_level.txt 2>&1
  sed -n '8,60p' $R/gpurun_out/${RND:-r06}_elast_bl${BL}_by_level.txt
done
cd $R
[ -z "$BLS" ] && python scripts/evidence/elast_ab.py 2>&1 | cut -c1-260 | tail -5
