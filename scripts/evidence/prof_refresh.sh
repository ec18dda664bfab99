#!/bin/bash
# per-kernel table of one numeric refresh: KIND=elast|poisson
R=${GRAFT_REPO_ROOT:-.}
cd /tmp && export TMPDIR=/tmp
for KIND in ${KINDS:-elast poisson}; do
  for K in 1 6; do
    D=$R/gpurun_out/${RND:-r06}_prof_refresh_${KIND}_$K; rm -rf $D
    KIND=$KIND K=$K timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r -- python $R/scripts/evidence/refresh_prof.py > $R/gpurun_out/${RND:-r06}_prof_refresh_${KIND}_$K.log 2>&1
    grep -E "^\{" $R/gpurun_out/${RND:-r06}_prof_refresh_${KIND}_$K.log | cut -c1-200
  done
  A=$(find $R/gpurun_out/${RND:-r06}_prof_refresh_${KIND}_1 -name "*kernel_stats*" | head -1); B=$(find $R/gpurun_out/${RND:-r06}_prof_refresh_${KIND}_6 -name "*kernel_stats*" | head -1)
  python $R/scripts/evidence/refresh_table.py $A $B 1 6 $R/gpurun_out/${RND:-r06}_refresh_${KIND}${SUFFIX}_by_kernel.csv 24 | tee $R/gpurun_out/${RND:-r06}_refresh_${KIND}${SUFFIX}_by_kernel.txt
done
