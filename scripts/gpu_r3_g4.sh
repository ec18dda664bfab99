#!/bin/bash
R=$GRAFT_REPO_ROOT
for g in 1 0 2; do
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof3_g4
PSOLVE_GATHER4=$g rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3_g4 -o a -- python $R/bench.py --precond amg --steps 3 --warmup 1 --no-cpu-baseline --no-north-star --no-extra > $R/gpurun_out/prof3_g4.log 2>&1
cd $R
echo "=== PSOLVE_GATHER4=$g"
python3 scripts/top_kernels.py $(find gpurun_out/prof3_g4 -name "*kernel_stats*" | head -1) 10 | grep spmv
find $R/gpurun_out/prof3_g4 -name "*kernel_trace*" -size +20M -delete
done
