#!/bin/bash
# Round-2 GPU pass: kernel-trace stats of the bench command (no CPU legs), summary CSV kept.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof2
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof2 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-north-star > $R/gpurun_out/prof2_bench.log 2>&1
cd $R
tail -c 1500 gpurun_out/prof2_bench.log
f=$(find gpurun_out/prof2 -name "*kernel_stats*" | head -1)
head -12 $f
cp $f gpurun_out/r02_bench_kernel_stats.csv
find gpurun_out/prof2 -name "*kernel_trace*" -size +20M -delete
