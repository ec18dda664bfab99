#!/bin/bash
# One GPU-box pass: parity tests, smoke, bench, rocprofv3 kernel trace of the same bench command.
set -x
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/smoke.log
python bench.py --steps 3 --warmup 1 2>&1 | tail -3 | tee gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof | head -30
find gpurun_out/prof -name "*kernel_stats*" | head -1 | xargs -I{} head -20 {}
# keep only the small summaries (the raw trace can be large)
find gpurun_out/prof -name "*kernel_trace*" -size +20M -delete
