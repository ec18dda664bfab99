#!/bin/bash
# round 3: renumber test, then kernel-trace stats of one 256^3 AMG-PCG bench run (setup + solves)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_amg.py -x -q -m gpu -k "renumbered" 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/profamg
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profamg -o amg -- python $R/bench.py --precond amg --steps 3 --warmup 1 --no-cpu-baseline --no-north-star --no-extra > $R/gpurun_out/profamg_bench.log 2>&1
cd $R
tail -c 400 gpurun_out/profamg_bench.log
f=$(find gpurun_out/profamg -name "*kernel_stats*" | head -1)
cp $f gpurun_out/r03_amg_kernel_stats.csv
python3 scripts/top_kernels.py gpurun_out/r03_amg_kernel_stats.csv 24
find gpurun_out/profamg -name "*kernel_trace*" -size +20M -delete
python bench.py --precond amg --steps 3 --warmup 1 --no-cpu-baseline --no-north-star --no-extra 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('amg bench ms_per_step', j['ms_per_step'], 'its', j['iterations'])"
