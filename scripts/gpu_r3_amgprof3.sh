#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof3_amg
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3_amg -o a -- python $R/bench.py --precond amg --steps 3 --warmup 1 --no-cpu-baseline --no-north-star --no-extra > $R/gpurun_out/prof3_amg.log 2>&1
cp $(find $R/gpurun_out/prof3_amg -name "*kernel_stats*" | head -1) $R/gpurun_out/r03_amg_kernel_stats.csv
grep '^{' $R/gpurun_out/prof3_amg.log | tail -1 > $R/gpurun_out/r03_bench_amg.json
find $R/gpurun_out/prof3_amg -name "*kernel_trace*" -size +20M -delete
cd $R
python3 scripts/top_kernels.py gpurun_out/r03_amg_kernel_stats.csv 12
python -c "import json; j=json.load(open('gpurun_out/r03_bench_amg.json')); print('amg bench under rocprof', j['ms_per_step'], j['iterations'])"
for i in 1 2; do python bench.py --precond amg --steps 5 --warmup 1 --no-cpu-baseline --no-north-star --no-extra 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('amg bench', j['ms_per_step'], j['iterations'], j['value'])"; done
python bench.py --precond amg --grid 216 --steps 5 --warmup 1 --no-cpu-baseline --no-north-star --no-extra 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('amg bench 216', j['ms_per_step'], j['iterations'], j['value'])"
