#!/bin/bash
# kernel stats of the 256^3 AMG-PCG bench with RENUMBER=0/1 (env PSOLVE_BENCH_RENUMBER read by bench.py --precond amg)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_amg.py -x -q -m gpu -k "renumbered" 2>&1 | tail -5
R=$GRAFT_REPO_ROOT
for rn in 0 1; do
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/profamg$rn
PSOLVE_BENCH_RENUMBER=$rn rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profamg$rn -o amg -- python $R/bench.py --precond amg --steps 3 --warmup 1 --no-cpu-baseline --no-north-star --no-extra > $R/gpurun_out/profamg${rn}_bench.log 2>&1
cd $R
f=$(find gpurun_out/profamg$rn -name "*kernel_stats*" | head -1)
cp $f gpurun_out/r03_amg_kernel_stats_rn$rn.csv
echo "=== renumber=$rn"
python3 scripts/top_kernels.py gpurun_out/r03_amg_kernel_stats_rn$rn.csv 12
find gpurun_out/profamg$rn -name "*kernel_trace*" -size +20M -delete
PSOLVE_BENCH_RENUMBER=$rn python bench.py --precond amg --steps 3 --warmup 1 --no-cpu-baseline --no-north-star --no-extra 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('amg bench ms_per_step', j['ms_per_step'], 'its', j['iterations'])"
done
