#!/bin/bash
# Round-end measurement batch: bench line, rocprof kernel stats of the bench and of an AMG-PCG run, north-star
# comparison (CPU port pinned to one socket), setup times, elasticity.  Everything under gpurun_out/.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 3 --warmup 1 2> gpurun_out/bench.err | tail -1 > gpurun_out/final_bench.json
timeout 120 python bench.py --precond amg --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final_bench_amg.json
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_final_bench.log 2>&1 )
( cd /tmp && export TMPDIR=/tmp && NS_N=256 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_amg -o amg -- python $R/scripts/gpu_amg_one.py > $R/gpurun_out/prof_amg.log 2>&1 )
find gpurun_out/prof_final gpurun_out/prof_amg -name "*kernel_trace*" -delete
NS_THREADS=64 OMP_PROC_BIND=close OMP_PLACES=cores timeout 300 python scripts/gpu_northstar.py > gpurun_out/final_northstar.log 2>&1
SIZES=128,216,256 timeout 300 python scripts/gpu_amg_setup_time.py > gpurun_out/final_setup.log 2>&1
timeout 300 python scripts/gpu_elasticity.py > gpurun_out/final_elasticity.log 2>&1
tail -c 600 gpurun_out/final_bench.json; echo; tail -4 gpurun_out/final_northstar.log | cut -c1-300; grep -E "device_setup=1|solve" gpurun_out/final_setup.log | head -9; tail -4 gpurun_out/final_elasticity.log | cut -c1-200
