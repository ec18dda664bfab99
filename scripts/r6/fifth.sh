#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06_gpu_suite.txt 2>&1
tail -4 gpurun_out/r06_gpu_suite.txt
KINDS=elast MODE=1 SUFFIX=_random_map bash scripts/evidence/prof_refresh.sh 2>&1 | tail -14
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_third.json 2> gpurun_out/r06_bench_third.err
cp bench_detail.json gpurun_out/r06_bench_third_detail.json
tail -c 3600 gpurun_out/r06_bench_third.json
