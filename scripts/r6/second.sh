#!/bin/bash
# round 6, after vec_blocks_per_cu = 2: the GPU suite, the bench line, the large single-device sizes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06_gpu_suite.txt 2>&1
tail -4 gpurun_out/r06_gpu_suite.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_second.json 2> gpurun_out/r06_bench_second.err
cp bench_detail.json gpurun_out/r06_bench_second_detail.json
tail -c 2500 gpurun_out/r06_bench_second.json
VARIANTS='[{},{"vec_blocks_per_cu":8}]' IT=64 python scripts/r6/large_sweep.py > gpurun_out/r06_large_single.jsonl 2>> gpurun_out/r06_bench_second.err
python scripts/evidence/gpu_large_single.py > gpurun_out/r06_large_single.txt 2>&1
cat gpurun_out/r06_large_single.txt
