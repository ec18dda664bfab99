#!/bin/bash
# round 6, first lease: the new bench line (compact, plain-CSR headline) + its contract tests + the adapter with injected defaults
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bench.py tests/test_adapter.py -m gpu -x -q > gpurun_out/r06_first_tests.txt 2>&1
tail -5 gpurun_out/r06_first_tests.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06_bench_first.json 2> gpurun_out/r06_bench_first.err
tail -c 3000 gpurun_out/r06_bench_first.json
tail -30 gpurun_out/r06_bench_first.err
cp bench_detail.json gpurun_out/r06_bench_first_detail.json 2>/dev/null
