#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_amg.py -m gpu -x -q -k "parallel_aggregation or direct_coarse or runtime_classes" > gpurun_out/r06_agg_tests.txt 2>&1
tail -6 gpurun_out/r06_agg_tests.txt
python scripts/r6/agg_eval.py > gpurun_out/r06_agg_eval.jsonl 2> gpurun_out/r06_agg_eval.err
tail -3 gpurun_out/r06_agg_eval.err
cat gpurun_out/r06_agg_eval.jsonl
