#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_amg.py tests/test_gpu_kernels.py -m gpu -x -q -k "fp32 or bsr3 or block" > gpurun_out/r06_fp32_tests.txt 2>&1
tail -5 gpurun_out/r06_fp32_tests.txt
CASES=elast_random,elast VARIANTS='[{},{"matrix_fp32":true},{"matrix_fp32":true,"aggregation":"compact","direct_coarse":true}]' python scripts/r6/agg_eval.py > gpurun_out/r06_fp32_eval.jsonl 2>&1
cat gpurun_out/r06_fp32_eval.jsonl
