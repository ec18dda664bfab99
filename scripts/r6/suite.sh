#!/bin/bash
# the GPU suite -> gpurun_out/r06_gpu_suite.txt (tail printed)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q "$@" > gpurun_out/r06_gpu_suite.txt 2>&1
tail -25 gpurun_out/r06_gpu_suite.txt
