#!/bin/bash
python -m pytest tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -25
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "distributed_amg or global_amg_on_shards or sharded" 2>&1 | tail -10
python -m pytest tests/test_gpu_configs.py tests/test_gpu_fem.py -x -q -m gpu 2>&1 | tail -5
