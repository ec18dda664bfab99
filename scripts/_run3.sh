timeout 300 python -m pytest tests/test_gpu_amg.py -x -q -m gpu 2>&1 | tail -3
PSOLVE_TIMING=1 SIZES=256 timeout 300 python scripts/gpu_amg_setup_time.py 2>&1 | grep -E "aggregation|device_setup=1" | head -8
SIZES=216,512 timeout 300 python scripts/gpu_amg_setup_time.py 2>&1 | grep -E "device_setup=1" | head -8
