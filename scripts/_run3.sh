PSOLVE_TIMING=1 SIZES=512 timeout 170 python scripts/gpu_amg_setup_time.py > gpurun_out/amg512.log 2>&1
grep -v "amg host " gpurun_out/amg512.log | tail -40
