python bench.py --grid 512 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench512.json 2> gpurun_out/bench512.err
tail -c 1500 gpurun_out/bench512.json
SIZES=512 timeout 900 python scripts/gpu_amg_setup_time.py 2>&1 | tail -8 > gpurun_out/amg512.log
cat gpurun_out/amg512.log
timeout 900 python scripts/gpu_northstar.py > gpurun_out/northstar.log 2>&1
tail -15 gpurun_out/northstar.log
