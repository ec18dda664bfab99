#!/bin/bash
# round 3, first GPU pass: the new tests, the bench line with its extra legs
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench.py -x -q -m gpu -k "permuted or bench" 2>&1 | tail -15
python bench.py > gpurun_out/r03_bench_first.json 2> gpurun_out/r03_bench_first.err
tail -c 600 gpurun_out/r03_bench_first.err
python - <<'PY'
import json
j = json.loads([l for l in open('gpurun_out/r03_bench_first.json') if l.startswith('{')][-1])
print(json.dumps({k: v for k, v in j.items() if k not in ('north_star',)}, indent=1)[:6000])
PY
