#!/bin/bash
# Round 6's evidence in one lease (what the files under profiles/r06_* were made with; ~8 GPU-minutes):
#   the GPU suite, the driver-style bench line, rocprofv3 kernel stats + PMC traffic of the bench command, the large
#   single-device sizes, the aggregation / fp32 evaluation, one numeric refresh under a random numbering.
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}" || exit 1
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06_gpu_suite.txt 2>&1; tail -4 gpurun_out/r06_gpu_suite.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
cp bench_detail.json gpurun_out/r06_bench_detail.json; tail -c 1500 gpurun_out/r06_bench.json
RND=r06 bash scripts/evidence/pmc_bench.sh csr 2>&1 | tail -6
VARIANTS='[{},{"vec_blocks_per_cu":8}]' IT=64 python scripts/r6/large_sweep.py > gpurun_out/r06_large_single.jsonl 2>> gpurun_out/r06_bench.err
python scripts/evidence/gpu_large_single.py > gpurun_out/r06_large_single.txt 2>&1; cat gpurun_out/r06_large_single.txt
python scripts/r6/agg_eval.py > gpurun_out/r06_agg_eval.jsonl 2>> gpurun_out/r06_bench.err; tail -3 gpurun_out/r06_agg_eval.jsonl
KINDS=elast MODE=1 SUFFIX=_random bash scripts/evidence/prof_refresh.sh 2>&1 | tail -12
