#!/bin/bash
# round-4 baseline of the restored tree: full -m gpu suite, the bench line, configs[2] per-level tables, host contract
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -6
timeout 600 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err; tail -c 400 gpurun_out/r04_bench.json
timeout 900 bash scripts/r4/prof_elast.sh
cd $R
timeout 600 python scripts/r4/host_contract.py 2>&1 | cut -c1-1800
