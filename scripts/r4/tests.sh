#!/bin/bash
# the whole -m gpu suite, log kept in gpurun_out/r04_gputests.log
R=${GRAFT_REPO_ROOT:-.}
cd $R; mkdir -p gpurun_out
timeout ${T:-1200} python -m pytest tests -q -m gpu -x ${K:+-k "$K"} > gpurun_out/r04_gputests.log 2>&1
tail -5 gpurun_out/r04_gputests.log
