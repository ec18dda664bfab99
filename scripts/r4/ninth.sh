#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
F='==|L0 restrict|L0 prolong|L1 cheb_step|L1 residual|L1 restrict|L2 cheb_step|L2 restrict|per live'
bash scripts/r4/prof_poisson.sh new '{}' 2>&1 | grep -E "$F"
N=216 bash scripts/r4/prof_poisson.sh new216 '{}' 2>&1 | grep -E "$F"
N=216 HIPJ='{"lab.rb_fill":1852}' bash scripts/r4/prof_poisson.sh old216 '{}' 2>&1 | grep -E "$F"
for p in "" '--grid 216'; do
timeout 300 python bench.py --precond amg --no-cpu-baseline --no-north-star --no-extra --steps 5 --warmup 1 $p | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('amg', j['config']['workload'][:40], round(j['ms_per_step'],2), 'ms', j['iterations'], 'its', j['box']['probe'])"
done
