#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
N=216 timeout 300 python scripts/r4/vrb_ab.py | tail -3
N=256 timeout 300 python scripts/r4/vrb_ab.py | tail -3
bash scripts/r4/tests.sh
