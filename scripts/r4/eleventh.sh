#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
timeout 900 python scripts/r4/tetmesh_rev.py 48 2>&1 | cut -c1-200 | tail -20
