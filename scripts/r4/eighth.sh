#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
F='==|L0 restrict|L0 prolong|L1 cheb_step|L1 residual|L1 restrict|L2 cheb_step|L2 restrict|per live'
bash scripts/r4/prof_poisson.sh base4 '{}' 2>&1 | grep -E "$F"
HIPJ='{"lab.rb_fill":2304}' bash scripts/r4/prof_poisson.sh f2304 '{}' 2>&1 | grep -E "$F"
HIPJ='{"lab.rb_fill":2304,"lab.dma_tile_max":2560}' bash scripts/r4/prof_poisson.sh f2304t2560 '{}' 2>&1 | grep -E "$F"
HIPJ='{"lab.rb_fill":2304,"lab.dma_tile_max":3072}' bash scripts/r4/prof_poisson.sh f2304t3072 '{}' 2>&1 | grep -E "$F"
HIPJ='{"lab.rb_fill":2304,"lab.dma_tile_max":2560,"lab.tile_headroom_pct":110}' bash scripts/r4/prof_poisson.sh f2304t2560h110 '{}' 2>&1 | grep -E "$F"
HIPJ='{"lab.rb_fill":4608,"lab.dma_tile_max":4096}' bash scripts/r4/prof_poisson.sh f4608t4096 '{}' 2>&1 | grep -E "$F"
HIPJ='{"lab.rb_fill":2304,"lab.dma_tile_max":2048,"lab.tile_headroom_pct":100}' bash scripts/r4/prof_poisson.sh f2304h100 '{}' 2>&1 | grep -E "$F"
python scripts/r4/elast_ab.py 2>&1 | cut -c1-150 | tail -2
