"""Round 6: the 3x3-block product of configs[2] under a random numbering, knobs toggled on ONE handle (same memory)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from polysolve_amd import HIPSolver
s = HIPSolver("")
s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 2000, "precond": "jacobi", "block_size": 3}})
s.generate_elasticity_q1_permuted(100, mode=1, seed=7); s.synchronize()
n = s.matrix_shape()[0]
x, y = s.device_array(n), s.device_array(n)
s.generate_rhs(7, x)
blocks = 26207180; nb = 1000000
bytes_ = 76 * blocks + 52 * nb
def t(label, **kw):
    if kw: s.set_parameters({"HIP": kw})
    s.time_spmv(x, y, 5)
    ms = min(s.time_spmv(x, y, 30) for _ in range(3))
    print(json.dumps({"setting": label, "ms": round(ms, 4), "frac": round(bytes_ / (ms * 1e-3) / 8e12, 4), "kernel": s.last_spmv_kernel()}), flush=True)
t("default")
for b in (4, 5, 6, 7, 8):
    t(f"spmv_blocks_per_cu {b}", spmv_blocks_per_cu=b)
t("back to 6", spmv_blocks_per_cu=6)
for tile in (1024, 1536, 2048, 3072, 4096):
    t(f"dma_tile_max {tile}", **{"lab.dma_tile_max": tile})
t("tile 2048 again", **{"lab.dma_tile_max": 2048})
for c in (2048, 4096, 16384, 32768):
    t(f"chunk_rows {c}", spmv_chunk_rows=c)
t("chunk 8192", spmv_chunk_rows=8192)
t("xcd_map 0", spmv_xcd_map=0)
t("xcd_map 1", spmv_xcd_map=1)
t("xcd_map 2", spmv_xcd_map=2)
t("nt 0", spmv_nt=0)
t("nt 1", spmv_nt=1)
t("nt auto", spmv_nt=-1)
