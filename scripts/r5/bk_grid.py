"""spmv_bsr3_kind on configs[2]: workgroups per CU"""
import json, sys, time
sys.path.insert(0, ".")
from polysolve_amd import HIPSolver
M = 100
for bpc in (0, 4, 6, 8, 12, 16):
    s = HIPSolver("")
    hip = {"block_size": 3, "tolerance": 1e-8, "max_iter": 50}
    if bpc: hip["spmv_blocks_per_cu"] = bpc
    s.set_parameters({"HIP": hip})
    s.generate_elasticity_q1(M); s.synchronize()
    n = s.matrix_shape()[0]
    b, y = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    for _ in range(5): s.spmv_device(b, y)
    s.synchronize(); t = time.perf_counter()
    for _ in range(100): s.spmv_device(b, y)
    s.synchronize(); t_spmv = (time.perf_counter() - t) / 100
    print(json.dumps({"bpc": bpc, "spmv_grid": s.get_param("spmv_grid"), "spmv_us": t_spmv * 1e6}), flush=True)
