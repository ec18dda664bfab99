"""27-point constant-coefficient operator (scalar Q1 on a hex grid): row kinds (spmv_csr_kind) against the dictionary kernel"""
import json, sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, ".")
from polysolve_amd import HIPSolver
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
def tri(m):
    return sp.diags([np.ones(m - 1), np.ones(m), np.ones(m - 1)], [-1, 0, 1], format="csr")
Q = sp.kron(sp.kron(tri(N), tri(N)), tri(N), format="csr")
Q.data[:] = -1.0
Q = (Q + sp.diags(np.full(Q.shape[0], 28.0))).tocsr()
Q.sort_indices()
n = Q.shape[0]
b = np.random.default_rng(1).uniform(-1, 1, n)
for vd in (False, True, False, True):
    s = HIPSolver("")
    s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 2000, "spmv_value_dict": vd}})
    s.analyze_pattern(Q, n)
    s.factorize(Q)
    db, dx, dy = s.to_device(b), s.device_array(n), s.device_array(n)
    for _ in range(3): s.spmv_device(db, dy)
    s.synchronize(); t = time.perf_counter()
    for _ in range(50): s.spmv_device(db, dy)
    s.synchronize(); t_spmv = (time.perf_counter() - t) / 50
    best = 1e9
    for _ in range(2):
        s.axpby_device(n, 0.0, db, 0.0, dx); s.synchronize()
        t = time.perf_counter(); s.solve_device(db, dx); best = min(best, time.perf_counter() - t)
    i = s.get_info()
    print(json.dumps({"N": N, "value_dict": vd, "kinds": s.get_param("spmv_row_kinds"), "spmv_us": t_spmv * 1e6, "solve_s": best,
                      "iters": int(i["num_iterations"]), "kernel": s.last_spmv_kernel()}), flush=True)
