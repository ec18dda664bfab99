"""use_graph on / off by size (Jacobi-PCG, default storage and plain CSR): where does the captured loop pay?"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from polysolve_amd import HIPSolver
for N in (16, 32, 48, 64, 96, 128, 160, 200, 256):
    n = N ** 3
    row = {"N": N}
    for storage, tag in (({"spmv_kernel": 1, "spmv_value_dict": False}, "csr"), ({}, "auto")):
        for graph in (1, 0):
            s = HIPSolver("")
            s.set_parameters({"HIP": dict(dict(tolerance=1e-8, max_iter=20000, use_graph=bool(graph)), **storage)})
            s.generate_poisson7(N)
            b, x = s.device_array(n), s.device_array(n)
            s.generate_rhs(42, b)
            best = 1e30
            for r in range(4):
                s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
                t = time.perf_counter(); s.solve_device(b, x); s.synchronize(); best = min(best, time.perf_counter() - t)
            row[f"{tag}_graph{graph}_ms"] = round(best * 1e3, 3)
            row["iterations"] = int(s.info_struct().num_iterations)
            b.free(); x.free(); del s
    print(json.dumps(row), flush=True)
