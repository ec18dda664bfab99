"""where does a non-temporal matrix stream start to pay?  Jacobi-PCG and AMG-PCG per-iteration time by grid size, spmv_nt 0 / 1
(the vector kernels keep their own rule)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED
out = []
for N in [int(v) for v in os.environ.get("NS", "112,128,144,160,176,192,216").split(",")]:
    for pre in ("jacobi", "amg"):
        row = {}
        for nt in [int(v) for v in os.environ.get("NTS", "0,1,0,1").split(",")]:
            s = HIPSolver("")
            top = {"tolerance": 1e-8, "max_iter": 20000, "spmv_nt": (-1 if nt == 9 else nt), "lab.alternate": 4 if nt == 9 else 0}
            if pre == "amg": top.update(precond="amg", amg=dict(AMG_RECOMMENDED))
            s.set_parameters({"HIP": top})
            s.generate_poisson7(N, N, N); s.synchronize()
            n, nnz, _ = s.matrix_shape()
            b, x = s.device_array(n), s.device_array(n)
            s.generate_rhs(42, b)
            ts = []
            for _ in range(3):
                s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
                t = time.time(); s.solve_device(b, x); s.synchronize(); ts.append(time.time() - t)
            row.setdefault(f"nt{nt}", []).append(round(min(ts) * 1e3, 3))
            its = s.get_info()["num_iterations"]; vec_nt = None
            b.free(); x.free(); del s
        mb = (8 * nnz + 22 * n) / 2**20
        print(N, pre, f"operator {mb:.0f} MiB (dictionary stream), vectors {8*n/2**20:.0f} MiB, its {its}", row, flush=True)
        out.append(dict(N=N, precond=pre, operator_mib=mb, ms=row, iterations=its))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_nt_crossover.json"), "w"), indent=1)
