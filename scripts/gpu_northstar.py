"""north_star comparison: 10 M-DOF 3-D Poisson (N=216), GPU AMG-PCG vs the CPU port of the reference AMGCL
configuration (oracle.cg_amgcl + oracle.AMG with AMGCL.cpp:32-65 defaults, tol 1e-8), same box.
Prints one JSON object.  Test infrastructure (uses the oracle as the CPU baseline, like bench.py)."""
import json, os, sys, time
# the CPU port is pathologically slow when its OpenMP team is spread unpinned over both sockets: pin to one
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")
os.environ.setdefault("NS_THREADS", "64")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as O
from polysolve_amd import HIPSolver

N = int(os.environ.get("NS_N", "216"))
out = {"N": N, "dof": N ** 3}
def gpu(name, params):
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(params, tolerance=1e-8, max_iter=20000)})
    t = time.time(); s.generate_poisson7(N); tf = time.time() - t
    n, nnz, _ = s.matrix_shape()
    b, x = s.device_array(n), s.to_device(np.zeros(n))
    s.generate_rhs(42, b)
    s.solve_device(b, x); x.upload(np.zeros(n))
    t = time.time(); s.solve_device(b, x); ts = time.time() - t
    i = s.get_info()
    out[name] = {"setup_s": tf, "solve_s": ts, "iterations": i["num_iterations"], "true_residual": i["true_residual"], "dof_per_s": n / ts}
    print(name, out[name], flush=True)
gpu("gpu_jacobi_pcg", {})
gpu("gpu_amg_pcg_amgcl_defaults", dict(precond="amg", amg=dict(ncycle=2, cheb_degree=16, cheb_power_iters=100)))
gpu("gpu_amg_pcg_tuned", dict(precond="amg", amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20)))
if os.environ.get("NS_CPU", "1") == "1":
    threads = int(os.environ.get("NS_THREADS", "0"))
    if threads: O.lib().orc_set_num_threads(threads)
    cores = O.lib().orc_num_threads()
    t = time.time(); A = O.poisson7(N); b = O.spmv(A, O.splitmix_vector(A.n, 42)); tg = time.time() - t
    t = time.time(); amg = O.AMG(A); tsu = time.time() - t
    t = time.time(); x, it, err = O.cg_amgcl(A, b, precond=amg, tol=1e-8, max_iter=1000); tso = time.time() - t
    out["cpu_amgcl_port"] = {"cores": cores, "setup_s": tsu, "solve_s": tso, "iterations": it, "final_res_norm": err, "dof_per_s": A.n / tso,
                             "note": "oracle restatement of AMGCL 1.4.3 defaults (W-cycle, Chebyshev-16); setup single-threaded, cycle OpenMP"}
    print("cpu", out["cpu_amgcl_port"], flush=True)
    t = time.time(); x, it, err = O.cg_eigen(A, b, tol=1e-8, max_iter=60); tj = time.time() - t
    out["cpu_jacobi_pcg_port_60its_s"] = tj
    for k in ("gpu_amg_pcg_amgcl_defaults", "gpu_amg_pcg_tuned", "gpu_jacobi_pcg"):
        out[k]["speedup_vs_cpu_amgcl_solve"] = tso / out[k]["solve_s"]
print("NORTHSTAR " + json.dumps(out))
