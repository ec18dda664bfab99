"""Round 3: "reorder" on a multi-device handle (loopback shards on one GPU): 128^3 Poisson under a random numbering
through the host contract -- factorize (hash, search on device 0, packing, upload), halo of the largest shard, solve."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as O
from polysolve_amd import Solver
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4
A0 = O.poisson7(N)
t = time.time(); A = O.permuted(A0, np.random.default_rng(5).permutation(A0.n).astype(np.int32)); print("host permute", time.time() - t, flush=True)
M = A.to_scipy().tocsc()
xs = O.splitmix_vector(A.n, 42); b = O.spmv(A, xs)
out = {}
for precond in ("jacobi", "amg"):
    for reorder in (0, 1):
        s = Solver.create({"solver": "HIP", "HIP": {"devices": [0] * W, "tolerance": 1e-8, "reorder": reorder, "precond": precond,
                                                    "amg": {"cheb_degree": 2, "cheb_lower": 0.1, "cheb_higher": 1.1, "cheb_power_iters": 20, "sa_relax": 1.3}}})
        s.analyze_pattern(M, A.n)
        t = time.time(); s.factorize(M); t1 = time.time() - t
        t = time.time(); s.factorize(M); t2 = time.time() - t
        x = np.zeros(A.n); s.solve(b, x)
        x = np.zeros(A.n); t = time.time(); s.solve(b, x); ts = time.time() - t
        i = s.get_info()
        r = {"factorize_first_s": t1, "factorize_again_s": t2, "reorder_s": s.get_param("reorder.seconds"), "n_halo_max": s.get_param("dist.n_halo"),
             "rows_per_shard": A.n // W, "solve_s": ts, "iterations": int(i["num_iterations"]), "true_residual": i["true_residual"],
             "err": float(np.abs(x - xs).max()), "device_bytes_shard0": s.get_param("stats.device_bytes")}
        out[f"{precond}/reorder{reorder}"] = r
        print(precond, reorder, json.dumps(r), flush=True)
        del s
json.dump(out, open("gpurun_out/r03_reorder_shards.json", "w"), indent=1)
