"""First-time AMG setup: device-side construction vs the all-host hierarchy (PSOLVE_TIMING=1 prints the laps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver

for N in [int(v) for v in os.environ.get("SIZES", "128,216,256").split(",")]:
    for dev in (1, 0):
        s = HIPSolver("")
        s.set_parameters({"HIP": {"precond": "amg", "tolerance": 1e-8, "max_iter": 500,
                                  "amg": dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20,
                                              device_setup=dev)}})
        for rep in range(2):
            t = time.time(); s.generate_poisson7(N); tf = time.time() - t
            reused = s.get_param("amg.last_setup_reused")
            print(f"N={N} device_setup={dev} rep={rep}: generate+factorize {tf:.3f} s (reused={reused})", flush=True)
        n, nnz, _ = s.matrix_shape()
        b = s.device_array(n); x = s.to_device(np.zeros(n))
        s.generate_rhs(42, b)
        s.solve_device(b, x)
        x.upload(np.zeros(n))
        t = time.time(); s.solve_device(b, x); ts = time.time() - t
        i = s.get_info()
        lv = [s.amg_level_info(l)[:2] for l in range(i["amg_levels"])]
        print(f"   solve {ts*1e3:.1f} ms iters={i['num_iterations']} true={i['true_residual']:.2e} levels={lv}", flush=True)
        del s
