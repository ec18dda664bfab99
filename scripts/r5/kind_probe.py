"""What bounds spmv_csr_kind: 200 back-to-back products at 256^3 with the gathers / the store switched off (lab.kind_probe)."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from polysolve_amd import HIPSolver
N = int(os.environ.get("N", "256"))
s = HIPSolver("")
s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 5}})
s.generate_poisson7(N)
n = s.matrix_shape()[0]
x, y = s.device_array(n), s.device_array(n)
s.generate_rhs(42, x)
for vd, label in ((0, "pat"), (1, "kind"), (2, "slots"), (3, "ring")):
    for probe in ((0,) if vd in (0, 3) else ((0, 3) if vd == 1 else (0, 2))):
        for sched in ((0,) if vd != 1 else (0, 1)):
            for unroll in ((1,) if vd != 1 else (1,)):
                s.set_parameters({"HIP": {"spmv_value_dict": bool(vd), "lab.kind_probe": probe, "lab.kind_sched": sched, "lab.kind_unroll": unroll,
                                          "lab.kind_slots": int(vd >= 2), "lab.kind_ring": 1 if vd == 3 else 0}})
                s.generate_poisson7(N)
                for _ in range(5):
                    s.spmv_device(x, y)
                s.synchronize()
                t = time.perf_counter()
                for _ in range(200):
                    s.spmv_device(x, y)
                s.synchronize()
                dt = (time.perf_counter() - t) / 200
                print(json.dumps({"kernel": label, "probe": probe, "sched": sched, "unroll": unroll, "us": dt * 1e6,
                                  "gbs_18n": 18 * n / dt / 1e9}), flush=True)
s.set_parameters({"HIP": {"lab.kind_probe": 0}})
