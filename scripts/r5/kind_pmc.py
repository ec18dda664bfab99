"""60 Jacobi-PCG iterations at 256^3 on the row-kind kernel (for rocprofv3 --pmc / --kernel-trace passes)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from polysolve_amd import HIPSolver
s = HIPSolver("")
hip = {"tolerance": 1e-30, "max_iter": 60}
u = os.environ.get("UNROLL")
if u:
    hip["lab.kind_unroll"] = int(u)
if os.environ.get("VD") == "0":
    hip["spmv_value_dict"] = False
s.set_parameters({"HIP": hip})
N = int(os.environ.get("N", "256"))
s.generate_poisson7(N)
n = s.matrix_shape()[0]
b, x = s.device_array(n), s.device_array(n)
s.generate_rhs(42, b)
for _ in range(2):
    s.axpby_device(n, 0.0, b, 0.0, x)
    try:
        s.solve_device(b, x)
    except Exception:
        pass
s.synchronize()
print(s.last_spmv_kernel(), s.get_info()["num_iterations"])
