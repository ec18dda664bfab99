"""Round 6: the contract kernel's speed against what else the process did before (torch's runtime first, other blocks alive)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mode = sys.argv[1]
if mode in ("torch", "torch_copy"):
    import torch
    torch.zeros(1, device="cuda")
    if mode == "torch_copy":
        a = torch.zeros(1 << 27, dtype=torch.float64, device="cuda"); b_ = torch.empty_like(a)
        for _ in range(10): b_.copy_(a)
        torch.cuda.synchronize(); del a, b_; torch.cuda.empty_cache()
from polysolve_amd import HIPSolver
N = 256
n, nnz = N ** 3, 7 * N ** 3 - 6 * N ** 2
keep = []
if mode == "ballast":   # 3 GB of other blocks allocated first and kept
    s0 = HIPSolver("")
    keep = [s0.device_array(1 << 27) for _ in range(3)]
for rep in range(3):
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(tolerance=1e-8, max_iter=20000, spmv_kernel=1, spmv_value_dict=False, profile_spmv=8)})
    s.generate_poisson7(N)
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    k1 = []
    for r in range(3):
        s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
        s.solve_device(b, x); s.synchronize()
        k1.append(s.info_struct().spmv_ms_avg)
    print(json.dumps({"mode": mode, "rep": rep, "k1_ms": [round(v, 4) for v in k1], "frac": (12 * nnz + 20 * n) / (min(k1) * 1e-3) / 8e12}), flush=True)
    if mode != "keep_handles":
        b.free(); x.free(); del s
    else:
        keep.append((s, b, x))
