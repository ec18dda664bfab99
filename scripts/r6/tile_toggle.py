"""Round 6: the LDS tile of the contract kernel, 2 048 against 1 792 entries, toggled on ONE handle (same memory, same everything else)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from polysolve_amd import HIPSolver
N = int(os.environ.get("N", "256"))
n, nnz = N ** 3, 7 * N ** 3 - 6 * N ** 2
for h in range(int(os.environ.get("HANDLES", "3"))):
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(tolerance=1e-8, max_iter=20000, spmv_kernel=1, spmv_value_dict=False, profile_spmv=8)})
    s.generate_poisson7(N)
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    out = []
    for r in range(6):
        tile = 2048 if r % 2 == 0 else 1792
        s.set_parameters({"HIP": {"lab.dma_tile_max": tile}})
        s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
        s.solve_device(b, x); s.synchronize()
        out.append((tile, round(s.info_struct().spmv_ms_avg, 4)))
    print(json.dumps({"handle": h, "k1_ms_by_tile": out}), flush=True)
    b.free(); x.free(); del s
