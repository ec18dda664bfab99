"""configs[2] under a random node numbering: the schedule knobs the block product exposes (interleaved, best of 3)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from polysolve_amd import HIPSolver
from bench_legs import AMG_RECOMMENDED
VARIANTS = json.loads(os.environ.get("VARIANTS", "null")) or [
    {}, {"spmv_blocks_per_cu": 4}, {"spmv_blocks_per_cu": 5}, {"spmv_blocks_per_cu": 8}, {"spmv_chunk_rows": 2048}, {"spmv_chunk_rows": 32768},
    {"spmv_xcd_map": 0}, {"spmv_nt": 0}, {"spmv_nt": 1}, {"amg": {"stream_nt": 0}}, {"blocks_per_cu": 4}]
hs = []
for v in VARIANTS:
    s = HIPSolver("")
    hip = {"tolerance": 1e-8, "max_iter": 2000, "precond": "amg", "block_size": 3, "amg": dict(AMG_RECOMMENDED)}
    for k, val in v.items():
        if k == "amg": hip["amg"].update(val)
        else: hip[k] = val
    s.set_parameters({"HIP": hip})
    s.generate_elasticity_q1_permuted(100, mode=1, seed=7); s.synchronize()
    n = s.matrix_shape()[0]
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    hs.append((v, s, b, x, []))
for r in range(3):
    for v, s, b, x, acc in hs:
        s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
        t = time.perf_counter(); s.solve_device(b, x); s.synchronize(); acc.append(time.perf_counter() - t)
for v, s, b, x, acc in hs:
    print(json.dumps({"variant": v, "solve_ms": round(min(acc) * 1e3, 2), "iterations": int(s.get_info()["num_iterations"])}), flush=True)
