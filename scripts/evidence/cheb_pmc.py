"""configs[2] hierarchy, then 20 launches of every level-0 cycle operation on the hierarchy's own operators (for rocprofv3 --pmc)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from polysolve_amd import HIPSolver
from bench_legs import AMG_RECOMMENDED
s = HIPSolver("")
s.set_parameters({"HIP": {"tolerance": 1e-8, "precond": "amg", "block_size": 3, "amg": dict(AMG_RECOMMENDED), "bsr3_variant": int(os.environ.get("VARIANT", "-1"))}})
s.generate_elasticity_q1(int(os.environ.get("M", "100"))); s.synchronize()
print(s.amg_time_level_ops(0, 20))
