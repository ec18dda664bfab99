import sys, time, json
sys.path.insert(0, ".")
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED
M = 100
for reorder in (2, 0, 2):
    s = HIPSolver("")
    s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 20000, "precond": "amg", "block_size": 3, "profile_spmv": 4,
                              "reorder": reorder, "amg": dict(AMG_RECOMMENDED)}})
    gen = lambda: s.generate_elasticity_q1_permuted(M, mode=1, seed=7)
    for step in ("warm", "reuse_off", "reuse_on", "refresh"):
        if step == "reuse_off": s.set_parameters({"HIP": {"amg": {"reuse": False}}})
        if step == "reuse_on": s.set_parameters({"HIP": {"amg": {"reuse": True}}})
        s.synchronize(); t = time.perf_counter(); gen(); s.synchronize(); dt = time.perf_counter() - t
        print(reorder, step, f"{dt:.4f}", "reorder.s", f"{s.get_param('reorder.seconds'):.4f}", "ondev", s.get_param('amg.levels_aggregated_on_device'),
              "reused", s.get_param("amg.last_setup_reused"), flush=True)
    del s
