"""Distributed AMG hierarchy (amg.dist_global 2) against the replicated one (1), the per-shard one (0) and one device:
N^3 Poisson over W shards ON ONE GPU (loopback communicator: timings are not multi-GPU timings; counts, levels and
device bytes per shard are)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver, LocalGroup

amg = dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20)
for N, W in ((128, 1), (128, 4), (128, 8), (256, 1), (256, 8)):
    for mode in ((2, 1, 0) if W > 1 else (2,)):
        if N == 256 and mode == 1:
            continue
        group = LocalGroup(W)
        cuts = [round(q * N / W) for q in range(W + 1)]
        out, errors = [None] * W, []
        refreshed = [None] * W

        def run(rank):
            try:
                s = HIPSolver("")
                if W > 1:
                    s.comm_init_local(group, rank)
                s.set_parameters({"HIP": {"precond": "amg", "tolerance": 1e-8, "amg": dict(amg, dist_global=mode)}})
                t = time.perf_counter()
                s.generate_poisson7(N, N, N, cuts[rank], cuts[rank + 1])
                s.synchronize()
                ts = time.perf_counter() - t
                n = s.matrix_shape()[0]
                b, x = s.device_array(n), s.device_array(n)
                s.generate_rhs(42, b)
                s.axpby_device(n, 0.0, b, 0.0, x)
                t = time.perf_counter()
                s.solve_device(b, x)
                tv = time.perf_counter() - t
                i = s.get_info()
                t = time.perf_counter()
                s.generate_poisson7(N, N, N, cuts[rank], cuts[rank + 1])  # same pattern again: the numeric refresh
                s.synchronize()
                tr = time.perf_counter() - t
                refreshed[rank] = (tr, s.get_param("amg.last_setup_reused"))
                out[rank] = (i["num_iterations"], i["true_residual"], [s.amg_level_info(l)[:2] for l in range(i["amg_levels"])],
                             s.get_param("stats.device_bytes") / 2**20, ts, tv, int(s.get_param("amg.distributed_levels")))
            except Exception as e:
                errors.append((rank, repr(e)))

        th = [threading.Thread(target=run, args=(r,)) for r in range(W)]
        [t.start() for t in th]
        [t.join() for t in th]
        if errors:
            print(N, W, mode, "ERROR", errors[:1], flush=True)
            continue
        o = out[0]
        print(f"N={N} W={W} dist_global={mode}: its={o[0]} res={o[1]:.2e} distributed_levels={o[6]} levels(rank0)={o[2]} "
              f"device MiB per shard max={max(q[3] for q in out):.0f} setup {max(q[4] for q in out):.2f} s solve {max(q[5] for q in out):.3f} s "
              f"refactorize {max(r[0] for r in refreshed):.2f} s (reused={int(refreshed[0][1])})", flush=True)
