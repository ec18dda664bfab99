"""The sharded path at a realistic size on ONE GPU (in-process loopback communicator, one thread per rank): 128^3
Poisson on 4 ranks, both recurrences, Jacobi and (additive Schwarz) AMG -- iteration counts and residuals next to
the single-GPU solve of the same system."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver, LocalGroup

N = int(os.environ.get("LB_N", "128"))
world = int(os.environ.get("LB_WORLD", "4"))
ref = HIPSolver("")
ref.set_parameters({"HIP": {"tolerance": 1e-8}})
ref.generate_poisson7(N)
n = ref.matrix_shape()[0]
b, x = ref.device_array(n), ref.to_device(np.zeros(n))
ref.generate_rhs(42, b)
ref.solve_device(b, x)
print(f"single GPU: {ref.get_info()['num_iterations']} iterations, true residual {ref.get_info()['true_residual']:.3e}", flush=True)
xref = x.download()
for label, params in [("jacobi, single reduction", dict(dist_single_reduction=1)), ("jacobi, two reductions", dict(dist_single_reduction=0)),
                      ("amg (Schwarz)", dict(precond="amg", amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20)))]:
    group = LocalGroup(world)
    out, err = [None] * world, []
    cuts = np.linspace(0, N, world + 1).round().astype(int)
    def run(rank):
        try:
            s = HIPSolver("")
            s.comm_init_local(group, rank)
            s.set_parameters({"HIP": dict(params, tolerance=1e-8)})
            s.generate_poisson7(N, N, N, int(cuts[rank]), int(cuts[rank + 1]))
            nl = s.matrix_shape()[0]
            bb, xx = s.device_array(nl), s.to_device(np.zeros(nl))
            s.generate_rhs(42, bb)
            t = time.time(); s.solve_device(bb, xx); dt = time.time() - t
            out[rank] = (s.get_info(), xx.download(), dt)
        except Exception as e:
            err.append((rank, repr(e)))
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join(timeout=300) for t in th]
    assert not err, err
    xs = np.concatenate([o[1] for o in out])
    i = out[0][0]
    print(f"{world} ranks, {label}: {i['num_iterations']} iterations, true residual {i['true_residual']:.3e}, "
          f"max |x - x_single| = {np.abs(xs - xref).max():.2e}, {out[0][2]*1e3:.0f} ms on one shared GPU", flush=True)
