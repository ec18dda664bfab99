"""BSR-3 product of configs[2] (Q1 elasticity M = 100): in-loop shape (SPMV_DOT), back-to-back launches timed by HIP events,
against the number of resident workgroups per CU -- is the kernel bound by the bytes it keeps in flight?"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from polysolve_amd import HIPSolver
from bench import BoxSampler
M = int(os.environ.get("M", "100"))
res = []
for wg in [int(v) for v in os.environ.get("WGS", "3,4,5,6").split(",")]:
    s = HIPSolver("Eigen::IdentityPreconditioner")
    s.set_parameters({"HIP": dict(block_size=3, spmv_blocks_per_cu=wg)})
    s.generate_elasticity_q1(M); s.synchronize()
    n = s.matrix_shape()[0]
    nb, nnzb = int(s.get_param("bsr3_nb")), int(s.get_param("bsr3_nnzb"))
    x, y = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, x)
    with BoxSampler() as box:
        ms = min(s.time_spmv(x, y, 50) for _ in range(3))
    by = 76 * nnzb + 52 * nb
    res.append(dict(wg_per_cu=wg, ms=ms, gbs=by / ms / 1e6, frac=by / ms / 1e6 / 8000, sclk=box.summary()["sclk_mhz"], power=box.summary()["power_w"]))
    print(json.dumps(res[-1]), flush=True)
    x.free(); y.free(); del s
