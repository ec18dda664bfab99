"""One rank of tests/test_gpu_multi.py::test_one_process_per_gpu_vs_oracle: python rccl_worker.py RANK WORLD DIR NX NY NZ.
One process per GPU; the RCCL unique id travels through a file (what bench.py does with torch.distributed)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from polysolve_amd import HIPSolver  # noqa: E402


def main():
    rank, world, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    nx, ny, nz = (int(v) for v in sys.argv[4:7])
    idf = os.path.join(d, "rccl_id.bin")
    if rank == 0:
        uid = HIPSolver.comm_unique_id()
        with open(idf + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idf + ".tmp", idf)
    else:
        t0 = time.time()
        while not os.path.exists(idf):
            if time.time() - t0 > 120:
                raise SystemExit("no unique id from rank 0")
            time.sleep(0.05)
        uid = open(idf, "rb").read()
    cuts = [round(q * nz / world) for q in range(world + 1)]
    for tag, prm in (("jacobi1", {"dist_single_reduction": 1}), ("jacobi2", {"dist_single_reduction": 0}),
                     ("amg", {"precond": "amg", "amg": {"coarse_enough": 200, "ncycle": 1, "cheb_degree": 2}})):
        s = HIPSolver("", device=rank)
        s.set_parameters({"HIP": dict(prm, tolerance=1e-8, max_iter=5000)})
        if tag == "jacobi1":
            s.comm_init(rank, world, uid)
            keep = s
        else:  # one communicator pair per process is enough: the later handles share the first one's id file anew
            idk = os.path.join(d, f"rccl_id_{tag}.bin")
            if rank == 0:
                u2 = HIPSolver.comm_unique_id()
                with open(idk + ".tmp", "wb") as f:
                    f.write(u2)
                os.replace(idk + ".tmp", idk)
            else:
                t0 = time.time()
                while not os.path.exists(idk):
                    if time.time() - t0 > 120:
                        raise SystemExit("no unique id from rank 0")
                    time.sleep(0.05)
                u2 = open(idk, "rb").read()
            s.comm_init(rank, world, u2)
        s.generate_poisson7(nx, ny, nz, cuts[rank], cuts[rank + 1])
        n = s.matrix_shape()[0]
        b, x = s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        s.axpby_device(n, 0.0, b, 0.0, x)
        s.solve_device(b, x)
        np.save(os.path.join(d, f"x_{tag}_{rank}.npy"), x.download())
        np.save(os.path.join(d, f"it_{tag}_{rank}.npy"), np.array(s.get_info()["solver_iter"]))
    del keep


if __name__ == "__main__":
    main()
