"""First contact with the GPU: SpMV bit-parity, PCG parity vs the oracle, then a 256^3 timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as O
from polysolve_amd import Solver

def run(N, check=True):
    s = Solver.create("HIP", "")
    A = O.poisson7(N)
    S = A.to_scipy()
    xs = O.splitmix_vector(A.n)
    b = O.spmv(A, xs)
    s.analyze_pattern(S, A.n); s.factorize(S)
    dx = s.to_device(xs); dy = s.device_array(A.n)
    s.spmv_device(dx, dy)
    y = dy.download()
    print(f"N={N} spmv max|diff| vs oracle = {np.abs(y-b).max():.3e} bitexact={np.array_equal(y,b)}")
    pq = s.spmv_dot_device(dx, dy)
    print("  spmv_dot", pq, O.dot(xs, b), abs(pq-O.dot(xs,b))/abs(pq))
    x = np.zeros(A.n)
    s.set_parameters({"HIP": {"tolerance": 1e-8}})
    s.solve(b, x)
    info = s.get_info()
    xo, it, err = O.cg_eigen(A, b, tol=1e-8)
    print(f"  gpu: iter={info['solver_iter']} passes={info['num_iterations']} err={info['solver_error']:.3e} true={info['true_residual']:.3e} status={info['solver_status']}")
    print(f"  orc: iter={it} err={err:.3e}  |x-xo|max={np.abs(x-xo).max():.3e} |x-x*|max={np.abs(x-xs).max():.3e}")

for N in (8, 32, 64):
    run(N)

# device generator parity
s = Solver.create("HIP", "")
s.generate_poisson7(16)
n, nnz, nh = s.matrix_shape()
A = O.poisson7(16)
print("gen shape", n, nnz, nh, A.n, A.nnz)
db = s.device_array(n); dxs = s.device_array(n)
s.generate_rhs(42, db, dxs)
xs = O.splitmix_vector(A.n); b = O.spmv(A, xs)
print("gen rhs bitexact", np.array_equal(db.download(), b), np.array_equal(dxs.download(), xs))

# big one
N = int(os.environ.get("BIGN", "256"))
s = Solver.create("HIP", "")
t = time.time(); s.generate_poisson7(N); print("generate+factorize s", time.time()-t)
n, nnz, nh = s.matrix_shape()
db = s.device_array(n); dx = s.device_array(n); dy = s.device_array(n)
s.generate_rhs(42, db, dx)
for bpc in (4, 8, 12, 16):
    s._set("blocks_per_cu", bpc)
    ms = s.time_spmv(dx, dy, 20)
    byts = 12*nnz + 20*n
    print(f"bpc={bpc} spmv ms={ms:.4f}  alg GB/s={byts/ms/1e6:.1f}")
s._set("blocks_per_cu", 8)
u, d = s.time_vecops(20)
print(f"vecops: update_r ms={u:.4f} ({32*n/u/1e6:.0f} GB/s)  update_xp ms={d:.4f} ({48*n/d/1e6:.0f} GB/s)")
dx.upload(np.zeros(n))
s._set("profile_spmv", 8)
t = time.time(); s.solve_device(db, dx); dt = time.time()-t
info = s.get_info()
print(f"solve N={N}: {dt:.3f}s iter={info['solver_iter']} err={info['solver_error']:.3e} true={info['true_residual']:.3e} "
      f"DOF/s={n/dt:.3e} ms/iter={dt*1e3/max(1,info['num_iterations']):.4f} spmv_ms={info['spmv_ms_avg']:.4f} samples={info['spmv_samples']}")
