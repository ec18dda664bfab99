"""ctypes front-end of the CPU oracle (oracle/*.c).  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product (``polysolve_amd``) never does.  PARITY UNPINNED -- see the header of
``oracle/psolve_oracle.c``: the reference's arithmetic for this path lives in Eigen 5.0.1 and
AMGCL 1.4.3, which are neither vendored in the reference nor present in this image.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpsolve_oracle.so")

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("psolve_oracle.c", "amg_oracle.c", "elasticity_oracle.c", "schwarz_oracle.c", "ic_oracle.c", "reorder_oracle.c", "amd_oracle.c", "cpu_tuned.c")]
    stale = force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()  # (an mtime check; a stale .so -- a source newer than it -- is rebuilt, not loaded)
        L = C.CDLL(_SO)
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_poisson7_nnz.restype = C.c_int64
        L.orc_poisson7_nnz.argtypes = [C.c_int] * 5
        L.orc_poisson7_fill.argtypes = [C.c_int] * 5 + [_i32p, _i32p, _f64p]
        L.orc_splitmix_fill.argtypes = [_f64p, C.c_int64, C.c_int64, C.c_uint64]
        L.orc_stream_triad.restype = C.c_double
        L.orc_stream_triad.argtypes = [C.c_int64, C.c_int]
        L.orc_spmv.argtypes = [C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p]
        L.orc_residual.argtypes = [C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p, _f64p]
        L.orc_dot.restype = C.c_double
        L.orc_dot.argtypes = [C.c_int64, _f64p, _f64p]
        L.orc_jacobi_setup.argtypes = [C.c_int64, _i32p, _i32p, _f64p, _f64p]
        L.orc_cg_eigen.argtypes = [C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_double, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_void_p]
        L.orc_cg_amgcl.argtypes = [C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_double, C.c_double, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
        L.orc_cg_jacobi_tuned.argtypes = [C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p, C.c_double, C.c_int64,
                                          C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_amg_create.restype = C.c_void_p
        L.orc_amg_create.argtypes = [C.c_int64, _i32p, _i32p, _f64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                     C.c_double, C.c_int]
        L.orc_amg_create_bs.restype = C.c_void_p
        L.orc_amg_create_bs.argtypes = L.orc_amg_create.argtypes + [C.c_int]
        L.orc_amg_create_ex.restype = C.c_void_p
        L.orc_amg_create_ex.argtypes = [C.c_int64, _i32p, _i32p, _f64p, _f64p, C.c_int]
        L.orc_compact_aggregates.restype = C.c_int64
        L.orc_compact_aggregates.argtypes = [C.c_int64, _i32p, _i32p, _f64p, C.c_double, _i32p, C.POINTER(C.c_int)]
        L.orc_parallel_aggregates.restype = C.c_int64
        L.orc_parallel_aggregates.argtypes = [C.c_int64, _i32p, _i32p, _f64p, C.c_double, _i32p, C.POINTER(C.c_int)]
        L.orc_amg_destroy.argtypes = [C.c_void_p]
        L.orc_amg_apply.argtypes = [C.c_void_p, _f64p, _f64p]
        L.orc_amg_num_levels.restype = C.c_int
        L.orc_amg_num_levels.argtypes = [C.c_void_p]
        L.orc_amg_level_shape.restype = C.c_int
        L.orc_amg_level_shape.argtypes = [C.c_void_p, C.c_int, C.c_int, np.ctypeslib.ndpointer(np.int64)]
        L.orc_amg_level_copy.restype = C.c_int
        L.orc_amg_level_copy.argtypes = [C.c_void_p, C.c_int, C.c_int, _i32p, _i32p, _f64p]
        L.orc_amg_level_scalars.argtypes = [C.c_void_p, C.c_int, _f64p]
        L.orc_plain_aggregates.restype = C.c_int64
        L.orc_plain_aggregates.argtypes = [C.c_int64, _i32p, _i32p, _f64p, C.c_double, _i32p]
        L.orc_spectral_radius.restype = C.c_double
        L.orc_spectral_radius.argtypes = [C.c_int64, _i32p, _i32p, _f64p, C.c_int, C.c_int]
        L.orc_chebyshev.argtypes = [C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p, C.c_int, C.c_double, C.c_double,
                                    C.c_double]
        L.orc_mt19937_uniform.argtypes = [C.c_uint32, C.c_int64, _f64p]
        L.orc_schwarz_create_bs.restype = C.c_void_p
        L.orc_schwarz_create_bs.argtypes = [C.c_int64, _i32p, _i32p, _f64p, C.c_int, C.c_int]
        L.orc_schwarz_destroy.argtypes = [C.c_void_p]
        L.orc_schwarz_levels.restype = C.c_int
        L.orc_schwarz_levels.argtypes = [C.c_void_p]
        L.orc_schwarz_apply.argtypes = [C.c_void_p, _f64p, _f64p]
        L.orc_ic_create.restype = C.c_void_p
        L.orc_ic_create.argtypes = [C.c_int64, _i32p, _i32p, _f64p, C.c_double]
        L.orc_ic_destroy.argtypes = [C.c_void_p]
        L.orc_ic_apply.argtypes = [C.c_void_p, _f64p, _f64p]
        L.orc_ic_info.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_ic_copy.argtypes = [C.c_void_p, _i32p, _i32p, _f64p, _f64p]
        L.orc_amd_order.restype = C.c_int
        L.orc_amd_order.argtypes = [C.c_int64, _i32p, _i32p, _i32p]
        L.orc_cuthill_mckee.restype = C.c_int
        L.orc_cuthill_mckee.argtypes = [C.c_int64, _i32p, _i32p, C.c_int, _i32p, np.ctypeslib.ndpointer(np.int64)]
        L.orc_elasticity_q1.restype = C.c_int64
        L.orc_elasticity_q1.argtypes = [C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


# ------------------------------------------------------------------------------------------------
@dataclass
class CSR:
    """int32 / fp64 CSR.  For a symmetric matrix these are also Eigen's ColMajor arrays
    (outerIndexPtr, innerIndexPtr, valuePtr) -- AMGCL.hpp:36-43."""
    n: int
    rowptr: np.ndarray
    col: np.ndarray
    val: np.ndarray
    ncols: int | None = None

    @property
    def nnz(self) -> int:
        return int(self.rowptr[-1])

    def to_scipy(self):
        import scipy.sparse as sp
        m = self.ncols if self.ncols is not None else self.n
        return sp.csr_matrix((self.val, self.col, self.rowptr), shape=(self.n, m))

    @staticmethod
    def from_scipy(A) -> "CSR":
        A = A.tocsr()
        A.sort_indices()
        return CSR(A.shape[0], A.indptr.astype(np.int32), A.indices.astype(np.int32),
                   np.ascontiguousarray(A.data, dtype=np.float64), A.shape[1])


def poisson7(nx: int, ny: int | None = None, nz: int | None = None, z0: int = 0, z1: int | None = None) -> CSR:
    """Rows z0..z1 of the 7-point Laplacian (diag 6, off-diag -1), global column ids."""
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    z1 = nz if z1 is None else z1
    L = lib()
    nnz = L.orc_poisson7_nnz(nx, ny, nz, z0, z1)
    nrows = (z1 - z0) * nx * ny
    rowptr = np.empty(nrows + 1, np.int32)
    col = np.empty(nnz, np.int32)
    val = np.empty(nnz, np.float64)
    L.orc_poisson7_fill(nx, ny, nz, z0, z1, rowptr, col, val)
    assert rowptr[-1] == nnz
    return CSR(nrows, rowptr, col, val, nx * ny * nz)


def splitmix_vector(n: int, seed: int = 42, start: int = 0) -> np.ndarray:
    x = np.empty(n, np.float64)
    lib().orc_splitmix_fill(x, start, n, seed)
    return x


def elasticity_q1(M: int, E: float = 1.0, nu: float = 0.3) -> CSR:
    L = lib()
    nnz = L.orc_elasticity_q1(M, E, nu, None, None, None)
    n = 3 * M ** 3
    rowptr = np.empty(n + 1, np.int32)
    col = np.empty(nnz, np.int32)
    val = np.empty(nnz, np.float64)
    L.orc_elasticity_q1(M, E, nu, rowptr.ctypes.data, col.ctypes.data, val.ctypes.data)
    return CSR(n, rowptr, col, val, n)


def stream_triad(n: int, reps: int = 5) -> float:
    """GB/s of a = b + s c over three arrays of n doubles (OpenMP, first touch by the streaming threads), best of reps"""
    return float(lib().orc_stream_triad(int(n), int(reps)))


def spmv(A: CSR, x: np.ndarray) -> np.ndarray:
    y = np.empty(A.n, np.float64)
    lib().orc_spmv(A.n, A.rowptr, A.col, A.val, np.ascontiguousarray(x, np.float64), y)
    return y


def dot(a, b) -> float:
    return lib().orc_dot(len(a), np.ascontiguousarray(a, np.float64), np.ascontiguousarray(b, np.float64))


def jacobi_setup(A: CSR) -> np.ndarray:
    d = np.empty(A.n, np.float64)
    lib().orc_jacobi_setup(A.n, A.rowptr, A.col, A.val, d)
    return d


# AMGCL.cpp:32-65 default_params() + amgcl's own defaults for what polysolve leaves unset
AMGCL_DEFAULTS = dict(max_levels=6, coarse_enough=3000, ncycle=2, npre=1, npost=1, eps_strong=0.0, sa_relax=1.0,
                      estimate_spectral_radius=1, sa_power_iters=0, cheb_degree=16, cheb_power_iters=100,
                      cheb_higher=2.0, cheb_lower=0.008333333333, cheb_scale=1, block_size=1,
                      # round 5 (orc_amg_create_ex): aggregation "amgcl" | "parallel"; coarsening "smoothed_aggregation" |
                      # "aggregation"; relax_type "chebyshev" | "damped_jacobi" | "spai0"; direct_coarse
                      aggregation="amgcl", coarsening="smoothed_aggregation", over_interp=0.0, relax_type="chebyshev",
                      damping=0.72, direct_coarse=0,
                      # round 6: relax_type "gauss_seidel" | "ilu0" (ilu_damping: amgcl::relaxation::ilu0::params::damping);
                      # precond_class "amg" | "relaxation" (/AMGCL/precond/class, amgcl::runtime::preconditioner)
                      ilu_damping=1.0, precond_class="amg")
_AMG_ENUMS = dict(aggregation={"amgcl": 0, "parallel": 1, "compact": 2}, coarsening={"smoothed_aggregation": 0, "aggregation": 1},
                  relax_type={"chebyshev": 0, "damped_jacobi": 1, "spai0": 2, "gauss_seidel": 3, "ilu0": 4},
                  precond_class={"amg": 0, "relaxation": 1})
_AMG_OPT_ORDER = ("max_levels", "coarse_enough", "ncycle", "npre", "npost", "eps_strong", "sa_relax",
                  "estimate_spectral_radius", "sa_power_iters", "cheb_degree", "cheb_power_iters", "cheb_higher",
                  "cheb_lower", "cheb_scale", "block_size", "aggregation", "coarsening", "over_interp", "relax_type",
                  "damping", "direct_coarse", "ilu_damping", "precond_class")


class AMG:
    """amgcl::amg<builtin<double>, smoothed_aggregation, chebyshev> restated (amg_oracle.c)."""

    def __init__(self, A: CSR, **params):
        p = dict(AMGCL_DEFAULTS)
        p.update(params)
        self.params = p
        self.A = A
        if p["block_size"] > 1 and A.n % p["block_size"]:
            raise ValueError("matrix size is not a multiple of block_size")
        unknown = set(p) - set(_AMG_OPT_ORDER)
        if unknown:
            raise ValueError(f"unknown AMG options {sorted(unknown)}")
        opts = np.array([float(_AMG_ENUMS[k][p[k]] if k in _AMG_ENUMS and isinstance(p[k], str) else p[k])
                         for k in _AMG_OPT_ORDER], np.float64)
        self._h = lib().orc_amg_create_ex(A.n, A.rowptr, A.col, A.val, opts, len(opts))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_amg_destroy(self._h)
            self._h = None

    @property
    def num_levels(self) -> int:
        return lib().orc_amg_num_levels(self._h)

    def level(self, l: int, what: str = "A") -> CSR | None:
        w = {"A": 0, "P": 1, "R": 2}[what]
        shape = np.zeros(3, np.int64)
        if not lib().orc_amg_level_shape(self._h, l, w, shape):
            return None
        nr, nc, nnz = (int(v) for v in shape)
        ptr = np.empty(nr + 1, np.int32)
        col = np.empty(max(nnz, 1), np.int32)
        val = np.empty(max(nnz, 1), np.float64)
        lib().orc_amg_level_copy(self._h, l, w, ptr, col, val)
        return CSR(nr, ptr, col[:nnz], val[:nnz], nc)

    def level_scalars(self, l: int) -> dict:
        out = np.zeros(4)
        lib().orc_amg_level_scalars(self._h, l, out)
        return dict(rho=out[0], d=out[1], c=out[2], omega=out[3])

    def apply(self, r: np.ndarray) -> np.ndarray:
        z = np.empty(self.A.n, np.float64)
        lib().orc_amg_apply(self._h, np.ascontiguousarray(r, np.float64), z)
        return z


class Schwarz:
    """Multilevel additive Schwarz on 64-unknown dense domains (schwarz_oracle.c): the restatement of this
    repository's precond = "schwarz"."""

    def __init__(self, A: CSR, levels: int = 3, block_size: int = 1):
        self.A = A
        self._h = lib().orc_schwarz_create_bs(A.n, A.rowptr, A.col, A.val, levels, block_size)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_schwarz_destroy(self._h)
            self._h = None

    @property
    def num_levels(self) -> int:
        return lib().orc_schwarz_levels(self._h)

    def apply(self, r: np.ndarray) -> np.ndarray:
        z = np.empty(self.A.n, np.float64)
        lib().orc_schwarz_apply(self._h, np.ascontiguousarray(r, np.float64), z)
        return z


def amd_order(A: CSR) -> np.ndarray:
    """Eigen::AMDOrdering<int> restated (amd_oracle.c): order[k] = the k-th pivot of the approximate minimum degree
    ordering of the symmetric pattern of A (both triangles and the diagonal stored)."""
    order = np.empty(A.n, np.int32)
    rc = lib().orc_amd_order(A.n, A.rowptr, A.col, order)
    assert rc == 0
    return order


class IC:
    """Eigen::IncompleteCholesky<double, Lower, Ordering> restated (ic_oracle.c): scaled, shifted, left-looking incomplete
    Cholesky that keeps as many entries per column as the matrix has.  ordering "natural" (NaturalOrdering<int>) or "amd"
    (AMDOrdering<int>, the class template's -- and so the reference's -- default: the matrix is factored in the order of
    amd_order(), right-hand sides are permuted in and solutions out, as IncompleteCholesky::_solve_impl does)."""

    def __init__(self, A: CSR, initial_shift: float = 1e-3, ordering: str = "natural"):
        self.order = None
        if ordering == "amd":
            self.order = amd_order(A)
            A = permuted(A, self.order)
        elif ordering != "natural":
            raise ValueError(ordering)
        self.A = A
        self._h = lib().orc_ic_create(A.n, A.rowptr, A.col, A.val, initial_shift)
        if not self._h:
            raise ValueError("a column has no stored diagonal entry")
        sh, nnz, att, ok = C.c_double(), C.c_int64(), C.c_int(), C.c_int()
        lib().orc_ic_info(self._h, C.byref(sh), C.byref(nnz), C.byref(att), C.byref(ok))
        self.shift, self.nnz, self.attempts, self.ok = sh.value, nnz.value, att.value, bool(ok.value)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_ic_destroy(self._h)
            self._h = None

    def factor(self):
        """(colptr, rowidx, vals, scale): L by columns (diagonal first), the scaling S"""
        colptr = np.empty(self.A.n + 1, np.int32)
        rowidx = np.empty(self.nnz, np.int32)
        vals = np.empty(self.nnz, np.float64)
        scale = np.empty(self.A.n, np.float64)
        lib().orc_ic_copy(self._h, colptr, rowidx, vals, scale)
        return colptr, rowidx, vals, scale

    def apply(self, r: np.ndarray) -> np.ndarray:
        z = np.empty(self.A.n, np.float64)
        r = np.ascontiguousarray(r, np.float64)
        if self.order is None:
            lib().orc_ic_apply(self._h, r, z)
            return z
        lib().orc_ic_apply(self._h, np.ascontiguousarray(r[self.order]), z)
        out = np.empty_like(z)
        out[self.order] = z
        return out


def _precond_args(A: CSR, precond):
    if precond is None or precond == "none":
        return 0, None, None, None
    if isinstance(precond, AMG):
        return 2, None, precond._h, precond
    if isinstance(precond, Schwarz):
        return 3, None, precond._h, precond
    if isinstance(precond, IC):
        return 4, None, precond._h, precond
    if precond == "jacobi":
        d = jacobi_setup(A)
        return 1, d.ctypes.data, None, d
    raise ValueError(precond)


def cg_eigen(A: CSR, b, x0=None, precond="jacobi", tol=1e-8, max_iter=10000, history=False):
    """Eigen::ConjugateGradient<.., Lower|Upper, DiagonalPreconditioner>::solveWithGuess restated.
    Returns (x, iterations(), error()[, ||r||^2 history])."""
    b = np.ascontiguousarray(b, np.float64)
    x = np.zeros(A.n) if x0 is None else np.array(x0, np.float64, copy=True)
    kind, dptr, amg, keep = _precond_args(A, precond)
    it, err = C.c_int64(0), C.c_double(0)
    hist = np.full(max_iter + 1, np.nan) if history else None
    lib().orc_cg_eigen(A.n, A.rowptr, A.col, A.val, b, x, kind, dptr, amg, tol, max_iter, C.byref(it),
                       C.byref(err), hist.ctypes.data if history else None)
    del keep
    if history:
        return x, it.value, err.value, hist[: it.value + 2]
    return x, it.value, err.value


def cg_jacobi_tuned(A: CSR, b, x0=None, tol=1e-8, max_iter=10000, loop_seconds=False):
    """bench.py's tuned CPU leg (cpu_tuned.c): the recurrence of cg_eigen with Jacobi, three fused passes per iteration,
    private first-touch copies.  Returns (x, iterations, error) like cg_eigen (+ the wall time of the iteration loop alone
    with loop_seconds=True: the copies are a per-factorize cost)."""
    b = np.ascontiguousarray(b, np.float64)
    x = np.zeros(A.n) if x0 is None else np.array(x0, np.float64, copy=True)
    it, err = C.c_int64(0), C.c_double(0)
    secs = C.c_double(0)
    lib().orc_cg_jacobi_tuned(A.n, A.rowptr, A.col, A.val, b, x, tol, max_iter, C.byref(it), C.byref(err), C.byref(secs))
    if loop_seconds:
        return x, it.value, err.value, secs.value
    return x, it.value, err.value


def cg_amgcl(A: CSR, b, x0=None, precond=None, tol=1e-10, abstol=0.0, max_iter=1000):
    """amgcl::solver::cg restated.  Returns (x, num_iterations, final_res_norm)."""
    b = np.ascontiguousarray(b, np.float64)
    x = np.zeros(A.n) if x0 is None else np.array(x0, np.float64, copy=True)
    kind, dptr, amg, keep = _precond_args(A, precond)
    it, err = C.c_int64(0), C.c_double(0)
    lib().orc_cg_amgcl(A.n, A.rowptr, A.col, A.val, b, x, kind, dptr, amg, tol, abstol, max_iter, C.byref(it),
                       C.byref(err))
    del keep
    return x, it.value, err.value


def chebyshev(A: CSR, rhs, x0, degree: int, rho: float, higher: float = 2.0, lower: float = 1.0 / 120):
    x = np.array(x0, np.float64, copy=True)
    lib().orc_chebyshev(A.n, A.rowptr, A.col, A.val, np.ascontiguousarray(rhs, np.float64), x, degree, rho, higher,
                        lower)
    return x


def plain_aggregates(A: CSR, eps_strong: float = 0.0):
    ids = np.empty(A.n, np.int32)
    cnt = lib().orc_plain_aggregates(A.n, A.rowptr, A.col, A.val, eps_strong, ids)
    return int(cnt), ids


def parallel_aggregates(A: CSR, eps_strong: float = 0.0):
    """(count, ids, rounds) of amg.aggregation = "parallel": distance-2 maximal independent set by hashed priorities on
    the strength graph, membership by the sweep's closed form (amg_oracle.c: parallel_aggregates_graph)."""
    ids = np.empty(A.n, np.int32)
    rounds = C.c_int(0)
    cnt = lib().orc_parallel_aggregates(A.n, A.rowptr, A.col, A.val, eps_strong, ids, C.byref(rounds))
    return int(cnt), ids, rounds.value


def compact_aggregates(A: CSR, eps_strong: float = 0.0):
    """(count, ids, rounds) of amg.aggregation = "compact" (round 6): one-hop aggregates around two generations of
    hashed-priority distance-2 independent sets, the rest by most connections (amg_oracle.c: compact_aggregates_graph)."""
    ids = np.empty(A.n, np.int32)
    rounds = C.c_int(0)
    cnt = lib().orc_compact_aggregates(A.n, A.rowptr, A.col, A.val, eps_strong, ids, C.byref(rounds))
    return int(cnt), ids, rounds.value


def cuthill_mckee(A: CSR, max_components: int = 64, reverse: bool = False):
    """The backend's optional renumbering (oracle/reorder_oracle.c): order[k] = old index of the vertex at new position
    k, and {levels, components, isolated, leftover}.  reverse: the same order read backwards (reverse Cuthill-McKee) --
    what the backend numbers by unless "reorder_reverse" is switched off."""
    order = np.empty(A.n, np.int32)
    info = np.zeros(4, np.int64)
    rc = lib().orc_cuthill_mckee(A.n, A.rowptr, A.col, max_components, order, info)
    assert rc == 0
    if reverse:
        order = np.ascontiguousarray(order[::-1])
    return order, dict(zip(("levels", "components", "isolated", "leftover"), (int(v) for v in info)))


def permuted(A: CSR, order: np.ndarray) -> CSR:
    """P A P^T with row k of the result = row order[k] of A, columns renamed and sorted inside every row."""
    M = A.to_scipy().tocsr()
    B = M[order][:, order].tocsr()
    B.sort_indices()
    return CSR.from_scipy(B)


def spectral_radius(A: CSR, scale: bool = True, power_iters: int = 0) -> float:
    return lib().orc_spectral_radius(A.n, A.rowptr, A.col, A.val, int(scale), power_iters)


def mt19937_uniform(seed: int, n: int) -> np.ndarray:
    out = np.empty(n)
    lib().orc_mt19937_uniform(seed, n, out)
    return out


def gr_30_30() -> CSR:
    """The Matrix-Market `gr_30_30` test matrix the reference loads in
    tests/test_linear_solver.cpp:547-549: 9-point Laplacian on a 30x30 grid, diag 8 / off-diag -1
    (900 rows, 7744 nnz).  Regenerated synthetically -- the file itself is not in this image."""
    import scipy.sparse as sp
    n = 30
    idx = np.arange(n * n).reshape(n, n)
    rows, cols, vals = [], [], []
    for di in (-1, 0, 1):
        for dj in (-1, 0, 1):
            src = idx[max(0, -di): n - max(0, di), max(0, -dj): n - max(0, dj)]
            dst = idx[max(0, di): n - max(0, -di), max(0, dj): n - max(0, -dj)]
            rows.append(src.ravel())
            cols.append(dst.ravel())
            vals.append(np.full(src.size, 8.0 if (di == 0 and dj == 0) else -1.0))
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n * n, n * n))
    return CSR.from_scipy(A)


# ------------------------------------------------------------------------------------------------
# Opportunistic true oracle: the REAL Eigen / AMGCL behind the same signatures, when their headers exist
_REF_SO = os.path.join(_HERE, "_ref", "libpsolve_trueoracle.so")
_ref_lib = None


def true_oracle():
    """ctypes handle of oracle/_ref/libpsolve_trueoracle.so (built on demand by `make -C oracle ref`), or None
    when it cannot be built.  `.ref_have_eigen()` / `.ref_have_amgcl()` say which half is real."""
    global _ref_lib
    if _ref_lib is None:
        try:
            src = os.path.join(_HERE, "true_oracle.cpp")
            if not os.path.exists(_REF_SO) or os.path.getmtime(_REF_SO) < os.path.getmtime(src):
                subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
            L = C.CDLL(_REF_SO)
            L.ref_eigen_cg.argtypes = [C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p, C.c_int, C.c_double, C.c_int64,
                                       C.POINTER(C.c_int64), C.POINTER(C.c_double)]
            L.ref_amgcl_solve.argtypes = [C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p, C.c_int, C.c_double, C.c_int64,
                                          C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_void_p, C.c_void_p]
            _ref_lib = L
        except Exception:
            _ref_lib = False
    return _ref_lib or None
