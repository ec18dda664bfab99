"""Generates the committed fixtures under tests/golden/ (run from the repo root:
`python tests/golden/make_golden.py`).

The reference holds NO golden vectors for this path (its tests are tolerance-only on matrices that
are downloaded at configure time) and its arithmetic lives in Eigen/AMGCL, which
are not importable here.  These fixtures therefore pin (a) the synthetic inputs, (b) an exact
solution computed independently with scipy.sparse.linalg.spsolve, and (c) what the CPU oracle
returned when the fixture was made (iteration counts, residual history), so that later edits of the
oracle or of the HIP path cannot drift silently.
"""
import json
import os
import sys

import numpy as np
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def one(name, A, b, amg_params):
    S = A.to_scipy().tocsc()
    x_exact = spla.spsolve(S, b)
    xe, it_e, err_e, hist = O.cg_eigen(A, b, precond="jacobi", tol=1e-8, max_iter=2000, history=True)
    xn, it_n, err_n = O.cg_eigen(A, b, precond="none", tol=1e-8, max_iter=2000)
    amg = O.AMG(A, **amg_params)
    xa, it_a, err_a = O.cg_amgcl(A, b, precond=amg, tol=1e-10, max_iter=1000)
    z = amg.apply(b)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), n=A.n, rowptr=A.rowptr, col=A.col, val=A.val, b=b, x_exact=x_exact,
        cg_jacobi_x=xe, cg_jacobi_iters=it_e, cg_jacobi_err=err_e, cg_jacobi_hist=hist,
        cg_none_iters=it_n, cg_none_err=err_n,
        amg_levels=amg.num_levels, amg_level_rows=np.array([amg.level(l).n for l in range(amg.num_levels)]),
        amg_level_nnz=np.array([amg.level(l).nnz for l in range(amg.num_levels)]),
        amg_apply_b=z, cg_amg_iters=it_a, cg_amg_err=err_a, cg_amg_x=xa,
        amg_params=json.dumps(amg_params))
    print(name, A.n, A.nnz, "cg_jacobi", it_e, "cg_none", it_n, "cg_amg", it_a, "levels", amg.num_levels)


def schwarz_fixture():
    """precond = "schwarz" on three of the fixtures above: z = M^-1 b and the PCG iteration count, after checking the
    oracle's operator against a dense numpy construction (sum_l P_l blockdiag_64(P_l^T A P_l)^-1 P_l^T)."""
    out = {}
    for name, bs, levels in (("poisson7_n12", 1, 3), ("gr_30_30", 1, 2), ("elasticity_q1_m5", 3, 2)):
        g = np.load(os.path.join(OUT, name + ".npz"))
        A = O.CSR(int(g["n"]), g["rowptr"], g["col"], g["val"], int(g["n"]))
        S = O.Schwarz(A, levels, block_size=bs)
        n, M, idx = A.n, A.to_scipy().toarray(), np.arange(A.n)
        ref = np.zeros((n, n))
        for l in range(S.num_levels):
            agg = ((idx // bs) >> (6 * l)) * bs + idx % bs if l else idx
            P = np.zeros((n, agg.max() + 1))
            P[idx, agg] = 1
            Al = P.T @ M @ P
            Binv = np.zeros_like(Al)
            for k in range(0, Al.shape[0], 64):
                sl = slice(k, min(k + 64, Al.shape[0]))
                Binv[sl, sl] = np.linalg.inv(Al[sl, sl])
            ref += P @ Binv @ P.T
        z = S.apply(g["b"])
        assert np.abs(z - ref @ g["b"]).max() <= 1e-12 * np.abs(z).max()
        x, it, err = O.cg_eigen(A, g["b"], precond=S, tol=1e-8, max_iter=2000)
        out[name + "_z"] = z
        out[name + "_iters"] = it
        out[name + "_levels"] = S.num_levels
        out[name + "_cfg"] = np.array([levels, bs])
        print("schwarz", name, "levels", S.num_levels, "pcg", it, "jacobi", int(g["cg_jacobi_iters"]))
    np.savez_compressed(os.path.join(OUT, "schwarz.npz"), **out)


def reorder_fixture():
    """"reorder": an unstructured system (P1 Laplace on Delaunay tetrahedra, tests/mesh_utils.py, hull nodes clamped, nodes
    in a random order) with the oracle's Cuthill-McKee order of it -- checked here against scipy's breadth_first_order
    from the same start vertex (the same definition; the clamped rows are isolated vertices and come first) --, the
    level count, and the iteration counts of the oracle's Jacobi / AMG solves of the permuted system."""
    from scipy.sparse.csgraph import breadth_first_order
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import mesh_utils as mu
    P, T, bd = mu.tet_mesh(9, seed=5)
    K, _ = mu.renumber_nodes(mu.p1_laplace(P, T, bd), 1, seed=6)
    A = O.CSR.from_scipy(K)
    order, info = O.cuthill_mckee(A)
    iso = int(info["isolated"])
    assert iso == int(bd.sum()) and info["components"] == 1 and np.array_equal(np.sort(order), np.arange(A.n))
    bfs = breadth_first_order(K, int(order[iso]), directed=False, return_predecessors=False)
    assert np.array_equal(order[iso:], bfs) and np.array_equal(order[:iso], np.flatnonzero(np.diff(K.indptr) == 1))
    B = O.permuted(A, order)
    b = O.spmv(A, O.splitmix_vector(A.n, 42))
    x_exact = spla.spsolve(K.tocsc(), b)
    _, it_j, _ = O.cg_eigen(B, b[order], tol=1e-9, max_iter=2000)
    prm = dict(coarse_enough=60, ncycle=1, cheb_degree=3, cheb_power_iters=20)
    amg = O.AMG(B, **prm)
    _, it_a, _ = O.cg_amgcl(B, b[order], precond=amg, tol=1e-9, max_iter=500)
    np.savez_compressed(os.path.join(OUT, "reorder_tets.npz"), n=A.n, rowptr=A.rowptr, col=A.col, val=A.val, b=b, x_exact=x_exact,
                        order=order, levels=info["levels"], isolated=iso, cg_jacobi_iters=it_j, cg_amg_iters=it_a,
                        amg_levels=amg.num_levels, amg_params=json.dumps(prm))
    print("reorder_tets", A.n, A.nnz, info, "cg_jacobi", it_j, "cg_amg", it_a)


if __name__ == "__main__":
    if "--schwarz-only" in sys.argv:
        schwarz_fixture()
        sys.exit(0)
    if "--reorder-only" in sys.argv:
        reorder_fixture()
        sys.exit(0)
    for N in (4, 8, 12):
        A = O.poisson7(N)
        xs = O.splitmix_vector(A.n, 42)
        one(f"poisson7_n{N}", A, O.spmv(A, xs), dict(coarse_enough=50))
    A = O.poisson7(6, 5, 7)  # ragged grid
    one("poisson7_6x5x7", A, O.spmv(A, O.splitmix_vector(A.n, 7)), dict(coarse_enough=20))
    G = O.gr_30_30()
    one("gr_30_30", G, np.ones(G.n), dict(coarse_enough=100))
    E = O.elasticity_q1(5)
    one("elasticity_q1_m5", E, O.spmv(E, O.splitmix_vector(E.n, 3)), dict(coarse_enough=60))
    schwarz_fixture()
    reorder_fixture()
