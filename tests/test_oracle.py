"""CPU tests of the oracle (oracle/*.c): against scipy, against an independent numpy restatement of
the same recurrences, against the committed golden fixtures, and against the inequalities the
reference's own tests assert (tests/test_linear_solver.cpp) -- the only pins that exist for this
path, since Eigen/AMGCL are not in the image ("parity unpinned")."""
import glob
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def _np_cg_eigen(S, b, x0, dinv, tol, max_iter):
    """Independent numpy restatement of Eigen::internal::conjugate_gradient."""
    x = x0.copy()
    r = b - S @ x
    rhs2 = b @ b
    if rhs2 == 0:
        return np.zeros_like(b), 0, 0.0
    thr = max(tol * tol * rhs2, np.finfo(float).tiny)
    rn2 = r @ r
    if rn2 < thr:
        return x, 0, np.sqrt(rn2 / rhs2)
    p = dinv * r
    absnew = r @ p
    i = 0
    while i < max_iter:
        tmp = S @ p
        alpha = absnew / (p @ tmp)
        x += alpha * p
        r -= alpha * tmp
        rn2 = r @ r
        if rn2 < thr:
            break
        z = dinv * r
        absold = absnew
        absnew = r @ z
        p = z + (absnew / absold) * p
        i += 1
    return x, i, np.sqrt(rn2 / rhs2)


def test_poisson_generator_matches_scipy_kron(oracle):
    for nx, ny, nz in [(4, 4, 4), (5, 3, 2), (1, 1, 7), (7, 1, 1), (2, 2, 1)]:
        A = oracle.poisson7(nx, ny, nz)

        def lap(n):
            return sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]) if n > 1 else sp.csr_matrix([[2.0]])
        I = sp.identity
        K = sp.kron(I(nz), sp.kron(I(ny), lap(nx))) + sp.kron(I(nz), sp.kron(lap(ny), I(nx))) + sp.kron(lap(nz), sp.kron(I(ny), I(nx)))
        # Dirichlet truncation keeps the diagonal at 6 regardless of the number of neighbours
        K = K - sp.diags(K.diagonal()) + 6 * sp.identity(nx * ny * nz)
        assert abs(A.to_scipy() - K).max() == 0
        K = K.tocsr()
        K.eliminate_zeros()
        assert A.nnz == K.nnz
        assert np.all(np.diff(A.col[A.rowptr[0]:A.rowptr[1]]) > 0)


def test_poisson_shards_concatenate(oracle):
    full = oracle.poisson7(5, 4, 6)
    parts = [oracle.poisson7(5, 4, 6, z0, z1) for z0, z1 in [(0, 2), (2, 3), (3, 6)]]
    assert np.array_equal(np.concatenate([p.col for p in parts]), full.col)
    assert np.array_equal(np.concatenate([p.val for p in parts]), full.val)
    assert sum(p.n for p in parts) == full.n


def test_splitmix_is_stateless_per_index(oracle):
    a = oracle.splitmix_vector(1000, 42)
    b = oracle.splitmix_vector(400, 42, start=600)
    assert np.array_equal(a[600:], b)
    assert a.min() >= -1 and a.max() < 1 and abs(a.mean()) < 0.1


def test_spmv_dot_jacobi(oracle):
    A = oracle.poisson7(9, 7, 5)
    S = A.to_scipy()
    x = oracle.splitmix_vector(A.n, 1)
    assert np.allclose(oracle.spmv(A, x), S @ x, rtol=0, atol=1e-13)
    assert np.isclose(oracle.dot(x, x), x @ x, rtol=1e-14)
    assert np.array_equal(oracle.jacobi_setup(A), 1.0 / S.diagonal())


@pytest.mark.parametrize("N", [6, 10, 16])
def test_cg_eigen_matches_numpy_restatement_and_scipy(oracle, N):
    A = oracle.poisson7(N)
    S = A.to_scipy()
    b = oracle.spmv(A, oracle.splitmix_vector(A.n))
    x, it, err = oracle.cg_eigen(A, b, tol=1e-10, max_iter=1000)
    x2, it2, err2 = _np_cg_eigen(S, b, np.zeros(A.n), 1.0 / S.diagonal(), 1e-10, 1000)
    assert it == it2
    assert np.isclose(err, err2, rtol=1e-6)
    assert np.allclose(x, x2, rtol=0, atol=1e-12)
    xe = spla.spsolve(S.tocsc(), b)
    assert np.linalg.norm(x - xe) / np.linalg.norm(xe) < 1e-8
    # reported error is the recurrence residual relative to ||b||
    assert np.isclose(err, np.linalg.norm(b - S @ x) / np.linalg.norm(b), rtol=1e-3)


@pytest.mark.parametrize("k", [1, 5, 17])
def test_cg_iterates_match_scipys_independent_cg(oracle, k):
    """An implementation nobody here wrote: scipy.sparse.linalg.cg run for exactly k iterations (no stopping test) with the
    same preconditioner must give the k-th iterate of both restated recurrences -- Eigen's (Jacobi) and AMGCL's (the
    V-cycle as a LinearOperator) -- to rounding."""
    A = oracle.gr_30_30()
    S = A.to_scipy()
    b = oracle.spmv(A, oracle.splitmix_vector(A.n))
    dinv = 1.0 / S.diagonal()
    xs, _ = spla.cg(S, b, rtol=0.0, atol=0.0, maxiter=k, M=spla.LinearOperator(S.shape, matvec=lambda r: dinv * r))
    xe, it, _ = oracle.cg_eigen(A, b, tol=1e-300, max_iter=k)
    assert it == k and np.abs(xe - xs).max() <= 1e-12 * np.abs(xs).max()
    amg = oracle.AMG(A, coarse_enough=100, ncycle=1, cheb_degree=3, cheb_power_iters=20)
    xs, _ = spla.cg(S, b, rtol=0.0, atol=0.0, maxiter=k, M=spla.LinearOperator(S.shape, matvec=amg.apply))
    xa, ita, _ = oracle.cg_amgcl(A, b, precond=amg, tol=1e-300, max_iter=k)
    assert ita == k and np.abs(xa - xs).max() <= 1e-11 * np.abs(xs).max()


def test_cg_eigen_corner_cases(oracle):
    A = oracle.poisson7(5)
    # zero rhs -> x = 0, 0 iterations, error 0 (even from a non-zero guess)
    x, it, err = oracle.cg_eigen(A, np.zeros(A.n), x0=np.ones(A.n))
    assert it == 0 and err == 0 and not x.any()
    # converged initial guess -> 0 iterations, guess untouched
    b = oracle.spmv(A, oracle.splitmix_vector(A.n))
    x1, it1, _ = oracle.cg_eigen(A, b, tol=1e-12)
    x2, it2, _ = oracle.cg_eigen(A, b, x0=x1, tol=1e-10)
    assert it1 > 0 and it2 == 0 and np.array_equal(x1, x2)
    # max_iter cap: iterations() == max_iter
    _, it3, err3 = oracle.cg_eigen(A, b, tol=1e-14, max_iter=3)
    assert it3 == 3 and err3 > 1e-14


def test_tuned_cpu_leg_runs_the_eigen_recurrence(oracle):
    """bench.py's `tuned_value` leg (oracle/cpu_tuned.c: fused passes, private first-touch copies) is the recurrence of
    cg_eigen: same iteration counts (+-1: other summation order), same x, same capped-iteration and zero-rhs behaviour."""
    for N in (6, 20, 40):
        A = oracle.poisson7(N)
        b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
        x1, i1, e1 = oracle.cg_eigen(A, b, tol=1e-8)
        x2, i2, e2 = oracle.cg_jacobi_tuned(A, b, tol=1e-8)
        assert abs(i1 - i2) <= 1 and np.abs(x1 - x2).max() < 1e-9 and abs(e1 - e2) < 1e-9
        x1, i1, e1 = oracle.cg_eigen(A, b, tol=1e-8, max_iter=4)
        x2, i2, e2 = oracle.cg_jacobi_tuned(A, b, tol=1e-8, max_iter=4)
        assert i1 == i2 == 4 and np.abs(x1 - x2).max() < 1e-12 and abs(e1 - e2) < 1e-12
    x, it, err = oracle.cg_jacobi_tuned(A, np.zeros(A.n), x0=np.ones(A.n))
    assert it == 0 and err == 0.0 and not x.any()  # Eigen: rhs == 0 -> x = 0
    x0 = oracle.splitmix_vector(A.n, 42)
    x, it, err = oracle.cg_jacobi_tuned(A, b, x0=x0)  # the exact solution as the guess: no iteration
    assert it == 0 and np.array_equal(x, x0)


def test_cg_amgcl_counts_one_more_pass_than_eigen(oracle):
    A = oracle.poisson7(8)
    b = oracle.spmv(A, oracle.splitmix_vector(A.n))
    xe, ite, erre = oracle.cg_eigen(A, b, tol=1e-8)
    xa, ita, erra = oracle.cg_amgcl(A, b, precond="jacobi", tol=1e-8)
    assert ita == ite + 1  # Eigen breaks before i++, AMGCL counts the pass
    assert np.isclose(erre, erra, rtol=1e-9)
    assert np.allclose(xe, xa, atol=1e-13)


def test_mt19937_matches_numpy_legacy_stream(oracle):
    rs = np.random.RandomState(5489)
    raw = rs._bit_generator.random_raw(20000)
    assert raw[9999] == 4123659995  # the C++ standard's check value for std::mt19937
    for seed in (0, 7):
        raw = np.random.RandomState(seed)._bit_generator.random_raw(200).astype(np.float64)
        canon = (raw[0::2] + raw[1::2] * 4294967296.0) / 18446744073709551616.0
        assert np.array_equal(oracle.mt19937_uniform(seed, 100), 2.0 * canon - 1.0)


def test_spectral_radius(oracle):
    A = oracle.poisson7(8)
    S = A.to_scipy()
    DinvA = sp.diags(1.0 / S.diagonal()) @ S
    lam = np.linalg.eigvalsh(DinvA.toarray()).max()
    assert np.isclose(oracle.spectral_radius(A, True, 0), 2.0)  # Gershgorin: (6 + 6) / 6
    rho = oracle.spectral_radius(A, True, 100)
    assert 0.9 * lam < rho <= lam * (1 + 1e-12)  # power iteration approaches from below


def test_plain_aggregates_cover_and_connect(oracle):
    A = oracle.poisson7(7)
    cnt, ids = oracle.plain_aggregates(A, 0.0)
    assert ids.min() == 0 and ids.max() == cnt - 1
    assert np.array_equal(np.unique(ids), np.arange(cnt))
    S = A.to_scipy()
    # every aggregate is connected through strong (= any off-diagonal) links
    for a in range(0, cnt, 7):
        members = np.flatnonzero(ids == a)
        sub = S[members][:, members]
        ncomp, _ = sp.csgraph.connected_components(sub, directed=False)
        assert ncomp == 1
    # isolated rows are removed (id < 0)
    D = oracle.CSR.from_scipy(sp.identity(5, format="csr") * 2.0)
    with np.errstate(all="ignore"):
        cnt, ids = oracle.plain_aggregates(D, 0.0)
    assert cnt == 0 and (ids < 0).all()


def test_chebyshev_is_the_amgcl_polynomial(oracle):
    A = oracle.poisson7(6)
    S = A.to_scipy()
    dinv = 1.0 / S.diagonal()
    rhs = oracle.splitmix_vector(A.n, 5)
    x0 = oracle.splitmix_vector(A.n, 6)
    rho, hi_f, lo_f, deg = 1.9, 2.0, 1.0 / 120, 5
    hi = rho * hi_f
    lo = rho * lo_f
    d, c = 0.5 * (hi + lo), 0.5 * (hi - lo)
    x = x0.copy()
    p = np.zeros(A.n)
    alpha = 0.0
    for k in range(deg):
        r = dinv * (rhs - S @ x)
        if k == 0:
            alpha, beta = 1 / d, 0.0
        elif k == 1:
            alpha = 2 * d / (2 * d * d - c * c)
            beta = alpha * d - 1
        else:
            alpha = 1 / (d - 0.25 * alpha * c * c)
            beta = alpha * d - 1
        p = alpha * r + beta * p
        x = x + p
    assert np.allclose(oracle.chebyshev(A, rhs, x0, deg, rho, hi_f, lo_f), x, rtol=0, atol=1e-13)


def test_amg_hierarchy_is_galerkin(oracle):
    A = oracle.poisson7(10)
    amg = oracle.AMG(A, coarse_enough=40)
    assert amg.num_levels >= 3
    for l in range(amg.num_levels - 1):
        Al, P, R = (amg.level(l, w).to_scipy() for w in "APR")
        Ac = amg.level(l + 1).to_scipy()
        assert abs(R - P.T).max() == 0
        assert abs(Ac - R @ Al @ P).max() < 1e-12
        assert abs(Ac - Ac.T).max() < 1e-12
        sc = amg.level_scalars(l)
        assert 0 < sc["omega"] < 1
    assert amg.level(amg.num_levels - 1, "P") is None


def test_amg_vcycle_is_spd_and_contracts(oracle):
    A = oracle.poisson7(8)
    amg = oracle.AMG(A, coarse_enough=40, ncycle=1, cheb_degree=3)
    S = A.to_scipy()
    M = np.column_stack([amg.apply(e) for e in np.eye(A.n)])
    assert np.abs(M - M.T).max() < 1e-10  # symmetric preconditioner (pre/post smoothing identical, R = P^T)
    ev = np.linalg.eigvalsh(0.5 * (M + M.T))
    assert ev.min() > 0
    E = np.eye(A.n) - M @ S.toarray()
    assert np.abs(np.linalg.eigvals(E)).max() < 0.95  # stationary iteration x += M(b - Ax) converges


# ---- the reference tests' inequalities, on the oracle (tests/test_linear_solver.cpp) -------------
def _random_spd_like_pre_factor(S, rng):
    """pre_factor (test_linear_solver.cpp:241-307): same pattern, diag in [10,500], off-diag in [-5,-0.1]."""
    S = sp.triu(S, k=1).tocoo()
    off = -rng.uniform(0.1, 5, S.nnz)
    U = sp.coo_matrix((off, (S.row, S.col)), shape=S.shape)
    return (U + U.T + sp.diags(rng.uniform(0.1, 5, S.shape[0]) * 100)).tocsr()


def test_reference_all_and_pre_factor_inequalities(oracle):
    rng = np.random.default_rng(42)
    base = oracle.poisson7(6).to_scipy()
    for k in range(4):
        S = _random_spd_like_pre_factor(base, rng) if k else base
        A = oracle.CSR.from_scipy(S)
        b = rng.uniform(-1, 1, A.n)  # Eigen setRandom
        x, it, err = oracle.cg_eigen(A, b, tol=1e-10, max_iter=1000)
        assert np.linalg.norm(S @ x - b) < 1e-8  # :160-162, :299-301
        amg = oracle.AMG(A, coarse_enough=30)
        x, it, err = oracle.cg_amgcl(A, b, precond=amg)  # polysolve defaults: tol 1e-10, maxiter 1000
        assert np.linalg.norm(S @ x - b) < 1e-8


def test_reference_amgcl_initial_guess(oracle):
    """test_linear_solver.cpp:400-455: a second solve started from the converged x reports 0 iterations."""
    A = oracle.poisson7(7)
    amg = oracle.AMG(A, coarse_enough=30)
    b = np.random.default_rng(0).uniform(-1, 1, A.n)
    x, it, _ = oracle.cg_amgcl(A, b, precond=amg)
    assert it > 0
    x2, it2, _ = oracle.cg_amgcl(A, b, x0=x, precond=amg)
    assert it2 == 0
    assert np.linalg.norm(A.to_scipy() @ x2 - b) < 1e-8


def test_reference_gr_30_30(oracle):
    """test_linear_solver.cpp:541-602 (scalar leg): b = 1, ||Ax-b||/||b|| < 1e-7, iterations > 0."""
    G = oracle.gr_30_30()
    assert (G.n, G.nnz) == (900, 7744)
    b = np.ones(G.n)
    amg = oracle.AMG(G, coarse_enough=100)
    x, it, err = oracle.cg_amgcl(G, b, precond=amg)
    assert it > 0
    assert np.linalg.norm(G.to_scipy() @ x - b) / np.linalg.norm(b) < 1e-7


def test_elasticity_generator_is_block3_spd(oracle):
    E = oracle.elasticity_q1(4)
    S = E.to_scipy()
    assert E.n == 3 * 64 and abs(S - S.T).max() < 1e-15
    assert np.linalg.eigvalsh(S.toarray()).min() > 0
    # rigid translations are in the kernel of the un-clamped operator: free rows sum to ~0 per component
    free = np.flatnonzero(S.diagonal() != 1.0)
    far = [r for r in free if (r // 3) % 4 >= 2]  # nodes not adjacent to the clamped face
    for c in range(3):
        t = np.zeros(E.n)
        t[c::3] = 1.0
        assert np.abs((S @ t)[far]).max() < 1e-12


# ---- golden fixtures --------------------------------------------------------------------------------
def _fixtures(golden_dir):
    return sorted(f for f in glob.glob(os.path.join(golden_dir, "*.npz")) if not f.endswith("schwarz.npz"))


def test_golden_fixtures_present(golden_dir):
    assert len(_fixtures(golden_dir)) >= 6


@pytest.mark.parametrize("name", ["poisson7_n4", "poisson7_n8", "poisson7_n12", "poisson7_6x5x7", "gr_30_30",
                                  "elasticity_q1_m5"])
def test_oracle_reproduces_golden(oracle, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    A = oracle.CSR(int(g["n"]), g["rowptr"], g["col"], g["val"], int(g["n"]))
    b = g["b"]
    S = A.to_scipy()
    # (b) independent exact solution
    assert np.linalg.norm(S @ g["x_exact"] - b) / np.linalg.norm(b) < 1e-12
    x, it, err, hist = oracle.cg_eigen(A, b, tol=1e-8, max_iter=2000, history=True)
    assert it == int(g["cg_jacobi_iters"])
    assert np.allclose(hist, g["cg_jacobi_hist"], rtol=1e-9)
    assert np.allclose(x, g["cg_jacobi_x"], rtol=0, atol=1e-12)
    assert np.linalg.norm(x - g["x_exact"]) / np.linalg.norm(g["x_exact"]) < 1e-5
    _, it_n, _ = oracle.cg_eigen(A, b, precond="none", tol=1e-8, max_iter=2000)
    assert it_n == int(g["cg_none_iters"])
    params = json.loads(str(g["amg_params"]))
    amg = oracle.AMG(A, **params)
    assert amg.num_levels == int(g["amg_levels"])
    assert [amg.level(l).n for l in range(amg.num_levels)] == list(g["amg_level_rows"])
    assert [amg.level(l).nnz for l in range(amg.num_levels)] == list(g["amg_level_nnz"])
    assert np.allclose(amg.apply(b), g["amg_apply_b"], rtol=1e-10, atol=1e-13)
    xa, it_a, _ = oracle.cg_amgcl(A, b, precond=amg, tol=1e-10, max_iter=1000)
    assert it_a == int(g["cg_amg_iters"])
    assert np.linalg.norm(xa - g["x_exact"]) / np.linalg.norm(g["x_exact"]) < 1e-8


def test_schwarz_oracle_is_the_multilevel_additive_operator(oracle):
    """oracle.Schwarz against an independent dense construction: sum_l P_l blockdiag_64(P_l^T A P_l)^-1 P_l^T with
    piecewise-constant P_l (index >> 6l), and its effect in PCG (fewer iterations than Jacobi, same solution)."""
    for bs, A in ((1, oracle.poisson7(10, 9, 11)), (3, oracle.elasticity_q1(7))):
        _check_schwarz(oracle, A, bs)


def _check_schwarz(oracle, A, bs):
    n = A.n
    M = A.to_scipy().toarray()
    S = oracle.Schwarz(A, 3, block_size=bs)
    assert S.num_levels == 2  # 990 -> 16 / 1029 -> 18 unknowns: one domain covers level 1, nothing coarser is added
    idx = np.arange(n)
    ref = np.zeros((n, n))
    for l in range(S.num_levels):
        agg = ((idx // bs) >> (6 * l)) * bs + idx % bs if l else idx  # components kept apart on the coarse levels
        nl = agg.max() + 1
        P = np.zeros((n, nl))
        P[idx, agg] = 1
        Al = P.T @ M @ P
        Binv = np.zeros_like(Al)
        for b in range(0, nl, 64):
            sl = slice(b, min(b + 64, nl))
            Binv[sl, sl] = np.linalg.inv(Al[sl, sl])
        ref += P @ Binv @ P.T
    cols = list(range(0, n, 41))
    Z = np.column_stack([S.apply(np.eye(n)[:, i]) for i in cols])
    assert np.abs(Z - ref[:, cols]).max() < 1e-13
    b = oracle.spmv(A, oracle.splitmix_vector(n, 42))
    x, it, _ = oracle.cg_eigen(A, b, precond=S, tol=1e-10)
    xj, itj, _ = oracle.cg_eigen(A, b, tol=1e-10)
    assert it < itj and np.abs(x - xj).max() < 1e-7


@pytest.mark.parametrize("name", ["poisson7_n12", "gr_30_30", "elasticity_q1_m5"])
def test_oracle_reproduces_golden_schwarz(oracle, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    k = np.load(os.path.join(golden_dir, "schwarz.npz"))
    A = oracle.CSR(int(g["n"]), g["rowptr"], g["col"], g["val"], int(g["n"]))
    levels, bs = (int(v) for v in k[name + "_cfg"])
    S = oracle.Schwarz(A, levels, block_size=bs)
    assert S.num_levels == int(k[name + "_levels"])
    assert np.array_equal(S.apply(g["b"]), k[name + "_z"])
    _, it, _ = oracle.cg_eigen(A, g["b"], precond=S, tol=1e-8, max_iter=2000)
    assert it == int(k[name + "_iters"])


@pytest.mark.parametrize("bs", [1, 3])
def test_ordered_relaxations_of_the_oracle_against_dense_algebra(bs):
    """Round 6: the oracle's gauss_seidel and ilu0 (amg_oracle.c gs_sweep / ilu0_factor / ilu0_solve, restated from
    amgcl/relaxation/{gauss_seidel,ilu0}.hpp) checked against independent dense algebra: symmetric Gauss-Seidel as two (block)
    triangular solves, ILU(0) as the textbook IKJ elimination restricted to A's (block) pattern."""
    import scipy.sparse as sp
    import oracle
    A = oracle.poisson7(6, 5, 4) if bs == 1 else oracle.elasticity_q1(3)
    M = sp.csr_matrix(A.to_scipy())
    M.sort_indices()
    n = A.n
    Ad = M.toarray()
    r = np.random.default_rng(3).standard_normal(n)
    blk = np.arange(n) // bs
    lower = blk[:, None] > blk[None, :]
    upper = blk[:, None] < blk[None, :]
    diag = blk[:, None] == blk[None, :]
    # gauss_seidel as a preconditioner: x = 0, forward sweep, backward sweep
    z = oracle.AMG(A, relax_type="gauss_seidel", precond_class="relaxation", block_size=bs).apply(r)
    x1 = np.linalg.solve(Ad * (lower | diag), r)
    x2 = np.linalg.solve(Ad * (upper | diag), r - (Ad * lower) @ x1)
    assert np.linalg.norm(z - x2) <= 1e-12 * np.linalg.norm(x2)
    # ilu0: (block) IKJ elimination on the pattern
    nb = n // bs
    pat = np.zeros((nb, nb), bool)
    Mc = M.tocoo()
    pat[Mc.row // bs, Mc.col // bs] = True
    LU = Ad.copy()
    B = lambda i, j: (slice(i * bs, (i + 1) * bs), slice(j * bs, (j + 1) * bs))
    for i in range(nb):
        for k in range(i):
            if not pat[i, k]:
                continue
            LU[B(i, k)] = LU[B(i, k)] @ np.linalg.inv(LU[B(k, k)])
            for j in range(k + 1, nb):
                if pat[i, j] and pat[k, j]:
                    LU[B(i, j)] -= LU[B(i, k)] @ LU[B(k, j)]
    Lm = LU * lower + np.eye(n)
    Um = LU * (upper | diag)
    zz = np.linalg.solve(Um, np.linalg.solve(Lm, r))
    z = oracle.AMG(A, relax_type="ilu0", precond_class="relaxation", block_size=bs).apply(r)
    assert np.linalg.norm(z - zz) <= 1e-11 * np.linalg.norm(zz)
    # inside a hierarchy both make a convergent PCG, ilu0 with fewer iterations than gauss_seidel's sweeps
    b = oracle.spmv(A, oracle.splitmix_vector(n, 1))
    its = {}
    for rt in ("gauss_seidel", "ilu0"):
        ref = oracle.AMG(A, relax_type=rt, coarse_enough=40, ncycle=1, block_size=bs)
        assert ref.num_levels >= 2
        x, its[rt], err = oracle.cg_amgcl(A, b, precond=ref, tol=1e-10, max_iter=200)
        assert err <= 1e-10 and np.linalg.norm(M @ x - b) <= 1e-9 * np.linalg.norm(b)
    assert its["ilu0"] <= its["gauss_seidel"]
