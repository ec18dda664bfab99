"""CPU test of the product's host-side AMG setup (polysolve_amd/csrc/amg_setup.cpp) against the oracle
(oracle/amg_oracle.c): same aggregates => same level sizes, same P and Galerkin operators.
The product sorts the columns of every row; AMGCL/the oracle keep first-touch order, so matrices are
compared as scipy matrices (order-free)."""
import numpy as np
import pytest
import scipy.sparse as sp


def _mat(lvl):
    nr, nc, ptr, col, val, _ = lvl
    return sp.csr_matrix((val, col, ptr), shape=(nr, nc))


@pytest.mark.parametrize("case", ["poisson10", "poisson_ragged", "gr3030", "elasticity"])
def test_host_hierarchy_matches_oracle(oracle, case):
    from polysolve_amd import HostHierarchy
    A, ce = {
        "poisson10": (oracle.poisson7(10), 40),
        "poisson_ragged": (oracle.poisson7(13, 7, 9), 30),
        "gr3030": (oracle.gr_30_30(), 100),
        "elasticity": (oracle.elasticity_q1(5), 60),
    }[case]
    ref = oracle.AMG(A, coarse_enough=ce)
    H = HostHierarchy(A.n, A.rowptr, A.col, A.val, coarse_enough=ce)
    assert H.num_levels == ref.num_levels
    for l in range(H.num_levels):
        Ap = _mat(H.level(l, "A"))
        Ao = ref.level(l, "A").to_scipy()
        assert Ap.shape == Ao.shape
        assert abs(Ap - Ao).max() <= 1e-13 * abs(Ao).max()
        # sorted columns in the product's rows
        nr, nc, ptr, col, val, om = H.level(l, "A")
        for r in range(0, nr, max(1, nr // 50)):
            assert np.all(np.diff(col[ptr[r]:ptr[r + 1]]) > 0)
        if l + 1 < H.num_levels:
            Pp, Po = _mat(H.level(l, "P")), ref.level(l, "P").to_scipy()
            assert abs(Pp - Po).max() <= 1e-14
            Rp = _mat(H.level(l, "R"))
            assert abs(Rp - Pp.T).max() == 0
            assert np.isclose(H.level(l, "P")[5], ref.level_scalars(l)["omega"], rtol=1e-15)
        else:
            assert H.level(l, "P") is None


def test_host_hierarchy_limits(oracle):
    from polysolve_amd import HostHierarchy
    A = oracle.poisson7(12)
    assert HostHierarchy(A.n, A.rowptr, A.col, A.val, coarse_enough=5000).num_levels == 1
    H = HostHierarchy(A.n, A.rowptr, A.col, A.val, coarse_enough=10, max_levels=2)
    assert H.num_levels == 2 and H.level(1, "P") is None
    # a diagonal matrix has no strong connections: empty level -> stops, smoother only
    D = oracle.CSR.from_scipy(sp.identity(50, format="csr") * 3.0)
    assert HostHierarchy(D.n, D.rowptr, D.col, D.val, coarse_enough=10).num_levels == 1


@pytest.mark.parametrize("M,ce", [(5, 60), (8, 200)])
def test_host_block3_hierarchy_matches_oracle(oracle, M, ce):
    """AMGCL_Block<3> coarsening (AMGCL.cpp:243-302): aggregation on the block graph, block-smoothed P."""
    from polysolve_amd import HostHierarchy
    A = oracle.elasticity_q1(M)
    ref = oracle.AMG(A, coarse_enough=ce, block_size=3)
    H = HostHierarchy(A.n, A.rowptr, A.col, A.val, coarse_enough=ce, block_size=3)
    assert H.num_levels == ref.num_levels >= 2
    for l in range(H.num_levels):
        Ap, Ao = _mat(H.level(l, "A")), ref.level(l, "A").to_scipy()
        assert Ap.shape == Ao.shape and Ap.shape[0] % 3 == 0
        assert abs(Ap - Ao).max() <= 1e-12 * abs(Ao).max()
        if l + 1 < H.num_levels:
            Pp, Po = _mat(H.level(l, "P")), ref.level(l, "P").to_scipy()
            assert Pp.nnz == Po.nnz  # full 3x3 blocks on both sides
            assert abs(Pp - Po).max() <= 1e-13
            assert np.isclose(H.level(l, "P")[5], ref.level_scalars(l)["omega"], rtol=1e-14)
    # block coarsening keeps far more coarse dofs than the scalar one on the same matrix
    Hs = HostHierarchy(A.n, A.rowptr, A.col, A.val, coarse_enough=ce, block_size=1)
    assert H.level(1, "A")[0] > Hs.level(1, "A")[0]


def _anisotropic(oracle, N=9, eps=0.02):
    """7-point operator with a weak z-coupling: with eps_strong > 0 the z links are filtered out."""
    A = oracle.poisson7(N)
    S = A.to_scipy().tocoo()
    plane = N * N
    w = np.where(np.abs(S.row - S.col) == plane, eps, 1.0)
    data = S.data * w
    M = sp.coo_matrix((data, (S.row, S.col)), shape=S.shape).tocsr()
    M = M - sp.diags(M.diagonal()) + sp.diags(-(M - sp.diags(M.diagonal())).sum(axis=1).A1 + 0.5)
    M.sort_indices()
    return oracle.CSR.from_scipy(M.tocsr())


@pytest.mark.parametrize("eps_strong", [0.08, 0.25])
def test_host_hierarchy_with_strength_filter(oracle, eps_strong):
    """eps_strong > 0 (amgcl/coarsening/plain_aggregates.hpp: a_ij^2 > eps^2 |a_ii a_jj|, eps halved per
    level): weak links are left out of the aggregates and folded into the filtered diagonal of P."""
    from polysolve_amd import HostHierarchy
    A = _anisotropic(oracle)
    ref = oracle.AMG(A, coarse_enough=30, eps_strong=eps_strong)
    H = HostHierarchy(A.n, A.rowptr, A.col, A.val, coarse_enough=30, eps_strong=eps_strong)
    assert H.num_levels == ref.num_levels >= 2
    ref0 = oracle.AMG(A, coarse_enough=30, eps_strong=0.0)
    assert ref.level(1).n != ref0.level(1).n  # the filter really changes the aggregates
    for l in range(H.num_levels):
        Ap, Ao = _mat(H.level(l, "A")), ref.level(l, "A").to_scipy()
        assert Ap.shape == Ao.shape
        assert abs(Ap - Ao).max() <= 1e-12 * abs(Ao).max()
        if l + 1 < H.num_levels:
            assert abs(_mat(H.level(l, "P")) - ref.level(l, "P").to_scipy()).max() <= 1e-13


@pytest.mark.parametrize("case,bs", [("poisson", 1), ("gr3030", 1), ("elasticity", 3), ("tets", 1)])
@pytest.mark.parametrize("mode", ["parallel", "aggregation", "parallel+aggregation", "compact", "compact+aggregation"])
def test_host_hierarchy_round5_modes_match_oracle(oracle, case, bs, mode):
    """Round 5: amg.aggregation = "parallel" (this repository's hashed-priority distance-2 independent set, restated in
    oracle/amg_oracle.c: parallel_aggregates_graph -- integer work, identical aggregates) and amg.coarsening = "aggregation"
    (amgcl/coarsening/aggregation.hpp: P = tentative prolongation, Galerkin operator scaled by 1 / over_interp): the
    product's host construction against the oracle's, level by level."""
    from polysolve_amd import HostHierarchy
    if case == "tets":
        import os
        d = np.load(os.path.join(os.path.dirname(__file__), "golden", "reorder_tets.npz"))
        A = oracle.CSR(int(d["n"]), d["rowptr"].astype(np.int32), d["col"].astype(np.int32), d["val"].astype(np.float64))
        ce = 80
    else:
        A, ce = {"poisson": (oracle.poisson7(12, 9, 11), 40), "gr3030": (oracle.gr_30_30(), 60),
                 "elasticity": (oracle.elasticity_q1(6), 60)}[case]
    # (round 6: "compact" -- one-hop aggregates around two generations of such sets, oracle: compact_aggregates_graph)
    kw = dict(aggregation="parallel" if "parallel" in mode else ("compact" if "compact" in mode else "amgcl"),
              coarsening="aggregation" if mode.endswith("aggregation") else "smoothed_aggregation")
    ref = oracle.AMG(A, coarse_enough=ce, block_size=bs, **kw)
    H = HostHierarchy(A.n, A.rowptr, A.col, A.val, coarse_enough=ce, block_size=bs, **kw)
    assert H.num_levels == ref.num_levels and H.num_levels >= 2
    for l in range(H.num_levels):
        Ap, Ao = _mat(H.level(l, "A")), ref.level(l, "A").to_scipy()
        assert Ap.shape == Ao.shape
        assert abs(Ap - Ao).max() <= 1e-13 * abs(Ao).max()
        if l + 1 < H.num_levels:
            Pp, Po = _mat(H.level(l, "P")), ref.level(l, "P").to_scipy()
            assert Pp.shape == Po.shape and abs(Pp - Po).max() <= 1e-14
            if kw["coarsening"] == "aggregation":
                assert set(np.unique(Pp.data)) <= {0.0, 1.0}


@pytest.mark.parametrize("case", ["poisson7", "nodes27", "gr3030"])
def test_compact_aggregates_are_one_hop_balls_of_the_sweeps_size(oracle, case):
    """What "compact" promises (round 6), checked on the graph itself: every vertex is assigned; the aggregates whose seed
    came out of an independent set are their seed's whole one-hop ball among the vertices still free at that time, so every
    aggregate is connected; and the packing is the sweep's -- the aggregate count within 25 % of plain_aggregates' on a 7-point
    grid and on the 27-point node graph of Q1 elasticity, where "parallel" (two-hop aggregates of a random packing) has 0.6 x."""
    if case == "poisson7":
        A = oracle.poisson7(16, 13, 11)
    elif case == "gr3030":
        A = oracle.gr_30_30()
    else:  # the node graph of Q1 elasticity on a 14^3 grid: 27-point
        E = oracle.elasticity_q1(14).to_scipy().tocoo()
        G = sp.csr_matrix((np.ones(E.nnz), (E.row // 3, E.col // 3)), shape=(14 ** 3, 14 ** 3))
        G.sum_duplicates()
        G.data[:] = -1.0
        G = sp.csr_matrix(G - sp.diags(G.diagonal()) + sp.diags(np.full(14 ** 3, 50.0)))
        G.sort_indices()
        A = oracle.CSR.from_scipy(G)
    M = A.to_scipy()
    cnt, ids, rounds = oracle.compact_aggregates(A)
    live = np.diff(M.indptr) > 1
    assert rounds <= 24 and cnt > 0 and ids[live].min() >= 0 and ids.max() == cnt - 1 and np.all(ids[~live] == -2)
    c0, _ = oracle.plain_aggregates(A)
    c1, _, _ = oracle.parallel_aggregates(A)
    assert 0.8 * c0 <= cnt <= 1.25 * c0, (cnt, c0, c1)
    if case == "nodes27":
        assert c1 < 0.75 * c0  # (what "compact" is for)
    # every aggregate is connected (a one-hop ball, a one-hop ball among leftovers, or vertices hanging on to one)
    G = (abs(M) > 0).astype(np.int32).tocsr()
    from scipy.sparse.csgraph import connected_components
    for a in range(0, cnt, max(1, cnt // 40)):
        members = np.flatnonzero(ids == a)
        ncomp, _ = connected_components(G[members][:, members], directed=False)
        assert ncomp == 1, (a, members)
    sizes = np.bincount(ids[ids >= 0], minlength=cnt)
    deg = int(np.diff(M.indptr).max())
    assert sizes.min() >= 1 and sizes.max() <= 2 * deg  # a ball of at most deg vertices plus what hangs on to it


def test_parallel_aggregates_are_a_distance_two_maximal_independent_set(oracle):
    """What the parallel aggregation promises, checked on the graph itself: no two seeds within two hops of each other, every
    vertex within two hops of a seed, every aggregate connected to its seed, a dozen rounds at most."""
    A = oracle.poisson7(14, 11, 9)
    M = A.to_scipy()
    cnt, ids, rounds = oracle.parallel_aggregates(A)
    assert rounds <= 12 and cnt > 0 and ids.min() >= 0 and ids.max() == cnt - 1
    G = (abs(M) > 0).astype(np.int32)
    G.setdiag(0)
    G.eliminate_zeros()
    G2 = ((G @ G + G) > 0).astype(np.int32)
    # the seed of an aggregate: its member all of whose neighbours belong to it
    sizes = np.bincount(ids, minlength=cnt)
    assert sizes.min() >= 1 and 6 <= A.n / cnt <= 16
    c0, ids0 = oracle.plain_aggregates(A)
    assert 0.6 * c0 <= cnt <= 1.1 * c0  # random packing is looser than the lexicographic sweep's lattice, not by much
