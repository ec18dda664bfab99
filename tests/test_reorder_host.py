"""The oracle's statement of the backend's optional renumbering (oracle/reorder_oracle.c) pinned by an independent
implementation: scipy's breadth_first_order walks a component from a start vertex and appends the unvisited
neighbours of a dequeued vertex in row order -- the same definition -- and scipy's reverse Cuthill-McKee gives the
bandwidth a good ordering of the same graph reaches."""
import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import breadth_first_order, reverse_cuthill_mckee


def _bandwidth(M):
    c = M.tocoo()
    return int(np.abs(c.row - c.col).max())


def test_order_is_breadth_first_from_the_min_degree_vertex(oracle):
    A = oracle.poisson7(13, 9, 7)
    rng = np.random.default_rng(0)
    B = oracle.permuted(A, rng.permutation(A.n).astype(np.int32))
    order, info = oracle.cuthill_mckee(B)
    assert np.array_equal(np.sort(order), np.arange(B.n)) and info["components"] == 1 and info["isolated"] == 0
    deg = np.diff(B.rowptr)
    start = int(np.flatnonzero(deg == deg.min())[0])
    assert order[0] == start
    bfs = breadth_first_order(B.to_scipy(), start, directed=False, return_predecessors=False)
    assert np.array_equal(order, bfs)
    # levels = eccentricity of the start vertex + 1: a corner of the grid
    assert info["levels"] == (13 - 1) + (9 - 1) + (7 - 1) + 1
    C = oracle.permuted(B, order)
    rcm = reverse_cuthill_mckee(B.to_scipy(), symmetric_mode=True).astype(np.int32)
    assert _bandwidth(C.to_scipy()) <= 1.1 * _bandwidth(oracle.permuted(B, rcm).to_scipy())
    assert _bandwidth(C.to_scipy()) < 0.2 * _bandwidth(B.to_scipy())


def test_isolated_rows_components_and_leftover(oracle):
    blocks = [oracle.poisson7(4, 3, 2).to_scipy(), sp.identity(3, format="csr"), oracle.poisson7(5, 1, 1).to_scipy()]
    M = sp.block_diag(blocks, format="csr")
    A = oracle.CSR.from_scipy(M)
    order, info = oracle.cuthill_mckee(A)
    assert info == {"levels": info["levels"], "components": 2, "isolated": 3, "leftover": 0}
    assert list(order[:3]) == [24, 25, 26]          # the identity rows first, ascending
    assert order[3] == 27                            # then the component of the fewest-entries vertex (the chain's end)
    assert set(order[3:8]) == set(range(27, 32)) and set(order[8:]) == set(range(24))
    order1, info1 = oracle.cuthill_mckee(A, max_components=1)
    assert info1["components"] == 1 and info1["leftover"] == 24 and list(order1[8:]) == list(range(24))
    assert np.array_equal(order1[:8], order[:8])


def test_oracle_reproduces_the_golden_order(oracle, golden_dir):
    """tests/golden/reorder_tets.npz (an unstructured P1 Laplace system; made by tests/golden/make_golden.py, where the
    order was also checked against scipy's breadth_first_order): the oracle's order, level count and iteration counts."""
    import os
    g = np.load(os.path.join(golden_dir, "reorder_tets.npz"))
    A = oracle.CSR(int(g["n"]), g["rowptr"], g["col"], g["val"], int(g["n"]))
    order, info = oracle.cuthill_mckee(A)
    assert np.array_equal(order, g["order"]) and info["levels"] == int(g["levels"]) and info["isolated"] == int(g["isolated"])
    B = oracle.permuted(A, order)
    x, it, _ = oracle.cg_eigen(B, g["b"][order], tol=1e-9, max_iter=2000)
    assert it == int(g["cg_jacobi_iters"])
    xo = np.empty(A.n)
    xo[order] = x
    assert np.abs(xo - g["x_exact"]).max() <= 1e-7 * np.abs(g["x_exact"]).max()
