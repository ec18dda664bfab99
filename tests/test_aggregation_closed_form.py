"""The theorem behind polysolve_amd/csrc/amg_aggregate.hip, checked on the CPU against the oracle's sequential
sweep (amgcl/coarsening/plain_aggregates.hpp restated in oracle/amg_oracle.c):

  * i is a seed  <=>  i has a strong connection and none of its PREDECESSORS j < i (vertices that reach i in one
    or two hops of the strength graph) is a seed;
  * a vertex claimed directly (a seed has it as a strong neighbour) belongs to the LAST such seed; otherwise a
    seed keeps itself and any other vertex belongs to the FIRST seed that reaches it in two hops;
  * aggregates are numbered in seed order, the ones emptied by later claims are dropped.

Symmetric patterns (the SPD case) and unsymmetric ones (where seeds can be stolen and aggregates can empty).
The device code evaluates exactly this by dependency rounds; tests/test_gpu_amg.py then compares the resulting
hierarchy with the host sweep's bit for bit."""
import numpy as np
import pytest
import scipy.sparse as sp


def closed_form_aggregates(M):
    M = sp.csr_matrix(M)
    n = M.shape[0]
    succ = [set() for _ in range(n)]  # strong neighbours: eps_strong = 0 -> every stored off-diagonal with v*v > 0
    for i in range(n):
        for j, v in zip(M.indices[M.indptr[i]:M.indptr[i + 1]], M.data[M.indptr[i]:M.indptr[i + 1]]):
            if j != i and v * v > 0:
                succ[i].add(int(j))
    removed = [len(s) == 0 for s in succ]
    pred1 = [set() for _ in range(n)]
    for i in range(n):
        for j in succ[i]:
            pred1[j].add(i)
    pred2 = [set() for _ in range(n)]  # i <- c <- j : j reaches i through c (c has a successor, so it is not removed)
    for i in range(n):
        for c in pred1[i]:
            pred2[i] |= pred1[c]
    seed = [False] * n
    for i in range(n):
        if removed[i]:
            continue
        seed[i] = not any(seed[j] for j in (pred1[i] | pred2[i]) if j < i)
    rank, k = {}, 0
    for i in range(n):
        if seed[i]:
            rank[i] = k
            k += 1
    ids = np.full(n, -2, np.int64)
    for v in range(n):
        if removed[v]:
            continue
        direct = [s for s in pred1[v] if seed[s]]
        if direct:
            ids[v] = rank[max(direct)]
        elif seed[v]:
            ids[v] = rank[v]
        else:
            ids[v] = rank[min(s for s in pred2[v] if seed[s] and s != v)]
    used = np.zeros(k, bool)
    used[ids[ids >= 0]] = True
    new = np.cumsum(used) - 1
    ids[ids >= 0] = new[ids[ids >= 0]]
    return int(used.sum()), ids


def _random_pattern(n, deg, seed, symmetric):
    rng = np.random.default_rng(seed)
    r = np.repeat(np.arange(n), deg)
    c = rng.integers(0, n, n * deg)
    v = rng.uniform(0.1, 1.0, n * deg)
    M = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsr()
    if symmetric:
        M = M + M.T
    M = sp.lil_matrix(M)
    for k in rng.integers(0, n, max(1, n // 20)):  # a few vertices without any connection (removed)
        M[k, :] = 0.0
        M[:, k] = 0.0
    M = (sp.csr_matrix(M) + sp.identity(n) * 10.0).tocsr()
    M.eliminate_zeros()
    M.sort_indices()
    return M


@pytest.mark.parametrize("symmetric", [True, False])
@pytest.mark.parametrize("n,deg", [(40, 1), (120, 2), (300, 3), (250, 6), (90, 12)])
def test_closed_form_equals_the_sequential_sweep(oracle, symmetric, n, deg):
    for seed in range(6):
        M = _random_pattern(n, deg, 1000 * n + 10 * deg + seed, symmetric)
        cnt, ids = oracle.plain_aggregates(oracle.CSR.from_scipy(M), 0.0)
        cnt2, ids2 = closed_form_aggregates(M)
        assert cnt2 == cnt, (symmetric, n, deg, seed)
        assert np.array_equal(ids2, np.asarray(ids, np.int64)), (symmetric, n, deg, seed)


def test_closed_form_on_grids(oracle):
    for shape in [(7, 7, 7), (13, 5, 4), (30, 1, 1), (1, 1, 50)]:
        A = oracle.poisson7(*shape)
        cnt, ids = oracle.plain_aggregates(A, 0.0)
        cnt2, ids2 = closed_form_aggregates(A.to_scipy())
        assert cnt2 == cnt and np.array_equal(ids2, np.asarray(ids, np.int64)), shape
