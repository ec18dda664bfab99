"""CPU lab (oracle only): AMG-PCG iteration counts of Q1 elasticity under node numberings -- VERDICT r3 item 6."""
import os, sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle as O
M = int(os.environ.get("M", "20"))
REC = dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_higher=1.1, cheb_power_iters=20, sa_relax=1.3, block_size=3)
A = O.elasticity_q1(M)
nn = A.n // 3
S = A.to_scipy()
def node_graph(S):
    B = sp.bsr_matrix(S, blocksize=(3, 3))
    G = sp.csr_matrix((np.ones(len(B.indices)), B.indices, B.indptr), shape=(nn, nn))
    return O.CSR.from_scipy(G)
def expand(order):
    return (3 * order[:, None] + np.arange(3)[None, :]).reshape(-1).astype(np.int32)
def iters(order_nodes, tag):
    Ap = O.permuted(A, expand(order_nodes)) if order_nodes is not None else A
    xs = O.splitmix_vector(A.n, 42); b = O.spmv(Ap, xs)
    t = time.time()
    amg = O.AMG(Ap, **REC)
    x, it, err = O.cg_amgcl(Ap, b, precond=amg, tol=1e-8)
    l1 = amg.level(1, "A")
    print(f"{tag:34s} its={it:3d} levels={amg.num_levels} n1={l1.n//3 if l1 else 0:6d} ({time.time()-t:.1f}s)", flush=True)
    return it
rng = np.random.default_rng(1)
iters(None, "grid")
shuf = rng.permutation(nn).astype(np.int32)
iters(shuf, "random")
As = O.permuted(A, expand(shuf))
Gs = node_graph(As.to_scipy())
cm, info = O.cuthill_mckee(Gs)
iters(shuf[cm], f"random+CM ({info['levels']} levels)")
iters(shuf[cm[::-1]], "random+RCM")
if os.environ.get("SHORT"): sys.exit(0)
# candidate: CM, then cluster by the aggregates of a first sweep in CM order
Gc = O.permuted(Gs, cm)
cnt, ids = O.plain_aggregates(O.CSR(Gc.n, Gc.rowptr, Gc.col, np.where(Gc.col == np.repeat(np.arange(Gc.n), np.diff(Gc.rowptr)), 30.0, -1.0)))
o2 = np.argsort(ids, kind="stable").astype(np.int32)
iters(shuf[cm[o2]], f"random+CM+cluster ({cnt} aggs)")
# candidate: nested BFS (levels -> BFS inside each level's induced subgraph -> again)
import scipy.sparse.csgraph as cg
def nested(G, verts, depth):
    """order the vertex set `verts` of graph G (scipy csr): BFS levels from a min-degree vertex; inside a level recurse"""
    if len(verts) <= 2 or depth == 0:
        return list(verts)
    sub = G[verts][:, verts].tocsr()
    out = []
    left = np.ones(len(verts), bool)
    deg = np.diff(sub.indptr)
    while left.any():
        cand = np.flatnonzero(left)
        s = cand[np.argmin(deg[cand])]
        dist = cg.breadth_first_order(sub, s, directed=False, return_predecessors=False)
        d = cg.shortest_path(sub, method="D", unweighted=True, indices=s)
        reach = np.flatnonzero(np.isfinite(d) & left)
        for lev in range(int(d[reach].max()) + 1):
            vs = reach[d[reach] == lev]
            out += nested(G, verts[vs], depth - 1) if len(vs) > 2 else list(verts[vs])
        left[reach] = False
    return out
Gsp = Gs.to_scipy()
for depth in (1, 2, 3):
    o = np.array(nested(Gsp, np.arange(nn), depth), np.int32)
    iters(shuf[o], f"random+nested BFS depth {depth}")
