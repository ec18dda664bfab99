"""world_size-N CPU rehearsal (gloo) of what amg_dist.hip does on the shards of a row-partitioned matrix -- the coarse
level built shard by shard and its halo plan -- and of the row-partitioned PCG on a block-3 operator.  Launched by
tests/test_dist_gloo.py:  dist_worker_amg.py poisson nx ny nz | elasticity M

The PRODUCT code under test is the host-side planner psolve_hip_plan_halo (polysolve_amd.plan_halo) -- on level 0 AND on
level 1, whose column space is the aggregates' (a non-slab partition: the sizes come from an all-gather of the aggregate
counts) -- driven through the steps of DistAmg::setup (amg_dist.hpp): aggregates confined to the shard and numbered rank
after rank, P = (I - omega D^-1 A) P_tent with the GLOBAL Gershgorin omega, halo rows of P and of A P fetched from their
owners by a variable-length row exchange, R = the transpose of [P local ; P halo rows] restricted to the owned coarse nodes,
A_1 = R (A P).  gloo collectives stand in for RCCL, numpy / scipy for the kernels.  Checked against the same construction
done globally on rank 0's copy of the whole matrix.  Test infrastructure: nothing here is on the product path.
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
from polysolve_amd import plan_halo  # noqa: E402


def p2p(send, recv_sizes, dtype, rank, world):
    """grouped point-to-point: send[q] -> q (None / empty: nothing), receive recv_sizes[q] items from q"""
    reqs, out = [], [None] * world
    for q in range(world):
        if q == rank:
            continue
        if recv_sizes[q] > 0:
            out[q] = torch.empty(int(recv_sizes[q]), dtype=dtype)
            reqs.append(dist.irecv(out[q], src=q))
        if send[q] is not None and send[q].numel() > 0:
            reqs.append(dist.isend(send[q].contiguous(), dst=q))
    for r in reqs:
        r.wait()
    return out


def all_gather_i64(v, world):
    out = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(out, torch.tensor([int(v)], dtype=torch.int64))
    return np.array([int(t) for t in out], dtype=np.int64)


class Halo:
    """Context::setup_halo re-enacted for one column space: plan (product code), counts matrix, request lists."""

    def __init__(self, rank, world, offsets, cols_global):
        self.rank, self.world, self.offsets = rank, world, np.asarray(offsets, np.int64)
        self.row0, self.row1 = int(offsets[rank]), int(offsets[rank + 1])
        self.halo, self.recv_counts = plan_halo(rank, world, self.offsets, np.asarray(cols_global, np.int32))
        # (the plan lists exactly the off-rank columns, sorted, grouped by owner)
        off = np.unique(cols_global[(cols_global < self.row0) | (cols_global >= self.row1)])
        assert np.array_equal(self.halo, off)
        owner = np.searchsorted(self.offsets, self.halo, side="right") - 1
        assert np.array_equal(np.bincount(owner, minlength=world), self.recv_counts) and np.all(np.diff(owner) >= 0)
        self.recv_off = np.concatenate([[0], np.cumsum(self.recv_counts)[:-1]]).astype(np.int64)
        allc = [torch.zeros(world, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allc, torch.from_numpy(self.recv_counts.astype(np.int64).copy()))
        self.send_counts = np.array([int(allc[q][rank]) for q in range(world)])
        want = [torch.from_numpy(self.halo[self.recv_off[q]: self.recv_off[q] + self.recv_counts[q]].copy()) for q in range(world)]
        got = p2p(want, self.send_counts, torch.int32, rank, world)
        self.send_idx = [None if g is None else (g.numpy().astype(np.int64) - self.row0) for g in got]
        for s in self.send_idx:
            assert s is None or (s.min() >= 0 and s.max() < self.row1 - self.row0)

    def local_cols(self, cols_global):
        n = self.row1 - self.row0
        c = np.asarray(cols_global, np.int64)
        loc = (c >= self.row0) & (c < self.row1)
        return np.where(loc, c - self.row0, n + np.searchsorted(self.halo, c))

    def extend(self, v, dtype=torch.float64):
        """[own entries ; halo entries] of a distributed vector (Context::exchange_halo)"""
        n = self.row1 - self.row0
        chunks = [None if s is None else torch.from_numpy(np.ascontiguousarray(v[s])) for s in self.send_idx]
        got = p2p(chunks, self.recv_counts, dtype, self.rank, self.world)
        ext = np.empty(n + self.halo.size, dtype=v.dtype)
        ext[:n] = v
        for q in range(self.world):
            if got[q] is not None:
                ext[n + self.recv_off[q]: n + self.recv_off[q] + self.recv_counts[q]] = got[q].numpy()
        return ext

    def fetch_rows(self, M):
        """the rows of the row-distributed sparse matrix M (own rows, GLOBAL columns) at this rank's halo positions:
        a variable-length row exchange (lengths, then columns and values), as DistAmg fetches the halo rows of P and of A P"""
        M = sp.csr_matrix(M)
        lens = np.diff(M.indptr).astype(np.int64)
        hl = self.extend(lens, torch.int64)[self.row1 - self.row0:]
        send_c, send_v, recv_sizes = [], [], np.zeros(self.world, np.int64)
        for q in range(self.world):
            s = self.send_idx[q]
            if s is None:
                send_c.append(None)
                send_v.append(None)
            else:
                sub = M[s]
                send_c.append(torch.from_numpy(sub.indices.astype(np.int64)))
                send_v.append(torch.from_numpy(sub.data.astype(np.float64)))
            recv_sizes[q] = int(hl[self.recv_off[q]: self.recv_off[q] + self.recv_counts[q]].sum())
        gc = p2p(send_c, recv_sizes, torch.int64, self.rank, self.world)
        gv = p2p(send_v, recv_sizes, torch.float64, self.rank, self.world)
        cols = np.concatenate([g.numpy() for g in gc if g is not None] or [np.zeros(0, np.int64)])
        vals = np.concatenate([g.numpy() for g in gv if g is not None] or [np.zeros(0)])
        ptr = np.concatenate([[0], np.cumsum(hl)]).astype(np.int64)
        return sp.csr_matrix((vals, cols, ptr), shape=(self.halo.size, M.shape[1]))


def shard_aggregates(Adiag, bs):
    """AMGCL's sweep on the strong connections INSIDE the shard (eps_strong 0: every stored off-diagonal entry); block value
    types aggregate nodes (the block graph).  Returns the aggregate of every node, -1 for a node without neighbours (amgcl
    removes it: a Dirichlet row)."""
    C = sp.coo_matrix(Adiag)
    nb = Adiag.shape[0] // bs
    G = sp.csr_matrix((np.ones(C.nnz), (C.row // bs, C.col // bs)), shape=(nb, nb))
    G.sum_duplicates()
    G.sort_indices()
    G.data[:] = -1.0
    G = sp.csr_matrix(G - sp.diags(G.diagonal()) + sp.diags(np.full(nb, 100.0)))
    G.sort_indices()
    cnt, ids = O.plain_aggregates(O.CSR.from_scipy(G), 0.0)
    ids = np.where(ids >= 0, ids, -1).astype(np.int64)
    assert ids.max() == cnt - 1
    return int(cnt), ids


def smoothed_prolongation_rows(A_rows, row0, bs, agg_of_node, n_coarse, omega, dinv_rows):
    """the given rows of P = (I - omega D^-1 A) P_tent (A_rows: those rows of A with GLOBAL columns);
    P_tent(node i, aggregate(i)) = I_bs, a removed node has an empty row"""
    n_glob = A_rows.shape[1]
    rows = np.arange(n_glob)
    agg = agg_of_node[rows // bs]
    keep = agg >= 0
    Pt = sp.csr_matrix((np.ones(int(keep.sum())), (rows[keep], agg[keep] * bs + rows[keep] % bs)), shape=(n_glob, n_coarse))
    n = A_rows.shape[0]
    I = sp.csr_matrix((np.ones(n), (np.arange(n), row0 + np.arange(n))), shape=(n, n_glob))
    return sp.csr_matrix((I - omega * sp.diags(dinv_rows) @ A_rows) @ Pt)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    kind = sys.argv[1]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if kind == "poisson":
        nx, ny, nz = (int(v) for v in sys.argv[2:5])
        Af = O.poisson7(nx, ny, nz)
        bs, plane, nplanes = 1, nx * ny, nz
    else:
        M = int(sys.argv[2])
        Af = O.elasticity_q1(M)
        bs, plane, nplanes = 3, 3 * M * M, M
    G = Af.to_scipy().tocsr()
    G.sort_indices()
    N = G.shape[0]
    cuts = np.linspace(0, nplanes, world + 1).round().astype(int)
    cuts[1:-1] += (np.arange(1, world) % 2)  # uneven on purpose
    row0, row1 = int(cuts[rank]) * plane, int(cuts[rank + 1]) * plane
    n = row1 - row0
    A = sp.csr_matrix(G[row0:row1])  # this rank's rows, GLOBAL columns (what the caller hands to factorize on a shard)
    offsets0 = np.concatenate([all_gather_i64(row0, world), [N]])

    # ---- level 0: halo plan (product code), distributed block-3 / scalar Jacobi-PCG against the oracle ------------------
    H0 = Halo(rank, world, offsets0, A.indices)
    Aloc = sp.csr_matrix((A.data, H0.local_cols(A.indices), A.indptr), shape=(n, n + H0.halo.size))

    def gsum(*vals):
        t = torch.tensor(list(vals), dtype=torch.float64)
        dist.all_reduce(t)
        return [float(v) for v in t]

    xs = O.splitmix_vector(n, 42, start=row0)
    b = Aloc @ H0.extend(xs)
    dinv = 1.0 / A[:, row0:row1].diagonal()
    tol, max_iter = 1e-8, 2000
    x = np.zeros(n)
    r = b - Aloc @ H0.extend(x)
    (rhs2, rn2) = gsum(float(b @ b), float(r @ r))
    thr = max(tol * tol * rhs2, np.finfo(float).tiny)
    p = dinv * r
    (absnew,) = gsum(float(r @ p))
    it = 0
    while it < max_iter and rn2 >= thr:
        q = Aloc @ H0.extend(p)
        (pq,) = gsum(float(p @ q))
        alpha = absnew / pq
        x += alpha * p
        r -= alpha * q
        z = dinv * r
        absold = absnew
        rn2, absnew = gsum(float(r @ r), float(r @ z))
        if rn2 < thr:
            break
        p = z + (absnew / absold) * p
        it += 1

    # ---- level 1, shard by shard (DistAmg::setup) ------------------------------------------------------------------------
    cnt, ids_local = shard_aggregates(sp.csr_matrix(A[:, row0:row1]), bs)   # aggregates confined to the shard
    counts = all_gather_i64(cnt, world)
    agg0 = int(counts[:rank].sum())                                        # ... numbered rank after rank
    offsets1 = np.concatenate([[0], np.cumsum(counts)]) * bs
    nc = int(offsets1[-1])
    ids_glob_numbering = np.where(ids_local >= 0, ids_local + agg0, -1)
    # the aggregate of every node this rank reads: own nodes + halo nodes (an int exchange over the level-0 plan)
    node_ids_ext = H0.extend(np.repeat(ids_glob_numbering, bs).astype(np.int64), torch.int64)
    agg_for_P = np.full(N // bs, -1, np.int64)  # (nodes this rank never reads stay -1: no row of A here has a column there)
    agg_for_P[np.arange(row0, row1) // bs] = node_ids_ext[:n]
    agg_for_P[H0.halo // bs] = node_ids_ext[n:]
    # global Gershgorin bound of D^-1 A (an all-reduce max), omega as smoothed_aggregation takes it
    g_loc = float((np.abs(sp.diags(dinv) @ A).sum(axis=1)).max())
    t = torch.tensor([g_loc], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    omega = (4.0 / 3.0) / float(t)
    P = smoothed_prolongation_rows(A, row0, bs, agg_for_P, nc, omega, dinv)            # own rows of P, global coarse columns
    Phalo = H0.fetch_rows(P)                                              # rows of P at the halo positions
    Pext = sp.vstack([P, Phalo]).tocsr()                                  # rows in the order of Aloc's columns
    AP = sp.csr_matrix(Aloc @ Pext)                                       # own rows of A P
    APhalo = H0.fetch_rows(AP)
    APext = sp.vstack([AP, APhalo]).tocsr()
    c0, c1 = int(offsets1[rank]), int(offsets1[rank + 1])
    R = sp.csr_matrix(Pext.T)[c0:c1]                                       # R = P^T restricted to the coarse nodes owned here
    A1 = sp.csr_matrix(R @ APext)                                          # own rows of A_1 = R (A P), global coarse columns
    A1.sum_duplicates()
    A1.sort_indices()
    # rows of P this rank holds only touch the coarse nodes of ranks it borders
    H1 = Halo(rank, world, offsets1, A1.indices)                           # level-1 plan: product code on a non-slab partition
    A1loc = sp.csr_matrix((A1.data, H1.local_cols(A1.indices), A1.indptr), shape=(c1 - c0, c1 - c0 + H1.halo.size))
    v1 = O.splitmix_vector(c1 - c0, 7, start=c0)
    y1 = A1loc @ H1.extend(v1)                                             # a level-1 product with its halo exchange
    # restriction of a fine vector and prolongation back, as the cycle does them (R: own coarse rows over [own ; halo] fine)
    fine = O.splitmix_vector(n, 9, start=row0)
    coarse = R @ H0.extend(fine)
    # ... and P's own column halo: a fine row interpolates from aggregates of the neighbouring rank
    HP = Halo(rank, world, offsets1, P.indices)
    Ploc = sp.csr_matrix((P.data, HP.local_cols(P.indices), P.indptr), shape=(n, c1 - c0 + HP.halo.size))
    back = Ploc @ HP.extend(coarse)

    # ---- gather everything on rank 0 and compare with the same construction done globally ------------------------------
    def gather_vec(v, sizes):
        out = [torch.zeros(int(max(sizes)), dtype=torch.float64) for _ in range(world)]
        pad = torch.zeros(int(max(sizes)), dtype=torch.float64)
        pad[: v.size] = torch.from_numpy(np.ascontiguousarray(v, np.float64))
        dist.all_gather(out, pad)
        return np.concatenate([o.numpy()[: int(s)] for o, s in zip(out, sizes)])

    sizes0 = np.diff(offsets0)
    sizes1 = np.diff(offsets1)
    xg = gather_vec(x, sizes0)
    idg = gather_vec(np.repeat(ids_glob_numbering, bs).astype(np.float64), sizes0).astype(np.int64)
    backg = gather_vec(back, sizes0)
    y1g = gather_vec(y1, sizes1)
    cg = gather_vec(coarse, sizes1)
    nnz1 = all_gather_i64(A1.nnz, world)
    if rank == 0:
        bf = G @ O.splitmix_vector(N, 42)
        xo, ito, _ = O.cg_eigen(Af, bf, tol=tol, max_iter=max_iter)
        assert abs(it - ito) <= max(1, ito // 100), (it, ito)
        assert np.linalg.norm(bf - G @ xg) / np.linalg.norm(bf) < 1.5e-8 and np.abs(xg - xo).max() < 1e-6 * np.abs(xo).max()
        # the global construction with the SAME aggregates
        dg = 1.0 / G.diagonal()
        om = (4.0 / 3.0) / float(np.abs(sp.diags(dg) @ G).sum(axis=1).max())
        assert abs(om - omega) <= 1e-15 * om
        Pg = smoothed_prolongation_rows(G, 0, bs, idg[::bs], nc, om, dg)
        A1g = sp.csr_matrix(Pg.T @ (G @ Pg))
        A1g.sum_duplicates()
        v1g = O.splitmix_vector(nc, 7)
        ref = A1g @ v1g
        assert np.abs(y1g - ref).max() <= 1e-12 * np.abs(ref).max(), np.abs(y1g - ref).max()
        cref = Pg.T @ O.splitmix_vector(N, 9)
        assert np.abs(cg - cref).max() <= 1e-12 * max(np.abs(cref).max(), 1.0)
        assert np.abs(backg - Pg @ cref).max() <= 1e-12 * max(np.abs(cref).max(), 1.0)
        # the coarse operator is symmetric and keeps the constant (rigid translation) in its near-kernel like the fine one
        assert abs(A1g - A1g.T).max() <= 1e-12 * abs(A1g).max()
        print(f"DIST_AMG_OK world={world} kind={kind} bs={bs} rows0={N} pcg_iters={it} oracle_iters={ito} aggregates={nc // bs} "
              f"level1_rows={nc} level1_nnz={int(nnz1.sum())} halo0={H0.halo.size} halo1={H1.halo.size} haloP={HP.halo.size}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
