"""world_size-N CPU rehearsal of the row-partitioned PCG (gloo).  Launched by tests/test_dist_gloo.py.

It drives the SAME host-side planning code the GPU path uses (psolve_hip_plan_halo, through
polysolve_amd.plan_halo) and re-enacts Context::setup_halo / exchange_halo / solve_device step by step
with gloo collectives in place of RCCL and numpy in place of the kernels, then checks the result
against the global oracle solve.  Test infrastructure: nothing here is on the product path.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
from polysolve_amd import plan_halo  # noqa: E402


def exchange(send_chunks, recv_sizes, dtype, rank, world):
    """grouped point-to-point: send_chunks[q] -> q, receive recv_sizes[q] from q"""
    reqs, out = [], [None] * world
    for q in range(world):
        if q == rank:
            continue
        if recv_sizes[q] > 0:
            out[q] = torch.empty(int(recv_sizes[q]), dtype=dtype)
            reqs.append(dist.irecv(out[q], src=q))
        if send_chunks[q] is not None and send_chunks[q].numel() > 0:
            reqs.append(dist.isend(send_chunks[q].contiguous(), dst=q))
    for r in reqs:
        r.wait()
    return out


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    nx, ny, nz = (int(v) for v in sys.argv[1:4])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # slab partition, uneven on purpose
    cuts = np.linspace(0, nz, world + 1).round().astype(int)
    cuts[1:-1] += (np.arange(1, world) % 2)
    cuts = np.clip(cuts, 0, nz)
    z0, z1 = int(cuts[rank]), int(cuts[rank + 1])
    plane = nx * ny
    A = O.poisson7(nx, ny, nz, z0, z1)
    row0, row1, n = z0 * plane, z1 * plane, A.n

    # 1. partition
    mine = torch.tensor([row0], dtype=torch.int64)
    allb = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allb, mine)
    row_offsets = np.array([int(t) for t in allb] + [plane * nz], dtype=np.int64)
    # 2. halo plan (product code)
    halo, recv_counts = plan_halo(rank, world, row_offsets, A.col)
    recv_off = np.concatenate([[0], np.cumsum(recv_counts)[:-1]])
    # 3. counts matrix
    allc = [torch.zeros(world, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allc, torch.from_numpy(recv_counts.copy()))
    send_counts = np.array([int(allc[q][rank]) for q in range(world)])
    # 4. request lists
    reqs_out = [torch.from_numpy(halo[recv_off[q]: recv_off[q] + recv_counts[q]].copy()) for q in range(world)]
    got = exchange(reqs_out, send_counts, torch.int32, rank, world)
    send_idx = [None if g is None else (g.numpy() - row0) for g in got]
    for s in send_idx:
        assert s is None or (s.min() >= 0 and s.max() < n)
    # 5. column remap
    col = A.col.astype(np.int64)
    local = (col >= row0) & (col < row1)
    col_l = np.where(local, col - row0, n + np.searchsorted(halo, col))
    Aloc = O.CSR(n, A.rowptr, col_l.astype(np.int32), A.val, n + halo.size)
    S = Aloc.to_scipy()

    def extend(v):
        chunks = [None if s is None else torch.from_numpy(v[s]) for s in send_idx]
        got = exchange(chunks, recv_counts, torch.float64, rank, world)
        ext = np.empty(n + halo.size)
        ext[:n] = v
        for q in range(world):
            if got[q] is not None:
                ext[n + recv_off[q]: n + recv_off[q] + recv_counts[q]] = got[q].numpy()
        return ext

    def gdot(a, b):
        t = torch.tensor([float(a @ b)], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t)

    xs = O.splitmix_vector(n, 42, start=row0)
    xs_ext = np.concatenate([xs, np.array([O.splitmix_vector(1, 42, start=int(g))[0] for g in halo])])
    b = S @ xs_ext
    # also check extend() against the analytically known halo values
    assert np.array_equal(extend(xs), xs_ext)

    # Eigen recurrence, distributed (solver.hip:solve_device with dist == true)
    dinv = 1.0 / 6.0
    tol, max_iter = 1e-8, 1000
    x = np.zeros(n)
    r = b - S @ extend(x)
    rhs2 = gdot(b, b)
    thr = max(tol * tol * rhs2, np.finfo(float).tiny)
    rn2 = gdot(r, r)
    p = dinv * r
    absnew = gdot(r, p)
    i = 0
    while i < max_iter and rn2 >= thr:
        q = S @ extend(p)
        alpha = absnew / gdot(p, q)
        x += alpha * p
        r -= alpha * q
        rn2 = gdot(r, r)
        if rn2 < thr:
            break
        z = dinv * r
        absold, absnew = absnew, gdot(r, z)
        p = z + (absnew / absold) * p
        i += 1

    # Chronopoulos-Gear single-reduction recurrences, distributed (solver.hip: Context::cg1_loop,
    # kernels.hip: cg1_update_kernel): ONE all-reduce of (r.u, r.r, w.u) per iteration
    def gsum3(a, b2, c):
        t = torch.tensor([a, b2, c], dtype=torch.float64)
        dist.all_reduce(t)
        return (float(v) for v in t)

    x1 = np.zeros(n)
    r1 = b - S @ extend(x1)
    u = dinv * r1
    w = S @ extend(u)
    gamma, rr, delta = gsum3(float(r1 @ u), float(r1 @ r1), float(w @ u))
    p1 = np.zeros(n)
    s1 = np.zeros(n)
    gamma_old = alpha1 = 1.0
    i1 = 0
    reductions = 1
    while i1 < max_iter and rr >= thr:
        if i1 == 0:
            beta, alpha1 = 0.0, gamma / delta
        else:
            beta = gamma / gamma_old
            alpha1 = gamma / (delta - beta * gamma / alpha1)
        p1 = u + beta * p1
        s1 = w + beta * s1
        x1 += alpha1 * p1
        r1 -= alpha1 * s1
        u = dinv * r1
        w = S @ extend(u)
        gamma_old = gamma
        gamma, rr, delta = gsum3(float(r1 @ u), float(r1 @ r1), float(w @ u))
        reductions += 1
        i1 += 1
    assert reductions == i1 + 1  # one collective per iteration (the two-reduction loop above needs 2 i + 2)
    assert abs(i1 - (i + 1)) <= 2, (i1, i)  # same Krylov iterates; Eigen's count excludes the converging pass
    assert np.abs(x1 - x).max() <= 1e-7 * max(np.abs(x).max(), 1.0)

    # additive Schwarz structure of precond = amg on shards: the preconditioner acts on the diagonal block only
    # (halo columns dropped), i.e. z = M_rank^-1 r_rank with no communication; here M_rank = exact block solve
    import scipy.sparse.linalg as spla
    Sdiag = S[:, :n].tocsc()
    lu = spla.splu(Sdiag)
    x2 = np.zeros(n)
    r2 = b - S @ extend(x2)
    z2 = lu.solve(r2)
    p2 = z2.copy()
    rz = gdot(r2, z2)
    i2 = 0
    while i2 < max_iter and gdot(r2, r2) >= thr:
        q2 = S @ extend(p2)
        a2 = rz / gdot(p2, q2)
        x2 += a2 * p2
        r2 -= a2 * q2
        z2 = lu.solve(r2)
        rz_old, rz = rz, gdot(r2, z2)
        p2 = z2 + (rz / rz_old) * p2
        i2 += 1
    assert i2 < i / 2, (i2, i)  # block solves beat Jacobi by far
    assert np.abs(x2 - x).max() <= 1e-6 * max(np.abs(x).max(), 1.0)

    sizes = [int(row_offsets[q + 1] - row_offsets[q]) for q in range(world)]
    xs_all = [torch.zeros(max(sizes), dtype=torch.float64) for q in range(world)]  # gloo wants equal sizes
    xpad = torch.zeros(max(sizes), dtype=torch.float64)
    xpad[:n] = torch.from_numpy(x)
    dist.all_gather(xs_all, xpad)
    xs_all = [t[:sz] for t, sz in zip(xs_all, sizes)]
    if rank == 0:
        Af = O.poisson7(nx, ny, nz)
        bf = O.spmv(Af, O.splitmix_vector(Af.n, 42))
        xo, ito, erro = O.cg_eigen(Af, bf, tol=tol, max_iter=max_iter)
        xg = np.concatenate([t.numpy() for t in xs_all])
        res = np.linalg.norm(bf - Af.to_scipy() @ xg) / np.linalg.norm(bf)
        print(f"DIST_OK world={world} iters={i} oracle_iters={ito} single_reduction_iters={i1} schwarz_iters={i2} "
              f"res={res:.3e} dx={np.abs(xg - xo).max():.3e} halo={halo.size}")
        assert abs(i - ito) <= 1
        assert res < 1.5e-8
        assert np.abs(xg - xo).max() < 1e-7
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
