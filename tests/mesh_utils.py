"""Unstructured tetrahedral test meshes (test infrastructure; numpy / scipy only).

PolyFEM's workload is FEM on unstructured tetrahedral meshes; the synthetic inputs of SURVEY.md 8(d) are lattices.  This
module builds what a mesh generator would hand over: a Delaunay tetrahedralisation (scipy.spatial.Delaunay -- Qhull) of
a jittered point cloud in the unit cube, P1 stiffness matrices on it (Laplace; linear elasticity with 3 x 3 node blocks,
node-interleaved xyz like the reference's assembly), the hull nodes clamped by identity rows and columns exactly as
FEMSolver.cpp:136-161 eliminates Dirichlet nodes, and an optional random renumbering of the nodes."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
from scipy.spatial import Delaunay


def tet_mesh(m: int, jitter: float = 0.3, seed: int = 0):
    """(points [n, 3], tets [T, 4], boundary mask [n]): Delaunay of an m^3 lattice whose interior points are moved by up
    to `jitter` cell widths (no two tetrahedra alike, 14-16 neighbours per node, no slivers worth the name)."""
    rng = np.random.default_rng(seed)
    g = np.linspace(0.0, 1.0, m)
    P = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    ijk = np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij"), axis=-1).reshape(-1, 3)
    boundary = ((ijk == 0) | (ijk == m - 1)).any(axis=1)
    h = 1.0 / (m - 1)
    P = P + np.where(boundary[:, None], 0.0, jitter * h * rng.uniform(-1.0, 1.0, P.shape))
    T = Delaunay(P).simplices.astype(np.int64)
    E = P[T[:, 1:]] - P[T[:, :1]]
    vol = np.abs(np.linalg.det(E)) / 6.0
    T = T[vol > 1e-9 * h ** 3]  # (degenerate hull slivers of the flat faces)
    return P, T, boundary


def _gradients(P, T):
    E = P[T[:, 1:]] - P[T[:, :1]]                 # [T, 3, 3], rows = edge vectors
    vol = np.abs(np.linalg.det(E)) / 6.0
    Ginv = np.linalg.inv(E)                        # columns = gradients of lambda_1..3
    g = np.concatenate([-Ginv.sum(axis=2, keepdims=True), Ginv], axis=2).transpose(0, 2, 1)  # [T, 4, 3]
    return g, vol


def _clamp(K, fixed_dofs):
    """identity rows and columns for the fixed unknowns (FEMSolver.cpp:136-161), explicit zeros dropped"""
    n = K.shape[0]
    keep = np.ones(n)
    keep[fixed_dofs] = 0.0
    D = sp.diags(keep)
    K = (D @ K @ D + sp.diags(1.0 - keep)).tocsr()
    K.eliminate_zeros()
    K.sort_indices()
    return K


def _assemble_blocks(T, n_nodes, Ke):
    """sum of the element blocks Ke[t, i, j, ...] into one block per node pair: (indptr, indices, blocks), sorted columns"""
    tail = Ke.shape[3:]
    key = (T[:, :, None] * n_nodes + T[:, None, :]).ravel()
    order = np.argsort(key, kind="stable")
    key = key[order]
    vals = Ke.reshape((-1,) + tail)[order]
    first = np.flatnonzero(np.concatenate([[True], key[1:] != key[:-1]]))
    blocks = np.add.reduceat(vals, first, axis=0)
    ukey = key[first]
    rows, cols = ukey // n_nodes, ukey % n_nodes
    indptr = np.zeros(n_nodes + 1, np.int64)
    np.add.at(indptr, rows + 1, 1)
    return np.cumsum(indptr), cols, blocks


def p1_laplace(P, T, boundary):
    g, vol = _gradients(P, T)
    Ke = vol[:, None, None] * (g[:, :, None, :] * g[:, None, :, :]).sum(axis=3)  # [T, 4, 4]
    indptr, indices, vals = _assemble_blocks(T, len(P), Ke)
    K = sp.csr_matrix((vals, indices, indptr), shape=(len(P), len(P)))
    return _clamp(K, np.flatnonzero(boundary))


def p1_elasticity(P, T, boundary, E: float = 1.0, nu: float = 0.3):
    """node-interleaved (x, y, z per node): 3 x 3 blocks K_ij = vol (lam g_i g_j^T + mu g_j g_i^T + mu (g_i . g_j) I)"""
    lam, mu = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))
    g, vol = _gradients(P, T)
    gg = g[:, :, None, :, None] * g[:, None, :, None, :]                    # g_i[a] g_j[b]
    dot = (g[:, :, None, :] * g[:, None, :, :]).sum(axis=3)
    Ke = lam * gg
    Ke += mu * gg.transpose(0, 1, 2, 4, 3)
    Ke += (mu * dot)[..., None, None] * np.eye(3)
    Ke *= vol[:, None, None, None, None]
    indptr, indices, blocks = _assemble_blocks(T, len(P), Ke)
    n = 3 * len(P)
    K = sp.bsr_matrix((blocks, indices, indptr), shape=(n, n)).tocsr()
    fixed = (3 * np.flatnonzero(boundary)[:, None] + np.arange(3)[None, :]).ravel()
    return _clamp(K, fixed)


def renumber_nodes(K, block: int = 1, seed: int = 1):
    """the same matrix with its nodes in a random order (whole nodes move): returns (K', dof order)"""
    nb = K.shape[0] // block
    pn = np.random.default_rng(seed).permutation(nb)
    dof = (block * pn[:, None] + np.arange(block)[None, :]).ravel()
    Kp = K[dof][:, dof].tocsr()
    Kp.sort_indices()
    return Kp, dof
