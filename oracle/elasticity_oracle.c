/*
 * elasticity_oracle.c -- synthetic "PolyFEM-style" block-3 SPD stiffness matrix for the parity
 * tests (BASELINE.json configs[2], SURVEY.md 8(d) "Elasticity-Q1(M)").  TEST INFRASTRUCTURE ONLY.
 *
 * Trilinear (Q1) hexahedra on an M^3-node unit cube, isotropic linear elasticity (E, nu),
 * 2x2x2 Gauss quadrature, node-interleaved dofs (3*node + component).  The face x = 0 is
 * clamped the way the reference's dirichlet_solve does it (src/polysolve/linear/FEMSolver.cpp:
 * 136-161): entries whose row or column is a Dirichlet dof are dropped and a unit diagonal is
 * inserted, which keeps the matrix SPD.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int32_t idx_t;

static void element_stiffness(double h, double E, double nu, double Ke[24][24])
{
    const double lam = E * nu / ((1 + nu) * (1 - 2 * nu)), mu = E / (2 * (1 + nu));
    double D[6][6];
    memset(D, 0, sizeof(D));
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) D[a][b] = lam + (a == b ? 2 * mu : 0.0);
    for (int a = 3; a < 6; ++a) D[a][a] = mu;
    memset(Ke, 0, 24 * 24 * sizeof(double));
    const double g = 1.0 / sqrt(3.0);
    /* local node l: (l&1, (l>>1)&1, (l>>2)&1) in {0,1}^3, reference coords xi = 2*bit - 1 */
    for (int q = 0; q < 8; ++q) {
        double xi[3] = {(q & 1) ? g : -g, (q & 2) ? g : -g, (q & 4) ? g : -g};
        double dN[8][3];
        for (int l = 0; l < 8; ++l) {
            double s[3] = {(l & 1) ? 1.0 : -1.0, (l & 2) ? 1.0 : -1.0, (l & 4) ? 1.0 : -1.0};
            for (int d = 0; d < 3; ++d) {
                double v = 0.125 * s[d];
                for (int e = 0; e < 3; ++e)
                    if (e != d) v *= (1 + s[e] * xi[e]);
                dN[l][d] = v * (2.0 / h); /* d/dx = d/dxi * 2/h */
            }
        }
        double B[6][24];
        memset(B, 0, sizeof(B));
        for (int l = 0; l < 8; ++l) {
            B[0][3 * l + 0] = dN[l][0];
            B[1][3 * l + 1] = dN[l][1];
            B[2][3 * l + 2] = dN[l][2];
            B[3][3 * l + 0] = dN[l][1]; B[3][3 * l + 1] = dN[l][0]; /* xy */
            B[4][3 * l + 1] = dN[l][2]; B[4][3 * l + 2] = dN[l][1]; /* yz */
            B[5][3 * l + 0] = dN[l][2]; B[5][3 * l + 2] = dN[l][0]; /* xz */
        }
        double w = (h / 2) * (h / 2) * (h / 2); /* detJ, unit Gauss weights */
        double DB[6][24];
        for (int a = 0; a < 6; ++a)
            for (int c = 0; c < 24; ++c) {
                double s = 0;
                for (int b = 0; b < 6; ++b) s += D[a][b] * B[b][c];
                DB[a][c] = s;
            }
        for (int r = 0; r < 24; ++r)
            for (int c = 0; c < 24; ++c) {
                double s = 0;
                for (int a = 0; a < 6; ++a) s += B[a][r] * DB[a][c];
                Ke[r][c] += w * s;
            }
    }
}

/* Pass 1 (col == NULL): returns nnz.  Pass 2: fills rowptr[3*M^3+1], col, val. Columns sorted. */
int64_t orc_elasticity_q1(int M, double E, double nu, idx_t *rowptr, idx_t *col, double *val)
{
    double Ke[24][24];
    const double h = 1.0 / (M - 1);
    element_stiffness(h, E, nu, Ke);
    int64_t p = 0;
    if (rowptr) rowptr[0] = 0;
    for (int k = 0; k < M; ++k)
        for (int j = 0; j < M; ++j)
            for (int i = 0; i < M; ++i) {
                int64_t a = i + (int64_t)M * (j + (int64_t)M * k);
                double acc[27][3][3];
                char present[27];
                memset(acc, 0, sizeof(acc));
                memset(present, 0, sizeof(present));
                /* the (up to) 8 elements around node a: element origin (i-ei, j-ej, k-ek) */
                for (int e = 0; e < 8; ++e) {
                    int ei = e & 1, ej = (e >> 1) & 1, ek = (e >> 2) & 1;
                    int ox = i - ei, oy = j - ej, oz = k - ek;
                    if (ox < 0 || oy < 0 || oz < 0 || ox >= M - 1 || oy >= M - 1 || oz >= M - 1) continue;
                    int la = ei | (ej << 1) | (ek << 2);
                    for (int lb = 0; lb < 8; ++lb) {
                        int bi = ox + (lb & 1), bj = oy + ((lb >> 1) & 1), bk = oz + ((lb >> 2) & 1);
                        int slot = (bi - i + 1) + 3 * ((bj - j + 1) + 3 * (bk - k + 1));
                        present[slot] = 1;
                        for (int c = 0; c < 3; ++c)
                            for (int d = 0; d < 3; ++d) acc[slot][c][d] += Ke[3 * la + c][3 * lb + d];
                    }
                }
                for (int c = 0; c < 3; ++c) {
                    if (i == 0) { /* clamped dof: unit diagonal */
                        if (col) { col[p] = (idx_t)(3 * a + c); val[p] = 1.0; }
                        ++p;
                    } else {
                        for (int slot = 0; slot < 27; ++slot) {
                            if (!present[slot]) continue;
                            int bi = i + slot % 3 - 1, bj = j + (slot / 3) % 3 - 1, bk = k + slot / 9 - 1;
                            if (bi == 0) continue; /* Dirichlet column dropped */
                            int64_t b = bi + (int64_t)M * (bj + (int64_t)M * bk);
                            for (int d = 0; d < 3; ++d) {
                                if (col) { col[p] = (idx_t)(3 * b + d); val[p] = acc[slot][c][d]; }
                                ++p;
                            }
                        }
                    }
                    if (rowptr) rowptr[3 * a + c + 1] = (idx_t)p;
                }
            }
    return p;
}
