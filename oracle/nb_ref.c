/*
 * ORACLE — TEST INFRASTRUCTURE ONLY. Never imported, linked or called by the product path
 * (molecular-dynamics_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it, as the checker / reported CPU baseline.
 *
 * PARITY UNPINNED BY THE REFERENCE: the pipeline delegates this arithmetic to GROMACS
 * (`gromacs>=2021`, /root/reference/envs/environment.yml:24), which is neither vendored nor
 * installed, and the pipeline holds no golden vectors. This file restates the published
 * algorithm that the .mdp contract switches on, in float64:
 *   - plain cut-off LJ (Lorentz-Berthelot, potential-shift) and Ewald real-space Coulomb
 *     (`coulombtype = PME`, `rcoulomb = rvdw = 1.0`, /root/reference/mdp/production.mdp:24-27;
 *     1.2 nm in /root/reference/mdp/pull_short.mdp:19-20), potential shifted at the cut-off;
 *   - excluded-pair Ewald correction -qq erf(beta r)/r and the Ewald self term;
 *   - triclinic PBC for the rhombic dodecahedron (/root/reference/config.yaml:16).
 * It is pinned instead by closed-form known answers in tests/ (two-particle energies, NaCl
 * Madelung constant, finite differences).
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define ONE_4PI_EPS0 138.935458
#define M_2_SQRTPI_ 1.12837916709551257390

static void wrap_brick(const double *box, double *p)
{
    /* box rows a=(ax,0,0) b=(bx,by,0) c=(cx,cy,cz): z by c, then y by b, then x by a */
    double s = floor(p[2] / box[8]);
    p[0] -= s * box[6]; p[1] -= s * box[7]; p[2] -= s * box[8];
    s = floor(p[1] / box[4]);
    p[0] -= s * box[3]; p[1] -= s * box[4];
    s = floor(p[0] / box[0]);
    p[0] -= s * box[0];
}

typedef struct { double v[3]; int half; } shift_t;

static int build_shifts(const double *box, shift_t *sh)
{
    int n = 0;
    for (int k = -1; k <= 1; k++)
        for (int j = -2; j <= 2; j++)
            for (int i = -3; i <= 3; i++) {
                shift_t s;
                for (int d = 0; d < 3; d++)
                    s.v[d] = i * box[0 + d] + j * box[3 + d] + k * box[6 + d];
                s.half = (k > 0) || (k == 0 && j > 0) || (k == 0 && j == 0 && i > 0);
                if (i == 0 && j == 0 && k == 0) s.half = 2; /* central image */
                sh[n++] = s;
            }
    return n;
}

static inline int is_excluded(const int *es, const int *el, int i, int j)
{
    for (int k = es[i]; k < es[i + 1]; k++)
        if (el[k] == j) return 1;
    return 0;
}

/* Minimum-image vector by exhaustive search over nearby lattice vectors. */
void orc_min_image(const double *box, const double *d_in, double *d_out)
{
    double best = 1e300;
    double t[3] = { d_in[0], d_in[1], d_in[2] };
    /* first bring into the brick-centred cell */
    double s = rint(t[2] / box[8]);
    t[0] -= s * box[6]; t[1] -= s * box[7]; t[2] -= s * box[8];
    s = rint(t[1] / box[4]);
    t[0] -= s * box[3]; t[1] -= s * box[4];
    s = rint(t[0] / box[0]);
    t[0] -= s * box[0];
    for (int k = -1; k <= 1; k++)
        for (int j = -2; j <= 2; j++)
            for (int i = -2; i <= 2; i++) {
                double c[3];
                for (int d = 0; d < 3; d++)
                    c[d] = t[d] + i * box[0 + d] + j * box[3 + d] + k * box[6 + d];
                double r2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
                if (r2 < best) { best = r2; d_out[0] = c[0]; d_out[1] = c[1]; d_out[2] = c[2]; }
            }
}

/*
 * f (n*3) is accumulated into. ener[0]=LJ (shifted), ener[1]=Coulomb real space (shifted,
 * non-excluded pairs), ener[2]=excluded-pair correction + self term. vir (9) accumulates
 * -1/2 sum d (x) F over pairs. flags bit0: pair terms, bit1: exclusion correction + self.
 * Returns the number of non-excluded pairs with r < rc (each counted once).
 */
long orc_nonbonded(int n, const double *x, const double *q, const double *sigma, const double *eps,
                   const double *box, double rc, double beta, const int *es, const int *el,
                   double *f, double *ener, double *vir, int flags, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    double *xw = (double *)malloc(sizeof(double) * 3 * (size_t)n);
    for (int i = 0; i < n; i++) {
        xw[3 * i] = x[3 * i]; xw[3 * i + 1] = x[3 * i + 1]; xw[3 * i + 2] = x[3 * i + 2];
        wrap_brick(box, xw + 3 * i);
    }
    int nc[3];
    double w[3];
    const double L[3] = { box[0], box[4], box[8] };
    for (int d = 0; d < 3; d++) {
        nc[d] = (int)floor(L[d] / rc);
        if (nc[d] < 1) nc[d] = 1;
        if (nc[d] > 64) nc[d] = 64;
        w[d] = L[d] / nc[d];
    }
    int ncell = nc[0] * nc[1] * nc[2];
    int *cstart = (int *)calloc((size_t)ncell + 1, sizeof(int));
    int *cidx = (int *)malloc(sizeof(int) * (size_t)n);
    int *cell_of = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; i++) {
        int c[3];
        for (int d = 0; d < 3; d++) {
            c[d] = (int)(xw[3 * i + d] / w[d]);
            if (c[d] >= nc[d]) c[d] = nc[d] - 1;
            if (c[d] < 0) c[d] = 0;
        }
        cell_of[i] = (c[2] * nc[1] + c[1]) * nc[0] + c[0];
        cstart[cell_of[i] + 1]++;
    }
    for (int c = 0; c < ncell; c++) cstart[c + 1] += cstart[c];
    int *fill = (int *)calloc((size_t)ncell, sizeof(int));
    for (int i = 0; i < n; i++) cidx[cstart[cell_of[i]] + fill[cell_of[i]]++] = i;
    free(fill);

    shift_t sh[7 * 5 * 3];
    int nsh = build_shifts(box, sh);
    const double rc2 = rc * rc;
    const double sh_coul = erfc(beta * rc) / rc;
    double *fbuf = (double *)calloc((size_t)nthreads * 3 * (size_t)n, sizeof(double));
    double e_lj = 0, e_c = 0, e_x = 0;
    double v[9] = { 0 };
    long npairs = 0;

    if (flags & 1) {
#pragma omp parallel num_threads(nthreads) reduction(+ : e_lj, e_c, npairs, v[:9])
        {
            double *ft = fbuf + (size_t)omp_get_thread_num() * 3 * (size_t)n;
#pragma omp for schedule(dynamic, 64)
            for (int i = 0; i < n; i++) {
                const double *xi = xw + 3 * i;
                for (int s = 0; s < nsh; s++) {
                    if (!sh[s].half) continue;
                    double c[3] = { xi[0] - sh[s].v[0], xi[1] - sh[s].v[1], xi[2] - sh[s].v[2] };
                    int lo[3], hi[3], empty = 0;
                    for (int d = 0; d < 3; d++) {
                        lo[d] = (int)floor((c[d] - rc) / w[d]);
                        hi[d] = (int)floor((c[d] + rc) / w[d]);
                        if (lo[d] < 0) lo[d] = 0;
                        if (hi[d] > nc[d] - 1) hi[d] = nc[d] - 1;
                        if (lo[d] > hi[d]) empty = 1;
                    }
                    if (empty) continue;
                    for (int cz = lo[2]; cz <= hi[2]; cz++)
                        for (int cy = lo[1]; cy <= hi[1]; cy++)
                            for (int cx = lo[0]; cx <= hi[0]; cx++) {
                                int cc = (cz * nc[1] + cy) * nc[0] + cx;
                                for (int k = cstart[cc]; k < cstart[cc + 1]; k++) {
                                    int j = cidx[k];
                                    if (sh[s].half == 2 && j <= i) continue;
                                    double dx = c[0] - xw[3 * j], dy = c[1] - xw[3 * j + 1], dz = c[2] - xw[3 * j + 2];
                                    double r2 = dx * dx + dy * dy + dz * dz;
                                    if (r2 >= rc2) continue;
                                    if (is_excluded(es, el, i, j)) continue;
                                    npairs++;
                                    double r = sqrt(r2), rinv = 1.0 / r, rinv2 = rinv * rinv;
                                    double sg = 0.5 * (sigma[i] + sigma[j]);
                                    double ep = sqrt(eps[i] * eps[j]);
                                    double fs = 0.0;
                                    if (ep > 0.0) {
                                        double s2 = sg * sg * rinv2, s6 = s2 * s2 * s2;
                                        double c2 = sg * sg / rc2, c6 = c2 * c2 * c2;
                                        e_lj += 4.0 * ep * ((s6 * s6 - s6) - (c6 * c6 - c6));
                                        fs += 4.0 * ep * (12.0 * s6 * s6 - 6.0 * s6) * rinv2;
                                    }
                                    double qq = ONE_4PI_EPS0 * q[i] * q[j];
                                    if (qq != 0.0) {
                                        double br = beta * r;
                                        double ec = erfc(br);
                                        e_c += qq * (ec * rinv - sh_coul);
                                        fs += qq * (ec * rinv + M_2_SQRTPI_ * beta * exp(-br * br)) * rinv2;
                                    }
                                    double fx = fs * dx, fy = fs * dy, fz = fs * dz;
                                    ft[3 * i] += fx; ft[3 * i + 1] += fy; ft[3 * i + 2] += fz;
                                    ft[3 * j] -= fx; ft[3 * j + 1] -= fy; ft[3 * j + 2] -= fz;
                                    v[0] -= 0.5 * dx * fx; v[1] -= 0.5 * dx * fy; v[2] -= 0.5 * dx * fz;
                                    v[3] -= 0.5 * dy * fx; v[4] -= 0.5 * dy * fy; v[5] -= 0.5 * dy * fz;
                                    v[6] -= 0.5 * dz * fx; v[7] -= 0.5 * dz * fy; v[8] -= 0.5 * dz * fz;
                                }
                            }
                }
            }
        }
    }
    if (flags & 2) {
        /* excluded-pair correction and self term (serial: O(N) pairs) */
        double *ft = fbuf;
        for (int i = 0; i < n; i++) {
            e_x -= ONE_4PI_EPS0 * beta / sqrt(M_PI) * q[i] * q[i];
            for (int k = es[i]; k < es[i + 1]; k++) {
                int j = el[k];
                if (j <= i) continue;
                double qq = ONE_4PI_EPS0 * q[i] * q[j];
                if (qq == 0.0) continue;
                double d0[3] = { x[3 * i] - x[3 * j], x[3 * i + 1] - x[3 * j + 1], x[3 * i + 2] - x[3 * j + 2] };
                double d[3];
                orc_min_image(box, d0, d);
                double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
                double r = sqrt(r2), br = beta * r;
                double er = erf(br);
                e_x -= qq * er / r;
                /* F_i = -dE/dr * d/r, E = -qq erf(br)/r */
                double fs = -qq * (er / r - M_2_SQRTPI_ * beta * exp(-br * br)) / r2;
                double fx = fs * d[0], fy = fs * d[1], fz = fs * d[2];
                ft[3 * i] += fx; ft[3 * i + 1] += fy; ft[3 * i + 2] += fz;
                ft[3 * j] -= fx; ft[3 * j + 1] -= fy; ft[3 * j + 2] -= fz;
                v[0] -= 0.5 * d[0] * fx; v[1] -= 0.5 * d[0] * fy; v[2] -= 0.5 * d[0] * fz;
                v[3] -= 0.5 * d[1] * fx; v[4] -= 0.5 * d[1] * fy; v[5] -= 0.5 * d[1] * fz;
                v[6] -= 0.5 * d[2] * fx; v[7] -= 0.5 * d[2] * fy; v[8] -= 0.5 * d[2] * fz;
            }
        }
    }
    for (int t = 0; t < nthreads; t++) {
        const double *ft = fbuf + (size_t)t * 3 * (size_t)n;
        for (size_t k = 0; k < 3 * (size_t)n; k++) f[k] += ft[k];
    }
    ener[0] += e_lj; ener[1] += e_c; ener[2] += e_x;
    for (int k = 0; k < 9; k++) vir[k] += v[k];
    free(fbuf); free(xw); free(cstart); free(cidx); free(cell_of);
    return npairs;
}

/* Cardinal B-spline weights of order 4 and their derivatives at fractional offset t in [0,1):
 * w[k] multiplies grid point (floor(u) + k - 3 + ... ) -- the caller fixes the convention. */
static void bspline4(double t, double *w, double *dw)
{
    /* M4(t+3-k)... written as the standard recursion */
    double w2[3], w3[3];
    w2[0] = 1.0 - t; w2[1] = t; w2[2] = 0;
    /* order 3 */
    w3[0] = 0.5 * (1.0 - t) * w2[0];
    w3[1] = 0.5 * ((t + 1.0) * w2[0] + (2.0 - t) * w2[1]);
    w3[2] = 0.5 * t * w2[1];
    /* derivative of order 4 from order 3 */
    dw[0] = -w3[0]; dw[1] = w3[0] - w3[1]; dw[2] = w3[1] - w3[2]; dw[3] = w3[2];
    /* order 4 */
    w[0] = (1.0 - t) * w3[0] / 3.0;
    w[1] = ((t + 2.0) * w3[0] + (2.0 - t) * w3[1]) / 3.0;
    w[2] = ((t + 1.0) * w3[1] + (3.0 - t) * w3[2]) / 3.0;
    w[3] = t * w3[2] / 3.0;
}

/* Spread charges on an (nx,ny,nz) grid (z fastest) with order-4 B-splines.
 * recip = inverse box (3x3 row-major) such that fractional s = x . recip.
 * Grid point indices for atom with u = s*K: (floor(u) + k) mod K, k = 0..3. */
void orc_pme_spread(int n, const double *x, const double *q, const double *recip,
                    int nx, int ny, int nz, double *grid)
{
    const int K[3] = { nx, ny, nz };
    for (int i = 0; i < n; i++) {
        double w[3][4], dw[3][4];
        int i0[3];
        for (int d = 0; d < 3; d++) {
            double s = x[3 * i] * recip[0 + d] + x[3 * i + 1] * recip[3 + d] + x[3 * i + 2] * recip[6 + d];
            s -= floor(s);
            double u = s * K[d];
            int fl = (int)floor(u);
            if (fl >= K[d]) fl = K[d] - 1;
            bspline4(u - fl, w[d], dw[d]);
            i0[d] = fl;
        }
        for (int a = 0; a < 4; a++) {
            int ia = (i0[0] + a) % nx;
            for (int b = 0; b < 4; b++) {
                int ib = (i0[1] + b) % ny;
                double qa = q[i] * w[0][a] * w[1][b];
                for (int c = 0; c < 4; c++) {
                    int ic = (i0[2] + c) % nz;
                    grid[((size_t)ia * ny + ib) * nz + ic] += qa * w[2][c];
                }
            }
        }
    }
}

/* Gather forces from the convolved potential grid phi: F_i = -q_i sum dtheta/dx phi. */
void orc_pme_gather(int n, const double *x, const double *q, const double *recip,
                    int nx, int ny, int nz, const double *phi, double *f)
{
    const int K[3] = { nx, ny, nz };
    for (int i = 0; i < n; i++) {
        double w[3][4], dw[3][4];
        int i0[3];
        for (int d = 0; d < 3; d++) {
            double s = x[3 * i] * recip[0 + d] + x[3 * i + 1] * recip[3 + d] + x[3 * i + 2] * recip[6 + d];
            s -= floor(s);
            double u = s * K[d];
            int fl = (int)floor(u);
            if (fl >= K[d]) fl = K[d] - 1;
            bspline4(u - fl, w[d], dw[d]);
            i0[d] = fl;
        }
        double g[3] = { 0, 0, 0 }; /* dE/du_d */
        for (int a = 0; a < 4; a++) {
            int ia = (i0[0] + a) % nx;
            for (int b = 0; b < 4; b++) {
                int ib = (i0[1] + b) % ny;
                for (int c = 0; c < 4; c++) {
                    int ic = (i0[2] + c) % nz;
                    double p = phi[((size_t)ia * ny + ib) * nz + ic];
                    g[0] += dw[0][a] * w[1][b] * w[2][c] * p;
                    g[1] += w[0][a] * dw[1][b] * w[2][c] * p;
                    g[2] += w[0][a] * w[1][b] * dw[2][c] * p;
                }
            }
        }
        /* du_d/dx_e = K_d * recip[e][d] */
        for (int e = 0; e < 3; e++) {
            double s = 0;
            for (int d = 0; d < 3; d++) s += g[d] * K[d] * recip[3 * e + d];
            f[3 * i + e] -= q[i] * s;
        }
    }
}
