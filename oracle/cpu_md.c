/*
 * CPU BASELINE — TEST / BENCH INFRASTRUCTURE ONLY. Never imported, linked or called by the product path
 * (molecular-dynamics_amd/); bench.py's cpu_baseline leg times it and tests/ check it against the float64 oracle.
 *
 * "Own CPU restatement — not GROMACS": the per-step work that `gmx mdrun -nb cpu -pme cpu -update cpu -ntomp N` does
 * for the pipeline's stages (/root/reference/Snakefile:232-233 runs it with `-ntmpi 1 -ntomp 4`), written the way a
 * plain OpenMP C code would do it in single precision: a buffered Verlet pair list rebuilt every `nstlist` steps
 * (`cutoff-scheme = Verlet`, /root/reference/mdp/production.mdp:21-23), LJ + Ewald real space over that list
 * (production.mdp:24-27), order-4 B-spline charge spreading and force gathering for PME (production.mdp:28-29; the 3D
 * transforms and the influence function stay with numpy's pocketfft in oracle/cpu_md.py), SETTLE for rigid water and
 * the leap-frog update. It has no SIMD-specific kernels and no cluster pair list, so it is slower than GROMACS' own
 * CPU path on the same cores; it is a reported baseline, not a target.
 *
 * Build: make -C oracle   (gcc -O3 -march=x86-64-v3 -ffast-math -fopenmp -shared)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define ONE_4PI_EPS0 138.935458f
#define TWO_SQRTPI 1.1283791670955126f

typedef struct {
    int n;
    int *start;        /* n + 1 */
    int *nbr;          /* neighbour j | shift index << 24 */
    size_t cap;
    float shifts[128][3];
    int nshift;
} pairlist_t;

static void wrap_brick(const float *box, float *p)
{
    float s = floorf(p[2] / box[8]);
    p[0] -= s * box[6]; p[1] -= s * box[7]; p[2] -= s * box[8];
    s = floorf(p[1] / box[4]);
    p[0] -= s * box[3]; p[1] -= s * box[4];
    s = floorf(p[0] / box[0]);
    p[0] -= s * box[0];
}

void *cmd_list_new(void)
{
    pairlist_t *L = (pairlist_t *)calloc(1, sizeof(pairlist_t));
    return L;
}

void cmd_list_free(void *h)
{
    pairlist_t *L = (pairlist_t *)h;
    if (!L) return;
    free(L->start); free(L->nbr); free(L);
}

static inline int excluded(const int *es, const int *el, int i, int j)
{
    for (int k = es[i]; k < es[i + 1]; k++) if (el[k] == j) return 1;
    return 0;
}

/*
 * Half pair list within rlist. xw (n*3, output) receives the positions wrapped into the brick cell that the list's
 * shift vectors refer to; shift_of (n*3 ints, output) the lattice shift applied to each atom, so that later steps
 * can re-wrap consistently. Returns the number of listed pairs.
 */
long cmd_list_build(void *h, int n, const float *x, const float *box, float rlist, const int *es, const int *el,
                    float *xw, int *shift_of, int nthreads)
{
    pairlist_t *L = (pairlist_t *)h;
    if (nthreads < 1) nthreads = 1;
    for (int i = 0; i < n; i++) {
        float p[3] = { x[3 * i], x[3 * i + 1], x[3 * i + 2] };
        wrap_brick(box, p);
        /* integer lattice shift that was applied: solve p = x + a A + b B + c C */
        const float dc = (p[2] - x[3 * i + 2]) / box[8];
        const float c = rintf(dc);
        const float db = (p[1] - x[3 * i + 1] - c * box[7]) / box[4];
        const float b = rintf(db);
        const float da = (p[0] - x[3 * i] - c * box[6] - b * box[3]) / box[0];
        shift_of[3 * i] = (int)rintf(da); shift_of[3 * i + 1] = (int)b; shift_of[3 * i + 2] = (int)c;
        xw[3 * i] = p[0]; xw[3 * i + 1] = p[1]; xw[3 * i + 2] = p[2];
    }
    /* lattice shifts, half of them (the central image lists j > i) */
    L->nshift = 0;
    int central = -1;
    for (int k = 0; k <= 1; k++)
        for (int j = -2; j <= 2; j++)
            for (int i = -3; i <= 3; i++) {
                const int half = (k > 0) || (k == 0 && j > 0) || (k == 0 && j == 0 && i >= 0);
                if (!half) continue;
                const float sx = i * box[0] + j * box[3] + k * box[6], sy = j * box[4] + k * box[7], sz = k * box[8];
                const float gx = fmaxf(0.0f, fabsf(sx) - box[0]), gy = fmaxf(0.0f, fabsf(sy) - box[4]), gz = fmaxf(0.0f, fabsf(sz) - box[8]);
                if (gx * gx + gy * gy + gz * gz >= rlist * rlist) continue;
                if (i == 0 && j == 0 && k == 0) central = L->nshift;
                L->shifts[L->nshift][0] = sx; L->shifts[L->nshift][1] = sy; L->shifts[L->nshift][2] = sz;
                L->nshift++;
            }
    int nc[3]; float w[3];
    const float Ld[3] = { box[0], box[4], box[8] };
    for (int d = 0; d < 3; d++) {
        nc[d] = (int)floorf(Ld[d] / (0.5f * rlist));
        if (nc[d] < 1) nc[d] = 1;
        if (nc[d] > 128) nc[d] = 128;
        w[d] = Ld[d] / nc[d];
    }
    const int ncell = nc[0] * nc[1] * nc[2];
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
    /* two passes: count, then fill */
    free(L->start);
    L->start = (int *)calloc((size_t)n + 1, sizeof(int));
    L->n = n;
    const float rl2 = rlist * rlist;
    for (int pass = 0; pass < 2; pass++) {
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 256)
        for (int i = 0; i < n; i++) {
            const float *xi = xw + 3 * i;
            int cnt = 0;
            int *out = pass ? L->nbr + L->start[i] : NULL;
            for (int s = 0; s < L->nshift; s++) {
                const float c[3] = { xi[0] - L->shifts[s][0], xi[1] - L->shifts[s][1], xi[2] - L->shifts[s][2] };
                int lo[3], hi[3], empty = 0;
                for (int d = 0; d < 3; d++) {
                    lo[d] = (int)floorf((c[d] - rlist) / w[d]);
                    hi[d] = (int)floorf((c[d] + rlist) / w[d]);
                    if (lo[d] < 0) lo[d] = 0;
                    if (hi[d] > nc[d] - 1) hi[d] = nc[d] - 1;
                    if (lo[d] > hi[d]) empty = 1;
                }
                if (empty) continue;
                for (int cz = lo[2]; cz <= hi[2]; cz++)
                    for (int cy = lo[1]; cy <= hi[1]; cy++)
                        for (int cx = lo[0]; cx <= hi[0]; cx++) {
                            const int cc = (cz * nc[1] + cy) * nc[0] + cx;
                            for (int k = cstart[cc]; k < cstart[cc + 1]; k++) {
                                const int j = cidx[k];
                                if (s == central && j <= i) continue;
                                const float dx = c[0] - xw[3 * j], dy = c[1] - xw[3 * j + 1], dz = c[2] - xw[3 * j + 2];
                                if (dx * dx + dy * dy + dz * dz >= rl2) continue;
                                if (excluded(es, el, i, j)) continue;
                                if (out) out[cnt] = j | (s << 24);
                                cnt++;
                            }
                        }
            }
            if (!pass) L->start[i + 1] = cnt;
        }
        if (!pass) {
            for (int i = 0; i < n; i++) L->start[i + 1] += L->start[i];
            const size_t need = (size_t)L->start[n] + 16;
            if (need > L->cap) { free(L->nbr); L->nbr = (int *)malloc(sizeof(int) * need); L->cap = need; }
        }
    }
    free(cstart); free(cidx); free(cell_of);
    return L->start[n];
}

/* Positions of a later step brought into the frame of the list: x + the lattice shift stored at list time. */
void cmd_rewrap(int n, const float *x, const float *box, const int *shift_of, float *xw)
{
    for (int i = 0; i < n; i++) {
        const int a = shift_of[3 * i], b = shift_of[3 * i + 1], c = shift_of[3 * i + 2];
        xw[3 * i] = x[3 * i] + a * box[0] + b * box[3] + c * box[6];
        xw[3 * i + 1] = x[3 * i + 1] + b * box[4] + c * box[7];
        xw[3 * i + 2] = x[3 * i + 2] + c * box[8];
    }
}

/* erfc(z) = exp(-z^2) t P(t), t = 1 / (1 + 0.41 z): single-precision fit, relative error 3.5e-7 on 0 <= z <= 4.4 */
static inline float erfcx_poly(float z)
{
    const float t = 1.0f / (1.0f + 0.41f * z);
    float p = 7.959749251e-02f;
    p = p * t - 3.893341305e-01f; p = p * t + 6.172475282e-01f; p = p * t - 3.034404266e-01f;
    p = p * t + 3.679784521e-01f; p = p * t + 1.573482378e-01f; p = p * t + 2.398621772e-01f; p = p * t + 2.307406958e-01f;
    return p * t;
}

/*
 * LJ + Ewald real-space forces over the pair list. hs = sigma/2, se = 2 sqrt(eps) per atom (Lorentz-Berthelot).
 * f (n*3) is overwritten with the sum over threads. ener (2: LJ, Coulomb; both potential-shifted) only when
 * want_energy. Returns pairs within rc.
 */
long cmd_nonbonded(void *h, const float *xw, const float *q, const float *hs, const float *se, float rc, float beta,
                   float *f, double *ener, int want_energy, int nthreads, float *scratch /* nthreads * n * 3 */)
{
    const pairlist_t *L = (const pairlist_t *)h;
    const int n = L->n;
    if (nthreads < 1) nthreads = 1;
    const float rc2 = rc * rc;
    const float shc = erfcf(beta * rc) / rc;
    double e_lj = 0.0, e_c = 0.0;
    long npairs = 0;
    memset(scratch, 0, sizeof(float) * 3 * (size_t)n * nthreads);
#pragma omp parallel num_threads(nthreads) reduction(+ : e_lj, e_c, npairs)
    {
        float *ft = scratch + (size_t)omp_get_thread_num() * 3 * (size_t)n;
        float tfx[256], tfy[256], tfz[256];
#pragma omp for schedule(dynamic, 128)
        for (int i = 0; i < n; i++) {
            const float xi = xw[3 * i], yi = xw[3 * i + 1], zi = xw[3 * i + 2];
            const float qi = ONE_4PI_EPS0 * q[i], hsi = hs[i], sei = se[i];
            float fix = 0.0f, fiy = 0.0f, fiz = 0.0f;
            for (int k0 = L->start[i]; k0 < L->start[i + 1]; k0 += 256) {
                const int m = (L->start[i + 1] - k0 < 256) ? L->start[i + 1] - k0 : 256;
                const int *nb = L->nbr + k0;
                /* pass 1: the arithmetic, free of write conflicts */
#pragma omp simd reduction(+ : fix, fiy, fiz, e_lj, e_c, npairs)
                for (int k = 0; k < m; k++) {
                    const int j = nb[k] & 0xFFFFFF, s = nb[k] >> 24;
                    const float dx = xi - L->shifts[s][0] - xw[3 * j], dy = yi - L->shifts[s][1] - xw[3 * j + 1],
                                dz = zi - L->shifts[s][2] - xw[3 * j + 2];
                    const float r2 = dx * dx + dy * dy + dz * dz;
                    const int in = r2 < rc2;
                    const float r2s = in ? r2 : rc2;
                    const float rinv = 1.0f / sqrtf(r2s), rinv2 = rinv * rinv;
                    const float sg = hsi + hs[j], e4 = sei * se[j];
                    const float s2 = sg * sg * rinv2, s6 = s2 * s2 * s2, t6 = e4 * s6;
                    const float qq = qi * q[j];
                    const float br = beta * r2s * rinv;
                    const float ex = expf(-br * br);
                    const float ec = ex * erfcx_poly(br);
                    float fs = (t6 * (12.0f * s6 - 6.0f) + qq * rinv * (ec + TWO_SQRTPI * br * ex)) * rinv2;
                    fs = in ? fs : 0.0f;
                    const float fx = fs * dx, fy = fs * dy, fz = fs * dz;
                    fix += fx; fiy += fy; fiz += fz;
                    tfx[k] = fx; tfy[k] = fy; tfz[k] = fz;
                    npairs += in;
                    if (want_energy) {
                        const float c2 = sg * sg / rc2, c6 = c2 * c2 * c2;
                        e_lj += in ? (double)(t6 * (s6 - 1.0f) - e4 * c6 * (c6 - 1.0f)) : 0.0;
                        e_c += in ? (double)(qq * (ec * rinv - shc)) : 0.0;
                    }
                }
                /* pass 2: reaction forces */
                for (int k = 0; k < m; k++) {
                    const int j = nb[k] & 0xFFFFFF;
                    ft[3 * j] -= tfx[k]; ft[3 * j + 1] -= tfy[k]; ft[3 * j + 2] -= tfz[k];
                }
            }
            ft[3 * i] += fix; ft[3 * i + 1] += fiy; ft[3 * i + 2] += fiz;
        }
    }
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int k = 0; k < 3 * n; k++) {
        float a = 0.0f;
        for (int t = 0; t < nthreads; t++) a += scratch[(size_t)t * 3 * (size_t)n + k];
        f[k] = a;
    }
    if (ener) { ener[0] = e_lj; ener[1] = e_c; }
    return npairs;
}

/* Ewald correction of excluded pairs (-qq erf(beta r)/r, molecules are whole: no PBC) added to f; returns the energy
 * including the self term. */
double cmd_excl_correction(int n, const float *x, const float *q, float beta, const int *es, const int *el, float *f)
{
    double e = 0.0;
    for (int i = 0; i < n; i++) {
        e -= (double)ONE_4PI_EPS0 * beta / sqrt(M_PI) * q[i] * q[i];
        for (int k = es[i]; k < es[i + 1]; k++) {
            const int j = el[k];
            if (j <= i) continue;
            const float qq = ONE_4PI_EPS0 * q[i] * q[j];
            if (qq == 0.0f) continue;
            const float dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
            const float r2 = dx * dx + dy * dy + dz * dz, r = sqrtf(r2), br = beta * r;
            const float er = erff(br);
            e -= qq * er / r;
            const float fs = -qq * (er / r - TWO_SQRTPI * beta * expf(-br * br)) / r2;
            f[3 * i] += fs * dx; f[3 * i + 1] += fs * dy; f[3 * i + 2] += fs * dz;
            f[3 * j] -= fs * dx; f[3 * j + 1] -= fs * dy; f[3 * j + 2] -= fs * dz;
        }
    }
    return e;
}

static inline void bspline4(float t, float *w, float *dw)
{
    const float w20 = 1.0f - t, w21 = t;
    const float w30 = 0.5f * (1.0f - t) * w20, w31 = 0.5f * ((t + 1.0f) * w20 + (2.0f - t) * w21), w32 = 0.5f * t * w21;
    dw[0] = -w30; dw[1] = w30 - w31; dw[2] = w31 - w32; dw[3] = w32;
    w[0] = (1.0f - t) * w30 / 3.0f;
    w[1] = ((t + 2.0f) * w30 + (2.0f - t) * w31) / 3.0f;
    w[2] = ((t + 1.0f) * w31 + (3.0f - t) * w32) / 3.0f;
    w[3] = t * w32 / 3.0f;
}

static inline void frac_index(const float *x, const float *recip, const int *K, int *i0, float w[3][4], float dw[3][4])
{
    for (int d = 0; d < 3; d++) {
        float s = x[0] * recip[0 + d] + x[1] * recip[3 + d] + x[2] * recip[6 + d];
        s -= floorf(s);
        const float u = s * K[d];
        int fl = (int)floorf(u);
        if (fl >= K[d]) fl = K[d] - 1;
        bspline4(u - fl, w[d], dw[d]);
        i0[d] = fl;
    }
}

/* Charge spreading onto grid[(ix * ny + iy) * nz + iz] (overwritten). Threads own slabs of ix: an atom is handled by
 * every thread whose slab its four x-planes touch. */
void cmd_pme_spread(int n, const float *x, const float *q, const float *recip, int nx, int ny, int nz, float *grid, int nthreads)
{
    const int K[3] = { nx, ny, nz };
    if (nthreads < 1) nthreads = 1;
    memset(grid, 0, sizeof(float) * (size_t)nx * ny * nz);
#pragma omp parallel num_threads(nthreads)
    {
        const int t = omp_get_thread_num(), T = omp_get_num_threads();
        const int x0 = (int)((long)nx * t / T), x1 = (int)((long)nx * (t + 1) / T);
        for (int i = 0; i < n; i++) {
            if (q[i] == 0.0f) continue;
            /* cheap test on the x index before the splines */
            float s = x[3 * i] * recip[0] + x[3 * i + 1] * recip[3] + x[3 * i + 2] * recip[6];
            s -= floorf(s);
            int fl = (int)floorf(s * nx);
            if (fl >= nx) fl = nx - 1;
            int touches = 0;
            for (int a = 0; a < 4; a++) { const int ia = (fl + a) % nx; if (ia >= x0 && ia < x1) touches = 1; }
            if (!touches) continue;
            float w[3][4], dw[3][4]; int i0[3];
            frac_index(x + 3 * i, recip, K, i0, w, dw);
            for (int a = 0; a < 4; a++) {
                const int ia = (i0[0] + a) % nx;
                if (ia < x0 || ia >= x1) continue;
                for (int b = 0; b < 4; b++) {
                    const int ib = (i0[1] + b) % ny;
                    const float qa = q[i] * w[0][a] * w[1][b];
                    float *row = grid + ((size_t)ia * ny + ib) * nz;
                    for (int c = 0; c < 4; c++) row[(i0[2] + c) % nz] += qa * w[2][c];
                }
            }
        }
    }
}

void cmd_pme_gather(int n, const float *x, const float *q, const float *recip, int nx, int ny, int nz, const float *phi, float *f,
                    int nthreads)
{
    const int K[3] = { nx, ny, nz };
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int i = 0; i < n; i++) {
        if (q[i] == 0.0f) continue;
        float w[3][4], dw[3][4]; int i0[3];
        frac_index(x + 3 * i, recip, K, i0, w, dw);
        float g[3] = { 0, 0, 0 };
        for (int a = 0; a < 4; a++) {
            const int ia = (i0[0] + a) % nx;
            for (int b = 0; b < 4; b++) {
                const int ib = (i0[1] + b) % ny;
                const float *row = phi + ((size_t)ia * ny + ib) * nz;
                for (int c = 0; c < 4; c++) {
                    const float p = row[(i0[2] + c) % nz];
                    g[0] += dw[0][a] * w[1][b] * w[2][c] * p;
                    g[1] += w[0][a] * dw[1][b] * w[2][c] * p;
                    g[2] += w[0][a] * w[1][b] * dw[2][c] * p;
                }
            }
        }
        for (int e = 0; e < 3; e++) {
            float s = 0;
            for (int d = 0; d < 3; d++) s += g[d] * K[d] * recip[3 * e + d];
            f[3 * i + e] -= q[i] * s;
        }
    }
}

/* v += f/m dt; xp = x + v dt (all atoms) */
void cmd_leapfrog(int n, const float *x, float *v, const float *f, const float *invm, float dt, float *xp, int nthreads)
{
#pragma omp parallel for num_threads(nthreads < 1 ? 1 : nthreads) schedule(static)
    for (int i = 0; i < n; i++)
        for (int d = 0; d < 3; d++) {
            v[3 * i + d] += f[3 * i + d] * invm[i] * dt;
            xp[3 * i + d] = x[3 * i + d] + v[3 * i + d] * dt;
        }
}

/* Rigid three-site water: Newton iteration on the three constraint equations with the old bond vectors as
 * directions (the root SETTLE gives analytically); corrects xp and v. */
void cmd_settle(int nw, const int *idx, const float *x, float *xp, float *v, const float *invm, float doh, float dhh, float dt,
                int nthreads)
{
    const float inv_dt = 1.0f / dt;
#pragma omp parallel for num_threads(nthreads < 1 ? 1 : nthreads) schedule(static)
    for (int t = 0; t < nw; t++) {
        const int o = idx[3 * t], h1 = idx[3 * t + 1], h2 = idx[3 * t + 2];
        const float imo = invm[o], imh = invm[h1];
        float r0[3][3], s[3][3];
        for (int d = 0; d < 3; d++) {
            r0[0][d] = x[3 * o + d] - x[3 * h1 + d]; r0[1][d] = x[3 * o + d] - x[3 * h2 + d]; r0[2][d] = x[3 * h1 + d] - x[3 * h2 + d];
            s[0][d] = xp[3 * o + d] - xp[3 * h1 + d]; s[1][d] = xp[3 * o + d] - xp[3 * h2 + d]; s[2][d] = xp[3 * h1 + d] - xp[3 * h2 + d];
        }
        const float L2[3] = { doh * doh, doh * doh, dhh * dhh };
        const float A[3][3] = { { imo + imh, imo, -imh }, { imo, imo + imh, imh }, { -imh, imh, 2.0f * imh } };
        float lam[3] = { 0, 0, 0 };
        for (int it = 0; it < 8; it++) {
            float c[3][3], sig[3], J[3][3];
            for (int k = 0; k < 3; k++) {
                for (int d = 0; d < 3; d++) c[k][d] = s[k][d] + A[k][0] * lam[0] * r0[0][d] + A[k][1] * lam[1] * r0[1][d] + A[k][2] * lam[2] * r0[2][d];
                sig[k] = c[k][0] * c[k][0] + c[k][1] * c[k][1] + c[k][2] * c[k][2] - L2[k];
            }
            for (int k = 0; k < 3; k++)
                for (int m = 0; m < 3; m++) J[k][m] = 2.0f * A[k][m] * (c[k][0] * r0[m][0] + c[k][1] * r0[m][1] + c[k][2] * r0[m][2]);
            const float det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                              J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
            const float id = 1.0f / det, b0 = -sig[0], b1 = -sig[1], b2 = -sig[2];
            lam[0] += id * (b0 * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (b1 * J[2][2] - J[1][2] * b2) + J[0][2] * (b1 * J[2][1] - J[1][1] * b2));
            lam[1] += id * (J[0][0] * (b1 * J[2][2] - J[1][2] * b2) - b0 * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) + J[0][2] * (J[1][0] * b2 - b1 * J[2][0]));
            lam[2] += id * (J[0][0] * (J[1][1] * b2 - b1 * J[2][1]) - J[0][1] * (J[1][0] * b2 - b1 * J[2][0]) + b0 * (J[1][0] * J[2][1] - J[1][1] * J[2][0]));
            if (it > 0 && fabsf(sig[0]) < 1e-7f * L2[2] && fabsf(sig[1]) < 1e-7f * L2[2] && fabsf(sig[2]) < 1e-7f * L2[2]) break;
        }
        for (int d = 0; d < 3; d++) {
            const float dO = imo * (lam[0] * r0[0][d] + lam[1] * r0[1][d]);
            const float dH1 = imh * (lam[2] * r0[2][d] - lam[0] * r0[0][d]);
            const float dH2 = -imh * (lam[1] * r0[1][d] + lam[2] * r0[2][d]);
            xp[3 * o + d] += dO; xp[3 * h1 + d] += dH1; xp[3 * h2 + d] += dH2;
            v[3 * o + d] += dO * inv_dt; v[3 * h1 + d] += dH1 * inv_dt; v[3 * h2 + d] += dH2 * inv_dt;
        }
    }
}
