// Leap-frog update, constraints (SETTLE-type water + LINCS), V-rescale thermostat, C-rescale and
// Parrinello-Rahman barostats, COM-motion removal and the per-step energy bookkeeping.
// Replaces what `integrator = md`, `dt`, `constraints = h-bonds`, `constraint_algorithm = lincs`,
// `lincs_order/iter`, `tcoupl = V-rescale`, `pcoupl`, `nstcomm` switch on in gmx mdrun
// (/root/reference/mdp/production.mdp:3-5,16-19,31-40; npt_complex.mdp:64-66).
#include "mdx_internal.h"
#include <algorithm>

// v' = lambda_g v + f/m dt - dt_pc M v  (lambda_g: V-rescale factor of the atom's group, drawn from the kinetic
// energy of v itself, so it scales v and not the force kick; M: Parrinello-Rahman coupling matrix, held between
// coupling steps and applied every step over dt_pc = dt); xp = x + v' dt.
// Total force = original-order array + sorted nonbonded buffer.
__global__ __launch_bounds__(256) void k_leapfrog(int n, const float4 *__restrict__ x, float4 *__restrict__ xp,
                                                  float4 *__restrict__ v, float *__restrict__ f, const float *__restrict__ f_pme,
                                                  const float *__restrict__ f_s, const int *__restrict__ sorted_of,
                                                  const int *__restrict__ tcgrp, const DevState *__restrict__ d, float dt,
                                                  int use_lambda, int use_pr, float dt_pc)
{
    MDX_STAMP(d, 11);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 xi = x[i];
    float4 vi = v[i];
    const int g = sorted_of[i];
    const float fx = f[3 * i] + f_pme[3 * i] + f_s[3 * g], fy = f[3 * i + 1] + f_pme[3 * i + 1] + f_s[3 * g + 1],
                fz = f[3 * i + 2] + f_pme[3 * i + 2] + f_s[3 * g + 2];
    // the accumulation buffer of the bonded-type kernels is consumed here: leave it cleared for the
    // next step instead of paying a separate fill on the critical path
    f[3 * i] = 0.0f; f[3 * i + 1] = 0.0f; f[3 * i + 2] = 0.0f;
    const float im = vi.w;
    float lam = 1.0f;
    if (use_lambda) lam = (float)d->lambda[tcgrp[i]];
    float nx = fmaf(lam, vi.x, fx * im * dt), ny = fmaf(lam, vi.y, fy * im * dt), nz = fmaf(lam, vi.z, fz * im * dt);
    if (use_pr) {
        const double *M = d->pr_M;
        nx -= dt_pc * (float)(M[0] * vi.x + M[1] * vi.y + M[2] * vi.z);
        ny -= dt_pc * (float)(M[3] * vi.x + M[4] * vi.y + M[5] * vi.z);
        nz -= dt_pc * (float)(M[6] * vi.x + M[7] * vi.y + M[8] * vi.z);
    }
    vi.x = nx; vi.y = ny; vi.z = nz;
    v[i] = vi;
    xp[i] = make_float4(xi.x + nx * dt, xi.y + ny * dt, xi.z + nz * dt, xi.w);
}

// Rigid three-site water: solve the three holonomic constraint equations
//   |x'_a - x'_b|^2 = d_ab^2,  x' = xp + (1/m) sum_k lambda_k grad_k(x_old)
// for the Lagrange multipliers by Newton iteration (the SETTLE solution of Miyamoto & Kollman 1992
// is the same root, reached here iteratively in float32 with differences formed first).
// One thread per molecule; also corrects velocities and, on virial steps, accumulates the
// constraint virial -1/2 sum r_old (x) G with G = m dx / dt^2.
template <bool VIRIAL>
__device__ __forceinline__ void d_settle(const int t, int nw, const int4 *__restrict__ idx, const float4 *__restrict__ x,
                                         float4 *__restrict__ xp, float4 *__restrict__ v, float doh, float dhh,
                                         float inv_dt, DevState *dw)
{
    float v9[9] = { 0 };
    if (t < nw) {
        const int4 a = idx[t];
        const float4 vo = v[a.x], vh1 = v[a.y], vh2 = v[a.z];
        const float imo = vo.w, imh = vh1.w;
        const float3 o0 = xyz(x[a.x]), h10 = xyz(x[a.y]), h20 = xyz(x[a.z]);
        const float4 po = xp[a.x], ph1 = xp[a.y], ph2 = xp[a.z];
        // old bond vectors (constraint gradients) and unconstrained new bond vectors
        const float3 r0[3] = { o0 - h10, o0 - h20, h10 - h20 };
        float3 s[3] = { xyz(po) - xyz(ph1), xyz(po) - xyz(ph2), xyz(ph1) - xyz(ph2) };
        const float L2[3] = { doh * doh, doh * doh, dhh * dhh };
        // d s_k / d lambda_m = A[k][m] * r0[m]
        const float A[3][3] = { { imo + imh, imo, -imh }, { imo, imo + imh, imh }, { -imh, imh, 2.0f * imh } };
        float lam[3] = { 0.0f, 0.0f, 0.0f };
        for (int it = 0; it < 6; it++) {
            float3 c[3];
            float sig[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                c[k] = s[k] + (A[k][0] * lam[0]) * r0[0] + (A[k][1] * lam[1]) * r0[1] + (A[k][2] * lam[2]) * r0[2];
                sig[k] = dot3(c[k], c[k]) - L2[k];
            }
            float J[3][3];
#pragma unroll
            for (int k = 0; k < 3; k++)
#pragma unroll
                for (int m = 0; m < 3; m++) J[k][m] = 2.0f * A[k][m] * dot3(c[k], r0[m]);
            // solve J dl = -sig (Cramer)
            const float det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                              J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
            const float id = 1.0f / det;
            const float b0 = -sig[0], b1 = -sig[1], b2 = -sig[2];
            const float d0 = b0 * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (b1 * J[2][2] - J[1][2] * b2) + J[0][2] * (b1 * J[2][1] - J[1][1] * b2);
            const float d1 = J[0][0] * (b1 * J[2][2] - J[1][2] * b2) - b0 * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) + J[0][2] * (J[1][0] * b2 - b1 * J[2][0]);
            const float d2 = J[0][0] * (J[1][1] * b2 - b1 * J[2][1]) - J[0][1] * (J[1][0] * b2 - b1 * J[2][0]) + b0 * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
            lam[0] += d0 * id; lam[1] += d1 * id; lam[2] += d2 * id;
            const float tol = 1e-7f * L2[2];
            if (fabsf(sig[0]) < tol && fabsf(sig[1]) < tol && fabsf(sig[2]) < tol && it > 0) break;
        }
        // displacements: dO = imo (l0 r0 + l1 r1), dH1 = imh (-l0 r0 + l2 r2), dH2 = imh (-l1 r1 - l2 r2)
        const float3 dO = imo * (lam[0] * r0[0] + lam[1] * r0[1]);
        const float3 dH1 = imh * (lam[2] * r0[2] - lam[0] * r0[0]);
        const float3 dH2 = -imh * (lam[1] * r0[1] + lam[2] * r0[2]);
        xp[a.x] = make_float4(po.x + dO.x, po.y + dO.y, po.z + dO.z, po.w);
        xp[a.y] = make_float4(ph1.x + dH1.x, ph1.y + dH1.y, ph1.z + dH1.z, ph1.w);
        xp[a.z] = make_float4(ph2.x + dH2.x, ph2.y + dH2.y, ph2.z + dH2.z, ph2.w);
        v[a.x] = make_float4(vo.x + dO.x * inv_dt, vo.y + dO.y * inv_dt, vo.z + dO.z * inv_dt, vo.w);
        v[a.y] = make_float4(vh1.x + dH1.x * inv_dt, vh1.y + dH1.y * inv_dt, vh1.z + dH1.z * inv_dt, vh1.w);
        v[a.z] = make_float4(vh2.x + dH2.x * inv_dt, vh2.y + dH2.y * inv_dt, vh2.z + dH2.z * inv_dt, vh2.w);
        if (VIRIAL) {
            // constraint forces along the old bonds: G_k = lam_k r0_k / dt^2 acting on the first atom
            const float s2 = inv_dt * inv_dt;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float3 g = (lam[k] * s2) * r0[k];
                v9[0] += r0[k].x * g.x; v9[1] += r0[k].x * g.y; v9[2] += r0[k].x * g.z;
                v9[3] += r0[k].y * g.x; v9[4] += r0[k].y * g.y; v9[5] += r0[k].y * g.z;
                v9[6] += r0[k].z * g.x; v9[7] += r0[k].z * g.y; v9[8] += r0[k].z * g.z;
            }
        }
    }
    if (VIRIAL) {
        __shared__ double red[16];
        for (int k = 0; k < 9; k++) {
            const double r = block_sum_d((double)v9[k], red);
            if (threadIdx.x == 0) atomicAdd(&MDX_VIR(dw, k), -0.5 * r);
        }
    }
}

// LINCS (Hess et al. 1997) for small coupled components (h-bond constraints couple only through a
// shared heavy atom): one thread per connected component of at most MDX_LINCS_MAXK constraints,
// matrix expansion of order `order`, `niter` correction iterations for rotational lengthening.
// Components are stored by ascending size. Those of at most 4 constraints (CH, CH2, CH3, NH3
// groups: nearly all of an h-bonds topology) run in the MAXK = 4 instantiation, whose loops have
// constant bounds so that the per-component arrays live in registers; the rest use MAXK = 16.
#define LINCS_FOR(k) _Pragma("unroll") for (int k = 0; k < (MAXK <= 4 ? MAXK : K); k++) if (MAXK > 4 || k < K)
template <bool VIRIAL, int MAXK>
__device__ __forceinline__ void d_lincs(const int t, int comp0, int ncomp, const int *__restrict__ comp_start, const int2 *__restrict__ cij,
                                        const float *__restrict__ clen, const float4 *__restrict__ x, float4 *__restrict__ xp,
                                        float4 *__restrict__ v, int order, int niter, float inv_dt, DevState *dw)
{
    float v9[9] = { 0 };
    if (t < comp0 + ncomp) {
        const int c0 = comp_start[t];
        const int K = comp_start[t + 1] - c0;
        int ia[MAXK], ib[MAXK];
        float3 B[MAXK];
        float S[MAXK], len[MAXK], rhs[MAXK], sol[MAXK], tmp[MAXK];
        float ima[MAXK], imb[MAXK], lamtot[MAXK];
        float Acoef[MAXK][MAXK];
        LINCS_FOR(k) {
            const int2 p = cij[c0 + k];
            ia[k] = p.x; ib[k] = p.y; len[k] = clen[c0 + k];
            ima[k] = v[p.x].w; imb[k] = v[p.y].w;
            S[k] = rsqrtf(ima[k] + imb[k]);
            float3 d = xyz(x[p.x]) - xyz(x[p.y]);
            B[k] = rsqrtf(dot3(d, d)) * d;
            lamtot[k] = 0.0f;
        }
        LINCS_FOR(k)
            LINCS_FOR(m) {
                float c = 0.0f;
                if (m != k) {
                    // sign +1 when the shared atom is first in both or second in both
                    if (ia[k] == ia[m]) c = -ima[k];
                    else if (ib[k] == ib[m]) c = -imb[k];
                    else if (ia[k] == ib[m]) c = ima[k];
                    else if (ib[k] == ia[m]) c = imb[k];
                    c *= S[k] * S[m] * dot3(B[k], B[m]);
                }
                Acoef[k][m] = c;
            }
        for (int pass = 0; pass <= niter; pass++) {
            LINCS_FOR(k) {
                const float3 d = xyz(xp[ia[k]]) - xyz(xp[ib[k]]);
                if (pass == 0) rhs[k] = S[k] * (dot3(B[k], d) - len[k]);
                else {
                    const float p2 = 2.0f * len[k] * len[k] - dot3(d, d);
                    rhs[k] = S[k] * (len[k] - sqrtf(fmaxf(p2, 0.0f)));
                }
                sol[k] = rhs[k];
            }
            for (int o = 0; o < order; o++) {
                LINCS_FOR(k) {
                    float a = 0.0f;
                    LINCS_FOR(m) a += Acoef[k][m] * rhs[m];
                    tmp[k] = a;
                }
                LINCS_FOR(k) { rhs[k] = tmp[k]; sol[k] += tmp[k]; }
            }
            LINCS_FOR(k) {
                const float l = S[k] * sol[k];
                lamtot[k] += l;
                const float3 da = (-ima[k] * l) * B[k], db = (imb[k] * l) * B[k];
                float4 pa = xp[ia[k]], pb = xp[ib[k]];
                pa.x += da.x; pa.y += da.y; pa.z += da.z; pb.x += db.x; pb.y += db.y; pb.z += db.z;
                xp[ia[k]] = pa; xp[ib[k]] = pb;
                float4 va = v[ia[k]], vb = v[ib[k]];
                va.x += da.x * inv_dt; va.y += da.y * inv_dt; va.z += da.z * inv_dt;
                vb.x += db.x * inv_dt; vb.y += db.y * inv_dt; vb.z += db.z * inv_dt;
                v[ia[k]] = va; v[ib[k]] = vb;
            }
        }
        if (VIRIAL) {
            const float s2 = inv_dt * inv_dt;
            LINCS_FOR(k) {
                // force on atom a: -lam B / dt^2 ; r_ab = |r_old| B (old bond = constraint length)
                const float g = -lamtot[k] * s2 * len[k];
                v9[0] += B[k].x * B[k].x * g; v9[1] += B[k].x * B[k].y * g; v9[2] += B[k].x * B[k].z * g;
                v9[3] += B[k].y * B[k].x * g; v9[4] += B[k].y * B[k].y * g; v9[5] += B[k].y * B[k].z * g;
                v9[6] += B[k].z * B[k].x * g; v9[7] += B[k].z * B[k].y * g; v9[8] += B[k].z * B[k].z * g;
            }
        }
    }
    if (VIRIAL) {
        __shared__ double red[16];
        for (int k = 0; k < 9; k++) {
            const double r = block_sum_d((double)v9[k], red);
            if (threadIdx.x == 0) atomicAdd(&MDX_VIR(dw, k), -0.5 * r);
        }
    }
}

template <bool VIRIAL>
__global__ __launch_bounds__(256) void k_settle(int nw, const int4 *__restrict__ idx, const float4 *__restrict__ x,
                                                float4 *__restrict__ xp, float4 *__restrict__ v, float doh, float dhh,
                                                float inv_dt, DevState *dw)
{
    MDX_STAMP(dw, 12);
    d_settle<VIRIAL>(blockIdx.x * blockDim.x + threadIdx.x, nw, idx, x, xp, v, doh, dhh, inv_dt, dw);
}

template <bool VIRIAL, int MAXK>
__global__ __launch_bounds__(128) void k_lincs(int comp0, int ncomp, const int *__restrict__ comp_start, const int2 *__restrict__ cij,
                                               const float *__restrict__ clen, const float4 *__restrict__ x, float4 *__restrict__ xp,
                                               float4 *__restrict__ v, int order, int niter, float inv_dt, DevState *dw)
{
    MDX_STAMP(dw, 13);
    d_lincs<VIRIAL, MAXK>(comp0 + blockIdx.x * blockDim.x + threadIdx.x, comp0, ncomp, comp_start, cij, clen, x, xp, v, order, niter, inv_dt, dw);
}

// SETTLE and both LINCS instantiations in one launch (they touch disjoint atoms): consecutive ranges of 128-thread
// workgroups. The update sits on the critical path of every step and each launch costs more than its arithmetic.
struct ConstrainArgs {
    int nbw, nb4;               // workgroups of the settle range, of the small-component range (the rest: large components)
    int nw; const int4 *settle_idx; float doh, dhh;
    int ncomp_small, ncomp; const int *comp_start; const int2 *cij; const float *clen; int order, niter;
    const float4 *x; float4 *xp; float4 *v; float inv_dt; DevState *dw;
};
template <bool VIRIAL>
__global__ __launch_bounds__(128) void k_constrain(ConstrainArgs A)
{
    MDX_STAMP(A.dw, 12);
    const int b = blockIdx.x;
    if (b < A.nbw) d_settle<VIRIAL>(b * 128 + threadIdx.x, A.nw, A.settle_idx, A.x, A.xp, A.v, A.doh, A.dhh, A.inv_dt, A.dw);
    else if (b < A.nbw + A.nb4)
        d_lincs<VIRIAL, 4>((b - A.nbw) * 128 + threadIdx.x, 0, A.ncomp_small, A.comp_start, A.cij, A.clen, A.x, A.xp, A.v, A.order, A.niter, A.inv_dt, A.dw);
    else
        d_lincs<VIRIAL, MDX_LINCS_MAXK>(A.ncomp_small + (b - A.nbw - A.nb4) * 128 + threadIdx.x, A.ncomp_small, A.ncomp - A.ncomp_small,
                                        A.comp_start, A.cij, A.clen, A.x, A.xp, A.v, A.order, A.niter, A.inv_dt, A.dw);
}

// Half-step kinetic energy per T-coupling group (+ tensor) from the current velocities.
__global__ __launch_bounds__(256) void k_ke(int n, const float4 *__restrict__ v, const int *__restrict__ tcgrp, int ntc,
                                            DevState *dw, int old_slot, int tensor)
{
    // grid-stride: a fixed, small number of workgroups keeps the count of contended double atomics low
    __shared__ double red[16];
    double keg[MDX_MAX_TC];
    double t6[6] = { 0, 0, 0, 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < MDX_MAX_TC; k++) keg[k] = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 vi = v[i];
        const float m = vi.w > 0.0f ? 1.0f / vi.w : 0.0f;
        const float ke = 0.5f * m * (vi.x * vi.x + vi.y * vi.y + vi.z * vi.z);
        const int g = tcgrp[i];
#pragma unroll
        for (int k = 0; k < MDX_MAX_TC; k++) keg[k] += (g == k) ? (double)ke : 0.0;
        if (tensor) {
            t6[0] += 0.5f * m * vi.x * vi.x; t6[1] += 0.5f * m * vi.x * vi.y; t6[2] += 0.5f * m * vi.x * vi.z;
            t6[3] += 0.5f * m * vi.y * vi.y; t6[4] += 0.5f * m * vi.y * vi.z; t6[5] += 0.5f * m * vi.z * vi.z;
        }
    }
    const int ng = ntc > 0 ? ntc : 1;
#pragma unroll
    for (int k = 0; k < MDX_MAX_TC; k++) {
        if (k >= ng) break;
        const double r = block_sum_d(keg[k], red);
        if (threadIdx.x == 0 && r != 0.0) atomicAdd(old_slot ? &dw->ekinh_old[k] : &dw->ekinh[k], r);
    }
    if (tensor) {
        const int map[6][2] = { { 0, 0 }, { 1, 3 }, { 2, 6 }, { 4, 4 }, { 5, 7 }, { 8, 8 } };
        for (int k = 0; k < 6; k++) {
            const double r = block_sum_d(t6[k], red);
            if (threadIdx.x == 0) {
                atomicAdd(&MDX_EKT(dw, map[k][0]), 0.5 * r);   // average of two half steps
                if (map[k][1] != map[k][0]) atomicAdd(&MDX_EKT(dw, map[k][1]), 0.5 * r);
            }
        }
    }
}

// V-rescale (Bussi, Donadio, Parrinello 2007): one thread draws the new kinetic energy of every
// group from the half-step kinetic energy just computed and stores lambda for the coming step.
__global__ void k_tcouple(DevState *dw, int ntc, const double *ref_t, const double *tau_t, const double *ndf,
                          double dt_tc, unsigned long long seed, long long step)
{
    MDX_STAMP(dw, 15);
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int g = 0; g < ntc; g++) {
        const double ek = dw->ekinh[g];
        double lam = 1.0;
        if (ek > 0.0 && tau_t[g] > 0.0 && ndf[g] > 0.0) {
            const double ekref = 0.5 * ndf[g] * MDX_BOLTZ * ref_t[g];
            const double c = exp(-dt_tc / tau_t[g]);
            RngStream rng(seed, 0x7c0000ull + g, (uint64_t)step);
            const double r1 = rng.gauss();
            const double rs = rng.chi2(ndf[g] - 1.0);
            const double eknew = ek + (1.0 - c) * (ekref * (rs + r1 * r1) / ndf[g] - ek) +
                                 2.0 * r1 * sqrt(c * (1.0 - c) * ek * ekref / ndf[g]);
            lam = sqrt(fmax(eknew, 0.0) / ek);
            dw->therm_integral[g] -= (eknew - ek);
        }
        dw->lambda[g] = lam;
    }
}

// Barostat parameters of one coupling step (kind 0: none).
struct PcoupleArgs { int kind; double tau_p, ref_p, compress, dt_pc, kT; unsigned long long seed; long long step; };
__device__ void pcouple_scalars(DevState *dw, int kind, double tau_p, double ref_p, double compress, double dt_pc, double kT,
                                unsigned long long seed, long long step);

// Finish an energy step: kinetic energy as the leap-frog average, temperature, pressure tensor,
// dispersion correction, totals and the conserved quantity. `frame` (pinned host memory, or null) receives a copy of
// the finished energy terms; with pc.kind != 0 the barostat's scalars of this step follow in the same launch.
__global__ void k_finish_energy(DevState *dw, int ntc, double ndf_total, int dispcorr, double c6avg, int natoms,
                                double rc, double q2sum, double beta, int have_pme, double *__restrict__ frame, PcoupleArgs pc)
{
    // fold the replicated accumulators (one quantity per thread), then thread 0 finishes
    {
        const int t = threadIdx.x;
        if (t < MDX_E_N) { double a = 0.0; for (int r = 0; r < MDX_NREP; r++) a += dw->ener_rep[r][t]; dw->ener[t] += a; }
        else if (t < MDX_E_N + 9) { const int k = t - MDX_E_N; double a = 0.0; for (int r = 0; r < MDX_NREP; r++) a += dw->vir_rep[r][k]; dw->vir[k] += a; }
        else if (t < MDX_E_N + 18) { const int k = t - MDX_E_N - 9; double a = 0.0; for (int r = 0; r < MDX_NREP; r++) a += dw->ekt_rep[r][k]; dw->ekin_tensor[k] += a; }
        else if (t < MDX_E_N + 20) {
            const int k = t - MDX_E_N - 18; unsigned long long a = 0;
            for (int r = 0; r < MDX_NREP; r++) a += dw->npairs_rep[r][k];
            if (k == 0) dw->npairs_rc += a; else dw->npairs_lj += a;
        }
    }
    __syncthreads();
    if (blockIdx.x != 0) return;
    if (threadIdx.x == 0) {
    const int ng = ntc > 0 ? ntc : 1;
    double ek = 0.0;
    for (int g = 0; g < ng; g++) ek += 0.5 * (dw->ekinh_old[g] + dw->ekinh[g]);
    dw->ener[MDX_E_KIN] = ek;
    const double vol = dw->boxd[0] * dw->boxd[4] * dw->boxd[8];
    double edisp = 0.0, pdisp = 0.0;
    if (dispcorr) {
        const double nn = (double)natoms * natoms;
        const double etail = -(2.0 / 3.0) * M_PI * nn / vol * c6avg / (rc * rc * rc);
        edisp = 2.0 * etail;  // tail + the constant removed by the potential-shift inside rc
        if (dispcorr == 2) pdisp = -(4.0 / 3.0) * M_PI * nn / (vol * vol) * c6avg / (rc * rc * rc) * MDX_PRESFAC;
    }
    dw->ener[MDX_E_DISPCORR] = edisp;
    if (have_pme) dw->ener[MDX_E_COUL_EXCL] -= 138.935458 * beta / sqrt(M_PI) * q2sum;
    double pot = 0.0;
    for (int k = MDX_E_LJ; k <= MDX_E_DISPCORR; k++) pot += dw->ener[k];
    dw->ener[MDX_E_POT] = pot;
    dw->ener[MDX_E_TEMP] = ndf_total > 0 ? 2.0 * ek / (ndf_total * MDX_BOLTZ) : 0.0;
    double ptr = 0.0;
    for (int a = 0; a < 9; a++) {
        dw->pres[a] = 2.0 / vol * (dw->ekin_tensor[a] - dw->vir[a]) * MDX_PRESFAC + ((a % 4 == 0) ? pdisp : 0.0);
        if (a % 4 == 0) ptr += dw->pres[a];
    }
    dw->ener[MDX_E_PRES] = ptr / 3.0;
    dw->ener[MDX_E_VOLUME] = vol;
    double cons = pot + ek + dw->baro_integral;
    for (int g = 0; g < ng; g++) cons += dw->therm_integral[g];
    dw->ener[MDX_E_CONSERVED] = cons;
    }
    __syncthreads();
    if (frame && threadIdx.x < MDX_E_N) frame[threadIdx.x] = dw->ener[threadIdx.x];
    if (threadIdx.x != 0) return;
    if (pc.kind) pcouple_scalars(dw, pc.kind, pc.tau_p, pc.ref_p, pc.compress, pc.dt_pc, pc.kT, pc.seed, pc.step);
}

// Barostat scalars (one thread). kind 1: C-rescale (Bernetti & Bussi 2020), isotropic:
//   d eps = -(beta/tau_p)(P0 - P) dt + sqrt(2 kT beta dt / (V tau_p)) dW,  mu = exp(d eps / 3).
// kind 2: Parrinello-Rahman, isotropic: box-velocity ODE with W^-1 = 4 pi^2 beta / (3 tau_p^2 L).
__device__ void pcouple_scalars(DevState *dw, int kind, double tau_p, double ref_p, double compress, double dt_pc, double kT,
                                unsigned long long seed, long long step)
{
    double *b = dw->boxd;
    const double vol = b[0] * b[4] * b[8];
    const double P = (dw->pres[0] + dw->pres[4] + dw->pres[8]) / 3.0;
    double mu[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    for (int a = 0; a < 9; a++) dw->pr_M[a] = 0.0;
    if (kind == 1) {
        RngStream rng(seed, 0xba0000ull, (uint64_t)step);
        const double gauss = rng.gauss();
        const double deps = -(compress / tau_p) * (ref_p - P) * dt_pc +
                            sqrt(2.0 * kT * compress * MDX_PRESFAC * dt_pc / (vol * tau_p)) * gauss;
        const double m = exp(deps / 3.0);
        mu[0] = mu[4] = mu[8] = m;
        // Energy the scaling puts into the system, taken out of the conserved quantity: positions scale by m, so the
        // potential changes by -sum_i F_i . (m - 1) r_i = 2 (m - 1) tr(Xi) (Xi = -1/2 sum r x F, constraint part
        // included); velocities scale by 1/m, so the kinetic energy changes by K (m^-2 - 1). To first order the sum is
        // -P dV.
        const double trvir = dw->vir[0] + dw->vir[4] + dw->vir[8];
        const double ekin = dw->ekin_tensor[0] + dw->ekin_tensor[4] + dw->ekin_tensor[8];
        dw->baro_integral -= 2.0 * (m - 1.0) * trvir + ekin * (1.0 / (m * m) - 1.0);
    } else if (kind == 2) {
        // Isotropic Parrinello-Rahman as GROMACS defines it [GMX-doc, "Parrinello-Rahman pressure coupling"]:
        // W^-1 = 4 pi^2 beta / (3 tau_p^2 L_max); the "force" on the box is t = b^-1T (P - P_ref) (the box rows are the
        // cell vectors, so the off-diagonal pressure elements enter through the off-diagonal elements of b^-1 in a
        // triclinic cell); the isotropic variant turns its diagonal into one relative acceleration
        // a_rel = (1/3) sum_d t_dd / b_dd shared by every box element: d(b')/dt += dt_pc V W^-1 a_rel b.
        const double maxl = fmax(b[0], fmax(b[4], b[8]));
        const double winv = 4.0 * M_PI * M_PI * compress / (3.0 * tau_p * tau_p * maxl);
        const double ia = 1.0 / b[0], ie = 1.0 / b[4], ii = 1.0 / b[8];
        const double binv[9] = { ia, 0, 0, -b[3] * ia * ie, ie, 0, (b[3] * b[7] - b[4] * b[6]) * ia * ie * ii, -b[7] * ie * ii, ii };
        double arel = 0.0;
        for (int dd = 0; dd < 3; dd++) {
            double t = 0.0;
            for (int k = 0; k < 3; k++) t += binv[3 * k + dd] * (dw->pres[3 * k + dd] - (k == dd ? ref_p : 0.0));
            arel += t / b[4 * dd];
        }
        arel /= 3.0;
        for (int a = 0; a < 9; a++) dw->pr_boxv[a] += dt_pc * vol * winv * arel * b[a];
        // every element moves at the same relative rate: mu = 1 + rate dt_pc, M = b^-1 (b db'/dt + db/dt b') b'^-1 = 2 rate
        const double rate = dw->pr_boxv[0] / b[0];
        const double m = 1.0 + rate * dt_pc;
        mu[0] = mu[4] = mu[8] = m;
        dw->pr_M[0] = dw->pr_M[4] = dw->pr_M[8] = 2.0 * rate;
        // conserved-energy part of the barostat: P_ref V + kinetic energy of the box, 1/2 sum (db_ij/dt)^2 / W^-1
        double kbox = 0.0;
        for (int a = 0; a < 9; a++) kbox += 0.5 * dw->pr_boxv[a] * dw->pr_boxv[a] / (winv * MDX_PRESFAC);
        dw->baro_integral = kbox + ref_p * vol * m * m * m / MDX_PRESFAC;
    }
    for (int a = 0; a < 9; a++) dw->mu[a] = mu[a];
    for (int a = 0; a < 3; a++) { dw->posres_com_old[a] = dw->posres_com[a]; dw->posres_com[a] *= mu[0]; }
    // new box (rows scaled isotropically)
    for (int a = 0; a < 9; a++) b[a] *= mu[0];
    DevBox nb;
    for (int a = 0; a < 9; a++) nb.h[a] = (float)b[a];
    const double ia = 1.0 / b[0], ie = 1.0 / b[4], ii = 1.0 / b[8];
    // inverse of lower-triangular rows a=(b0,0,0) b=(b3,b4,0) c=(b6,b7,b8)
    double inv[9] = { ia, 0, 0, -b[3] * ia * ie, ie, 0, (b[3] * b[7] - b[4] * b[6]) * ia * ie * ii, -b[7] * ie * ii, ii };
    for (int a = 0; a < 9; a++) nb.inv[a] = (float)inv[a];
    nb.vol = (float)(b[0] * b[4] * b[8]);
    for (int a = 0; a < 5; a++) nb.pad[a] = 0.0f;
    dw->box = nb;
}

__global__ void k_pcouple(DevState *dw, PcoupleArgs pc)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    pcouple_scalars(dw, pc.kind, pc.tau_p, pc.ref_p, pc.compress, pc.dt_pc, pc.kT, pc.seed, pc.step);
}

// Scale coordinates (and restraint references) by mu; C-rescale also scales velocities by 1/mu.
// refcoord_scaling 1 (`all`, /root/reference/mdp/npt_complex.mdp:72) scales every reference position, 2 (`com`,
// /root/reference/mdp/npt.mdp:40) moves them rigidly with their scaled centre.
__global__ void k_scale(int n, float4 *__restrict__ x, float4 *__restrict__ v, const DevState *__restrict__ d, int scale_v,
                        int nposres, float4 *__restrict__ posres_ref, int refcoord_scaling)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float m = (float)d->mu[0];
    if (i < n) {
        float4 p = x[i]; p.x *= m; p.y *= m; p.z *= m; x[i] = p;
        if (scale_v) { float4 w = v[i]; const float im = 1.0f / m; w.x *= im; w.y *= im; w.z *= im; v[i] = w; }
    }
    if (i < nposres) {
        float4 r = posres_ref[i];
        if (refcoord_scaling == 1) { r.x *= m; r.y *= m; r.z *= m; }
        else {
            r.x += (float)(d->posres_com[0] - d->posres_com_old[0]);
            r.y += (float)(d->posres_com[1] - d->posres_com_old[1]);
            r.z += (float)(d->posres_com[2] - d->posres_com_old[2]);
        }
        posres_ref[i] = r;
    }
}

__global__ __launch_bounds__(256) void k_comm_sum(int n, const float4 *__restrict__ v, DevState *dw)
{
    __shared__ double red[16];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double p[4] = { 0, 0, 0, 0 };
    if (i < n) {
        const float4 vi = v[i];
        const double m = vi.w > 0.0f ? 1.0 / vi.w : 0.0;
        p[0] = m * vi.x; p[1] = m * vi.y; p[2] = m * vi.z; p[3] = m;
    }
    for (int k = 0; k < 4; k++) {
        const double r = block_sum_d(p[k], red);
        if (threadIdx.x == 0) atomicAdd(&dw->com_v[k], r);
    }
}
__global__ void k_comm_sub(int n, float4 *__restrict__ v, const DevState *__restrict__ d)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double im = 1.0 / d->com_v[3];
    float4 vi = v[i];
    vi.x -= (float)(d->com_v[0] * im); vi.y -= (float)(d->com_v[1] * im); vi.z -= (float)(d->com_v[2] * im);
    v[i] = vi;
}

__global__ void k_merge_forces(int n, const float *__restrict__ f, const float *__restrict__ f_pme, const float *__restrict__ f_s,
                               const int *__restrict__ sorted_of, float *__restrict__ out, int have_sorted)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = f[3 * i] + f_pme[3 * i], b = f[3 * i + 1] + f_pme[3 * i + 1], c = f[3 * i + 2] + f_pme[3 * i + 2];
    if (have_sorted) { const int g = sorted_of[i]; a += f_s[3 * g]; b += f_s[3 * g + 1]; c += f_s[3 * g + 2]; }
    out[3 * i] = a; out[3 * i + 1] = b; out[3 * i + 2] = c;
}

int mdx_launch_update(mdx_system *s, bool do_tc, bool do_pr, bool virial)
{
    const int nb = (s->n + 255) / 256;
    hipLaunchKernelGGL(k_leapfrog, dim3(nb), dim3(256), 0, s->stream, s->n, s->x, s->xp, s->v, s->f, s->f_pme, s->f_s, s->sorted_of,
                       s->tcgrp, s->d, s->dt, do_tc ? 1 : 0, do_pr ? 1 : 0, s->dt);
    int rc = mdx_launch_constrain(s, 1.0f / s->dt, virial, s->lincs_iter);
    if (rc) return rc;
    // the constrained new positions become the state
    float4 *t = s->x; s->x = s->xp; s->xp = t;
    return MDX_OK;
}

// Constrain s->xp against the reference geometry s->x; velocities receive dx * inv_dt
// (inv_dt = 0 leaves them alone, inv_dt = -1/dt serves the start-up velocity projection).
int mdx_launch_constrain(mdx_system *s, float inv_dt, bool virial, int lincs_iter)
{
    if (s->nsettle > 0 && s->ncomp > 0) {
        ConstrainArgs A;
        A.nbw = (s->nsettle + 127) / 128; A.nb4 = (s->ncomp_small + 127) / 128;
        const int nb16 = (s->ncomp - s->ncomp_small + 127) / 128;
        A.nw = s->nsettle; A.settle_idx = s->settle_idx; A.doh = s->settle_doh; A.dhh = s->settle_dhh;
        A.ncomp_small = s->ncomp_small; A.ncomp = s->ncomp; A.comp_start = s->comp_start; A.cij = s->cons_ij; A.clen = s->cons_len;
        A.order = s->lincs_order; A.niter = lincs_iter; A.x = s->x; A.xp = s->xp; A.v = s->v; A.inv_dt = inv_dt; A.dw = s->d;
        if (virial) hipLaunchKernelGGL(k_constrain<true>, dim3(A.nbw + A.nb4 + nb16), dim3(128), 0, s->stream, A);
        else hipLaunchKernelGGL(k_constrain<false>, dim3(A.nbw + A.nb4 + nb16), dim3(128), 0, s->stream, A);
        MDX_CHECK(s, hipGetLastError());
        return MDX_OK;
    }
    if (s->nsettle > 0) {
        const int nbw = (s->nsettle + 255) / 256;
        if (virial) hipLaunchKernelGGL(k_settle<true>, dim3(nbw), dim3(256), 0, s->stream, s->nsettle, s->settle_idx, s->x, s->xp, s->v, s->settle_doh, s->settle_dhh, inv_dt, s->d);
        else hipLaunchKernelGGL(k_settle<false>, dim3(nbw), dim3(256), 0, s->stream, s->nsettle, s->settle_idx, s->x, s->xp, s->v, s->settle_doh, s->settle_dhh, inv_dt, s->d);
    }
#define LAUNCH_LINCS(MAXK_, first, count)                                                                                  \
    if ((count) > 0) {                                                                                                      \
        const int nbc_ = ((count) + 127) / 128;                                                                             \
        if (virial) hipLaunchKernelGGL((k_lincs<true, MAXK_>), dim3(nbc_), dim3(128), 0, s->stream, first, count, s->comp_start, s->cons_ij, s->cons_len, s->x, s->xp, s->v, s->lincs_order, lincs_iter, inv_dt, s->d); \
        else hipLaunchKernelGGL((k_lincs<false, MAXK_>), dim3(nbc_), dim3(128), 0, s->stream, first, count, s->comp_start, s->cons_ij, s->cons_len, s->x, s->xp, s->v, s->lincs_order, lincs_iter, inv_dt, s->d); \
    }
    LAUNCH_LINCS(4, 0, s->ncomp_small)
    LAUNCH_LINCS(MDX_LINCS_MAXK, s->ncomp_small, s->ncomp - s->ncomp_small)
    MDX_CHECK(s, hipGetLastError());
    return MDX_OK;
}

int mdx_launch_ke(mdx_system *s, bool old_slot, bool tensor, hipStream_t st)
{
    const int nb = std::min((s->n + 255) / 256, 128);
    hipLaunchKernelGGL(k_ke, dim3(nb), dim3(256), 0, st ? st : s->stream, s->n, s->v, s->tcgrp, s->ntc, s->d, old_slot ? 1 : 0, tensor ? 1 : 0);
    MDX_CHECK(s, hipGetLastError());
    return MDX_OK;
}

// Zero the accumulators that the coming step's kernels add into.
__global__ void k_clear(DevState *dw, int energies, int ekin_new, int ekin_old)
{
    const int t = threadIdx.x;
    if (energies) {
        if (t < MDX_E_N) dw->ener[t] = 0.0;
        if (t < 9) { dw->vir[t] = 0.0; dw->ekin_tensor[t] = 0.0; }
        if (t == 0) { dw->npairs_rc = 0; dw->npairs_lj = 0; }
        if (t < MDX_NREP) {     // one replica row per thread
            for (int k = 0; k < MDX_E_N; k++) dw->ener_rep[t][k] = 0.0;
            for (int k = 0; k < 9; k++) { dw->vir_rep[t][k] = 0.0; dw->ekt_rep[t][k] = 0.0; }
            dw->npairs_rep[t][0] = 0; dw->npairs_rep[t][1] = 0;
        }
    }
    if (t < MDX_MAX_TC) {
        if (ekin_new) dw->ekinh[t] = 0.0;
        if (ekin_old) dw->ekinh_old[t] = 0.0;
    }
}

int mdx_launch_clear(mdx_system *s, bool energies, bool ekin_new, bool ekin_old)
{
    hipLaunchKernelGGL(k_clear, dim3(1), dim3(64), 0, s->stream, s->d, energies ? 1 : 0, ekin_new ? 1 : 0, ekin_old ? 1 : 0);
    MDX_CHECK(s, hipGetLastError());
    return MDX_OK;
}

int mdx_launch_tcouple(mdx_system *s)
{
    if (s->ntc <= 0) return MDX_OK;
    hipLaunchKernelGGL(k_tcouple, dim3(1), dim3(64), 0, s->stream, s->d, s->ntc, s->d_tc, s->d_tc + MDX_MAX_TC,
                       s->d_tc + 2 * MDX_MAX_TC, (double)s->nsttcouple * s->dt, (unsigned long long)s->seed, s->step);
    MDX_CHECK(s, hipGetLastError());
    return MDX_OK;
}

static PcoupleArgs pcouple_args(mdx_system *s)
{
    PcoupleArgs pc;
    pc.kind = s->pcoupl; pc.tau_p = s->tau_p; pc.ref_p = s->ref_p; pc.compress = s->compress;
    pc.dt_pc = (double)s->nstpcouple * s->dt; pc.kT = MDX_BOLTZ * (s->ntc > 0 ? s->ref_t[0] : 300.0);
    pc.seed = (unsigned long long)s->seed; pc.step = s->step;
    return pc;
}

static int launch_scale(mdx_system *s)
{
    const int nb = (std::max(s->n, s->nposres) + 255) / 256;
    hipLaunchKernelGGL(k_scale, dim3(nb), dim3(256), 0, s->stream, s->n, s->x, s->v, s->d, s->pcoupl == 1 ? 1 : 0,
                       s->refcoord_scaling ? s->nposres : 0, s->posres_ref, s->refcoord_scaling);
    MDX_CHECK(s, hipGetLastError());
    return MDX_OK;
}

// frame: pinned host destination of the finished energy terms (null: none); with_pcouple: the barostat's scalars and
// the scaling of coordinates follow (what mdx_launch_pcouple would do after this call).
int mdx_launch_finish_energy(mdx_system *s, double *frame, bool with_pcouple)
{
    double ndf_total = 0;
    for (int g = 0; g < s->ntc; g++) ndf_total += s->ndf[g];
    if (s->ntc == 0) ndf_total = s->ndf_total;
    PcoupleArgs pc = pcouple_args(s);
    if (!with_pcouple) pc.kind = 0;
    hipLaunchKernelGGL(k_finish_energy, dim3(1), dim3(64), 0, s->stream, s->d, s->ntc, ndf_total, s->dispcorr, s->dispcorr_c6avg,
                       s->n, (double)s->rc, s->q2sum, (double)s->beta, s->have_pme ? 1 : 0, frame, pc);
    MDX_CHECK(s, hipGetLastError());
    return with_pcouple ? launch_scale(s) : MDX_OK;
}

int mdx_launch_pcouple(mdx_system *s)
{
    if (s->pcoupl == 0) return MDX_OK;
    hipLaunchKernelGGL(k_pcouple, dim3(1), dim3(64), 0, s->stream, s->d, pcouple_args(s));
    return launch_scale(s);
}

// phase 0: momentum and mass sums into DevState::com_v; phase 1: subtract the common velocity
int mdx_launch_comm_phase(mdx_system *s, int phase)
{
    const int nb = (s->n + 255) / 256;
    if (phase == 0) {
        MDX_CHECK(s, hipMemsetAsync((char *)s->d + offsetof(DevState, com_v), 0, sizeof(double) * 4, s->stream));
        hipLaunchKernelGGL(k_comm_sum, dim3(nb), dim3(256), 0, s->stream, s->n, s->v, s->d);
    } else hipLaunchKernelGGL(k_comm_sub, dim3(nb), dim3(256), 0, s->stream, s->n, s->v, s->d);
    MDX_CHECK(s, hipGetLastError());
    return MDX_OK;
}
int mdx_launch_comm(mdx_system *s)
{
    int rc = mdx_launch_comm_phase(s, 0);
    return rc ? rc : mdx_launch_comm_phase(s, 1);
}

int mdx_launch_merge_forces(mdx_system *s, float *out3n)
{
    const int nb = (s->n + 255) / 256;
    hipLaunchKernelGGL(k_merge_forces, dim3(nb), dim3(256), 0, s->stream, s->n, s->f, s->f_pme, s->f_s, s->sorted_of, out3n, 1);
    MDX_CHECK(s, hipGetLastError());
    return MDX_OK;
}
