// Leap-frog update, constraints (SETTLE-type water + LINCS), V-rescale thermostat, C-rescale and
// Parrinello-Rahman barostats, COM-motion removal and the per-step energy bookkeeping.
// Replaces what `integrator = md`, `dt`, `constraints = h-bonds`, `constraint_algorithm = lincs`,
// `lincs_order/iter`, `tcoupl = V-rescale`, `pcoupl`, `nstcomm` switch on in gmx mdrun
// (/root/reference/mdp/production.mdp:3-5,16-19,31-40; npt_complex.mdp:64-66).
#include "mdx_internal.h"
#include <hip/hip_ext.h>
#include <algorithm>
#include <vector>

// v' = lambda_g v + f/m dt - dt_pc M v  (lambda_g: V-rescale factor of the atom's group, drawn from the kinetic
// energy of v itself, so it scales v and not the force kick; M: Parrinello-Rahman coupling matrix, held between
// coupling steps and applied every step over dt_pc = dt); xp = x + v' dt.
// Total force = original-order array + sorted nonbonded buffer.
struct LeapArgs {
    float *f; const float *f_pme; float *f_s; const int *sorted_of; const int *tcgrp; const DevState *d;
    float dt; int use_lambda, use_pr; float dt_pc; int clear_sorted;
};
// One atom: reads and clears its force entries (the sorted nonbonded buffer only with clear_sorted), returns the new
// velocity (w: inverse mass), the unconstrained new position (w: charge) and the atom's slot in the cluster order.
__device__ __forceinline__ void d_leap(const LeapArgs &L, const int i, const float4 xi, float4 &vi, float4 &xpo, int &g)
{
    g = L.sorted_of[i];
    const float fx = L.f[3 * i] + L.f_pme[3 * i] + L.f_s[3 * g], fy = L.f[3 * i + 1] + L.f_pme[3 * i + 1] + L.f_s[3 * g + 1],
                fz = L.f[3 * i + 2] + L.f_pme[3 * i + 2] + L.f_s[3 * g + 2];
    // the accumulation buffer of the bonded-type kernels is consumed here: leave it cleared for the
    // next step instead of paying a separate fill on the critical path
    L.f[3 * i] = 0.0f; L.f[3 * i + 1] = 0.0f; L.f[3 * i + 2] = 0.0f;
    if (L.clear_sorted) { L.f_s[3 * g] = 0.0f; L.f_s[3 * g + 1] = 0.0f; L.f_s[3 * g + 2] = 0.0f; }
    const float im = vi.w, dt = L.dt;
    float lam = 1.0f;
    if (L.use_lambda) lam = (float)L.d->lambda[L.tcgrp[i]];
    float nx = fmaf(lam, vi.x, fx * im * dt), ny = fmaf(lam, vi.y, fy * im * dt), nz = fmaf(lam, vi.z, fz * im * dt);
    if (L.use_pr) {
        const double *M = L.d->pr_M;
        nx -= L.dt_pc * (float)(M[0] * vi.x + M[1] * vi.y + M[2] * vi.z);
        ny -= L.dt_pc * (float)(M[3] * vi.x + M[4] * vi.y + M[5] * vi.z);
        nz -= L.dt_pc * (float)(M[6] * vi.x + M[7] * vi.y + M[8] * vi.z);
    }
    vi.x = nx; vi.y = ny; vi.z = nz;
    xpo = make_float4(xi.x + nx * dt, xi.y + ny * dt, xi.z + nz * dt, xi.w);
}

__global__ __launch_bounds__(256) void k_leapfrog(int n, const float4 *__restrict__ x, float4 *__restrict__ xp,
                                                  float4 *__restrict__ v, LeapArgs L)
{
    MDX_STAMP(L.d, 11);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 vi = v[i], xo;
    int g;
    d_leap(L, i, x[i], vi, xo, g);
    v[i] = vi;
    xp[i] = xo;
}

// Rigid three-site water: solve the three holonomic constraint equations
//   |x'_a - x'_b|^2 = d_ab^2,  x' = xp + (1/m) sum_k lambda_k grad_k(x_old)
// for the Lagrange multipliers by Newton iteration (the SETTLE solution of Miyamoto & Kollman 1992
// is the same root, reached here iteratively in float32 with differences formed first).
// One thread per molecule; also corrects velocities and, on virial steps, accumulates the
// constraint virial -1/2 sum r_old (x) G with G = m dx / dt^2.
// The solve: old positions o0, h10, h20, unconstrained new positions po, ph1, ph2, inverse masses -> displacements of
// the three atoms, the multipliers and the old bond vectors (for the virial).
__device__ __forceinline__ void settle_solve(const float3 o0, const float3 h10, const float3 h20, const float3 po, const float3 ph1,
                                             const float3 ph2, const float imo, const float imh, const float doh, const float dhh,
                                             float3 &dO, float3 &dH1, float3 &dH2, float *lam, float3 *r0)
{
    // old bond vectors (constraint gradients) and unconstrained new bond vectors
    r0[0] = o0 - h10; r0[1] = o0 - h20; r0[2] = h10 - h20;
    float3 s[3] = { po - ph1, po - ph2, ph1 - ph2 };
    const float L2[3] = { doh * doh, doh * doh, dhh * dhh };
    // d s_k / d lambda_m = A[k][m] * r0[m]
    const float A[3][3] = { { imo + imh, imo, -imh }, { imo, imo + imh, imh }, { -imh, imh, 2.0f * imh } };
    lam[0] = lam[1] = lam[2] = 0.0f;
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
    dO = imo * (lam[0] * r0[0] + lam[1] * r0[1]);
    dH1 = imh * (lam[2] * r0[2] - lam[0] * r0[0]);
    dH2 = -imh * (lam[1] * r0[1] + lam[2] * r0[2]);
}

template <bool VIRIAL>
__device__ __forceinline__ void settle_core(const float3 o0, const float3 h10, const float3 h20, float4 &po, float4 &ph1, float4 &ph2,
                                            float4 &vo, float4 &vh1, float4 &vh2, float doh, float dhh, float inv_dt, float *v9)
{
    float3 dO, dH1, dH2, r0[3];
    float lam[3];
    settle_solve(o0, h10, h20, xyz(po), xyz(ph1), xyz(ph2), vo.w, vh1.w, doh, dhh, dO, dH1, dH2, lam, r0);
    po = make_float4(po.x + dO.x, po.y + dO.y, po.z + dO.z, po.w);
    ph1 = make_float4(ph1.x + dH1.x, ph1.y + dH1.y, ph1.z + dH1.z, ph1.w);
    ph2 = make_float4(ph2.x + dH2.x, ph2.y + dH2.y, ph2.z + dH2.z, ph2.w);
    vo = make_float4(vo.x + dO.x * inv_dt, vo.y + dO.y * inv_dt, vo.z + dO.z * inv_dt, vo.w);
    vh1 = make_float4(vh1.x + dH1.x * inv_dt, vh1.y + dH1.y * inv_dt, vh1.z + dH1.z * inv_dt, vh1.w);
    vh2 = make_float4(vh2.x + dH2.x * inv_dt, vh2.y + dH2.y * inv_dt, vh2.z + dH2.z * inv_dt, vh2.w);
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

template <bool VIRIAL>
__device__ __forceinline__ void d_settle(const int t, int nw, const int4 *__restrict__ idx, const float4 *__restrict__ x,
                                         float4 *__restrict__ xp, float4 *__restrict__ v, float doh, float dhh,
                                         float inv_dt, DevState *dw)
{
    float v9[9] = { 0 };
    if (t < nw) {
        const int4 a = idx[t];
        float4 vo = v[a.x], vh1 = v[a.y], vh2 = v[a.z];
        float4 po = xp[a.x], ph1 = xp[a.y], ph2 = xp[a.z];
        settle_core<VIRIAL>(xyz(x[a.x]), xyz(x[a.y]), xyz(x[a.z]), po, ph1, ph2, vo, vh1, vh2, doh, dhh, inv_dt, v9);
        xp[a.x] = po; xp[a.y] = ph1; xp[a.z] = ph2;
        v[a.x] = vo; v[a.y] = vh1; v[a.z] = vh2;
    }
    if (VIRIAL) {
        // the nine virial sums in one reduction; thread k adds the k-th
        __shared__ double red[16 * 9];
        double d9[9];
#pragma unroll
        for (int k = 0; k < 9; k++) d9[k] = (double)v9[k];
        const double r = block_sum_n<9>(d9, red);
        if (threadIdx.x < 9) atomicAdd(&MDX_VIR(dw, threadIdx.x), -0.5 * r);
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
        // the nine virial sums in one reduction; thread k adds the k-th
        __shared__ double red[16 * 9];
        double d9[9];
#pragma unroll
        for (int k = 0; k < 9; k++) d9[k] = (double)v9[k];
        const double r = block_sum_n<9>(d9, red);
        if (threadIdx.x < 9) atomicAdd(&MDX_VIR(dw, threadIdx.x), -0.5 * r);
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
// LARGE: the launch has a range of components of more than 4 constraints, whose 16 x 16 coupling matrix per thread sets
// the register allocation (168, with scratch memory) of every workgroup of the launch, the waters' included; a topology
// constrained on bonds to hydrogens only has none and takes the kernel without that range.
template <bool VIRIAL, bool LARGE>
__global__ __launch_bounds__(128) void k_constrain(ConstrainArgs A)
{
    MDX_STAMP(A.dw, 12);
    const int b = blockIdx.x;
    if (b < A.nbw) d_settle<VIRIAL>(b * 128 + threadIdx.x, A.nw, A.settle_idx, A.x, A.xp, A.v, A.doh, A.dhh, A.inv_dt, A.dw);
    else if (b < A.nbw + A.nb4)
        d_lincs<VIRIAL, 4>((b - A.nbw) * 128 + threadIdx.x, 0, A.ncomp_small, A.comp_start, A.cij, A.clen, A.x, A.xp, A.v, A.order, A.niter, A.inv_dt, A.dw);
    else if (LARGE)
        d_lincs<VIRIAL, MDX_LINCS_MAXK>(A.ncomp_small + (b - A.nbw - A.nb4) * 128 + threadIdx.x, A.ncomp_small, A.ncomp - A.ncomp_small,
                                        A.comp_start, A.cij, A.clen, A.x, A.xp, A.v, A.order, A.niter, A.inv_dt, A.dw);
}

// Leap-frog, constraints and the next step's sorted coordinates in one launch, for the steps that accumulate no virial.
// Every atom is advanced by the thread that constrains it: a water by its SETTLE thread (positions and velocities stay
// in registers between the two), the atoms of a LINCS component by the component's thread, all other atoms by threads
// of a fourth range. With write_sorted the thread also refreshes its atoms' entries of the cluster-ordered coordinate
// array, clears their slots of the sorted force buffer and block 0 resets the unit queues of the nonbonded kernel: the
// next step then needs no prepare launch (filler slots keep the position they had at the pair search; they carry no
// charge and no LJ well, so where they sit inside their cluster's bounds does not matter).
struct FusedArgs {
    ConstrainArgs C;
    LeapArgs L;
    int nbw3;                   // workgroups of the water range: 42 waters each (three lanes per water)
    int nbfree, nfree; const int *free_atoms;
    const int *comp_atom_start; const int *comp_atoms;
    const int2 *comp_local;     // per constraint: positions of its two atoms in the component's atom list
    int write_sorted; const int *shift_s; float4 *xq_s;
    int settle_first;           // >= 0: atom `role` of water w is settle_first + 3 w + role (no index load)
    double *ke_part;            // KE variant: MDX_MAX_TC partial sums per workgroup of the new half-step kinetic energy per T-group
};
__device__ __forceinline__ void d_sorted_write(const FusedArgs &A, const int g, const float4 p)
{
    const float *h = A.C.dw->box.h;
    int sa, sb, sc;
    unpack_shift(A.shift_s[g], sa, sb, sc);
    float4 o;
    o.x = p.x + sa * h[0] + sb * h[3] + sc * h[6];
    o.y = p.y + sb * h[4] + sc * h[7];
    o.z = p.z + sc * h[8];
    o.w = p.w;
    A.xq_s[g] = o;
}
// 1/2 m v^2 of one atom (v.w: inverse mass) added to its T-group's entry (compile-time indices: the array stays in registers)
__device__ __forceinline__ void d_ke_add(float *keg, const int g, const float4 v)
{
    const float ke = v.w > 0.0f ? 0.5f * (v.x * v.x + v.y * v.y + v.z * v.z) / v.w : 0.0f;
#pragma unroll
    for (int k = 0; k < MDX_MAX_TC; k++) keg[k] += (g == k) ? ke : 0.0f;
}
// A constraint component in the fused update: its thread advances the component's atoms, keeps their new positions and
// velocities in a private column of LDS (slot k of atom k of the component's atom list; `nthr` columns side by side, so
// neighbouring threads use neighbouring banks), runs LINCS on those and stores every atom once. The separate LINCS
// kernel updates positions and velocities in global memory constraint by constraint, and every such read-modify-write
// waits a memory round trip for the one before it (an atom shared by two constraints): about 10 us per launch, which
// in a fused kernel would sit behind the leap-frog of the same thread.
// (keg, when given: the kinetic energy 1/2 m v^2 of the component's atoms after the update, added per T-group)
template <int MAXK>
__device__ __forceinline__ void d_fused_lincs(const FusedArgs &A, const int t, const int comp0, const int ncomp, float *sh,
                                              const int tid, const int nthr, float *keg = nullptr)
{
    constexpr int NA = MAXK + 1;
    const ConstrainArgs &C = A.C;
    if (t >= comp0 + ncomp) return;
    auto S = [&](int slot, int c) -> float & { return sh[(slot * 7 + c) * nthr + tid]; };
    const int a0 = A.comp_atom_start[t], na = A.comp_atom_start[t + 1] - a0;
    int ai[NA], gi[NA];
    float qa[NA];
#pragma unroll
    for (int k = 0; k < NA; k++) {
        ai[k] = 0; gi[k] = 0; qa[k] = 0.0f;
        if (k < na) {
            const int i = A.comp_atoms[a0 + k];
            float4 vi = C.v[i], xo;
            d_leap(A.L, i, C.x[i], vi, xo, gi[k]);
            ai[k] = i; qa[k] = xo.w;
            S(k, 0) = xo.x; S(k, 1) = xo.y; S(k, 2) = xo.z; S(k, 3) = vi.x; S(k, 4) = vi.y; S(k, 5) = vi.z; S(k, 6) = vi.w;
        }
    }
    const int c0 = C.comp_start[t];
    const int K = C.comp_start[t + 1] - c0;
    int la[MAXK], lb[MAXK];
    float3 B[MAXK];
    float Sd[MAXK], len[MAXK], rhs[MAXK], sol[MAXK], tmp[MAXK], ima[MAXK], imb[MAXK];
    float Acoef[MAXK][MAXK];
    LINCS_FOR(k) {
        const int2 p = C.cij[c0 + k];
        const int2 q = A.comp_local[c0 + k];
        la[k] = q.x; lb[k] = q.y; len[k] = C.clen[c0 + k];
        ima[k] = S(q.x, 6); imb[k] = S(q.y, 6);
        Sd[k] = rsqrtf(ima[k] + imb[k]);
        const float3 d = xyz(C.x[p.x]) - xyz(C.x[p.y]);
        B[k] = rsqrtf(dot3(d, d)) * d;
    }
    LINCS_FOR(k)
        LINCS_FOR(m) {
            float c = 0.0f;
            if (m != k) {
                // sign +1 when the shared atom is first in both or second in both
                if (la[k] == la[m]) c = -ima[k];
                else if (lb[k] == lb[m]) c = -imb[k];
                else if (la[k] == lb[m]) c = ima[k];
                else if (lb[k] == la[m]) c = imb[k];
                c *= Sd[k] * Sd[m] * dot3(B[k], B[m]);
            }
            Acoef[k][m] = c;
        }
    for (int pass = 0; pass <= C.niter; pass++) {
        LINCS_FOR(k) {
            const float3 d = make_float3(S(la[k], 0) - S(lb[k], 0), S(la[k], 1) - S(lb[k], 1), S(la[k], 2) - S(lb[k], 2));
            if (pass == 0) rhs[k] = Sd[k] * (dot3(B[k], d) - len[k]);
            else {
                const float p2 = 2.0f * len[k] * len[k] - dot3(d, d);
                rhs[k] = Sd[k] * (len[k] - sqrtf(fmaxf(p2, 0.0f)));
            }
            sol[k] = rhs[k];
        }
        for (int o = 0; o < C.order; o++) {
            LINCS_FOR(k) {
                float a = 0.0f;
                LINCS_FOR(m) a += Acoef[k][m] * rhs[m];
                tmp[k] = a;
            }
            LINCS_FOR(k) { rhs[k] = tmp[k]; sol[k] += tmp[k]; }
        }
        LINCS_FOR(k) {
            const float l = Sd[k] * sol[k];
            const float3 da = (-ima[k] * l) * B[k], db = (imb[k] * l) * B[k];
            S(la[k], 0) += da.x; S(la[k], 1) += da.y; S(la[k], 2) += da.z;
            S(lb[k], 0) += db.x; S(lb[k], 1) += db.y; S(lb[k], 2) += db.z;
            S(la[k], 3) += da.x * C.inv_dt; S(la[k], 4) += da.y * C.inv_dt; S(la[k], 5) += da.z * C.inv_dt;
            S(lb[k], 3) += db.x * C.inv_dt; S(lb[k], 4) += db.y * C.inv_dt; S(lb[k], 5) += db.z * C.inv_dt;
        }
    }
#pragma unroll
    for (int k = 0; k < NA; k++)
        if (k < na) {
            const float4 xo = make_float4(S(k, 0), S(k, 1), S(k, 2), qa[k]);
            C.xp[ai[k]] = xo;
            C.v[ai[k]] = make_float4(S(k, 3), S(k, 4), S(k, 5), S(k, 6));
            if (A.write_sorted) d_sorted_write(A, gi[k], xo);
            if (keg) d_ke_add(keg, A.L.tcgrp[ai[k]], make_float4(S(k, 3), S(k, 4), S(k, 5), S(k, 6)));
        }
}
#define MDX_FUSED_LARGE_THREADS 32      // threads of a workgroup that take a component of more than 4 constraints (LDS columns of 17 atoms)
// KE: the launch also leaves the half-step kinetic energy of the new velocities per T-group, as one set of partial sums per
// workgroup (plain stores; the kernel that draws the thermostat's factors folds them): on the steps that end a coupling
// period the kinetic energy then needs no pass of its own over the velocities behind the update.
// LARGE: the launch has a range of components of more than 4 constraints. Their path holds a 16 x 16 coupling matrix per
// thread (255 registers and scratch memory), which as part of the one kernel would set the register allocation of every
// workgroup, the waters' included: topologies constrained on bonds to hydrogens only have no such component (a heavy atom
// carries at most four) and take the kernel without that path.
template <bool KE, bool LARGE>
__global__ __launch_bounds__(128) void k_update_fused(FusedArgs A)
{
    MDX_STAMP(A.C.dw, 11);
    __shared__ float sh[5 * 7 * 128];       // LINCS ranges: 5 atoms x 7 values x 128 threads, or 17 x 7 x 32
    static_assert((MDX_LINCS_MAXK + 1) * 7 * MDX_FUSED_LARGE_THREADS <= 5 * 7 * 128, "LDS columns of the large components");
    const ConstrainArgs &C = A.C;
    // (workgroups are dispatched in index order: the ranges are taken back to front, so that the large constraint components,
    // one thread walking through up to 17 atoms, start first and the waters fill in behind them)
    const int b = gridDim.x - 1 - blockIdx.x;
    float keg[MDX_MAX_TC];
#pragma unroll
    for (int k = 0; k < MDX_MAX_TC; k++) keg[k] = 0.0f;
    if (A.write_sorted && b == 0 && threadIdx.x < 8) C.dw->nb_head[threadIdx.x * 32] = 0;
    if (b < A.nbw3) {
        // One lane per atom, three consecutive lanes per water (21 waters per wave, the last lane idles): the leap-frog of
        // an atom is its lane's, the three lanes exchange old and new positions and each solves the water's constraints
        // for its own atom (lanes are cheaper than the latency of one thread walking through three atoms).
        const int lane = threadIdx.x & 63, role = lane % 3;
        const int w = (b * 2 + (threadIdx.x >> 6)) * 21 + lane / 3;
        const bool valid = lane < 63 && w < C.nw;
        int i = 0, g = 0;
        float4 xi = make_float4(0.0f, 0.0f, 0.0f, 0.0f), vi = xi, xpo = xi;
        if (valid) {
            if (A.settle_first >= 0) i = A.settle_first + 3 * w + role;
            else { const int4 a = C.settle_idx[w]; i = role == 0 ? a.x : (role == 1 ? a.y : a.z); }
            xi = C.x[i]; vi = C.v[i];
            d_leap(A.L, i, xi, vi, xpo, g);
        }
        const int base = lane - role;
        float3 xo[3], xn[3];
        float im[2];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int src = min(base + r, 63);
            xo[r] = make_float3(__shfl(xi.x, src, 64), __shfl(xi.y, src, 64), __shfl(xi.z, src, 64));
            xn[r] = make_float3(__shfl(xpo.x, src, 64), __shfl(xpo.y, src, 64), __shfl(xpo.z, src, 64));
            if (r < 2) im[r] = __shfl(vi.w, src, 64);
        }
        if (valid) {
            float3 dO, dH1, dH2, r0[3];
            float lam[3];
            settle_solve(xo[0], xo[1], xo[2], xn[0], xn[1], xn[2], im[0], im[1], C.doh, C.dhh, dO, dH1, dH2, lam, r0);
            const float3 dd = role == 0 ? dO : (role == 1 ? dH1 : dH2);
            xpo = make_float4(xpo.x + dd.x, xpo.y + dd.y, xpo.z + dd.z, xpo.w);
            vi = make_float4(vi.x + dd.x * C.inv_dt, vi.y + dd.y * C.inv_dt, vi.z + dd.z * C.inv_dt, vi.w);
            C.xp[i] = xpo; C.v[i] = vi;
            if (A.write_sorted) d_sorted_write(A, g, xpo);
            if (KE) d_ke_add(keg, A.L.tcgrp[i], vi);
        }
    } else if (b < A.nbw3 + C.nb4) {
        d_fused_lincs<4>(A, (b - A.nbw3) * 128 + threadIdx.x, 0, C.ncomp_small, sh, threadIdx.x, 128, KE ? keg : nullptr);
    } else if (b < A.nbw3 + C.nb4 + A.nbfree) {
        const int t = (b - A.nbw3 - C.nb4) * 128 + threadIdx.x;
        if (t < A.nfree) {
            const int i = A.free_atoms[t];
            float4 vi = C.v[i], xo; int g;
            d_leap(A.L, i, C.x[i], vi, xo, g);
            C.v[i] = vi; C.xp[i] = xo;
            if (A.write_sorted) d_sorted_write(A, g, xo);
            if (KE) d_ke_add(keg, A.L.tcgrp[i], vi);
        }
    } else if (LARGE) {
        if (threadIdx.x < MDX_FUSED_LARGE_THREADS)
            d_fused_lincs<MDX_LINCS_MAXK>(A, C.ncomp_small + (b - A.nbw3 - C.nb4 - A.nbfree) * MDX_FUSED_LARGE_THREADS + threadIdx.x, C.ncomp_small,
                                          C.ncomp - C.ncomp_small, sh, threadIdx.x, MDX_FUSED_LARGE_THREADS, KE ? keg : nullptr);
    }
    if (KE) {
        // (every thread of the workgroup arrives here: no path above returns early)
        __shared__ double red[16 * MDX_MAX_TC];
        double all[MDX_MAX_TC];
#pragma unroll
        for (int k = 0; k < MDX_MAX_TC; k++) all[k] = (double)keg[k];
        const double r = block_sum_n<MDX_MAX_TC>(all, red);
        if (threadIdx.x < MDX_MAX_TC) A.ke_part[(size_t)b * MDX_MAX_TC + threadIdx.x] = r;
    }
}

// Half-step kinetic energy per T-coupling group (+ tensor) from the current velocities.
__global__ __launch_bounds__(256) void k_ke(int n, const float4 *__restrict__ v, const int *__restrict__ tcgrp, int ntc,
                                            DevState *dw, int old_slot, int tensor)
{
    // grid-stride: a fixed, small number of workgroups keeps the count of contended double atomics low
    __shared__ double red[16 * (MDX_MAX_TC + 6)];
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
    // group kinetic energies and the six tensor sums in one reduction; thread k adds the k-th
    const int ng = ntc > 0 ? ntc : 1;
    double all[MDX_MAX_TC + 6];
#pragma unroll
    for (int k = 0; k < MDX_MAX_TC; k++) all[k] = keg[k];
#pragma unroll
    for (int k = 0; k < 6; k++) all[MDX_MAX_TC + k] = t6[k];
    const double r = block_sum_n<MDX_MAX_TC + 6>(all, red);
    const int t = threadIdx.x;
    if (t < ng && r != 0.0) atomicAdd(old_slot == 2 ? &dw->ekinh_tc[t] : (old_slot ? &dw->ekinh_old[t] : &dw->ekinh[t]), r);
    if (tensor && t >= MDX_MAX_TC && t < MDX_MAX_TC + 6) {
        const int map[6][2] = { { 0, 0 }, { 1, 3 }, { 2, 6 }, { 4, 4 }, { 5, 7 }, { 8, 8 } };
        const int k = t - MDX_MAX_TC;
        atomicAdd(&MDX_EKT(dw, map[k][0]), 0.5 * r);   // average of two half steps
        if (map[k][1] != map[k][0]) atomicAdd(&MDX_EKT(dw, map[k][1]), 0.5 * r);
    }
}

// V-rescale (Bussi, Donadio, Parrinello 2007): one thread per group draws the group's new kinetic energy
// from the half-step kinetic energy just computed and stores lambda for the coming step.
// tc_slot: the kinetic energy comes from the accumulator of the side-stream thermostat, which is left zeroed for the next use.
__device__ __forceinline__ void d_tcouple_draw(DevState *dw, const int g, const double ek, const double *ref_t, const double *tau_t,
                                               const double *ndf, double dt_tc, unsigned long long seed, long long step)
{
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
__global__ void k_tcouple(DevState *dw, int ntc, const double *ref_t, const double *tau_t, const double *ndf,
                          double dt_tc, unsigned long long seed, long long step, int tc_slot)
{
    MDX_STAMP(dw, 15);
    // one thread per group (the groups' draws are independent streams)
    if (blockIdx.x != 0 || (int)threadIdx.x >= ntc) return;
    const int g = threadIdx.x;
    const double ek = tc_slot ? dw->ekinh_tc[g] : dw->ekinh[g];
    if (tc_slot) dw->ekinh_tc[g] = 0.0;
    d_tcouple_draw(dw, g, ek, ref_t, tau_t, ndf, dt_tc, seed, step);
}
// The same from the partial sums a fused update (KE variant) left per workgroup: one workgroup folds them, then one
// thread per group draws.
__global__ __launch_bounds__(256) void k_tcouple_fold(DevState *dw, int ntc, const double *ref_t, const double *tau_t, const double *ndf,
                                                      double dt_tc, unsigned long long seed, long long step,
                                                      const double *__restrict__ ke_part, int nparts)
{
    MDX_STAMP(dw, 15);
    __shared__ double red[16 * MDX_MAX_TC];
    __shared__ double ek_s[MDX_MAX_TC];
    double acc[MDX_MAX_TC];
#pragma unroll
    for (int k = 0; k < MDX_MAX_TC; k++) acc[k] = 0.0;
    for (int p = threadIdx.x; p < nparts; p += blockDim.x)
#pragma unroll
        for (int k = 0; k < MDX_MAX_TC; k++) acc[k] += ke_part[(size_t)p * MDX_MAX_TC + k];
    const double r = block_sum_n<MDX_MAX_TC>(acc, red);
    if (threadIdx.x < MDX_MAX_TC) ek_s[threadIdx.x] = r;
    __syncthreads();
    if ((int)threadIdx.x < ntc) d_tcouple_draw(dw, threadIdx.x, ek_s[threadIdx.x], ref_t, tau_t, ndf, dt_tc, seed, step);
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
    // Fold the replicated accumulators through LDS: lane r loads every quantity of replica r (independent loads, one
    // memory round trip for all of them) and stores them transposed; lane k then sums the 64 replicas of quantity k from
    // LDS. (One thread per quantity walking through the replicas in global memory took 64 dependent loads, 12 of the
    // 13 us of this launch; butterfly sums of 36 doubles over the wave take as long.) One wave per launch.
    {
        constexpr int NQ = MDX_E_N + 18;
        static_assert(NQ <= 64, "one lane per folded quantity");
        static_assert(MDX_NREP == 64, "the fold below takes one replica per lane of a 64-wide wave");
        __shared__ double fold[NQ * 65];
        __shared__ unsigned long long foldn[2 * 64];
        const int t = threadIdx.x & 63;
#pragma unroll
        for (int k = 0; k < MDX_E_N; k++) fold[k * 65 + t] = dw->ener_rep[t][k];
#pragma unroll
        for (int k = 0; k < 9; k++) { fold[(MDX_E_N + k) * 65 + t] = dw->vir_rep[t][k]; fold[(MDX_E_N + 9 + k) * 65 + t] = dw->ekt_rep[t][k]; }
        foldn[t] = dw->npairs_rep[t][0]; foldn[64 + t] = dw->npairs_rep[t][1];
        __syncthreads();
        if (t < NQ) {
            double a = 0.0;
            for (int r = 0; r < MDX_NREP; r++) a += fold[t * 65 + r];
            if (t < MDX_E_N) dw->ener[t] += a;
            else if (t < MDX_E_N + 9) dw->vir[t - MDX_E_N] += a;
            else dw->ekin_tensor[t - MDX_E_N - 9] += a;
        } else if (t < NQ + 2) {
            const int k = t - NQ; unsigned long long a = 0;
            for (int r = 0; r < MDX_NREP; r++) a += foldn[k * 64 + r];
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

// (grid-stride over a fixed, small number of workgroups: every workgroup ends with four atomics on the same four
// addresses, which retire one after the other; a workgroup per 256 atoms took 29 us for 139 k atoms)
__global__ __launch_bounds__(256) void k_comm_sum(int n, const float4 *__restrict__ v, DevState *dw)
{
    __shared__ double red[16 * 4];
    double p[4] = { 0, 0, 0, 0 };
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 vi = v[i];
        const double m = vi.w > 0.0f ? 1.0 / vi.w : 0.0;
        p[0] += m * vi.x; p[1] += m * vi.y; p[2] += m * vi.z; p[3] += m;
    }
    const double r = block_sum_n<4>(p, red);
    if (threadIdx.x < 4) atomicAdd(&dw->com_v[threadIdx.x], r);
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

static LeapArgs leap_args(mdx_system *s, bool do_tc, bool do_pr, bool clear_sorted)
{
    LeapArgs L;
    L.f = s->f; L.f_pme = s->f_pme; L.f_s = s->f_s; L.sorted_of = s->sorted_of; L.tcgrp = s->tcgrp; L.d = s->d;
    L.dt = s->dt; L.use_lambda = do_tc ? 1 : 0; L.use_pr = do_pr ? 1 : 0; L.dt_pc = s->dt; L.clear_sorted = clear_sorted ? 1 : 0;
    return L;
}

int mdx_launch_update(mdx_system *s, bool do_tc, bool do_pr, bool virial)
{
    const int nb = (s->n + 255) / 256;
    hipLaunchKernelGGL(k_leapfrog, dim3(nb), dim3(256), 0, s->stream, s->n, (const float4 *)s->x, s->xp, s->v, leap_args(s, do_tc, do_pr, false));
    int rc = mdx_launch_constrain(s, 1.0f / s->dt, virial, s->lincs_iter);
    if (rc) return rc;
    // the constrained new positions become the state
    float4 *t = s->x; s->x = s->xp; s->xp = t;
    return MDX_OK;
}

static void constrain_args(mdx_system *s, ConstrainArgs &A, float inv_dt, int lincs_iter)
{
    A.nbw = (s->nsettle + 127) / 128; A.nb4 = (s->ncomp_small + 127) / 128;
    A.nw = s->nsettle; A.settle_idx = s->settle_idx; A.doh = s->settle_doh; A.dhh = s->settle_dhh;
    A.ncomp_small = s->ncomp_small; A.ncomp = s->ncomp; A.comp_start = s->comp_start; A.cij = s->cons_ij; A.clen = s->cons_len;
    A.order = s->lincs_order; A.niter = lincs_iter; A.x = s->x; A.xp = s->xp; A.v = s->v; A.inv_dt = inv_dt; A.dw = s->d;
}

// Which thread advances which atom in the fused update: every atom must belong to exactly one water, one constraint
// component or to neither. Builds the per-component atom lists and the list of unconstrained atoms; 1: usable.
int mdx_fused_plan(mdx_system *s)
{
    if (s->fused_plan != 0) return s->fused_plan;
    s->fused_plan = -1;
    if (s->dd_on || s->nsettle != s->nsettle_cap || (int)s->h_settle.size() != 3 * s->nsettle) return s->fused_plan;
    if ((int)s->h_cstart.size() != (s->ncomp ? s->ncomp + 1 : 0) && !(s->ncomp == 0 && s->h_cstart.size() <= 1)) return s->fused_plan;
    std::vector<int> owned(s->n, 0);
    for (int a : s->h_settle) owned[a]++;
    std::vector<int> cas(1, 0), ca, cl(2 * (size_t)s->ncons, 0);
    for (int c = 0; c < s->ncomp; c++) {
        const size_t first = ca.size();
        for (int k = s->h_cstart[c]; k < s->h_cstart[c + 1]; k++) {
            const int pair[2] = { s->h_cij[k].x, s->h_cij[k].y };
            for (int e = 0; e < 2; e++) {
                auto it = std::find(ca.begin() + first, ca.end(), pair[e]);
                if (it == ca.end()) { ca.push_back(pair[e]); owned[pair[e]]++; it = ca.end() - 1; }
                cl[2 * (size_t)k + e] = (int)(it - (ca.begin() + first));
            }
        }
        if ((int)(ca.size() - first) > s->h_cstart[c + 1] - s->h_cstart[c] + 1) return s->fused_plan;     // (cannot happen for a connected component)
        cas.push_back((int)ca.size());
    }
    std::vector<int> fr;
    for (int i = 0; i < s->n; i++) {
        if (owned[i] > 1) return s->fused_plan;       // an atom in a water and a component: the separate kernels handle that
        if (owned[i] == 0) fr.push_back(i);
    }
    auto up = [&](int **p, const std::vector<int> &h) {
        if (*p) { (void)hipFree(*p); *p = nullptr; }
        if (hipMalloc((void **)p, sizeof(int) * std::max<size_t>(h.size(), 1)) != hipSuccess) return false;
        return h.empty() || hipMemcpy(*p, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice) == hipSuccess;
    };
    if (!up(&s->comp_atom_start, cas) || !up(&s->comp_atoms, ca) || !up(&s->free_atoms, fr) || !up(&s->comp_local, cl)) return s->fused_plan;
    s->nfree = (int)fr.size();
    s->fused_plan = 1;
    return s->fused_plan;
}

// The fused update of a step without virial (mdx_fused_plan must have returned 1). write_sorted: also leave the sorted
// coordinates, the cleared sorted force buffer and the reset unit queues for the next step; fork_signal: the kernel's
// completion releases the PME and bonded streams (their next kernels read the new positions only).
int mdx_launch_update_fused(mdx_system *s, bool do_tc, bool do_pr, bool write_sorted, bool fork_signal, bool with_ke)
{
    FusedArgs A;
    constrain_args(s, A.C, 1.0f / s->dt, s->lincs_iter);
    A.L = leap_args(s, do_tc, do_pr, write_sorted);
    A.nfree = s->nfree; A.nbfree = (s->nfree + 127) / 128; A.free_atoms = s->free_atoms;
    A.comp_atom_start = s->comp_atom_start; A.comp_atoms = s->comp_atoms; A.comp_local = (const int2 *)s->comp_local;
    A.write_sorted = write_sorted ? 1 : 0; A.shift_s = s->shift_s; A.xq_s = s->xq_s;
    A.settle_first = s->settle_first;
    const int nb16 = (s->ncomp - s->ncomp_small + MDX_FUSED_LARGE_THREADS - 1) / MDX_FUSED_LARGE_THREADS;
    A.nbw3 = (s->nsettle + 41) / 42;
    const int nblocks = A.nbw3 + A.C.nb4 + A.nbfree + nb16;
    A.ke_part = nullptr;
    if (with_ke) {
        if (s->ke_part_cap < nblocks) {
            if (s->ke_part) (void)hipFree(s->ke_part);
            s->ke_part = nullptr; s->ke_part_cap = 0;
            MDX_CHECK(s, hipMalloc((void **)&s->ke_part, sizeof(double) * MDX_MAX_TC * (size_t)nblocks));
            s->ke_part_cap = nblocks;
        }
        A.ke_part = s->ke_part; s->ke_parts = nblocks;
    }
    void (*kern)(FusedArgs) = nb16 > 0 ? (with_ke ? k_update_fused<true, true> : k_update_fused<false, true>)
                                       : (with_ke ? k_update_fused<true, false> : k_update_fused<false, false>);
    if (nblocks > 0) {
        if (fork_signal) {
            hipExtLaunchKernelGGL(kern, dim3(nblocks), dim3(128), 0, s->stream, nullptr, s->ev_fork, 0, A);
            MDX_CHECK(s, hipGetLastError());
            MDX_CHECK(s, hipStreamWaitEvent(s->st_pme, s->ev_fork, 0));
            MDX_CHECK(s, hipStreamWaitEvent(s->st_bond, s->ev_fork, 0));
        } else {
            hipLaunchKernelGGL(kern, dim3(nblocks), dim3(128), 0, s->stream, A);
            MDX_CHECK(s, hipGetLastError());
        }
    }
    float4 *t = s->x; s->x = s->xp; s->xp = t;
    return MDX_OK;
}

// Constrain s->xp against the reference geometry s->x; velocities receive dx * inv_dt
// (inv_dt = 0 leaves them alone, inv_dt = -1/dt serves the start-up velocity projection).
int mdx_launch_constrain(mdx_system *s, float inv_dt, bool virial, int lincs_iter)
{
    if (s->nsettle > 0 && s->ncomp > 0) {
        ConstrainArgs A;
        constrain_args(s, A, inv_dt, lincs_iter);
        const int nb16 = (s->ncomp - s->ncomp_small + 127) / 128;
        void (*kern)(ConstrainArgs) = nb16 > 0 ? (virial ? k_constrain<true, true> : k_constrain<false, true>)
                                               : (virial ? k_constrain<true, false> : k_constrain<false, false>);
        hipLaunchKernelGGL(kern, dim3(A.nbw + A.nb4 + nb16), dim3(128), 0, s->stream, A);
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

int mdx_launch_ke(mdx_system *s, bool old_slot, bool tensor, hipStream_t st, bool tc_slot)
{
    const int nb = std::min((s->n + 255) / 256, 128);
    hipLaunchKernelGGL(k_ke, dim3(nb), dim3(256), 0, st ? st : s->stream, s->n, s->v, s->tcgrp, s->ntc, s->d, tc_slot ? 2 : (old_slot ? 1 : 0), tensor ? 1 : 0);
    MDX_CHECK(s, hipGetLastError());
    return MDX_OK;
}

// Zero the accumulators that the coming step's kernels add into (the body, mdx_clear_accumulators, is shared with the
// prepare kernel, which clears them itself when it is launched for such a step anyway).
__global__ void k_clear(DevState *dw, int energies, int ekin_new, int ekin_old)
{
    mdx_clear_accumulators(dw, threadIdx.x, energies, ekin_new, ekin_old);
}

int mdx_launch_clear(mdx_system *s, bool energies, bool ekin_new, bool ekin_old)
{
    hipLaunchKernelGGL(k_clear, dim3(1), dim3(64), 0, s->stream, s->d, energies ? 1 : 0, ekin_new ? 1 : 0, ekin_old ? 1 : 0);
    MDX_CHECK(s, hipGetLastError());
    return MDX_OK;
}

int mdx_launch_tcouple(mdx_system *s, hipStream_t st, bool tc_slot)
{
    if (s->ntc <= 0) return MDX_OK;
    hipLaunchKernelGGL(k_tcouple, dim3(1), dim3(64), 0, st ? st : s->stream, s->d, s->ntc, s->d_tc, s->d_tc + MDX_MAX_TC,
                       s->d_tc + 2 * MDX_MAX_TC, (double)s->nsttcouple * s->dt, (unsigned long long)s->seed, s->step, tc_slot ? 1 : 0);
    MDX_CHECK(s, hipGetLastError());
    return MDX_OK;
}

// The thermostat's factors from the partial sums of the last fused update with kinetic energy, on stream `st`.
// `step`: the step counter the draw is keyed by (that of the step which ended the coupling period).
int mdx_launch_tcouple_fold(mdx_system *s, hipStream_t st, long long step)
{
    if (s->ntc <= 0 || !s->ke_part) return MDX_OK;
    hipLaunchKernelGGL(k_tcouple_fold, dim3(1), dim3(256), 0, st, s->d, s->ntc, s->d_tc, s->d_tc + MDX_MAX_TC, s->d_tc + 2 * MDX_MAX_TC,
                       (double)s->nsttcouple * s->dt, (unsigned long long)s->seed, step, (const double *)s->ke_part, s->ke_parts);
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
        hipLaunchKernelGGL(k_comm_sum, dim3(std::min(nb, 64)), dim3(256), 0, s->stream, s->n, s->v, s->d);
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
