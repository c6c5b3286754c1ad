// Device code of the bonded launch (bonds, angles, dihedrals, 1-4 pairs, position restraints): included by bonded.hip, whose
// kernel runs it on the bonded stream, and by pme.hip, whose spreading launch carries it along on the steps without energies.
#pragma once
#include "mdx_internal.h"

// Forces of a workgroup's interactions are summed in LDS first. The interactions of a topology come in atom order, so
// the 256 interactions of a workgroup touch a short range of atoms; a window of MDX_BWIN atoms starting at the lowest
// atom index of the workgroup collects them (LDS float atomics), and only its non-zero entries go to the force array
// with global atomics: a tenth of the global atomics, and far fewer of them on one address. Global float atomics are
// carried out at the memory side, where these (75 per protein atom before) also held up the flush of the charge
// spreading that runs at the same time (spreading 53 -> 30 us beside it). Atoms outside the window are added directly.
#define MDX_BWIN 1024
struct ForceWin {
    float *win; int base; float *f;
    __device__ __forceinline__ void add(int i, float3 v) const
    {
        const unsigned int k = (unsigned int)(i - base);
        if (k < (unsigned int)MDX_BWIN) { atomicAdd(&win[3 * k + 0], v.x); atomicAdd(&win[3 * k + 1], v.y); atomicAdd(&win[3 * k + 2], v.z); }
        else { atomicAdd(&f[3 * i + 0], v.x); atomicAdd(&f[3 * i + 1], v.y); atomicAdd(&f[3 * i + 2], v.z); }
    }
};
__device__ __forceinline__ void add_force(const ForceWin &f, int i, float3 v) { f.add(i, v); }

// Block-reduce an energy and a 3x3 virial contribution (sum d (x) F, not yet scaled by -1/2).
__device__ inline void accum_ev(double e, const float *v9, DevState *dw, int eidx, double e2 = 0.0, int eidx2 = -1)
{
    // two energies and the nine virial sums in one reduction; thread k adds the k-th
    __shared__ double red[16 * 11];
    double all[11];
    all[0] = e; all[1] = e2;
#pragma unroll
    for (int k = 0; k < 9; k++) all[2 + k] = (double)v9[k];
    const double r = block_sum_n<11>(all, red);
    const int t = threadIdx.x;
    if (r == 0.0) return;
    if (t == 0) atomicAdd(&MDX_ENER(dw, eidx), r);
    else if (t == 1) { if (eidx2 >= 0) atomicAdd(&MDX_ENER(dw, eidx2), r); }
    else if (t < 11) atomicAdd(&MDX_VIR(dw, t - 2), -0.5 * r);
}

__device__ __forceinline__ void outer_acc(float *v9, float3 d, float3 f)
{
    v9[0] += d.x * f.x; v9[1] += d.x * f.y; v9[2] += d.x * f.z;
    v9[3] += d.y * f.x; v9[4] += d.y * f.y; v9[5] += d.y * f.z;
    v9[6] += d.z * f.x; v9[7] += d.z * f.y; v9[8] += d.z * f.z;
}

template <bool ENERGY>
__device__ __forceinline__ void d_bonds(int t, int n, const int2 *__restrict__ ij, const float2 *__restrict__ p,
                                               const float4 *__restrict__ x, const ForceWin &f, DevState *dw)
{
    double e = 0.0; float v9[9] = { 0 };
    if (t < n) {
        const int2 a = ij[t]; const float2 par = p[t];
        const float3 d = xyz(x[a.x]) - xyz(x[a.y]);
        const float r = sqrtf(dot3(d, d));
        const float dr = r - par.x;
        const float3 fi = (-par.y * dr / r) * d;
        add_force(f, a.x, fi); add_force(f, a.y, -1.0f * fi);
        if (ENERGY) { e = 0.5 * par.y * dr * dr; outer_acc(v9, d, fi); }
    }
    if (ENERGY) accum_ev(e, v9, dw, MDX_E_BOND);
}

template <bool ENERGY>
__device__ __forceinline__ void d_angles(int t, int n, const int4 *__restrict__ ijk, const float2 *__restrict__ p,
                                                const float4 *__restrict__ x, const ForceWin &f, DevState *dw)
{
    double e = 0.0; float v9[9] = { 0 };
    if (t < n) {
        const int4 a = ijk[t]; const float2 par = p[t]; // theta0 (rad), k
        const float3 rij = xyz(x[a.x]) - xyz(x[a.y]);
        const float3 rkj = xyz(x[a.z]) - xyz(x[a.y]);
        const float nij2 = dot3(rij, rij), nkj2 = dot3(rkj, rkj);
        const float inv = rsqrtf(nij2 * nkj2);
        float c = dot3(rij, rkj) * inv;
        c = fminf(1.0f, fmaxf(-1.0f, c));
        const float th = acosf(c);
        const float dth = th - par.x;
        const float st = sqrtf(fmaxf(1.0f - c * c, 1e-12f));
        const float pre = par.y * dth / st;
        const float3 fi = pre * (inv * rkj - (c / nij2) * rij);
        const float3 fk = pre * (inv * rij - (c / nkj2) * rkj);
        add_force(f, a.x, fi); add_force(f, a.z, fk); add_force(f, a.y, -1.0f * (fi + fk));
        if (ENERGY) { e = 0.5 * par.y * dth * dth; outer_acc(v9, rij, fi); outer_acc(v9, rkj, fk); }
    }
    if (ENERGY) accum_ev(e, v9, dw, MDX_E_ANGLE);
}

template <bool ENERGY>
__device__ __forceinline__ void d_dihedrals(int t, int n, const int4 *__restrict__ ijkl, const float4 *__restrict__ p,
                                                   const float4 *__restrict__ x, const ForceWin &f, DevState *dw)
{
    double e = 0.0; float v9[9] = { 0 };
    if (t < n) {
        const int4 a = ijkl[t]; const float4 par = p[t]; // phi0 (rad), k, mult
        const float3 xj = xyz(x[a.y]);
        const float3 rij = xyz(x[a.x]) - xj;
        const float3 rkj = xyz(x[a.z]) - xj;
        const float3 rkl = xyz(x[a.z]) - xyz(x[a.w]);
        const float3 m = cross3(rij, rkj), nn = cross3(rkj, rkl);
        const float nrkj2 = dot3(rkj, rkj), nrkj = sqrtf(nrkj2);
        const float phi = atan2f(nrkj * dot3(rij, nn), dot3(m, nn));
        const float mult = par.z;
        const float arg = mult * phi - par.x;
        float sa, ca;
        sincosf(arg, &sa, &ca);
        const float ddphi = -par.y * mult * sa;
        const float iprm = dot3(m, m), iprn = dot3(nn, nn);
        const float3 fi = (-ddphi * nrkj / iprm) * m;
        const float3 fl = (ddphi * nrkj / iprn) * nn;
        const float pp = dot3(rij, rkj) / nrkj2, qq = dot3(rkl, rkj) / nrkj2;
        const float3 sv = pp * fi - qq * fl;
        const float3 fj = fi - sv, fk = fl + sv;
        add_force(f, a.x, fi); add_force(f, a.y, -1.0f * fj); add_force(f, a.z, -1.0f * fk); add_force(f, a.w, fl);
        if (ENERGY) {
            e = par.y * (1.0 + ca);
            // positions relative to atom j
            outer_acc(v9, rij, fi); outer_acc(v9, rkj, -1.0f * fk); outer_acc(v9, rkj - rkl, fl);
        }
    }
    if (ENERGY) accum_ev(e, v9, dw, MDX_E_DIH);
}

template <bool ENERGY>
__device__ __forceinline__ void d_pairs14(int t, int n, const int2 *__restrict__ ij, const float4 *__restrict__ p,
                                                 const float4 *__restrict__ x, const ForceWin &f, DevState *dw)
{
    double e = 0.0, e2 = 0.0; float v9[9] = { 0 };
    if (t < n) {
        const int2 a = ij[t]; const float4 par = p[t]; // c6, c12, qq*ONE_4PI_EPS0
        const float3 d = xyz(x[a.x]) - xyz(x[a.y]);
        const float rinv2 = 1.0f / dot3(d, d);
        const float rinv6 = rinv2 * rinv2 * rinv2;
        const float vlj12 = par.y * rinv6 * rinv6, vlj6 = par.x * rinv6;
        const float vc = par.z * sqrtf(rinv2);
        const float3 fi = ((12.0f * vlj12 - 6.0f * vlj6 + vc) * rinv2) * d;
        add_force(f, a.x, fi); add_force(f, a.y, -1.0f * fi);
        if (ENERGY) { e = vlj12 - vlj6; e2 = vc; outer_acc(v9, d, fi); }
    }
    if (ENERGY) accum_ev(e, v9, dw, MDX_E_LJ14, e2, MDX_E_COUL14);
}

template <bool ENERGY>
__device__ __forceinline__ void d_posres(int t, int n, const int *__restrict__ idx, const float4 *__restrict__ ref,
                                                const float4 *__restrict__ kk, const float4 *__restrict__ x,
                                                const ForceWin &f, DevState *dw, int refcoord_scaling)
{
    double e = 0.0; float v9[9] = { 0 };
    if (t < n) {
        const int a = idx[t];
        const float4 r = ref[t], k = kk[t];
        float3 d = xyz(x[a]) - xyz(r);
        d = mdx_min_image(dw->box, d);
        const float3 fi = make_float3(-k.x * d.x, -k.y * d.y, -k.z * d.z);
        add_force(f, a, fi);
        if (ENERGY) {
            e = 0.5 * (k.x * d.x * d.x + k.y * d.y * d.y + k.z * d.z * d.z);
            // The virial arm is the displacement plus the part of the reference that does not scale with the box:
            // nothing (`all`), the reference relative to the restraint centre (`com`), the whole reference (`no`).
            float3 arm = d;
            if (refcoord_scaling == 0) arm = d + xyz(r);
            else if (refcoord_scaling == 2)
                arm = d + xyz(r) - make_float3((float)dw->posres_com[0], (float)dw->posres_com[1], (float)dw->posres_com[2]);
            outer_acc(v9, arm, fi);
        }
    }
    if (ENERGY) accum_ev(e, v9, dw, MDX_E_POSRES);
}

// All bonded-type terms in one launch: consecutive ranges of workgroups serve bonds, angles, dihedrals,
// 1-4 pairs and position restraints (a workgroup is homogeneous, so the energy reductions stay per type).
// Five dependent launches of a few thousand threads each cost several times their arithmetic.
struct BondedArgs {
    int nb[5];                  // first workgroup of each type's range, then the end: nb[k] .. nb[k+1] (nb[5] = total in `nblocks`)
    int nblocks;
    int nbonds; const int2 *bond_ij; const float2 *bond_p;
    int nangles; const int4 *angle_ijk; const float2 *angle_p;
    int ndih; const int4 *dih_ijkl; const float4 *dih_p;
    int npairs; const int2 *pair_ij; const float4 *pair_p;
    int nposres; const int *posres_idx; const float4 *posres_ref; const float4 *posres_k; int refcoord_scaling;
    const float4 *x; float *f; DevState *dw;
};

// One workgroup's share of the bonded launch: block `b` of A's ranges, 256 threads; `win` holds 3 MDX_BWIN floats of LDS and
// `s_base` one int (the kernel's own arrays, or a piece of another kernel's LDS when the bonded work rides in its launch).
template <bool ENERGY>
__device__ __forceinline__ void bonded_block(const BondedArgs &A, const int b, float *win, int *s_base_p)
{
    int &s_base = *s_base_p;
    // lowest atom index among the workgroup's interactions (first index of each tuple: tuples are written in ascending order
    // of their atoms often enough; the others only miss the window and take the direct path)
    if (threadIdx.x == 0) s_base = 0x7fffffff;
    for (int k = threadIdx.x; k < 3 * MDX_BWIN; k += 256) win[k] = 0.0f;
    __syncthreads();
    int t, lo = 0x7fffffff;
    if (b < A.nb[1]) { t = (b - A.nb[0]) * 256 + threadIdx.x; if (t < A.nbonds) { const int2 a = A.bond_ij[t]; lo = min(a.x, a.y); } }
    else if (b < A.nb[2]) { t = (b - A.nb[1]) * 256 + threadIdx.x; if (t < A.nangles) { const int4 a = A.angle_ijk[t]; lo = min(a.x, min(a.y, a.z)); } }
    else if (b < A.nb[3]) { t = (b - A.nb[2]) * 256 + threadIdx.x; if (t < A.ndih) { const int4 a = A.dih_ijkl[t]; lo = min(min(a.x, a.y), min(a.z, a.w)); } }
    else if (b < A.nb[4]) { t = (b - A.nb[3]) * 256 + threadIdx.x; if (t < A.npairs) { const int2 a = A.pair_ij[t]; lo = min(a.x, a.y); } }
    else { t = (b - A.nb[4]) * 256 + threadIdx.x; if (t < A.nposres) lo = A.posres_idx[t]; }
    for (int o = 32; o > 0; o >>= 1) lo = min(lo, __shfl_xor(lo, o, 64));
    if ((threadIdx.x & 63) == 0 && lo != 0x7fffffff) atomicMin(&s_base, lo);
    __syncthreads();
    ForceWin F; F.win = win; F.base = s_base; F.f = A.f;
    if (b < A.nb[1]) d_bonds<ENERGY>(t, A.nbonds, A.bond_ij, A.bond_p, A.x, F, A.dw);
    else if (b < A.nb[2]) d_angles<ENERGY>(t, A.nangles, A.angle_ijk, A.angle_p, A.x, F, A.dw);
    else if (b < A.nb[3]) d_dihedrals<ENERGY>(t, A.ndih, A.dih_ijkl, A.dih_p, A.x, F, A.dw);
    else if (b < A.nb[4]) d_pairs14<ENERGY>(t, A.npairs, A.pair_ij, A.pair_p, A.x, F, A.dw);
    else d_posres<ENERGY>(t, A.nposres, A.posres_idx, A.posres_ref, A.posres_k, A.x, F, A.dw, A.refcoord_scaling);
    __syncthreads();
    if (F.base != 0x7fffffff) {
        float *dst = A.f + 3 * (size_t)F.base;
        for (int k = threadIdx.x; k < 3 * MDX_BWIN; k += 256) {
            const float v = win[k];
            if (v != 0.0f) atomicAdd(&dst[k], v);
        }
    }
}
