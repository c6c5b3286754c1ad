// Bonded interactions, 1-4 pairs, position restraints and the Ewald excluded-pair correction.
// One thread per interaction; forces are added with float atomics to the original-order force
// array. Functional forms follow the AMBER-in-GROMACS .top dialect the pipeline assembles
// (/root/reference/scripts/create_complex_topology.py:65-122; fix_ligand_topology.py:88-96):
//   bonds funct 1, angles funct 1, dihedrals funct 1/4/9, pairs funct 1, position_restraints funct 1.
// State coordinates are kept unwrapped with whole molecules, so no PBC is needed inside a
// molecule; restraints and exclusion pairs use the triclinic minimum image.
#include "mdx_internal.h"

__device__ __forceinline__ void add_force(float *f, int i, float3 v)
{
    atomicAdd(&f[3 * i + 0], v.x); atomicAdd(&f[3 * i + 1], v.y); atomicAdd(&f[3 * i + 2], v.z);
}

// Block-reduce an energy and a 3x3 virial contribution (sum d (x) F, not yet scaled by -1/2).
__device__ void accum_ev(double e, const float *v9, DevState *dw, int eidx, double e2 = 0.0, int eidx2 = -1)
{
    __shared__ double red[16];
    double r = block_sum_d(e, red);
    if (threadIdx.x == 0 && r != 0.0) atomicAdd(&MDX_ENER(dw, eidx), r);
    if (eidx2 >= 0) {
        r = block_sum_d(e2, red);
        if (threadIdx.x == 0 && r != 0.0) atomicAdd(&MDX_ENER(dw, eidx2), r);
    }
    for (int k = 0; k < 9; k++) {
        r = block_sum_d((double)v9[k], red);
        if (threadIdx.x == 0 && r != 0.0) atomicAdd(&MDX_VIR(dw, k), -0.5 * r);
    }
}

__device__ __forceinline__ void outer_acc(float *v9, float3 d, float3 f)
{
    v9[0] += d.x * f.x; v9[1] += d.x * f.y; v9[2] += d.x * f.z;
    v9[3] += d.y * f.x; v9[4] += d.y * f.y; v9[5] += d.y * f.z;
    v9[6] += d.z * f.x; v9[7] += d.z * f.y; v9[8] += d.z * f.z;
}

template <bool ENERGY>
__global__ __launch_bounds__(256) void k_bonds(int n, const int2 *__restrict__ ij, const float2 *__restrict__ p,
                                               const float4 *__restrict__ x, float *__restrict__ f, DevState *dw)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
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
__global__ __launch_bounds__(256) void k_angles(int n, const int4 *__restrict__ ijk, const float2 *__restrict__ p,
                                                const float4 *__restrict__ x, float *__restrict__ f, DevState *dw)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
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
__global__ __launch_bounds__(256) void k_dihedrals(int n, const int4 *__restrict__ ijkl, const float4 *__restrict__ p,
                                                   const float4 *__restrict__ x, float *__restrict__ f, DevState *dw)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
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
__global__ __launch_bounds__(256) void k_pairs14(int n, const int2 *__restrict__ ij, const float4 *__restrict__ p,
                                                 const float4 *__restrict__ x, float *__restrict__ f, DevState *dw)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
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
__global__ __launch_bounds__(256) void k_posres(int n, const int *__restrict__ idx, const float4 *__restrict__ ref,
                                                const float4 *__restrict__ kk, const float4 *__restrict__ x,
                                                float *__restrict__ f, DevState *dw)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    double e = 0.0; float v9[9] = { 0 };
    if (t < n) {
        const int a = idx[t];
        const float4 r = ref[t], k = kk[t];
        float3 d = xyz(x[a]) - xyz(r);
        d = mdx_min_image(dw->box, d);
        const float3 fi = make_float3(-k.x * d.x, -k.y * d.y, -k.z * d.z);
        add_force(f, a, fi);
        if (ENERGY) { e = 0.5 * (k.x * d.x * d.x + k.y * d.y * d.y + k.z * d.z * d.z); outer_acc(v9, d, fi); }
    }
    if (ENERGY) accum_ev(e, v9, dw, MDX_E_POSRES);
}

#define LAUNCH_EV(kern, n, ...)                                                                     \
    if ((n) > 0) {                                                                                  \
        const int nb_ = ((n) + 255) / 256;                                                          \
        if (energy) hipLaunchKernelGGL(kern<true>, dim3(nb_), dim3(256), 0, (s->bond_on_main ? s->stream : s->st_bond), n, __VA_ARGS__); \
        else hipLaunchKernelGGL(kern<false>, dim3(nb_), dim3(256), 0, (s->bond_on_main ? s->stream : s->st_bond), n, __VA_ARGS__);   \
    }

int mdx_launch_bonded(mdx_system *s, bool energy, unsigned flags)
{
    if (flags & MDX_F_BONDED) {
        LAUNCH_EV(k_bonds, s->nbonds, s->bond_ij, s->bond_p, s->x, s->f, s->d);
        LAUNCH_EV(k_angles, s->nangles, s->angle_ijk, s->angle_p, s->x, s->f, s->d);
        LAUNCH_EV(k_dihedrals, s->ndih, s->dih_ijkl, s->dih_p, s->x, s->f, s->d);
        LAUNCH_EV(k_pairs14, s->npairs, s->pair_ij, s->pair_p, s->x, s->f, s->d);
    }
    if (flags & MDX_F_POSRES) {
        LAUNCH_EV(k_posres, s->nposres, s->posres_idx, s->posres_ref, s->posres_k, s->x, s->f, s->d);
    }
    // the Ewald correction of excluded pairs is evaluated by the nonbonded kernel on the pair list
    MDX_CHECK(s, hipGetLastError());
    return MDX_OK;
}
