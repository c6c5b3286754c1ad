// Bonded interactions, 1-4 pairs, position restraints and the Ewald excluded-pair correction.
// One thread per interaction; forces are added with float atomics to the original-order force
// array. Functional forms follow the AMBER-in-GROMACS .top dialect the pipeline assembles
// (/root/reference/scripts/create_complex_topology.py:65-122; fix_ligand_topology.py:88-96):
//   bonds funct 1, angles funct 1, dihedrals funct 1/4/9, pairs funct 1, position_restraints funct 1.
// State coordinates are kept unwrapped with whole molecules, so no PBC is needed inside a
// molecule; restraints and exclusion pairs use the triclinic minimum image.
#include "mdx_internal.h"
#include "bonded_body.h"

template <bool ENERGY>
__global__ __launch_bounds__(256) void k_bonded(BondedArgs A)
{
    MDX_STAMP(A.dw, 10);
    __shared__ float win[3 * MDX_BWIN];
    __shared__ int s_base;
    bonded_block<ENERGY>(A, blockIdx.x, win, &s_base);
}

// The launch's arguments for the terms in `flags` (MDX_F_BONDED, MDX_F_POSRES); A.nblocks == 0: nothing to do.
void mdx_bonded_args(mdx_system *s, unsigned flags, BondedArgs &A)
{
    const bool bo = (flags & MDX_F_BONDED) != 0, pr = (flags & MDX_F_POSRES) != 0;
    A.nbonds = bo ? s->nbonds : 0; A.bond_ij = s->bond_ij; A.bond_p = s->bond_p;
    A.nangles = bo ? s->nangles : 0; A.angle_ijk = s->angle_ijk; A.angle_p = s->angle_p;
    A.ndih = bo ? s->ndih : 0; A.dih_ijkl = s->dih_ijkl; A.dih_p = s->dih_p;
    A.npairs = bo ? s->npairs : 0; A.pair_ij = s->pair_ij; A.pair_p = s->pair_p;
    A.nposres = pr ? s->nposres : 0; A.posres_idx = s->posres_idx; A.posres_ref = s->posres_ref; A.posres_k = s->posres_k;
    A.refcoord_scaling = s->refcoord_scaling;
    A.x = s->x; A.f = s->f; A.dw = s->d;
    const int cnt[5] = { A.nbonds, A.nangles, A.ndih, A.npairs, A.nposres };
    int acc = 0;
    for (int k = 0; k < 5; k++) { A.nb[k] = acc; acc += (cnt[k] + 255) / 256; }
    A.nblocks = acc;
}

int mdx_launch_bonded(mdx_system *s, bool energy, unsigned flags)
{
    BondedArgs A;
    mdx_bonded_args(s, flags, A);
    const int acc = A.nblocks;
    if (acc == 0) return MDX_OK;
    hipStream_t st = s->st_bond;
    if (energy) hipLaunchKernelGGL(k_bonded<true>, dim3(acc), dim3(256), 0, st, A);
    else hipLaunchKernelGGL(k_bonded<false>, dim3(acc), dim3(256), 0, st, A);
    // the Ewald correction of excluded pairs is evaluated by the nonbonded kernel on the pair list
    MDX_CHECK(s, hipGetLastError());
    return MDX_OK;
}
