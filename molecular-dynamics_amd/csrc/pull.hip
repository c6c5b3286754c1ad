// Centre-of-mass pulling between two groups along their distance vector.
// Replaces what `pull = yes`, `pull_coord1_geometry = distance`, `pull_coord1_type`,
// `pull_coord1_k`, `pull_coord1_start`, `pull_coord1_dim` switch on in gmx mdrun
// (/root/reference/mdp/pull_short.mdp:36-46). Supports constant-force (force = -k on the
// distance, i.e. attraction for k > 0, as the committed .mdp asks) and umbrella with a rate
// (the constant-velocity variant of /root/reference/config.yaml:38-39).
// One workgroup: mass-weighted COMs relative to a per-group PBC reference atom (double
// accumulation), the scalar force, its distribution over the atoms, and the (t, x, f, W) record
// that the host turns into PREFIX_pullx.xvg / PREFIX_pullf.xvg.
#include "mdx_internal.h"
#include "pull_body.h"

__global__ __launch_bounds__(256) void k_pull(PullArgs P, const float4 *__restrict__ x, const float4 *__restrict__ v,
                                              float *__restrict__ f, DevState *dw)
{
    pull_block(P, x, v, f, dw);
}

// The arguments of this step's pull; log_now: the step writes a record (the caller counts it: s->pull_log_count).
void mdx_pull_args(mdx_system *s, bool log_now, bool energy, PullArgs &P)
{
    P.n1 = s->pull_n[0]; P.n2 = s->pull_n[1]; P.g1 = s->pull_idx[0]; P.g2 = s->pull_idx[1];
    P.pbc1 = s->pull_pbcatom[0]; P.pbc2 = s->pull_pbcatom[1];
    P.type = s->pull_type; P.k = s->pull_k; P.rate = s->pull_rate; P.init_user = s->pull_init_user; P.start = s->pull_start;
    for (int d = 0; d < 3; d++) P.dim[d] = s->pull_dim[d];
    P.time = (double)s->step * s->dt; P.dt = s->dt;
    P.log = s->pull_log; P.log_cap = s->pull_log_cap; P.log_slot = s->pull_log_count; P.log_now = log_now ? 1 : 0;
    P.energy = energy ? 1 : 0;
}

int mdx_launch_pull(mdx_system *s, bool log_now, bool energy)
{
    if (!s->pull_on) return MDX_OK;
    PullArgs P;
    mdx_pull_args(s, log_now, energy, P);
    hipLaunchKernelGGL(k_pull, dim3(1), dim3(256), 0, s->st_bond, P, s->x, s->v, s->f, s->d);
    MDX_CHECK(s, hipGetLastError());
    if (log_now) s->pull_log_count++;
    return MDX_OK;
}
