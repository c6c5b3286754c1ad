// Spatial domain decomposition of one large system over several GPUs (one process and one handle per GPU).
//
// Replaces what gmx mdrun's domain decomposition would do for a box too large for one device; the pipeline
// itself only runs single-rank mdrun (/root/reference/Snakefile:232-233: -ntmpi 1), so this path serves the
// scaled synthetic water box. A handle holds `ncap` atom slots filled with whole molecules of one kind
// (the per-slot mass, charge and LJ pattern given at set-up repeats): first the molecules this domain owns
// ("home"), then copies of the molecules of other domains within the pair-list radius of it ("halo"),
// then unused slots. Halo atoms have inverse mass zero: they feel no update, take part in the short-range
// kernels as partners of home atoms, and are skipped by the charge spreading (their owner spreads them).
// Every periodic image is handled by the handle's own box, which is the box of the whole system.
//   mdx_dd_load      install a new partition (home + halo state), every pair-list lifetime
//   mdx_dd_pack      gather the positions of home atoms that are halo atoms elsewhere into a send buffer
//   mdx_dd_unpack    scatter received halo positions into the halo slots
//   mdx_dd_step_begin / mdx_dd_wait_spread / mdx_dd_step_end
//                    a step in two halves; between them the caller sums the charge grids of all domains
//                    (mdx_dd_grid gives the device pointer) so that every domain solves the full mesh problem
//   mdx_dd_mesh_slabs / mdx_dd_cgrid / mdx_dd_step_mid
//                    the mesh problem shared instead: every domain transforms its own planes of the half-complex grid
#include "mdx_internal.h"

int mdx_step_begin_internal(mdx_system *s);
int mdx_step_end_internal(mdx_system *s);
int mdx_step_mid_internal(mdx_system *s);

__global__ void k_dd_load(int ncap, int n_home, int n_total, const float4 *__restrict__ xin, const float4 *__restrict__ vin,
                          const float *__restrict__ q0, const float *__restrict__ invm0, const float2 *__restrict__ lj0,
                          float4 *__restrict__ x, float4 *__restrict__ v, float2 *__restrict__ lj)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncap) return;
    if (i < n_total) {
        const float4 p = xin[i];
        x[i] = make_float4(p.x, p.y, p.z, q0[i]);
        float4 w = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (i < n_home) { w = vin[i]; w.w = invm0[i]; }
        v[i] = w;
        // a halo atom carries a minus sign on 2 sqrt(eps) (also on a zero): the pair kernel takes the magnitude and, on
        // energy steps, reads the sign as "not owned here"
        const float2 l = lj0[i];
        lj[i] = make_float2(l.x, i < n_home ? l.y : -fabsf(l.y));
    } else {
        // unused slot: no charge, no LJ well, parked on top of a real atom (the kernels keep r = 0 finite)
        const float4 p = xin[i % n_total];
        x[i] = make_float4(p.x, p.y, p.z, 0.0f);
        v[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        lj[i] = make_float2(lj0[i].x, -0.0f);
    }
}

__global__ void k_dd_pack(int n, const int *__restrict__ idx, const float4 *__restrict__ x, float4 *__restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = x[idx[k]];
}

__global__ void k_dd_unpack(int n, int first, const float4 *__restrict__ in, float4 *__restrict__ x)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float4 p = in[k];
    const float q = x[first + k].w;
    x[first + k] = make_float4(p.x, p.y, p.z, q);
}

extern "C" int mdx_dd_load(mdx_system *s, int n_home, int n_total, const void *x4_dev, const void *v4_dev, int n_settle,
                           double ndf, double density)
{
    (void)hipSetDevice(s->device);
    if (!(n_home >= 0 && n_home <= n_total && n_total >= 1 && n_total <= s->n) || !x4_dev || !v4_dev) {
        s->err = "mdx_dd_load: need 0 <= n_home <= n_total <= capacity, n_total >= 1 and device buffers"; return MDX_ERR_ARG;
    }
    if (n_settle < 0 || n_settle > s->nsettle_cap) { s->err = "mdx_dd_load: more settles than were set up"; return MDX_ERR_ARG; }
    if (s->ncons > 0 || s->nbonds > 0 || s->nangles > 0 || s->ndih > 0 || s->npairs > 0 || s->nposres > 0 || s->pull_on) {
        s->err = "mdx_dd_load: the decomposed path carries rigid molecules of one kind only (no bonded terms, constraints or pull)";
        return MDX_ERR_UNSUPPORTED;
    }
    if (!s->dd_q0) {
        // the per-slot pattern as it was set up: charge, inverse mass, LJ parameters
        MDX_CHECK(s, hipMalloc((void **)&s->dd_q0, sizeof(float) * s->n));
        MDX_CHECK(s, hipMalloc((void **)&s->dd_invm0, sizeof(float) * s->n));
        MDX_CHECK(s, hipMalloc((void **)&s->dd_lj0, sizeof(float2) * s->n));
        MDX_CHECK(s, hipMemcpy(s->dd_q0, s->host_atoms.q.data(), sizeof(float) * s->n, hipMemcpyHostToDevice));
        MDX_CHECK(s, hipMemcpy(s->dd_invm0, s->host_atoms.invm.data(), sizeof(float) * s->n, hipMemcpyHostToDevice));
        MDX_CHECK(s, hipMemcpy(s->dd_lj0, s->lj, sizeof(float2) * s->n, hipMemcpyDeviceToDevice));
    }
    hipLaunchKernelGGL(k_dd_load, dim3((s->n + 255) / 256), dim3(256), 0, s->stream, s->n, n_home, n_total, (const float4 *)x4_dev,
                       (const float4 *)v4_dev, (const float *)s->dd_q0, (const float *)s->dd_invm0, (const float2 *)s->dd_lj0, s->x, s->v, s->lj);
    MDX_CHECK(s, hipGetLastError());
    s->dd_on = true;
    // Ewald self term of the atoms this domain owns
    s->q2sum = 0.0;
    for (int i = 0; i < n_home; i++) s->q2sum += (double)s->host_atoms.q[i] * s->host_atoms.q[i];
    s->nsettle = n_settle;
    s->search_density = density;
    s->steps_since_search = 1 << 30;
    s->forces_current = false;
    s->ndf_total = ndf;
    if (s->ntc == 1) {
        s->ndf[0] = ndf;
        MDX_CHECK(s, hipMemcpyAsync(s->d_tc + 2 * MDX_MAX_TC, &s->ndf[0], sizeof(double), hipMemcpyHostToDevice, s->stream));
    } else if (s->ntc > 1) { s->err = "mdx_dd_load: one temperature-coupling group at most"; return MDX_ERR_UNSUPPORTED; }
    return MDX_OK;
}

extern "C" int mdx_dd_pack(mdx_system *s, int n, const int *idx_dev, void *out4_dev)
{
    (void)hipSetDevice(s->device);
    if (n <= 0) return MDX_OK;
    hipLaunchKernelGGL(k_dd_pack, dim3((n + 255) / 256), dim3(256), 0, s->stream, n, idx_dev, (const float4 *)s->x, (float4 *)out4_dev);
    MDX_CHECK(s, hipGetLastError());
    return MDX_OK;
}

extern "C" int mdx_dd_unpack(mdx_system *s, int first, int n, const void *in4_dev)
{
    (void)hipSetDevice(s->device);
    if (n <= 0) return MDX_OK;
    if (first < 0 || first + n > s->n) { s->err = "mdx_dd_unpack: slot range outside the handle"; return MDX_ERR_ARG; }
    hipLaunchKernelGGL(k_dd_unpack, dim3((n + 255) / 256), dim3(256), 0, s->stream, n, first, (const float4 *)in4_dev, s->x);
    MDX_CHECK(s, hipGetLastError());
    s->forces_current = false;
    return MDX_OK;
}

extern "C" int mdx_dd_step_begin(mdx_system *s)
{
    (void)hipSetDevice(s->device);
    if (!s->dd_on) { s->err = "mdx_dd_step_begin: no partition loaded"; return MDX_ERR_ARG; }
    return mdx_step_begin_internal(s);
}

// Blocks until the charge spreading of the step begun last has finished (the grid may then be summed).
extern "C" int mdx_dd_wait_spread(mdx_system *s)
{
    (void)hipSetDevice(s->device);
    if (s->have_pme) MDX_CHECK(s, hipEventSynchronize(s->ev_spread));
    return MDX_OK;
}

extern "C" int mdx_dd_step_end(mdx_system *s)
{
    (void)hipSetDevice(s->device);
    if (!s->dd_on) { s->err = "mdx_dd_step_end: no partition loaded"; return MDX_ERR_ARG; }
    return mdx_step_end_internal(s);
}

// The mesh transform shared among `world` domains. The x passes of the 3D transform act on lines of one (y, z) each, and a
// sum over domains commutes with them, so a domain can transform its own partial charge grid along x first; what the domains
// then have to add up is the half-complex grid, stored as k_x planes of K2 x K1 values (y fastest), and each plane is a
// self-contained problem for the y and z transforms and the influence function. With this call
//   mdx_dd_step_begin   runs on through the x forward pass;
//   the caller          reduce-scatters the grid of mdx_dd_cgrid: `world` equal chunks of `planes_per_domain` planes, chunk r
//                       summed onto domain r (planes past K0 / 2 stay zero);
//   mdx_dd_step_mid     transforms the planes of this domain's chunk (y, z, influence function, z and y back);
//   the caller          all-gathers the chunks;
//   mdx_dd_step_end     starts at the x inverse pass.
// mdx_dd_wait_spread blocks until the stage begun last has finished. The reciprocal-space energy and virial of a domain
// are those of its planes and add up over the domains (mesh_energy of mdx_dd_set_roles on for every domain).
// Returns MDX_ERR_UNSUPPORTED for a grid whose planes the fused kernel cannot take (a dimension that does not split into
// two factors <= 10, or a plane beyond the LDS budget): the caller then keeps the sum of the real-space grids.
// world = 0 returns to that scheme.
extern "C" int mdx_dd_mesh_slabs(mdx_system *s, int world, int rank)
{
    (void)hipSetDevice(s->device);
    if (!s->have_pme) { s->err = "mdx_dd_mesh_slabs: PME is not set up"; return MDX_ERR_ARG; }
    if (world < 0 || (world > 0 && (rank < 0 || rank >= world))) { s->err = "mdx_dd_mesh_slabs: need 0 <= rank < world"; return MDX_ERR_ARG; }
    if (world > 0 && !(s->own_fft && s->fft_yz)) { s->err = "mdx_dd_mesh_slabs: this grid has no plane-wise transform"; return MDX_ERR_UNSUPPORTED; }
    MDX_CHECK(s, hipDeviceSynchronize());
    const int K0c = s->K[0] / 2 + 1;
    s->dd_mesh_world = world;
    s->dd_plane_chunk = world > 0 ? (K0c + world - 1) / world : 0;
    s->dd_plane_first = world > 0 ? rank * s->dd_plane_chunk : 0;
    s->dd_plane_count = world > 0 ? std::max(0, std::min(K0c, s->dd_plane_first + s->dd_plane_chunk) - s->dd_plane_first) : 0;
    return mdx_pme_init(s);         // (the half-complex grid is sized by the chunks)
}

// Device pointer of the half-complex grid (float2, plane-major: plane k_x at k_x * plane_elems), the number of planes in
// a domain's chunk and the values per plane (K1 * K2).
extern "C" int mdx_dd_cgrid(mdx_system *s, void **cgrid_dev, int *planes_per_domain, long long *plane_elems)
{
    if (!s->have_pme || s->dd_mesh_world <= 0) { s->err = "mdx_dd_cgrid: call mdx_dd_mesh_slabs first"; return MDX_ERR_ARG; }
    if (cgrid_dev) *cgrid_dev = s->cgrid;
    if (planes_per_domain) *planes_per_domain = s->dd_plane_chunk;
    if (plane_elems) *plane_elems = (long long)s->K[1] * s->K[2];
    return MDX_OK;
}

extern "C" int mdx_dd_step_mid(mdx_system *s)
{
    (void)hipSetDevice(s->device);
    if (!s->dd_on) { s->err = "mdx_dd_step_mid: no partition loaded"; return MDX_ERR_ARG; }
    return mdx_step_mid_internal(s);
}

// Which parts of a step this domain's handle does on its own (default: all of them, as a single domain needs):
//   mesh_energy 0: the reciprocal-space energy and virial are not accumulated here (every domain solves the same mesh
//                  problem; one of them reports it);
//   own_tcouple 0: mdx_dd_step_end leaves the thermostat factor alone; the caller sums the half-step kinetic energies of
//                  all domains into the scalars of mdx_dd_scalars and calls mdx_dd_tcouple, so that every domain scales
//                  by the same factor (same seed on every domain).
extern "C" int mdx_dd_set_roles(mdx_system *s, int mesh_energy, int own_tcouple)
{
    s->dd_mesh_energy = mesh_energy != 0;
    s->dd_own_tcouple = own_tcouple != 0;
    return MDX_OK;
}

// Device pointers of the scalars the domains have to agree on: the half-step kinetic energy per temperature-coupling
// group (MDX_MAX_TC doubles) and the momentum sum of the centre-of-mass removal (4 doubles: m v and m).
extern "C" int mdx_dd_scalars(mdx_system *s, void **ekinh_dev, void **com_dev)
{
    if (ekinh_dev) *ekinh_dev = (char *)s->d + offsetof(DevState, ekinh);
    if (com_dev) *com_dev = (char *)s->d + offsetof(DevState, com_v);
    return MDX_OK;
}

// The thermostat factor for the coming steps from the (summed) kinetic energy in the handle's scalars.
extern "C" int mdx_dd_tcouple(mdx_system *s, int before_first_step)
{
    (void)hipSetDevice(s->device);
    // the random stream of the factor is keyed as inside mdx_step: by the step just completed, or by the step about to
    // start for the factor drawn ahead of the first one
    if (!before_first_step) s->step--;
    const int rc = mdx_launch_tcouple(s);
    if (!before_first_step) s->step++;
    return rc;
}

// Pressure coupling across the domains (C-rescale): on a coupling step every domain's step leaves its share of the
// pressure tensor in the scalars (the shares add up: half-weighted home-halo pairs, home kinetic energy). The caller sums
// the 9 doubles at `pres_dev` over the domains and calls mdx_dd_pcouple: every domain draws the same scaling factor
// (C-rescale: same seed, keyed by the step just completed as inside mdx_step) or advances the same box velocity
// (Parrinello-Rahman: the box state is replicated on every domain and sees the same summed pressure), scales its box,
// its atoms (copies included) and, for C-rescale, its velocities, and returns the new box (9 doubles, rows = cell
// vectors). The call waits for the main stream.
extern "C" int mdx_dd_pressure(mdx_system *s, void **pres_dev)
{
    if (pres_dev) *pres_dev = (char *)s->d + offsetof(DevState, pres);
    return MDX_OK;
}
extern "C" int mdx_dd_pcouple(mdx_system *s, double *box9)
{
    (void)hipSetDevice(s->device);
    if (!s->dd_on) { s->err = "mdx_dd_pcouple: no partition loaded"; return MDX_ERR_ARG; }
    if (s->pcoupl != 1 && s->pcoupl != 2) { s->err = "mdx_dd_pcouple: no pressure coupling set up"; return MDX_ERR_UNSUPPORTED; }
    s->step--;
    const int rc = mdx_launch_pcouple(s);
    s->step++;
    if (rc) return rc;
    // the host copy of the box sizes the search grid and the spreading windows
    MDX_CHECK(s, hipMemcpyAsync(s->h->boxd, (char *)s->d + offsetof(DevState, boxd), sizeof(double) * 9, hipMemcpyDeviceToHost, s->stream));
    MDX_CHECK(s, hipStreamSynchronize(s->stream));
    if (box9) for (int a = 0; a < 9; a++) box9[a] = s->h->boxd[a];
    return MDX_OK;
}

// Kinetic energy of the home atoms at the current half step into the handle's scalars (cleared first).
extern "C" int mdx_dd_kinetic(mdx_system *s)
{
    (void)hipSetDevice(s->device);
    int rc = mdx_launch_clear(s, false, true, false); if (rc) return rc;
    return mdx_launch_ke(s, false, false);
}

// Centre-of-mass motion removal in two halves: sum m v and m of the home atoms into the scalars, then (after the
// caller's sum over domains) subtract the common velocity.
extern "C" int mdx_dd_com(mdx_system *s, int phase)
{
    (void)hipSetDevice(s->device);
    return mdx_launch_comm_phase(s, phase);
}

// Device pointer and element count of the real-space charge grid (x fastest), for the caller's sum over domains.
extern "C" int mdx_dd_grid(mdx_system *s, void **grid_dev, long long *nelem)
{
    if (!s->have_pme) { s->err = "mdx_dd_grid: PME is not set up"; return MDX_ERR_ARG; }
    if (grid_dev) *grid_dev = s->grid;
    if (nelem) *nelem = (long long)s->K[0] * s->K[1] * s->K[2];
    return MDX_OK;
}

// Blocks until the main stream is idle (packed buffers complete, state readable) without the checks and the
// scalar read-back of mdx_synchronize.
extern "C" int mdx_dd_stream_sync(mdx_system *s)
{
    (void)hipSetDevice(s->device);
    MDX_CHECK(s, hipStreamSynchronize(s->stream));
    return MDX_OK;
}

extern "C" int mdx_dd_streams(mdx_system *s, void **main_stream, void **pme_stream)
{
    if (main_stream) *main_stream = (void *)s->stream;
    if (pme_stream) *pme_stream = (void *)s->st_pme;
    return MDX_OK;
}

// Current device pointers of positions (float4: xyz, charge) and velocities (float4: xyz, inverse mass); they
// change with every step (double buffering), so they are valid until the next step call only.
extern "C" int mdx_dd_state_pointers(mdx_system *s, void **x4_dev, void **v4_dev)
{
    if (x4_dev) *x4_dev = s->x;
    if (v4_dev) *v4_dev = s->v;
    return MDX_OK;
}
