// C ABI of libmdx.so (declared in include/mdx.h) and the per-step launch sequence.
#include "mdx_internal.h"
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <mutex>

static thread_local std::string g_create_err;

#define ARG_CHECK(sys, cond, msg)                                   \
    do {                                                            \
        if (!(cond)) { (sys)->err = (msg); return MDX_ERR_ARG; }    \
    } while (0)

template <typename T>
static int dev_alloc(mdx_system *s, T **p, size_t count)
{
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (count == 0) count = 1;
    MDX_CHECK(s, hipMalloc((void **)p, sizeof(T) * count));
    return MDX_OK;
}
template <typename T>
static int dev_upload(mdx_system *s, T **p, const std::vector<T> &h)
{
    int rc = dev_alloc(s, p, h.size());
    if (rc) return rc;
    if (!h.empty()) MDX_CHECK(s, hipMemcpy(*p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
    return MDX_OK;
}

// ---------------------------------------------------------------- scratch for the entry point that allocates nothing
// mdx_get_forces_noalloc stages the merged forces here instead of in a buffer it allocates per call. The array is
// part of the code object: every GPU the library runs on holds its own copy, at its own address, until the process
// ends. Host code never uses the array's name as an address; the address on a device is taken once, with that device
// current (mdx_create), and kept in g_scratch_ptr.
#define MDX_NOALLOC_MAX_ATOMS (1 << 20)
#define MDX_MAX_DEVICES 64
static __device__ __align__(256) float scratch[3 * (size_t)MDX_NOALLOC_MAX_ATOMS];      // 12 MiB
static float *g_scratch_ptr[MDX_MAX_DEVICES] = { nullptr };
static hipError_t g_scratch_err[MDX_MAX_DEVICES] = { hipSuccess };
// One lock per device (static storage: taking it allocates nothing). It guards the look-up below and, in
// mdx_get_forces_noalloc, the section from the merge into the array to the end of the copy out of it, so that handles
// driven from different threads of one process take turns on a GPU's array. Other processes have their own copy.
static std::mutex g_scratch_lock[MDX_MAX_DEVICES];

// With `device` current: look the device's copy of `scratch` up once.
static void scratch_lookup(int device)
{
    if (device < 0 || device >= MDX_MAX_DEVICES) return;
    std::lock_guard<std::mutex> guard(g_scratch_lock[device]);
    if (g_scratch_ptr[device]) return;
    float *p = nullptr;
    const hipError_t e = hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch));
    g_scratch_err[device] = e;
    g_scratch_ptr[device] = (e == hipSuccess) ? p : nullptr;
}


extern "C" int mdx_version(void) { return 100; }

extern "C" const char *mdx_last_error(const mdx_system *sys)
{
    return sys ? sys->err.c_str() : g_create_err.c_str();
}

static int sync_and_check(mdx_system *s);

static void host_box(const double *b, DevBox *out)
{
    for (int a = 0; a < 9; a++) out->h[a] = (float)b[a];
    const double ia = 1.0 / b[0], ie = 1.0 / b[4], ii = 1.0 / b[8];
    const double inv[9] = { ia, 0, 0, -b[3] * ia * ie, ie, 0, (b[3] * b[7] - b[4] * b[6]) * ia * ie * ii, -b[7] * ie * ii, ii };
    for (int a = 0; a < 9; a++) out->inv[a] = (float)inv[a];
    out->vol = (float)(b[0] * b[4] * b[8]);
    for (int a = 0; a < 5; a++) out->pad[a] = 0.0f;
}

extern "C" mdx_system *mdx_create(int device, int natoms)
{
    if (natoms <= 0) { g_create_err = "natoms must be positive"; return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_err = "no HIP device available (libmdx has no CPU fallback)";
        return nullptr;
    }
    if (device < 0 || device >= ndev) { g_create_err = "device index out of range"; return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { g_create_err = "hipSetDevice failed"; return nullptr; }
    scratch_lookup(device);
    mdx_system *s = new mdx_system();
    s->device = device; s->n = natoms;
    // All three streams at the default priority. High-priority side streams bring nothing on a GPU of one's own (694
    // against 687 ns/day on the bench workload) and are ruinous on a shared one: hardware queues of a higher priority
    // pre-empt the running waves of every other queue, so two processes, or several handles in one process, keep
    // evicting each other's pair kernel (measured on MI355X: two 24 k-atom runs sharing the device take 59 ms to
    // seconds per step with high-priority side streams, 0.3 ms at equal priorities).
    bool ok = hipStreamCreate(&s->stream) == hipSuccess && hipStreamCreate(&s->st_pme) == hipSuccess &&
              hipStreamCreate(&s->st_bond) == hipSuccess &&
              hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming | hipEventReleaseToDevice) == hipSuccess &&
              hipEventCreateWithFlags(&s->ev_pme, hipEventDisableTiming | hipEventReleaseToDevice) == hipSuccess &&
              hipEventCreateWithFlags(&s->ev_bond, hipEventDisableTiming | hipEventReleaseToDevice) == hipSuccess &&
              hipEventCreateWithFlags(&s->ev_spread, hipEventDisableTiming | hipEventReleaseToDevice) == hipSuccess &&
              hipEventCreateWithFlags(&s->ev_prune, hipEventDisableTiming | hipEventReleaseToDevice) == hipSuccess;
    s->own_streams[0] = s->stream; s->own_streams[1] = s->st_pme; s->own_streams[2] = s->st_bond;
    // MDX_STREAMS=1: every kernel of the handle on one stream (diagnostic; three streams are the default)
    if (const char *e = getenv("MDX_STREAMS")) if (atoi(e) == 1 && ok) (void)mdx_set_stream_mode(s, 1);
    const int n = natoms;
    // MDX_NB_DYN=0 in the environment selects the static deal of the nonbonded kernel (processes sharing one GPU)
    if (const char *e = getenv("MDX_NB_DYN")) s->nb_dyn = atoi(e);
    if (const char *e = getenv("MDX_YZ_LDS_KB")) s->yz_lds_limit = std::max(0, std::min(158, atoi(e))) * 1024;
    s->ncol_cap = n / 128 + 256;
    s->npad = ((n + 63) / 64 + s->ncol_cap) * 64;
    s->cj_cap = (s->npad / 64) * MDX_CJ_PER_SC;
    s->work_cap = (s->npad / 64) * MDX_WORK_PER_SC;
    s->excl_cap = (s->npad / 64) * 4 * MDX_EXCL_PER_WAVE + 1;
    auto A = [&](auto **p, size_t c) { ok = ok && (dev_alloc(s, p, c) == MDX_OK); };
    A(&s->x, n); A(&s->xp, n); A(&s->v, n); A(&s->f, 3 * (size_t)n); A(&s->f_pme, 3 * (size_t)n); A(&s->lj, n); A(&s->molid, n); A(&s->tcgrp, n);
    A(&s->xq_s, s->npad); A(&s->lj_s, s->npad); A(&s->ljt_s, s->npad); A(&s->ljtype, n); A(&s->f_s, 3 * (size_t)s->npad); A(&s->perm, s->npad); A(&s->perm_src, s->npad);
    A(&s->shift_s, s->npad); A(&s->molid_s, s->npad); A(&s->sorted_of, n);
    A(&s->bb_lo, s->npad / 8); A(&s->bb_hi, s->npad / 8); A(&s->sbb_lo, s->npad / 64); A(&s->sbb_hi, s->npad / 64);
    A(&s->col_count, s->ncol_cap + 1); A(&s->col_astart, s->ncol_cap + 1); A(&s->col_scstart, s->ncol_cap + 1); A(&s->col_atoms, n);
    A(&s->atom_col, n); A(&s->atom_slot, n); A(&s->wrapped, n);
    A(&s->cj, s->cj_cap); A(&s->work, s->work_cap); A(&s->work_order, s->work_cap); A(&s->sc_nwork, s->npad / 64); A(&s->exclmask, (size_t)s->excl_cap * 64);
    A(&s->shift_ijk, MDX_MAX_SHIFTS); A(&s->shiftvec, MDX_MAX_SHIFTS);
    A(&s->excl_start, n + 1); A(&s->excl_list, 1); A(&s->d_tc, 3 * MDX_MAX_TC);
    A(&s->d, 1);
    ok = ok && hipHostMalloc((void **)&s->h, sizeof(DevState), hipHostMallocDefault) == hipSuccess;
    s->ener_log_cap = 8192;
    ok = ok && hipHostMalloc((void **)&s->ener_log, sizeof(double) * MDX_E_N * s->ener_log_cap, hipHostMallocDefault) == hipSuccess;
    if (!ok) {
        g_create_err = "allocation failed: " + s->err;
        mdx_destroy(s);
        return nullptr;
    }
    memset(s->h, 0, sizeof(DevState));
    (void)hipMemset(s->d, 0, sizeof(DevState));
    (void)hipMemset(s->excl_start, 0, sizeof(int) * (n + 1));
    (void)hipMemset(s->tcgrp, 0, sizeof(int) * n);
    (void)hipMemset(s->f, 0, sizeof(float) * 3 * n);
    (void)hipMemset(s->f_pme, 0, sizeof(float) * 3 * n);
    (void)hipMemset(s->f_s, 0, sizeof(float) * 3 * s->npad);
    (void)hipMemset(s->sorted_of, 0, sizeof(int) * n);
    return s;
}

extern "C" void mdx_destroy(mdx_system *s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    mdx_pme_free(s);
    mdx_frames_free(s);
    void *ptrs[] = { s->ljtype, s->ljt_s, s->nbfp, s->x, s->xp, s->v, s->f, s->f_pme, s->lj, s->molid, s->tcgrp, s->excl_start, s->excl_list, s->excl_pairs, s->xq_s, s->lj_s,
                     s->f_s, s->perm, s->perm_src, s->shift_s, s->molid_s, s->sorted_of, s->bb_lo, s->bb_hi, s->sbb_lo, s->sbb_hi,
                     s->col_count, s->col_astart, s->col_scstart, s->col_atoms, s->atom_col, s->atom_slot, s->wrapped, s->cj, s->work, s->work_order, s->sc_nwork,
                     s->exclmask, s->shift_ijk, s->shiftvec, s->bond_ij, s->bond_p, s->angle_ijk, s->angle_p, s->dih_ijkl, s->dih_p,
                     s->pair_ij, s->pair_p, s->posres_idx, s->posres_ref, s->posres_k, s->settle_idx, s->cons_ij, s->cons_len,
                     s->comp_start, s->comp_atom_start, s->comp_atoms, s->comp_local, s->free_atoms, s->d_tc, s->ke_part, s->pull_idx[0], s->pull_idx[1], s->gtab, s->htab, s->dd_q0, s->dd_invm0, s->dd_lj0,
                     s->cj_in[0], s->cj_in[1], s->work_in[0], s->work_in[1], s->order_in[0], s->order_in[1], s->d };
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (s->h) (void)hipHostFree(s->h);
    if (s->ener_log) (void)hipHostFree(s->ener_log);
    if (s->pull_log) (void)hipHostFree(s->pull_log);
    for (hipEvent_t e : s->ev_pool) (void)hipEventDestroy(e);
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    if (s->ev_pme) (void)hipEventDestroy(s->ev_pme);
    if (s->ev_bond) (void)hipEventDestroy(s->ev_bond);
    if (s->ev_spread) (void)hipEventDestroy(s->ev_spread);
    if (s->ev_prune) (void)hipEventDestroy(s->ev_prune);
    for (hipStream_t st : s->own_streams) if (st) (void)hipStreamDestroy(st);
    delete s;
}

// One stream per handle instead of three (diagnostic, mdx_debug_set key 10 / MDX_STREAMS=1): the PME chain and the
// bonded kernels follow the nonbonded kernel in the main stream's order, so no kernel of the handle ever waits for an
// event of another stream. Costs the overlap of those chains (489 against 703 ns/day on the bench workload).
int mdx_set_stream_mode(mdx_system *s, int one_stream)
{
    (void)hipSetDevice(s->device);
    MDX_CHECK(s, hipDeviceSynchronize());
    s->one_stream = one_stream != 0;
    s->stream = s->own_streams[0];
    s->st_pme = s->one_stream ? s->own_streams[0] : s->own_streams[1];
    s->st_bond = s->one_stream ? s->own_streams[0] : s->own_streams[2];
    s->pme_signalled = false;
    return MDX_OK;
}

// ---------------------------------------------------------------- topology
static HostAtoms &host_atoms(mdx_system *s) { return s->host_atoms; }

extern "C" int mdx_set_atoms(mdx_system *s, const float *mass, const float *charge, const float *sigma, const float *epsilon,
                             const int *molid, const int *tc_group)
{
    ARG_CHECK(s, mass && charge && sigma && epsilon && molid, "mdx_set_atoms: null pointer");
    (void)hipSetDevice(s->device);
    const int n = s->n;
    HostAtoms &ha = host_atoms(s);
    ha.q.assign(charge, charge + n);
    ha.invm.resize(n);
    std::vector<float2> lj(n);
    s->q2sum = 0.0;
    for (int i = 0; i < n; i++) {
        ARG_CHECK(s, mass[i] >= 0.0f && epsilon[i] >= 0.0f && sigma[i] >= 0.0f, "mdx_set_atoms: negative mass/sigma/epsilon");
        ha.invm[i] = mass[i] > 0.0f ? 1.0f / mass[i] : 0.0f;
        lj[i] = make_float2(0.5f * sigma[i], 2.0f * sqrtf(epsilon[i]));
        s->q2sum += (double)charge[i] * charge[i];
        if (tc_group) ARG_CHECK(s, tc_group[i] >= 0 && tc_group[i] < MDX_MAX_TC, "mdx_set_atoms: tc_group out of range");
    }
    MDX_CHECK(s, hipMemcpy(s->lj, lj.data(), sizeof(float2) * n, hipMemcpyHostToDevice));
    {
        // atom types of the force kernel: one per distinct (sigma, epsilon) with epsilon > 0, type 0 for atoms without a well
        std::vector<std::pair<float, float>> types(1, std::make_pair(0.0f, 0.0f));
        std::vector<int> ty(n);
        for (int i = 0; i < n; i++) {
            int t = 0;
            if (epsilon[i] > 0.0f) {
                const auto key = std::make_pair(sigma[i], epsilon[i]);
                t = (int)(std::find(types.begin() + 1, types.end(), key) - types.begin());
                if (t == (int)types.size()) types.push_back(key);
            }
            ty[i] = 8 * t;
            if ((int)types.size() > MDX_NB_MAXTYPES) break;
        }
        if (s->nbfp) { (void)hipFree(s->nbfp); s->nbfp = nullptr; }
        s->nt = (int)types.size() <= MDX_NB_MAXTYPES ? (int)types.size() : 0;
        if (s->nt > 0) {
            std::vector<float2> tab((size_t)s->nt * s->nt, make_float2(0.0f, 0.0f));
            for (int a = 1; a < s->nt; a++)
                for (int b = 1; b < s->nt; b++) {
                    const double sg = 0.5 * ((double)types[a].first + types[b].first);
                    const double e4 = 4.0 * sqrt((double)types[a].second * types[b].second);
                    const double s6 = sg * sg * sg * sg * sg * sg;
                    tab[(size_t)a * s->nt + b] = make_float2((float)(12.0 * e4 * s6 * s6), (float)(6.0 * e4 * s6));
                }
            MDX_CHECK(s, hipMalloc((void **)&s->nbfp, sizeof(float2) * tab.size()));
            MDX_CHECK(s, hipMemcpy(s->nbfp, tab.data(), sizeof(float2) * tab.size(), hipMemcpyHostToDevice));
            MDX_CHECK(s, hipMemcpy(s->ljtype, ty.data(), sizeof(int) * n, hipMemcpyHostToDevice));
        } else MDX_CHECK(s, hipMemset(s->ljtype, 0, sizeof(int) * n));
    }
    MDX_CHECK(s, hipMemcpy(s->molid, molid, sizeof(int) * n, hipMemcpyHostToDevice));
    if (tc_group) MDX_CHECK(s, hipMemcpy(s->tcgrp, tc_group, sizeof(int) * n, hipMemcpyHostToDevice));
    return MDX_OK;
}

extern "C" int mdx_set_exclusions(mdx_system *s, const int *es, const int *el)
{
    ARG_CHECK(s, es, "mdx_set_exclusions: null pointer");
    (void)hipSetDevice(s->device);
    const int n = s->n;
    ARG_CHECK(s, es[0] == 0, "mdx_set_exclusions: excl_start[0] must be 0");
    const int ne = es[n];
    std::vector<int2> pairs;
    for (int i = 0; i < n; i++) {
        ARG_CHECK(s, es[i + 1] >= es[i], "mdx_set_exclusions: excl_start not monotone");
        for (int k = es[i]; k < es[i + 1]; k++) {
            ARG_CHECK(s, el[k] >= 0 && el[k] < n, "mdx_set_exclusions: index out of range");
            if (el[k] > i) pairs.push_back(make_int2(i, el[k]));
        }
    }
    MDX_CHECK(s, hipMemcpy(s->excl_start, es, sizeof(int) * (n + 1), hipMemcpyHostToDevice));
    int rc = dev_alloc(s, &s->excl_list, (size_t)ne);
    if (rc) return rc;
    if (ne) MDX_CHECK(s, hipMemcpy(s->excl_list, el, sizeof(int) * ne, hipMemcpyHostToDevice));
    s->n_excl_pairs = (int)pairs.size();
    return dev_upload(s, &s->excl_pairs, pairs);
}

static int check_idx(mdx_system *s, int n, const int *idx, int per, const char *what)
{
    for (int k = 0; k < n * per; k++)
        if (idx[k] < 0 || idx[k] >= s->n) { s->err = std::string(what) + ": atom index out of range"; return MDX_ERR_ARG; }
    return MDX_OK;
}

extern "C" int mdx_set_bonds(mdx_system *s, int n, const int *idx, const float *b0, const float *kb)
{
    (void)hipSetDevice(s->device);
    ARG_CHECK(s, n >= 0 && (n == 0 || (idx && b0 && kb)), "mdx_set_bonds: bad arguments");
    int rc = check_idx(s, n, idx, 2, "mdx_set_bonds"); if (rc) return rc;
    std::vector<int2> ij(n); std::vector<float2> p(n);
    for (int k = 0; k < n; k++) { ij[k] = make_int2(idx[2 * k], idx[2 * k + 1]); p[k] = make_float2(b0[k], kb[k]); }
    s->nbonds = n;
    rc = dev_upload(s, &s->bond_ij, ij); if (rc) return rc;
    return dev_upload(s, &s->bond_p, p);
}

extern "C" int mdx_set_angles(mdx_system *s, int n, const int *idx, const float *th0, const float *k)
{
    (void)hipSetDevice(s->device);
    ARG_CHECK(s, n >= 0 && (n == 0 || (idx && th0 && k)), "mdx_set_angles: bad arguments");
    int rc = check_idx(s, n, idx, 3, "mdx_set_angles"); if (rc) return rc;
    std::vector<int4> ij(n); std::vector<float2> p(n);
    for (int a = 0; a < n; a++) {
        ij[a] = make_int4(idx[3 * a], idx[3 * a + 1], idx[3 * a + 2], 0);
        p[a] = make_float2((float)(th0[a] * M_PI / 180.0), k[a]);
    }
    s->nangles = n;
    rc = dev_upload(s, &s->angle_ijk, ij); if (rc) return rc;
    return dev_upload(s, &s->angle_p, p);
}

extern "C" int mdx_set_dihedrals(mdx_system *s, int n, const int *idx, const float *phi0, const float *k, const int *mult)
{
    (void)hipSetDevice(s->device);
    ARG_CHECK(s, n >= 0 && (n == 0 || (idx && phi0 && k && mult)), "mdx_set_dihedrals: bad arguments");
    int rc = check_idx(s, n, idx, 4, "mdx_set_dihedrals"); if (rc) return rc;
    std::vector<int4> ij(n); std::vector<float4> p(n);
    for (int a = 0; a < n; a++) {
        ij[a] = make_int4(idx[4 * a], idx[4 * a + 1], idx[4 * a + 2], idx[4 * a + 3]);
        p[a] = make_float4((float)(phi0[a] * M_PI / 180.0), k[a], (float)mult[a], 0.0f);
    }
    s->ndih = n;
    rc = dev_upload(s, &s->dih_ijkl, ij); if (rc) return rc;
    return dev_upload(s, &s->dih_p, p);
}

extern "C" int mdx_set_pairs14(mdx_system *s, int n, const int *idx, const float *c6, const float *c12, const float *qq)
{
    (void)hipSetDevice(s->device);
    ARG_CHECK(s, n >= 0 && (n == 0 || (idx && c6 && c12 && qq)), "mdx_set_pairs14: bad arguments");
    int rc = check_idx(s, n, idx, 2, "mdx_set_pairs14"); if (rc) return rc;
    std::vector<int2> ij(n); std::vector<float4> p(n);
    for (int a = 0; a < n; a++) {
        ij[a] = make_int2(idx[2 * a], idx[2 * a + 1]);
        p[a] = make_float4(c6[a], c12[a], qq[a] * MDX_ONE_4PI_EPS0, 0.0f);
    }
    s->npairs = n;
    rc = dev_upload(s, &s->pair_ij, ij); if (rc) return rc;
    return dev_upload(s, &s->pair_p, p);
}

extern "C" int mdx_set_posres(mdx_system *s, int n, const int *idx, const float *ref, const float *k, int refcoord_scaling)
{
    (void)hipSetDevice(s->device);
    ARG_CHECK(s, n >= 0 && (n == 0 || (idx && ref && k)), "mdx_set_posres: bad arguments");
    ARG_CHECK(s, refcoord_scaling >= 0 && refcoord_scaling <= 2, "mdx_set_posres: refcoord_scaling must be 0, 1 or 2");
    int rc = check_idx(s, n, idx, 1, "mdx_set_posres"); if (rc) return rc;
    std::vector<int> id(idx, idx + n); std::vector<float4> r(n), kk(n);
    for (int a = 0; a < n; a++) {
        r[a] = make_float4(ref[3 * a], ref[3 * a + 1], ref[3 * a + 2], 0.0f);
        kk[a] = make_float4(k[3 * a], k[3 * a + 1], k[3 * a + 2], 0.0f);
    }
    s->nposres = n; s->refcoord_scaling = refcoord_scaling;
    double com[6] = { 0, 0, 0, 0, 0, 0 };
    for (int a = 0; a < n; a++) for (int c = 0; c < 3; c++) com[c] += ref[3 * a + c];
    for (int c = 0; c < 3; c++) com[c] = com[3 + c] = n ? com[c] / n : 0.0;
    MDX_CHECK(s, hipMemcpy((char *)s->d + offsetof(DevState, posres_com), com, sizeof(com), hipMemcpyHostToDevice));
    rc = dev_upload(s, &s->posres_idx, id); if (rc) return rc;
    rc = dev_upload(s, &s->posres_ref, r); if (rc) return rc;
    return dev_upload(s, &s->posres_k, kk);
}

extern "C" int mdx_get_posres_ref(mdx_system *s, float *ref_xyz)
{
    (void)hipSetDevice(s->device);
    ARG_CHECK(s, s->nposres == 0 || ref_xyz, "mdx_get_posres_ref: null output");
    if (s->nposres == 0) return MDX_OK;
    std::vector<float4> r(s->nposres);
    MDX_CHECK(s, hipStreamSynchronize(s->stream));
    MDX_CHECK(s, hipMemcpy(r.data(), s->posres_ref, sizeof(float4) * s->nposres, hipMemcpyDeviceToHost));
    for (int a = 0; a < s->nposres; a++) { ref_xyz[3 * a] = r[a].x; ref_xyz[3 * a + 1] = r[a].y; ref_xyz[3 * a + 2] = r[a].z; }
    return MDX_OK;
}

extern "C" int mdx_set_settles(mdx_system *s, int n, const int *idx, float doh, float dhh)
{
    (void)hipSetDevice(s->device);
    ARG_CHECK(s, n >= 0 && (n == 0 || idx), "mdx_set_settles: bad arguments");
    ARG_CHECK(s, n == 0 || (doh > 0 && dhh > 0 && dhh < 2 * doh), "mdx_set_settles: bad geometry");
    int rc = check_idx(s, n, idx, 3, "mdx_set_settles"); if (rc) return rc;
    std::vector<int4> t(n);
    for (int a = 0; a < n; a++) t[a] = make_int4(idx[3 * a], idx[3 * a + 1], idx[3 * a + 2], 0);
    s->nsettle = n; s->nsettle_cap = n; s->settle_doh = doh; s->settle_dhh = dhh;
    s->h_settle.assign(idx, idx + 3 * (size_t)n); s->fused_plan = 0;
    // waters listed one after the other (O H H of water w at first + 3 w ...), as a solvated system's topology has them: the
    // fused update then computes a water's atom indices instead of loading them (one dependent memory round trip fewer)
    s->settle_first = n > 0 ? idx[0] : -1;
    for (int a = 0; a < n && s->settle_first >= 0; a++)
        for (int k = 0; k < 3; k++) if (idx[3 * a + k] != idx[0] + 3 * a + k) s->settle_first = -1;
    return dev_upload(s, &s->settle_idx, t);
}

extern "C" int mdx_set_constraints(mdx_system *s, int n, const int *idx, const float *length, int order, int iter)
{
    (void)hipSetDevice(s->device);
    ARG_CHECK(s, n >= 0 && (n == 0 || (idx && length)), "mdx_set_constraints: bad arguments");
    ARG_CHECK(s, order >= 1 && order <= 12 && iter >= 0 && iter <= 8, "mdx_set_constraints: lincs_order/lincs_iter out of range");
    int rc = check_idx(s, n, idx, 2, "mdx_set_constraints"); if (rc) return rc;
    // connected components over shared atoms (union-find on constraints)
    std::vector<int> parent(n);
    std::iota(parent.begin(), parent.end(), 0);
    auto find = [&](int a) { while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; } return a; };
    std::vector<int> owner(s->n, -1);
    for (int k = 0; k < n; k++)
        for (int e = 0; e < 2; e++) {
            const int a = idx[2 * k + e];
            if (owner[a] < 0) owner[a] = k;
            else parent[find(k)] = find(owner[a]);
        }
    std::vector<int> order_idx(n);
    std::iota(order_idx.begin(), order_idx.end(), 0);
    std::vector<int> root(n);
    for (int k = 0; k < n; k++) root[k] = find(k);
    // components by ascending size (then by root) so that the small ones form a contiguous front
    std::vector<int> csize(n, 0);
    for (int k = 0; k < n; k++) csize[root[k]]++;
    std::stable_sort(order_idx.begin(), order_idx.end(), [&](int a, int b) {
        const int sa = csize[root[a]], sb = csize[root[b]];
        return sa != sb ? sa < sb : root[a] < root[b];
    });
    std::vector<int2> cij(n); std::vector<float> cl(n); std::vector<int> cstart;
    for (int k = 0; k < n; k++) {
        const int c = order_idx[k];
        if (k == 0 || root[c] != root[order_idx[k - 1]]) cstart.push_back(k);
        cij[k] = make_int2(idx[2 * c], idx[2 * c + 1]); cl[k] = length[c];
    }
    cstart.push_back(n);
    for (size_t c = 0; c + 1 < cstart.size(); c++)
        if (cstart[c + 1] - cstart[c] > MDX_LINCS_MAXK) {
            s->err = "mdx_set_constraints: a coupled constraint component exceeds 16 constraints (only h-bond style topologies are supported)";
            return MDX_ERR_UNSUPPORTED;
        }
    s->ncons = n; s->ncomp = n ? (int)cstart.size() - 1 : 0; s->lincs_order = order; s->lincs_iter = iter;
    s->h_cij = cij; s->h_cstart = cstart; s->fused_plan = 0;
    s->ncomp_small = 0;
    for (int c = 0; c < s->ncomp; c++) if (cstart[c + 1] - cstart[c] <= 4) s->ncomp_small = c + 1;
    rc = dev_upload(s, &s->cons_ij, cij); if (rc) return rc;
    rc = dev_upload(s, &s->cons_len, cl); if (rc) return rc;
    return dev_upload(s, &s->comp_start, cstart);
}

// ---------------------------------------------------------------- run parameters
extern "C" int mdx_set_nonbonded(mdx_system *s, float rc, float rlist, float beta, int nstlist)
{
    ARG_CHECK(s, rc > 0 && rlist >= rc && beta >= 0 && nstlist >= 1, "mdx_set_nonbonded: need rc > 0, rlist >= rc, nstlist >= 1");
    (void)hipSetDevice(s->device);
    s->rc = rc; s->rlist = rlist; s->beta = beta; s->nstlist = nstlist;
    s->steps_since_search = 1 << 30;
    // g(w) = (erf(z)/z - 2/sqrt(pi) exp(-w)) / w, w = z^2, on MDX_GTAB_N intervals of r^2 in [0, rc^2]
    std::vector<float2> tab(MDX_GTAB_N);
    const double wmax = (double)beta * beta * rc * rc;
    auto g = [](double w) {
        const double c = 1.1283791670955126;   // 2/sqrt(pi)
        if (w < 0.05) return c * (2.0 / 3.0 + w * (-0.4 + w * (1.0 / 7.0 + w * (-1.0 / 27.0 + w * (1.0 / 132.0 - w / 780.0)))));
        const double z = sqrt(w);
        return (erf(z) / z - c * exp(-w)) / w;
    };
    auto h = [](double w) {
        const double c = 1.1283791670955126;
        if (w < 0.05) return c * (1.0 + w * (-1.0 / 3.0 + w * (0.1 + w * (-1.0 / 42.0 + w * (1.0 / 216.0 - w / 1320.0)))));
        const double z = sqrt(w);
        return erf(z) / z;
    };
    // energy kernel: beta^3 g and beta h per interval as lines c0 + c1 r^2 through the end points, written in r^2 itself
    // (erfc(beta r)/r = 1/r - beta h, so an energy step evaluates no exponential and takes no fractional part)
    std::vector<float4> tab4(MDX_GTAB_N);
    {
        const double b3 = (double)beta * beta * beta, scale = (double)MDX_GTAB_N / ((double)rc * rc);
        for (int k = 0; k < MDX_GTAB_N; k++) {
            const double g0 = g(wmax * k / MDX_GTAB_N), g1 = g(wmax * (k + 1) / MDX_GTAB_N);
            const double h0 = h(wmax * k / MDX_GTAB_N), h1 = h(wmax * (k + 1) / MDX_GTAB_N);
            const double cg = b3 * (g1 - g0), ch = (double)beta * (h1 - h0);
            tab4[k] = make_float4((float)(b3 * g0 - cg * k), (float)(cg * scale), (float)(beta * h0 - ch * k), (float)(ch * scale));
        }
    }
    if (!s->htab) MDX_CHECK(s, hipMalloc((void **)&s->htab, sizeof(float4) * MDX_GTAB_N));
    MDX_CHECK(s, hipMemcpy(s->htab, tab4.data(), sizeof(float4) * MDX_GTAB_N, hipMemcpyHostToDevice));
    // The force kernel's table carries beta^3 g (one multiplication less per pair), each interval as the line c0 + c1 r^2
    // through its end points, written in r^2 itself (no table coordinate, no fractional part to take per pair).
    {
        const double b3 = (double)beta * beta * beta;
        const double scale = (double)MDX_GTAB_N / ((double)rc * rc);
        for (int k = 0; k < MDX_GTAB_N; k++) {
            const double g0 = g(wmax * k / MDX_GTAB_N), g1 = g(wmax * (k + 1) / MDX_GTAB_N);
            const double c1 = b3 * (g1 - g0);
            tab[k] = make_float2((float)(b3 * g0 - c1 * k), (float)(c1 * scale));
        }
        if (!s->gtab) MDX_CHECK(s, hipMalloc((void **)&s->gtab, sizeof(float2) * MDX_GTAB_N));
        MDX_CHECK(s, hipMemcpy(s->gtab, tab.data(), sizeof(float2) * MDX_GTAB_N, hipMemcpyHostToDevice));
    }
    return MDX_OK;
}

extern "C" int mdx_set_pairlist_prune(mdx_system *s, float rlist_inner, int nstprune)
{
    (void)hipSetDevice(s->device);
    if (nstprune <= 0) { s->nstprune = 0; s->inner_valid = false; s->nb_chunk = MDX_NB_CHUNK; return MDX_OK; }
    ARG_CHECK(s, rlist_inner >= s->rc && rlist_inner <= s->rlist, "mdx_set_pairlist_prune: need rc <= rlist_inner <= rlist (set the outer list first)");
    ARG_CHECK(s, !s->dd_on, "mdx_set_pairlist_prune: not available for the domains of a decomposed run");
    int rc = sync_and_check(s); if (rc) return rc;
    for (int k = 0; k < 2; k++) {
        if ((rc = dev_alloc(s, &s->cj_in[k], (size_t)s->cj_cap))) return rc;
        if ((rc = dev_alloc(s, &s->work_in[k], (size_t)s->work_cap))) return rc;
        if ((rc = dev_alloc(s, &s->order_in[k], (size_t)s->work_cap))) return rc;
    }
    s->rlist_inner = rlist_inner; s->nstprune = nstprune;
    s->nb_chunk = 40;
    // Units of the outer list a little longer than those of a single list, so that what pruning leaves of a unit still
    // amortises the unit's fixed costs (bench workload: 999 / 1013 / 1025 / 1012 ns/day at 24 / 32 / 40 / 48 entries).
    s->inner_valid = false; s->steps_since_search = 1 << 30;
    return MDX_OK;
}

extern "C" int mdx_set_pme(mdx_system *s, int nx, int ny, int nz, int order)
{
    (void)hipSetDevice(s->device);
    if (nx <= 0) { s->K[0] = s->K[1] = s->K[2] = 0; mdx_pme_free(s); return MDX_OK; }
    ARG_CHECK(s, order == 4, "mdx_set_pme: only pme_order = 4 is implemented");
    ARG_CHECK(s, nx >= 8 && ny >= 8 && nz >= 8, "mdx_set_pme: grid dimensions must be >= 8");
    s->K[0] = nx; s->K[1] = ny; s->K[2] = nz;
    return mdx_pme_init(s);
}

extern "C" int mdx_set_integrator(mdx_system *s, float dt, int nstcalcenergy, int nstcomm)
{
    ARG_CHECK(s, dt > 0 && nstcalcenergy >= 0 && nstcomm >= 0, "mdx_set_integrator: bad arguments");
    s->dt = dt; s->nstcalcenergy = nstcalcenergy; s->nstcomm = nstcomm;
    return MDX_OK;
}

extern "C" int mdx_set_tcoupl(mdx_system *s, int ng, const double *ref_t, const double *tau_t, const double *ndf, int nsttcouple,
                              unsigned long long seed)
{
    (void)hipSetDevice(s->device);
    ARG_CHECK(s, ng >= 0 && ng <= MDX_MAX_TC, "mdx_set_tcoupl: at most 8 groups");
    ARG_CHECK(s, nsttcouple >= 1, "mdx_set_tcoupl: nsttcouple must be >= 1");
    s->ntc = ng; s->nsttcouple = nsttcouple; s->seed = seed;
    double buf[3 * MDX_MAX_TC] = { 0 };
    for (int g = 0; g < ng; g++) {
        ARG_CHECK(s, ref_t && tau_t && ndf, "mdx_set_tcoupl: null pointer");
        s->ref_t[g] = ref_t[g]; s->tau_t[g] = tau_t[g]; s->ndf[g] = ndf[g];
        buf[g] = ref_t[g]; buf[MDX_MAX_TC + g] = tau_t[g]; buf[2 * MDX_MAX_TC + g] = ndf[g];
    }
    if (ng == 0 && ndf) s->ndf_total = ndf[0];
    MDX_CHECK(s, hipMemcpy(s->d_tc, buf, sizeof(buf), hipMemcpyHostToDevice));
    s->first_step = true;
    s->tc_fold_pending = false;        // (factors still owed from the old settings: the first step draws them afresh)
    return MDX_OK;
}

extern "C" int mdx_set_pcoupl(mdx_system *s, int kind, double tau_p, double ref_p, double compress, int nstpcouple)
{
    ARG_CHECK(s, kind >= 0 && kind <= 2, "mdx_set_pcoupl: kind must be 0 (no), 1 (C-rescale) or 2 (Parrinello-Rahman)");
    ARG_CHECK(s, kind == 0 || (tau_p > 0 && compress > 0 && nstpcouple >= 1), "mdx_set_pcoupl: bad parameters");
    s->pcoupl = kind; s->tau_p = tau_p; s->ref_p = ref_p; s->compress = compress; s->nstpcouple = nstpcouple > 0 ? nstpcouple : 10;
    return MDX_OK;
}

extern "C" int mdx_set_dispcorr(mdx_system *s, int mode, double c6avg)
{
    ARG_CHECK(s, mode >= 0 && mode <= 2, "mdx_set_dispcorr: mode must be 0, 1 or 2");
    s->dispcorr = mode; s->dispcorr_c6avg = c6avg;
    return MDX_OK;
}

extern "C" int mdx_set_pull(mdx_system *s, int n1, const int *g1, int n2, const int *g2, int pbc1, int pbc2, int type, double k,
                            double rate, double init, int start, const int *dim, int nstout)
{
    (void)hipSetDevice(s->device);
    if (n1 == 0 && n2 == 0) { s->pull_on = 0; return MDX_OK; }
    ARG_CHECK(s, n1 > 0 && n2 > 0 && g1 && g2, "mdx_set_pull: both groups must be non-empty");
    ARG_CHECK(s, type == 0 || type == 1, "mdx_set_pull: type must be 0 (constant-force) or 1 (umbrella)");
    ARG_CHECK(s, nstout >= 1, "mdx_set_pull: nstout must be >= 1");
    int rc = check_idx(s, n1, g1, 1, "mdx_set_pull"); if (rc) return rc;
    rc = check_idx(s, n2, g2, 1, "mdx_set_pull"); if (rc) return rc;
    std::vector<int> a(g1, g1 + n1), b(g2, g2 + n2);
    rc = dev_upload(s, &s->pull_idx[0], a); if (rc) return rc;
    rc = dev_upload(s, &s->pull_idx[1], b); if (rc) return rc;
    s->pull_n[0] = n1; s->pull_n[1] = n2;
    s->pull_pbcatom[0] = (pbc1 >= 0 && pbc1 < n1) ? pbc1 : (n1 - 1) / 2;
    s->pull_pbcatom[1] = (pbc2 >= 0 && pbc2 < n2) ? pbc2 : (n2 - 1) / 2;
    s->pull_type = type; s->pull_k = k; s->pull_rate = rate; s->pull_init_user = init; s->pull_start = start;
    for (int d = 0; d < 3; d++) s->pull_dim[d] = dim ? dim[d] : 1;
    s->pull_nstout = nstout;
    if (!s->pull_log) {
        s->pull_log_cap = 1 << 16;
        MDX_CHECK(s, hipHostMalloc((void **)&s->pull_log, sizeof(double) * 4 * s->pull_log_cap, hipHostMallocMapped));
    }
    s->pull_log_count = 0;
    s->pull_on = 1;
    s->pull_reset_pending = true;
    return MDX_OK;
}

// ---------------------------------------------------------------- state
static int upload_box(mdx_system *s, const float *box9)
{
    ARG_CHECK(s, fabsf(box9[1]) + fabsf(box9[2]) + fabsf(box9[5]) < 1e-6f, "box must be lower triangular (a along x, b in the xy plane)");
    ARG_CHECK(s, box9[0] > 0 && box9[4] > 0 && box9[8] > 0, "box diagonal must be positive");
    for (int a = 0; a < 9; a++) s->h->boxd[a] = box9[a];
    host_box(s->h->boxd, &s->h->box);
    ARG_CHECK(s, box9[0] >= 2 * s->rlist * 0.999f && box9[4] >= 2 * s->rlist * 0.999f && box9[8] >= 2 * s->rlist * 0.999f,
              "box diagonal shorter than twice the pair-list radius");
    MDX_CHECK(s, hipMemcpyAsync(&s->d->box, &s->h->box, sizeof(DevBox), hipMemcpyHostToDevice, s->stream));
    MDX_CHECK(s, hipMemcpyAsync(s->d->boxd, s->h->boxd, sizeof(double) * 9, hipMemcpyHostToDevice, s->stream));
    MDX_CHECK(s, hipStreamSynchronize(s->stream));
    return MDX_OK;
}

extern "C" int mdx_set_state(mdx_system *s, const float *x, const float *v, const float *box9)
{
    (void)hipSetDevice(s->device);
    ARG_CHECK(s, x && box9, "mdx_set_state: null pointer");
    HostAtoms &ha = host_atoms(s);
    ARG_CHECK(s, (int)ha.q.size() == s->n, "mdx_set_state: call mdx_set_atoms first");
    if (s->frame_ring) { const int rcf = mdx_frames_flush(s, s->stream); if (rcf) return rcf; }       // captured frames read the state about to be replaced
    int rc = upload_box(s, box9); if (rc) return rc;
    const int n = s->n;
    std::vector<float4> hx(n), hv(n);
    for (int i = 0; i < n; i++) {
        hx[i] = make_float4(x[3 * i], x[3 * i + 1], x[3 * i + 2], ha.q[i]);
        hv[i] = v ? make_float4(v[3 * i], v[3 * i + 1], v[3 * i + 2], ha.invm[i]) : make_float4(0, 0, 0, ha.invm[i]);
        ARG_CHECK(s, std::isfinite(hx[i].x) && std::isfinite(hx[i].y) && std::isfinite(hx[i].z), "mdx_set_state: non-finite coordinate");
    }
    MDX_CHECK(s, hipMemcpy(s->x, hx.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    MDX_CHECK(s, hipMemcpy(s->v, hv.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    s->steps_since_search = 1 << 30; s->first_step = true; s->forces_current = false;
    return MDX_OK;
}

__global__ void k_pack_state(int n, const float *__restrict__ x3, const float *__restrict__ v3, float4 *__restrict__ x, float4 *__restrict__ v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = x[i]; p.x = x3[3 * i]; p.y = x3[3 * i + 1]; p.z = x3[3 * i + 2]; x[i] = p;
    if (v3) { float4 w = v[i]; w.x = v3[3 * i]; w.y = v3[3 * i + 1]; w.z = v3[3 * i + 2]; v[i] = w; }
}
__global__ void k_set_qm(int n, const float *__restrict__ q, const float *__restrict__ im, float4 *__restrict__ x, float4 *__restrict__ v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    x[i].w = q[i]; v[i].w = im[i];
}
__global__ void k_unpack_state(int n, const float4 *__restrict__ x, const float4 *__restrict__ v, float *__restrict__ x3, float *__restrict__ v3)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = x[i], w = v[i];
    x3[3 * i] = p.x; x3[3 * i + 1] = p.y; x3[3 * i + 2] = p.z;
    v3[3 * i] = w.x; v3[3 * i + 1] = w.y; v3[3 * i + 2] = w.z;
}

extern "C" int mdx_set_state_device(mdx_system *s, const void *xd, const void *vd, const float *box9)
{
    (void)hipSetDevice(s->device);
    ARG_CHECK(s, xd && box9, "mdx_set_state_device: null pointer");
    HostAtoms &ha = host_atoms(s);
    ARG_CHECK(s, (int)ha.q.size() == s->n, "mdx_set_state_device: call mdx_set_atoms first");
    if (s->frame_ring) { const int rcf = mdx_frames_flush(s, s->stream); if (rcf) return rcf; }       // captured frames read the state about to be replaced
    int rc = upload_box(s, box9); if (rc) return rc;
    const int n = s->n, nb = (n + 255) / 256;
    // charges / inverse masses ride in the .w lanes: (re)write them from the host copies
    float *tmp = nullptr;
    MDX_CHECK(s, hipMalloc((void **)&tmp, sizeof(float) * 2 * n));
    MDX_CHECK(s, hipMemcpy(tmp, ha.q.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    MDX_CHECK(s, hipMemcpy(tmp + n, ha.invm.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_set_qm, dim3(nb), dim3(256), 0, s->stream, n, tmp, tmp + n, s->x, s->v);
    if (!vd) MDX_CHECK(s, hipMemsetAsync(s->v, 0, sizeof(float4) * n, s->stream));
    if (!vd) hipLaunchKernelGGL(k_set_qm, dim3(nb), dim3(256), 0, s->stream, n, tmp, tmp + n, s->x, s->v);
    hipLaunchKernelGGL(k_pack_state, dim3(nb), dim3(256), 0, s->stream, n, (const float *)xd, (const float *)vd, s->x, s->v);
    MDX_CHECK(s, hipStreamSynchronize(s->stream));
    (void)hipFree(tmp);
    s->steps_since_search = 1 << 30; s->first_step = true; s->forces_current = false;
    return MDX_OK;
}

static int sync_and_check(mdx_system *s)
{
    if (s->frame_ring) { const int rc = mdx_frames_flush(s, s->stream); if (rc) return rc; }
    // (a pruning launch is the one piece of work that no step joins into the main stream: it ends on the bonded stream)
    if (s->prune_pending) { MDX_CHECK(s, hipStreamSynchronize(s->st_bond)); s->prune_pending = false; }
    // (thermostat factors still to be drawn from the last update's kinetic-energy sums: normally the next step's bonded stream does it)
    if (s->tc_fold_pending) { s->tc_fold_pending = false; const int rc = mdx_launch_tcouple_fold(s, s->stream, s->tc_fold_step); if (rc) return rc; }
    MDX_CHECK(s, hipMemcpyAsync(s->h, s->d, sizeof(DevState), hipMemcpyDeviceToHost, s->stream));
    MDX_CHECK(s, hipStreamSynchronize(s->stream));
    if (s->h->overflow) {
        s->err = "pair-list capacity exceeded (flags " + std::to_string(s->h->overflow) + "): 1 j-list, 2 work list, 4 masks, 8 column too tall, 16 clusters";
        return MDX_ERR_OVERFLOW;
    }
    return MDX_OK;
}

extern "C" int mdx_synchronize(mdx_system *s)
{
    (void)hipSetDevice(s->device);
    return sync_and_check(s);
}

extern "C" int mdx_get_state(mdx_system *s, float *x, float *v, float *box9)
{
    (void)hipSetDevice(s->device);
    int rc = sync_and_check(s); if (rc) return rc;
    const int n = s->n, nb = (n + 255) / 256;
    float *tmp = nullptr;
    MDX_CHECK(s, hipMalloc((void **)&tmp, sizeof(float) * 6 * n));
    hipLaunchKernelGGL(k_unpack_state, dim3(nb), dim3(256), 0, s->stream, n, s->x, s->v, tmp, tmp + 3 * n);
    if (x) MDX_CHECK(s, hipMemcpyAsync(x, tmp, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, s->stream));
    if (v) MDX_CHECK(s, hipMemcpyAsync(v, tmp + 3 * n, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, s->stream));
    MDX_CHECK(s, hipStreamSynchronize(s->stream));
    (void)hipFree(tmp);
    if (box9) for (int a = 0; a < 9; a++) box9[a] = (float)s->h->boxd[a];
    return MDX_OK;
}

extern "C" long long mdx_get_step(const mdx_system *s) { return s->step; }

// Run state: the coupling and pull scalars that live on the device between steps (layout: MDX_RS_* in mdx.h).
extern "C" int mdx_get_run_state(mdx_system *s, double *rs)
{
    (void)hipSetDevice(s->device);
    ARG_CHECK(s, rs, "mdx_get_run_state: null pointer");
    int rc = sync_and_check(s); if (rc) return rc;
    const DevState *h = s->h;
    for (int k = 0; k < MDX_RS_N; k++) rs[k] = 0.0;
    for (int g = 0; g < MDX_MAX_TC; g++) {
        rs[MDX_RS_LAMBDA + g] = h->lambda[g]; rs[MDX_RS_THERM_INTEGRAL + g] = h->therm_integral[g];
        rs[MDX_RS_EKINH + g] = h->ekinh[g]; rs[MDX_RS_EKINH_OLD + g] = h->ekinh_old[g];
    }
    for (int a = 0; a < 9; a++) { rs[MDX_RS_PR_BOXV + a] = h->pr_boxv[a]; rs[MDX_RS_MU + a] = h->mu[a]; }
    rs[MDX_RS_BARO_INTEGRAL] = h->baro_integral;
    rs[MDX_RS_PULL_INIT] = h->pull_init; rs[MDX_RS_PULL_WORK] = h->pull_work; rs[MDX_RS_PULL_VALUE] = h->pull_value;
    rs[MDX_RS_PULL_HAVE_INIT] = h->pull_have_init ? 1.0 : 0.0;
    for (int a = 0; a < 4; a++) rs[MDX_RS_COM_V + a] = h->com_v[a];
    return MDX_OK;
}

extern "C" int mdx_set_run_state(mdx_system *s, const double *rs)
{
    (void)hipSetDevice(s->device);
    ARG_CHECK(s, rs, "mdx_set_run_state: null pointer");
    int rc = sync_and_check(s); if (rc) return rc;
    DevState *h = s->h;
    for (int g = 0; g < MDX_MAX_TC; g++) { h->lambda[g] = rs[MDX_RS_LAMBDA + g]; h->therm_integral[g] = rs[MDX_RS_THERM_INTEGRAL + g]; }
    for (int a = 0; a < 9; a++) h->pr_boxv[a] = rs[MDX_RS_PR_BOXV + a];
    // the velocity coupling matrix follows from the box velocity (isotropic: every element shares one relative rate)
    const double rate = h->boxd[0] > 0.0 ? h->pr_boxv[0] / h->boxd[0] : 0.0;
    for (int a = 0; a < 9; a++) h->pr_M[a] = (a % 4 == 0 && s->pcoupl == 2) ? 2.0 * rate : 0.0;
    h->baro_integral = rs[MDX_RS_BARO_INTEGRAL];
    h->pull_init = rs[MDX_RS_PULL_INIT]; h->pull_work = rs[MDX_RS_PULL_WORK]; h->pull_value = rs[MDX_RS_PULL_VALUE];
    h->pull_have_init = rs[MDX_RS_PULL_HAVE_INIT] != 0.0 ? 1 : 0;
#define RS_UP(field) MDX_CHECK(s, hipMemcpyAsync((char *)s->d + offsetof(DevState, field), (char *)h + offsetof(DevState, field), sizeof(h->field), hipMemcpyHostToDevice, s->stream))
    RS_UP(lambda); RS_UP(therm_integral); RS_UP(pr_boxv); RS_UP(pr_M); RS_UP(baro_integral);
    RS_UP(pull_init); RS_UP(pull_work); RS_UP(pull_value); RS_UP(pull_have_init);
#undef RS_UP
    MDX_CHECK(s, hipStreamSynchronize(s->stream));
    // the restored scaling factors are the ones the interrupted run would have used next: do not redraw them, and
    // keep the restored pull reference instead of re-seeding it from the restart geometry
    s->first_step = false;
    if (h->pull_have_init) s->pull_reset_pending = false;
    return MDX_OK;
}

// ---------------------------------------------------------------- compute
// Next slot of the pinned ring of energy frames: the kernel that finishes the energies writes the frame there itself.
static double *next_energy_frame(mdx_system *s)
{
    const long long slot = s->ener_log_count % s->ener_log_cap;
    s->ener_log_count++;
    return s->ener_log + slot * MDX_E_N;
}

static std::vector<long long> &frame_steps(mdx_system *s) { return s->frame_steps; }

// The PME chain and the bonded-type kernels run on their own streams next to the nonbonded
// kernel: fork after the sorted coordinates are ready, join before the update.
int mdx_fork(mdx_system *s)
{
    MDX_CHECK(s, hipEventRecord(s->ev_fork, s->stream));
    MDX_CHECK(s, hipStreamWaitEvent(s->st_pme, s->ev_fork, 0));
    MDX_CHECK(s, hipStreamWaitEvent(s->st_bond, s->ev_fork, 0));
    return MDX_OK;
}
// The main stream waits for a pruning launch that is still unaccounted for: ahead of every kernel that rewrites the
// sorted coordinates the pruning kernel reads, and thereby ahead of the first nonbonded launch on the pruned list.
int mdx_wait_prune(mdx_system *s)
{
    if (!s->prune_pending) return MDX_OK;
    MDX_CHECK(s, hipStreamWaitEvent(s->stream, s->ev_prune, 0));
    s->prune_pending = false;
    return MDX_OK;
}
int mdx_join(mdx_system *s)
{
    if (!s->pme_signalled) MDX_CHECK(s, hipEventRecord(s->ev_pme, s->st_pme));
    s->pme_signalled = false;
    // (a step whose bonded work rode in the spreading launch and that put nothing else on the bonded stream has one wait)
    if (s->bond_busy) {
        MDX_CHECK(s, hipEventRecord(s->ev_bond, s->st_bond));
        MDX_CHECK(s, hipStreamWaitEvent(s->stream, s->ev_bond, 0));
    }
    s->bond_busy = true;
    MDX_CHECK(s, hipStreamWaitEvent(s->stream, s->ev_pme, 0));
    return MDX_OK;
}

extern "C" int mdx_compute_forces(mdx_system *s, unsigned terms)
{
    (void)hipSetDevice(s->device);
    int rc;
    s->bond_busy = true;        // (the bonded kernels below run on their own stream; a step that failed half-way may have left the flag down)
    if ((rc = mdx_launch_clear(s, true, true, true))) return rc;
    MDX_CHECK(s, hipMemsetAsync(s->f, 0, sizeof(float) * 3 * s->n, s->stream));
    if ((rc = mdx_search(s))) return rc;
    if ((rc = mdx_launch_prepare(s))) return rc;
    if (!(terms & MDX_F_PME) || !s->have_pme) MDX_CHECK(s, hipMemsetAsync(s->f_pme, 0, sizeof(float) * 3 * s->n, s->stream));
    if ((rc = mdx_fork(s))) return rc;
    if (terms & MDX_F_PME) if ((rc = mdx_launch_pme(s, true))) return rc;
    if ((rc = mdx_launch_bonded(s, true, terms))) return rc;
    if (terms & MDX_F_PULL) if ((rc = mdx_launch_pull(s, false, true))) return rc;
    s->count_pairs = true;       // a single evaluation also reports the pair statistics
    rc = mdx_launch_nonbonded(s, true, terms);
    s->count_pairs = s->count_pairs_steps;
    if (rc) return rc;
    if ((rc = mdx_join(s))) return rc;
    if ((rc = mdx_launch_ke(s, true, true))) return rc;
    if ((rc = mdx_launch_ke(s, false, true))) return rc;
    if ((rc = mdx_launch_finish_energy(s))) return rc;
    s->forces_current = true;
    s->f_dirty = true;
    return sync_and_check(s);
}

extern "C" int mdx_get_forces(mdx_system *s, float *f)
{
    (void)hipSetDevice(s->device);
    ARG_CHECK(s, f, "mdx_get_forces: null pointer");
    ARG_CHECK(s, s->forces_current, "mdx_get_forces: no forces computed at the current positions");
    float *tmp = nullptr;
    MDX_CHECK(s, hipMalloc((void **)&tmp, sizeof(float) * 3 * s->n));
    int rc = mdx_launch_merge_forces(s, tmp);
    if (rc) { (void)hipFree(tmp); return rc; }
    MDX_CHECK(s, hipMemcpyAsync(f, tmp, sizeof(float) * 3 * s->n, hipMemcpyDeviceToHost, s->stream));
    MDX_CHECK(s, hipStreamSynchronize(s->stream));
    (void)hipFree(tmp);
    return MDX_OK;
}

// mdx_get_forces without any allocation: the merged forces are staged in the file-scope scratch array of the
// handle's device (address looked up at mdx_create), then copied to the caller's buffer on the handle's stream.
// All handles of a process on one GPU share that array: a per-device lock makes their calls take turns.
extern "C" int mdx_get_forces_noalloc(mdx_system *s, float *f)
{
    (void)hipSetDevice(s->device);
    ARG_CHECK(s, f, "mdx_get_forces_noalloc: null pointer");
    if (s->n > MDX_NOALLOC_MAX_ATOMS) {
        s->err = "mdx_get_forces_noalloc: " + std::to_string(s->n) + " atoms exceed the fixed scratch of " +
                 std::to_string(MDX_NOALLOC_MAX_ATOMS) + " atoms (use mdx_get_forces)";
        return MDX_ERR_OVERFLOW;
    }
    ARG_CHECK(s, s->forces_current, "mdx_get_forces_noalloc: no forces computed at the current positions");
    if (s->device < 0 || s->device >= MDX_MAX_DEVICES) {
        s->err = "mdx_get_forces_noalloc: the fixed scratch exists on devices 0 .. " + std::to_string(MDX_MAX_DEVICES - 1) + " only (use mdx_get_forces)";
        return MDX_ERR_UNSUPPORTED;
    }
    std::lock_guard<std::mutex> guard(g_scratch_lock[s->device]);
    float *stage = g_scratch_ptr[s->device];
    if (!stage) {
        s->err = std::string("mdx_get_forces_noalloc: no scratch on this device (hipGetSymbolAddress: ") +
                 hipGetErrorString(g_scratch_err[s->device]) + ")";
        return MDX_ERR_HIP;
    }
    // every one of the 3n staged values is written by the merge kernel: nothing of an earlier call survives
    int rc = mdx_launch_merge_forces(s, stage);
    if (rc) return rc;
    MDX_CHECK(s, hipMemcpyAsync(f, stage, sizeof(float) * 3 * s->n, hipMemcpyDeviceToHost, s->stream));
    MDX_CHECK(s, hipStreamSynchronize(s->stream));
    return MDX_OK;
}

extern "C" int mdx_get_energies(mdx_system *s, double *e)
{
    (void)hipSetDevice(s->device);
    ARG_CHECK(s, e, "mdx_get_energies: null pointer");
    int rc = sync_and_check(s); if (rc) return rc;
    for (int k = 0; k < MDX_E_N; k++) e[k] = s->h->ener[k];
    return MDX_OK;
}

extern "C" int mdx_get_virial(mdx_system *s, double *vir9, double *pres9)
{
    (void)hipSetDevice(s->device);
    int rc = sync_and_check(s); if (rc) return rc;
    if (vir9) for (int k = 0; k < 9; k++) vir9[k] = s->h->vir[k];
    if (pres9) for (int k = 0; k < 9; k++) pres9[k] = s->h->pres[k];
    return MDX_OK;
}

extern "C" int mdx_get_list_stats(mdx_system *s, long long *st)
{
    (void)hipSetDevice(s->device);
    ARG_CHECK(s, st, "mdx_get_list_stats: null pointer");
    int rc = sync_and_check(s); if (rc) return rc;
    st[0] = s->h->nsc; st[1] = s->h->ncj; st[2] = s->h->nwork; st[3] = s->h->nexcl;
    st[4] = (long long)s->h->ntiles;
    st[5] = (long long)s->h->npairs_rc; st[6] = (long long)s->h->npairs_lj; st[7] = s->one_stream ? 1 : 3;
    return MDX_OK;
}

// Kernel timing uses a pool of events created when timing is switched on (creating an event costs tens of
// microseconds of host time, which has no place in the launch path); drain_timing reads the pairs used so far and
// makes the pool available again.
static int drain_timing(mdx_system *s)
{
    if (s->ev_nb.empty()) return MDX_OK;
    MDX_CHECK(s, hipStreamSynchronize(s->stream));
    for (auto &p : s->ev_nb) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, s->ev_pool[p.first], s->ev_pool[p.second]) == hipSuccess) { s->nb_ms_total += ms; s->nb_launches++; }
    }
    s->ev_nb.clear();
    return MDX_OK;
}

extern "C" int mdx_enable_kernel_timing(mdx_system *s, int on)
{
    (void)hipSetDevice(s->device);
    int rc = drain_timing(s); if (rc) return rc;
    s->time_nb = on != 0; s->nb_ms_total = 0; s->nb_launches = 0;
    s->time_stride = on > 1 ? on : 1;
    if (on && s->ev_pool.empty()) {
        s->ev_pool.resize(2 * 1024);
        for (hipEvent_t &e : s->ev_pool) MDX_CHECK(s, hipEventCreate(&e));
    }
    if (!on) {
        for (hipEvent_t e : s->ev_pool) (void)hipEventDestroy(e);
        s->ev_pool.clear();
    }
    return MDX_OK;
}

extern "C" int mdx_get_kernel_timing(mdx_system *s, double *ms, long long *launches)
{
    (void)hipSetDevice(s->device);
    int rc = drain_timing(s); if (rc) return rc;
    if (ms) *ms = s->nb_ms_total;
    if (launches) *launches = s->nb_launches;
    return MDX_OK;
}

// One MD step, or (domain decomposition) its two halves: phase 1 enqueues everything up to the charge
// spreading and the short-range kernels and records ev_spread behind the spreading; the caller then sums the
// charge grids of all domains; phase 2 enqueues the rest of the PME chain, the join, the update and the
// coupling kernels. Phase 0 is the whole step. `more` tells that another step follows in the same call.
// Domains that share the mesh transform (mdx_dd_mesh_slabs) cut the PME chain twice: phase 1 runs on through the x forward
// pass, phase 3 transforms this domain's planes (ev_spread recorded behind it as well), phase 2 starts at the x inverse pass.
static int step_phase(mdx_system *s, int phase, bool more, bool &forked_early, bool &prepared)
{
    int rc;
    const long long step = s->step;
    const bool do_search = s->steps_since_search >= s->nstlist;
    const bool do_ener = s->nstcalcenergy > 0 && (step % s->nstcalcenergy == 0);
    const bool do_pc = s->pcoupl != 0 && (step % s->nstpcouple == 0);
    const bool need_vir = do_ener || do_pc;
    const bool tc_apply = s->ntc > 0 && (step % s->nsttcouple == 0);
    bool pruned = false, do_prune = false, ke_in_update = false;
    const bool slabs = s->dd_on && s->dd_mesh_world > 0 && s->have_pme;
    if (phase == 3) {
        if (!slabs) return MDX_OK;
        if ((rc = mdx_launch_pme(s, need_vir && s->dd_mesh_energy, true, 2))) return rc;
        MDX_CHECK(s, hipEventRecord(s->ev_spread, s->st_pme));
        return MDX_OK;
    }
    if (phase != 2) {
        if (s->first_step && s->ntc > 0 && !(s->dd_on && !s->dd_own_tcouple)) {
            if ((rc = mdx_launch_clear(s, false, true, false))) return rc;
            if ((rc = mdx_launch_ke(s, false, false))) return rc;
            if ((rc = mdx_launch_tcouple(s))) return rc;
        }
        s->first_step = false;
        // (accumulators of a step with energies: cleared by the prepare kernel where one is launched for this step)
        if (need_vir && prepared) if ((rc = mdx_launch_clear(s, true, true, true))) return rc;
        // the leap-frog kernel leaves f cleared; only a force evaluation outside the step loop dirties it
        if (s->f_dirty) { MDX_CHECK(s, hipMemsetAsync(s->f, 0, sizeof(float) * 3 * s->n, s->stream)); s->f_dirty = false; }
        if (do_search) if ((rc = mdx_search(s))) return rc;
        // the side streams start when the prepare kernel completes (its completion signal is the fork event:
        // an event recorded by a separate marker costs the main stream several microseconds)
        // (a fused update of the previous step in this call has already left the sorted coordinates of this one)
        if (prepared) { if (!forked_early) if ((rc = mdx_fork(s))) return rc; }
        else if ((rc = mdx_launch_prepare(s, !forked_early, need_vir ? 7 : 0))) return rc;
        prepared = false;
        // Thermostat factors of the coupling period that the previous step ended: drawn on the bonded stream, which has just
        // been released behind that step's update; the join orders it ahead of this step's update, the first to use them.
        // On a step without energies or domain decomposition the bonded terms (and the pull) ride in the spreading launch; the bonded
        // stream then carries this step's work only when the thermostat's factors or a trajectory frame are due.
        const bool merge_bonded = phase == 0 && !need_vir && s->have_pme && !s->one_stream && !s->dd_on && !(s->dbg & 0x4000000);
        s->bond_busy = !merge_bonded || s->bond_carry;       // (bond_carry: the previous step left work there that this step's update needs)
        s->bond_carry = false;
        if (s->tc_fold_pending) { s->tc_fold_pending = false; s->bond_busy = true; if ((rc = mdx_launch_tcouple_fold(s, s->st_bond, s->tc_fold_step))) return rc; }
        // the nonbonded kernel is the critical path: enqueue it first, the side streams fill in
        if ((rc = mdx_launch_nonbonded(s, need_vir, MDX_F_NONBONDED | (s->have_pme ? MDX_F_EXCL : 0)))) return rc;
        if (phase == 0) {
            if ((rc = mdx_launch_pme(s, need_vir, false, 0, merge_bonded ? (MDX_F_BONDED | MDX_F_POSRES | (s->pull_on ? MDX_F_PULL : 0u)) : 0u))) return rc;
        } else if (s->have_pme) {
            if (slabs) { if ((rc = mdx_launch_pme(s, false, false, 1))) return rc; }
            else if ((rc = mdx_launch_pme_spread(s, s->st_pme))) return rc;
            MDX_CHECK(s, hipEventRecord(s->ev_spread, s->st_pme));
        }
        if (!merge_bonded) if ((rc = mdx_launch_bonded(s, need_vir, MDX_F_BONDED | MDX_F_POSRES))) return rc;
        // kinetic energy of the half step behind: reads only the velocities, which the update (behind the join) changes,
        // so it runs on the bonded stream beside the nonbonded kernel instead of ahead of it on the main stream
        if (need_vir) if ((rc = mdx_launch_ke(s, true, true, s->st_bond))) return rc;
        if (s->pull_on && !merge_bonded) {
            const bool log_now = (step % s->pull_nstout == 0);
            if ((rc = mdx_launch_pull(s, log_now, need_vir))) return rc;
        }
        // Rolling prune of the pair list (launched below, behind the join's marker on the bonded stream)
        do_prune = s->nstprune > 0 && phase == 0 && !s->dd_on && s->steps_since_search % s->nstprune == 0 &&
                   s->steps_since_search + 1 < s->nstlist;
        if (phase == 1) return MDX_OK;
    }
    if (phase == 2) if ((rc = slabs ? mdx_launch_pme(s, false, true, 3) : mdx_launch_pme(s, need_vir && s->dd_mesh_energy, true))) return rc;
    // trajectory frames captured ahead of this step: written to their pinned slots on the bonded stream, beside the nonbonded
    // kernel and ahead of the marker the update waits for (the update changes the box, and the next one the captured buffers)
    if (s->frame_ring) { bool any = false; if ((rc = mdx_frames_flush(s, s->st_bond, &any))) return rc; if (any) s->bond_busy = true; }
    if (phase == 2) s->bond_busy = true;
    if ((rc = mdx_join(s))) return rc;
    if (do_prune) {
        // Beside this step's nonbonded kernel (which still reads the list of the last prune, or on a search step the outer
        // list itself) the bonded stream derives the list of the coming steps from this step's sorted coordinates. The
        // launch stands behind the marker the join has just recorded on that stream, so the update does not wait for it:
        // it has to be complete when the sorted coordinates are next overwritten (mdx_wait_prune ahead of the fused update
        // or the next prepare kernel), which on a step that also accumulates energies is the next step.
        if ((rc = mdx_launch_prune(s, s->st_bond))) return rc;
        MDX_CHECK(s, hipEventRecord(s->ev_prune, s->st_bond));
        s->prune_pending = true;
        pruned = true;
    }
    // The PME and bonded kernels of the next step read only the new positions (the spread goes through the
    // sort permutation): when that step neither rebuilds the list nor accumulates energies they are
    // released here, ahead of the thermostat kernels and the coordinate sort, instead of by the prepare kernel.
    forked_early = false;
    {
        const long long nx = step + 1;
        const bool vir_next = (s->nstcalcenergy > 0 && nx % s->nstcalcenergy == 0) || (s->pcoupl != 0 && nx % s->nstpcouple == 0);
        const bool search_next = s->steps_since_search + 1 >= s->nstlist;
        const bool fork_now = phase == 0 && more && !vir_next && !search_next && !do_pc;
        // Steps that accumulate no virial take leap-frog and constraints in one launch, which (when another step of this
        // call follows on the same pair list and the cell stays as it is) also writes that step's sorted coordinates.
        const bool fused = phase == 0 && !need_vir && !s->dd_on && s->fused_update && mdx_fused_plan(s) == 1;
        // a step that ends a coupling period of the thermostat: the fused update also sums the new kinetic energy
        ke_in_update = fused && s->ntc > 0 && ((step + 1) % s->nsttcouple == 0);
        if (fused) {
            prepared = more && !search_next;
            if (prepared) if ((rc = mdx_wait_prune(s))) return rc;        // the launch rewrites the sorted coordinates
            if ((rc = mdx_launch_update_fused(s, tc_apply, s->pcoupl == 2, prepared, fork_now && prepared, ke_in_update))) return rc;
            if (fork_now && prepared) forked_early = true;
        } else if ((rc = mdx_launch_update(s, tc_apply, s->pcoupl == 2, need_vir))) return rc;
        if (fork_now && !forked_early) {
            MDX_CHECK(s, hipEventRecord(s->ev_fork, s->stream));
            MDX_CHECK(s, hipStreamWaitEvent(s->st_pme, s->ev_fork, 0));
            MDX_CHECK(s, hipStreamWaitEvent(s->st_bond, s->ev_fork, 0));
            forked_early = true;
        }
    }
    if (need_vir) {
        if ((rc = mdx_launch_ke(s, false, true))) return rc;
        // (a domain of a decomposed run leaves the scaling to the caller, who sums the pressure over the domains first: mdx_dd_pcouple)
        if ((rc = mdx_launch_finish_energy(s, do_ener ? next_energy_frame(s) : nullptr, do_pc && !s->dd_on))) return rc;
        if (do_ener) frame_steps(s).push_back(step);
    }
    if (ke_in_update) {
        // The factors for the coming period come from the sums the update has left. Nothing uses them before the next
        // update, so they are drawn beside the next step: on the bonded stream, at once when this step's update has
        // released that stream already, otherwise when the next step releases it.
        s->tc_fold_pending = true; s->tc_fold_step = step;
        if (forked_early) { s->tc_fold_pending = false; s->bond_carry = true; if ((rc = mdx_launch_tcouple_fold(s, s->st_bond, step))) return rc; }
    } else if (s->ntc > 0 && ((step + 1) % s->nsttcouple == 0) && !need_vir && !s->dd_on) {
        // the thermostat's own accumulator, which the kernel that draws the factor leaves zeroed: no clearing launch
        if ((rc = mdx_launch_ke(s, false, false, nullptr, true))) return rc;
        if ((rc = mdx_launch_tcouple(s, nullptr, true))) return rc;
    } else if (s->ntc > 0 && ((step + 1) % s->nsttcouple == 0)) {
        if (!need_vir) {
            if ((rc = mdx_launch_clear(s, false, true, false))) return rc;
            if ((rc = mdx_launch_ke(s, false, false))) return rc;
        }
        // (a domain of a decomposed run may leave the factor to the caller, who sums the kinetic energies first)
        if (!(s->dd_on && !s->dd_own_tcouple)) if ((rc = mdx_launch_tcouple(s))) return rc;
    }
    if (s->nstcomm > 0 && ((step + 1) % s->nstcomm == 0)) if ((rc = mdx_launch_comm(s))) return rc;
    s->step++;
    s->steps_since_search++;
    s->forces_current = false;
    if (pruned) { s->inner_cur ^= 1; s->inner_valid = true; }       // the list written beside this step serves from the next one
    return MDX_OK;
}

extern "C" int mdx_step(mdx_system *s, long long nsteps)
{
    (void)hipSetDevice(s->device);
    ARG_CHECK(s, nsteps >= 0, "mdx_step: nsteps must be >= 0");
    int rc;
    if (s->pull_on && s->pull_reset_pending) {
        // the pull reference lives on the device; make sure the flag starts cleared
        int zero = 0;
        MDX_CHECK(s, hipMemcpyAsync(&s->d->pull_have_init, &zero, sizeof(int), hipMemcpyHostToDevice, s->stream));
        MDX_CHECK(s, hipStreamSynchronize(s->stream));
        s->pull_reset_pending = false;
    }
    bool forked_early = false, prepared = false;
    for (long long it = 0; it < nsteps; it++)
        if ((rc = step_phase(s, 0, it + 1 < nsteps, forked_early, prepared))) return rc;
    return MDX_OK;
}

// Domain decomposition: the two halves of a step around the summation of the charge grids (csrc/domdec.hip).
int mdx_step_begin_internal(mdx_system *s) { bool fe = false, pr = false; return step_phase(s, 1, false, fe, pr); }
int mdx_step_end_internal(mdx_system *s) { bool fe = false, pr = false; return step_phase(s, 2, false, fe, pr); }
int mdx_step_mid_internal(mdx_system *s) { bool fe = false, pr = false; return step_phase(s, 3, false, fe, pr); }

extern "C" long long mdx_pop_energy_frames(mdx_system *s, double *out, long long max_frames)
{
    (void)hipSetDevice(s->device);
    if (sync_and_check(s) != MDX_OK) return -1;
    std::vector<long long> &steps = frame_steps(s);
    const long long have = (long long)steps.size();
    const long long avail = std::min<long long>(have, s->ener_log_cap);
    const long long ncopy = std::min(avail, max_frames);
    // frames are in the ring at indices (count - have ... count - 1); return the newest `ncopy` in order
    for (long long k = 0; k < ncopy; k++) {
        const long long fi = s->ener_log_count - ncopy + k;
        const long long slot = fi % s->ener_log_cap;
        const long long st = steps[(size_t)(have - ncopy + k)];
        double *o = out + k * (2 + MDX_E_N);
        o[0] = (double)st; o[1] = (double)st * s->dt;
        memcpy(o + 2, s->ener_log + slot * MDX_E_N, sizeof(double) * MDX_E_N);
    }
    steps.clear();
    return ncopy;
}

extern "C" long long mdx_pop_pull_records(mdx_system *s, double *out, long long max_records)
{
    (void)hipSetDevice(s->device);
    if (!s->pull_on) return 0;
    if (sync_and_check(s) != MDX_OK) return -1;
    // records [popped, count) are complete after the synchronisation above
    const long long count = s->pull_log_count;
    long long start = s->pull_log_popped;
    if (count - start > s->pull_log_cap) start = count - s->pull_log_cap;
    const long long n = std::min(count - start, max_records);
    for (long long k = 0; k < n; k++) memcpy(out + 4 * k, s->pull_log + 4 * ((start + k) % s->pull_log_cap), sizeof(double) * 4);
    s->pull_log_popped = start + n;
    return n;
}

// Diagnostic: average time of `iters` back-to-back runs of one stage of the step on its own
// stream, measured with HIP events. stage 1 pair search, 2 PME chain, 3 bonded + exclusion
// kernels, 4 update + constraints, 5 prepare, 6 PME spread only, 7 PME gather only (dbg bits select ablations),
// 8 fused update with the next step's sorted coordinates, 9 fused update without them.
extern "C" int mdx_debug_time_stage(mdx_system *s, int stage, int iters, double *avg_ms)
{
    (void)hipSetDevice(s->device);
    int rc = sync_and_check(s); if (rc) return rc;
    MDX_CHECK(s, hipDeviceSynchronize());
    hipStream_t st = s->stream;
    if (stage == 2 || stage == 6 || stage == 7) st = s->st_pme;
    if (stage == 3) st = s->st_bond;
    hipEvent_t a, b;
    MDX_CHECK(s, hipEventCreate(&a)); MDX_CHECK(s, hipEventCreate(&b));
    s->pme_part = stage == 6 ? 1 : (stage == 7 ? 2 : 0);
    auto run = [&]() -> int {
        switch (stage) {
        case 1: return mdx_search(s);
        case 2: case 6: case 7: return mdx_launch_pme(s, false);
        case 3: return mdx_launch_bonded(s, false, MDX_F_BONDED | MDX_F_POSRES);
        case 4: { int r = mdx_launch_update(s, false, false, false); float4 *t = s->x; s->x = s->xp; s->xp = t; return r; }
        case 5: return mdx_launch_prepare(s);
        case 8: case 9: {
            if (mdx_fused_plan(s) != 1) { s->err = "mdx_debug_time_stage: no fused update for this topology"; return MDX_ERR_ARG; }
            int r = mdx_launch_update_fused(s, false, false, stage == 8, false); float4 *t = s->x; s->x = s->xp; s->xp = t; return r;
        }
        default: s->err = "mdx_debug_time_stage: unknown stage"; return MDX_ERR_ARG;
        }
    };
    rc = run();
    MDX_CHECK(s, hipEventRecord(a, st));
    for (int i = 0; i < iters && rc == MDX_OK; i++) rc = run();
    MDX_CHECK(s, hipEventRecord(b, st));
    MDX_CHECK(s, hipEventSynchronize(b));
    float ms = 0.0f;
    MDX_CHECK(s, hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    s->pme_part = 0;
    s->f_dirty = true;
    MDX_CHECK(s, hipDeviceSynchronize());
    if (avg_ms) *avg_ms = ms / (iters > 0 ? iters : 1);
    return rc;
}
