// Internal state of one mdx system: one HIP device, one stream, all arrays resident in HBM.
#pragma once
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "mdx_math.h"
#include "../../include/mdx.h"
#include "../../include/mdx_debug.h"

#define MDX_MAX_TC 8
#define MDX_MAX_SHIFTS 48
#define MDX_NB_CHUNK 32          // j-clusters per nonbonded work unit
#define MDX_NB_MAXTYPES 32       // Lennard-Jones atom types the force kernel's LDS table holds (type 0: no well)
#define MDX_MAX_COL 2048         // atoms per grid column handled by the in-LDS sort
#define MDX_GTAB_N 4096          // points of the tabulated Ewald correction used by the force-only nonbonded kernel
#define MDX_LINCS_MAXK 16        // constraints per coupled component

#define MDX_NREP 64              // replicas of the energy-step accumulators (one per lane of the wave that folds them)
// Device-resident scalars: written by kernels, read by kernels, copied to the host on demand.
struct DevState {
    DevBox box;                   // float box used by every kernel
    double boxd[9];               // master copy (double) updated by pressure coupling
    // pair-search counters
    int nsc;                      // super-clusters (64 atoms)
    int ncj;                      // j-cluster entries
    int nwork;                    // nonbonded work units
    int nexcl;                    // exclusion-mask slots in use (slot 0 = all ones)
    int overflow;                 // bit flags: 1 cj list, 2 work list, 4 mask list, 8 column too tall
    int pad0[3];
    // per-step accumulators
    double ener[MDX_E_N];
    double vir[9];                // -1/2 sum r (x) F of all force terms + constraints
    double ekinh_old[MDX_MAX_TC]; // half-step kinetic energy per T-group before the update
    double ekinh[MDX_MAX_TC];     // ... after the update
    double ekinh_tc[MDX_MAX_TC];  // ... after the update, for the thermostat when it runs beside the next step (zeroed by its reader)
    double ekin_tensor[9];        // average of the two half-step tensors (energy steps)
    double lambda[MDX_MAX_TC];    // V-rescale scaling factor for the coming step
    double therm_integral[MDX_MAX_TC];
    double pres[9];               // pressure tensor (bar) of the last virial step
    double mu[9];                 // box/coordinate scaling matrix of the last coupling step
    double pr_boxv[9];            // Parrinello-Rahman box velocity
    double pr_M[9];               // Parrinello-Rahman velocity coupling matrix
    double baro_integral;
    double com_v[4];              // COM momentum accumulators (px,py,pz,mass)
    // pull
    double pull_com[2][4];        // mass-weighted sums (x,y,z,m) per group
    double pull_value, pull_force, pull_ref, pull_work;
    double pull_init;             // reference value at t=0 (set by pull start)
    int pull_have_init;
    int pad1;
    long long step;
    // pair counters of the last energy step: non-excluded pairs with r < rc, and those with LJ
    unsigned long long npairs_rc, npairs_lj;
    unsigned long long ntiles;    // 8x8 tiles in the pair list (statistics)
    int cost_hist[258];           // work units per tile count (last bin: 256 and more), filled by the list builder
    int cost_cursor[258];         // positions handed out per tile count by the sort of the units
    int cost_hist_in[258], cost_cursor_in[258];   // the same for the units of a pruned list (cleared ahead of every pruning)
    int nwork_in[2];              // units with a tile left in either pruned list
    int nb_head[8 * 32];          // dequeue counters of the nonbonded kernel's eight unit queues, one cache line each (cleared every step)
    // Replicated accumulators of the energy steps: thousands of workgroups adding doubles to one
    // address serialise in L2, so each adds to replica (block index mod MDX_NREP) and
    // k_finish_energy folds the replicas into ener / vir / ekin_tensor / npairs.
    double ener_rep[MDX_NREP][MDX_E_N];
    double vir_rep[MDX_NREP][9];
    double ekt_rep[MDX_NREP][9];
    unsigned long long npairs_rep[MDX_NREP][2];
    // `refcoord_scaling = com`: plain average of the restraint reference positions, scaled with the box, and its value
    // before the last coupling step (the scaling kernel shifts every reference by the difference)
    double posres_com[3], posres_com_old[3];
    long long ts_min[24], ts_max[24];   // diagnostic build (-DMDX_STAMPS): first start and last end of each kernel, 100 MHz clock
};
#ifdef __HIPCC__
// Diagnostic build only: every workgroup records when it starts and ends, per kernel id, so that one step's
// real timeline can be read without a tracing tool between the streams.
#ifdef MDX_STAMPS
struct MdxStamp {
    long long *mx;
    // first workgroup stamps the start, the last one (by index: about the last to be dispatched) the end; plain
    // stores, because thousands of workgroups updating one address would themselves stretch the kernel
    // (all_end 1: every workgroup stamps the end, for a kernel whose last workgroups may be idle; 2: every wave, for a
    // kernel whose waves draw work from queues and finish far apart)
    __device__ MdxStamp(const DevState *d, int id, int all_end = 0) {
        DevState *w = const_cast<DevState *>(d);
        mx = nullptr;
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) w->ts_min[id] = (long long)wall_clock64();
        if ((threadIdx.x == 0 && (all_end || (blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1))) ||
            (all_end == 2 && (threadIdx.x & 63) == 0)) mx = &w->ts_max[id];
    }
    __device__ ~MdxStamp() {
        if (!mx) return;
        // (all_end: every workgroup stores its end time, the last store to arrive stands; an atomic maximum on one address
        // from every workgroup stretched the kernel it was meant to time by tens of microseconds)
        *mx = (long long)wall_clock64();
    }
};
#define MDX_STAMP(d, id) MdxStamp stamp_((d), (id))
#define MDX_STAMP_ALL(d, id) MdxStamp stamp_((d), (id), 1)
#define MDX_STAMP_WAVES(d, id) MdxStamp stamp_((d), (id), 2)
#else
#define MDX_STAMP(d, id)
#define MDX_STAMP_ALL(d, id)
#define MDX_STAMP_WAVES(d, id)
#endif
__device__ __forceinline__ int mdx_rep() { return (int)((blockIdx.x + 5u * blockIdx.y) & (MDX_NREP - 1)); }
#define MDX_ENER(dw, t) ((dw)->ener_rep[mdx_rep()][t])
#define MDX_VIR(dw, k) ((dw)->vir_rep[mdx_rep()][k])
#define MDX_EKT(dw, k) ((dw)->ekt_rep[mdx_rep()][k])
#endif

// The pair list has a fixed region per i-super-cluster (and, for exclusion masks, per wave of the
// list builder), so building it needs no global counters.
#define MDX_CJ_PER_SC 1024
#define MDX_WORK_PER_SC 64
#define MDX_EXCL_PER_WAVE 48
struct NbEntry { int cj; unsigned int mask; };        // mask: bits 0-7 i-cluster mask, bits 8.. exclusion slot
struct NbWork { int sci; int shift; int cj_start; int cj_end; };

struct HostAtoms { std::vector<float> q, invm; };

struct mdx_system {
    int device = 0;
    HostAtoms host_atoms;
    std::vector<long long> frame_steps;   // steps of the energy frames not yet popped
    long long pull_log_popped = 0;
    bool pull_reset_pending = false;
    int dbg = 0;                       // diagnostic ablation bits (mdx_debug_set key 2)
    int nb_dyn = 1;                    // nonbonded kernel: rounds' worth of cheap units scheduled dynamically (mdx_debug_set key 3)
    int nb_dyn_split = 0;              // ... in 2^nb_dyn_split parts per unit (mdx_debug_set key 4)
    int pme_part = 0;                  // stage timing: 1 only the spreading, 2 only the gather of the PME chain
    long long *nb_wave_t = nullptr;    // diagnostic per-wave clock buffer (mdx_debug_wave_times)
    bool bond_carry = false;           // ... left there by the previous step (thermostat factors drawn behind its update)
    bool bond_busy = true;             // the bonded stream holds work of this step that the update has to wait for
    bool one_stream = false;           // st_pme and st_bond are the main stream (no cross-stream waits at all)
    hipStream_t own_streams[3] = { nullptr, nullptr, nullptr };   // the three streams created for the handle (destroyed with it)
    hipStream_t stream = nullptr;      // main stream: search, nonbonded, update
    hipStream_t st_pme = nullptr;      // PME chain (spread, FFTs, solve, gather) overlaps the nonbonded kernel
    hipStream_t st_bond = nullptr;     // bonded, restraint, exclusion and pull kernels
    hipEvent_t ev_fork = nullptr, ev_pme = nullptr, ev_bond = nullptr, ev_spread = nullptr, ev_prune = nullptr;
    bool prune_pending = false;        // a pruning launch on the bonded stream has not been waited for yet (mdx_wait_prune)
    float *f_pme = nullptr;            // 3N mesh forces (plain stores, so no race with the atomics on f)
    std::string err;
    int n = 0;                    // atoms
    int npad = 0;                 // capacity of sorted arrays (multiple of 64)
    // ---- state (original atom order)
    float4 *x = nullptr;          // xyz + charge
    float4 *xp = nullptr;         // scratch positions (unconstrained update)
    float4 *v = nullptr;          // xyz + inverse mass
    float *f = nullptr;           // 3N forces from bonded/PME/restraint kernels (original order)
    float2 *lj = nullptr;         // (sigma/2, 2 sqrt(eps)) per atom
    int *molid = nullptr;
    int *tcgrp = nullptr;         // T-coupling group per atom
    int *excl_start = nullptr, *excl_list = nullptr;
    int n_excl_pairs = 0;
    int2 *excl_pairs = nullptr;   // i<j list for the Ewald exclusion correction
    double q2sum = 0.0;
    // ---- sorted (cluster) layout
    float4 *xq_s = nullptr;       // npad
    float2 *lj_s = nullptr;
    // Lennard-Jones by atom type (force kernel of the step loop): atoms with the same (sigma, epsilon) share a type, type 0 has no LJ;
    // ljtype: 8 * type per atom, ljt_s: the same in cluster order, nbfp: nt x nt entries (12 C12, 6 C6); nt = 0: more types than the kernel's table holds
    int *ljtype = nullptr, *ljt_s = nullptr; float2 *nbfp = nullptr; int nt = 0;
    float *f_s = nullptr;         // 3*npad
    int *perm = nullptr;          // sorted -> original (-1 filler)
    int *perm_src = nullptr;      // sorted -> original atom whose position is used (always valid)
    int *shift_s = nullptr;       // packed lattice shift applied when wrapping
    int *molid_s = nullptr;
    int *sorted_of = nullptr;     // original -> sorted
    float4 *bb_lo = nullptr, *bb_hi = nullptr;     // per cluster
    float4 *sbb_lo = nullptr, *sbb_hi = nullptr;   // per super-cluster
    // grid
    int ncx = 0, ncy = 0, ncol_cap = 0;
    int *col_count = nullptr, *col_astart = nullptr, *col_scstart = nullptr, *col_atoms = nullptr;
    int *atom_col = nullptr, *atom_slot = nullptr;
    float4 *wrapped = nullptr;
    // lists
    NbEntry *cj = nullptr; int cj_cap = 0;
    NbWork *work = nullptr; int work_cap = 0;
    int *work_order = nullptr;    // unit indices by descending tile count
    int *sc_nwork = nullptr;      // work units in use per i-super-cluster
    unsigned char *exclmask = nullptr; int excl_cap = 0;
    int nshift = 0, central_shift = 0;
    int3 *shift_ijk = nullptr;    // device
    float4 *shiftvec = nullptr;   // device, refreshed each step from the box
    std::vector<int3> h_shift_ijk, h_shift_sent;     // lattice shifts of the current box; the list last copied to the device
    // nonbonded parameters
    float rc = 1.0f, rlist = 1.1f, beta = 3.12341f;
    float4 *htab = nullptr;       // ... with h(w) = erf(z)/z beside g, for the energy kernel
    float2 *gtab = nullptr;       // MDX_GTAB_N entries of beta^3 g (Ewald correction of the force) on a uniform r^2 grid up to rc^2: per interval the line c0 + c1 r^2
    int nstlist = 10;
    // Dual pair list (mdx_set_pairlist_prune): the list above is built at `rlist` for `nstlist` steps; every `nstprune`
    // steps a kernel on the bonded stream keeps the tiles of it that have a pair within `rlist_inner` right now, and the
    // nonbonded kernel of the following steps walks that shorter list. Two inner lists: one is read while the other is written.
    float rlist_inner = 0.0f; int nstprune = 0;
    int nb_chunk = MDX_NB_CHUNK;       // entries per work unit the builder cuts (<= 64: one list entry per lane)
    NbEntry *cj_in[2] = { nullptr, nullptr }; NbWork *work_in[2] = { nullptr, nullptr }; int *order_in[2] = { nullptr, nullptr };
    int inner_cur = 0;                 // the inner list the nonbonded kernel reads
    bool inner_valid = false;          // ... exists for the current outer list
    // PME
    int K[3] = {0, 0, 0};
    float *grid = nullptr; float2 *cgrid = nullptr;
    float *phi = nullptr;         // potential grid written by the own transform's last pass (the charge grid stays zeroed)
    bool grid_clean = false;      // the charge grid is known to hold zeros
    float *bmod[3] = {nullptr, nullptr, nullptr};
    hipfftHandle plan_r2c = 0, plan_c2r = 0;
    float2 *fftW[3] = {nullptr, nullptr, nullptr};   // twiddle tables exp(-2 pi i j / K) of the own transform
    bool fft_yz = false;          // ... and a k_x plane fits into LDS with factors <= 16: y and z passes fused into one launch
    bool fft_yz_big = false;      // ... in the form for planes beyond 96 KB or factors beyond 10
    bool own_fft = false;         // every grid dimension splits into two factors <= 32: the own transform is used
    bool have_pme = false;
    // bonded
    int nbonds = 0; int2 *bond_ij = nullptr; float2 *bond_p = nullptr;
    int nangles = 0; int4 *angle_ijk = nullptr; float2 *angle_p = nullptr;
    int ndih = 0; int4 *dih_ijkl = nullptr; float4 *dih_p = nullptr;
    int npairs = 0; int2 *pair_ij = nullptr; float4 *pair_p = nullptr;
    int nposres = 0; int *posres_idx = nullptr; float4 *posres_ref = nullptr; float4 *posres_k = nullptr;
    int refcoord_scaling = 0;     // 0 no, 1 all, 2 com
    int perm_n = -1;              // atom count the cluster order in `perm` was built for (-1: none yet)
    // constraints
    int nsettle_cap = 0;          // settles set up (domain decomposition activates the first nsettle of them)
    int nsettle = 0; int4 *settle_idx = nullptr; float settle_doh = 0, settle_dhh = 0;
    int settle_first = -1;        // first atom of the first water when the waters' atoms are consecutive in the order listed, else -1
    int ncons = 0; int ncomp = 0; int ncomp_small = 0;   // components are sorted by size; the first ncomp_small have <= 4 constraints
    int2 *cons_ij = nullptr; float *cons_len = nullptr; int *comp_start = nullptr;  // constraints sorted by component
    int lincs_order = 4, lincs_iter = 1;
    // fused update (leap-frog + constraints + next step's sorted coordinates in one launch; update.hip)
    std::vector<int> h_settle; std::vector<int2> h_cij; std::vector<int> h_cstart;   // host copies of the constraint topology
    int fused_plan = 0;           // 0: not built yet, 1: usable, -1: this topology needs the separate kernels
    int fused_update = 1;         // use it on steps without virial (mdx_debug_set key 12)
    int *comp_atom_start = nullptr, *comp_atoms = nullptr, *comp_local = nullptr, *free_atoms = nullptr; int nfree = 0;
    // integration / coupling
    float dt = 0.002f;
    int ntc = 0; double ref_t[MDX_MAX_TC] = {0}, tau_t[MDX_MAX_TC] = {0}, ndf[MDX_MAX_TC] = {0};
    int nsttcouple = 10; uint64_t seed = 1;
    double *d_tc = nullptr;       // device: ref_t, tau_t, ndf (3 x MDX_MAX_TC)
    double *ke_part = nullptr; int ke_part_cap = 0, ke_parts = 0;    // per-workgroup kinetic-energy sums of the fused update (KE variant)
    bool tc_fold_pending = false; long long tc_fold_step = 0;       // ... not yet folded into the thermostat's factors (mdx_flush_tcouple)
    double ndf_total = 0;
    int pcoupl = 0;               // 0 no, 1 C-rescale, 2 Parrinello-Rahman
    double tau_p = 1.0, ref_p = 1.0, compress = 4.5e-5; int nstpcouple = 10;
    int nstcomm = 100;
    int nstcalcenergy = 100;
    double dispcorr_c6avg = 0.0; int dispcorr = 0;   // 0 no, 1 Ener, 2 EnerPres
    // pull
    int pull_on = 0; int pull_n[2] = {0, 0}; int *pull_idx[2] = {nullptr, nullptr};
    int pull_pbcatom[2] = {0, 0};
    int pull_type = 0;            // 0 constant-force, 1 umbrella
    double pull_k = 0, pull_rate = 0, pull_init_user = 0; int pull_start = 0;
    int pull_dim[3] = {1, 1, 1};
    int pull_nstout = 50;
    double *pull_log = nullptr;   // pinned host ring: (time, value, force, work) records
    int pull_log_cap = 0;
    long long pull_log_count = 0;
    // scalars
    DevState *d = nullptr;        // device
    DevState *h = nullptr;        // pinned host mirror
    long long step = 0;           // host-side step counter (steps enqueued)
    long long steps_since_search = 1 << 30;
    bool first_step = true;
    bool forces_current = false;
    bool f_dirty = true;          // f holds something other than zeros at the start of the next step
    // energy history ring (pinned)
    double *ener_log = nullptr; int ener_log_cap = 0; long long ener_log_count = 0;
    // timing of the dominant kernel (HIP events on this stream)
    unsigned nb_launch_seq = 0;
    bool count_pairs = false;          // the next launch of the energy kernel counts the pairs inside the cutoff (statistics)
    bool count_pairs_steps = false;    // ... also on the energy steps of mdx_step (mdx_debug_set key 11); mdx_compute_forces always counts
    bool pme_signalled = false;        // ev_pme already attached to the last kernel of the PME chain
    bool time_nb = false; int time_stride = 7; std::vector<hipEvent_t> ev_pool; std::vector<std::pair<int, int>> ev_nb;
    double nb_ms_total = 0; long long nb_launches = 0;
    // domain decomposition (csrc/domdec.hip)
    bool dd_on = false;
    bool dd_mesh_energy = true;        // this domain accumulates the reciprocal-space energy and virial
    bool dd_own_tcouple = true;        // the step computes the thermostat factor itself (false: mdx_dd_tcouple after the caller's sum)
    // mesh transform shared among the domains (mdx_dd_mesh_slabs): of the K0 / 2 + 1 planes of the half-complex grid this
    // domain transforms [dd_plane_first, + dd_plane_count); the grid holds dd_mesh_world equal chunks of dd_plane_chunk planes
    int yz_lds_limit = 96 * 1024;      // largest plane (with its twiddle tables) the fused y/z kernel takes; MDX_YZ_LDS_KB overrides (a CU has 160 KB)
    int dd_mesh_world = 0, dd_plane_chunk = 0, dd_plane_first = 0, dd_plane_count = 0;
    float *dd_q0 = nullptr, *dd_invm0 = nullptr; float2 *dd_lj0 = nullptr;   // per-slot pattern as set up
    double search_density = 0.0;  // atoms per nm^3 for the sizing of the search grid (0: atoms / box volume)
    void *frame_ring = nullptr;   // pinned trajectory-frame slots (runctl.hip)
};

// Zero the accumulators that a step's kernels add into; threads t = 0 .. 63 of one workgroup.
__device__ __forceinline__ void mdx_clear_accumulators(DevState *dw, const int t, const int energies, const int ekin_new, const int ekin_old)
{
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

#define MDX_CHECK(sys, call)                                                                         \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            (sys)->err = std::string(#call) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" +     \
                         std::to_string(__LINE__) + ")";                                             \
            return MDX_ERR_HIP;                                                                      \
        }                                                                                            \
    } while (0)

// kernels' host launchers (each returns MDX_OK or an error code)
int mdx_set_stream_mode(mdx_system *s, int one_stream);
int mdx_search(mdx_system *s);
int mdx_launch_prune(mdx_system *s, hipStream_t st);
int mdx_wait_prune(mdx_system *s);
int mdx_fork(mdx_system *s);
int mdx_join(mdx_system *s);
int mdx_launch_prepare(mdx_system *s, bool fork = false, int clear_mask = 0);      // clear_mask: 1 energies, 2 new, 4 old half-step kinetic energy
int mdx_launch_nonbonded(mdx_system *s, bool energy, unsigned terms);
// stage 0: the whole chain; 1: spreading (unless done) and x forward; 2: y/z transforms and influence function on this
// domain's planes; 3: x inverse and gather
int mdx_launch_pme(mdx_system *s, bool energy, bool spread_done = false, int stage = 0, unsigned bonded_flags = 0);
int mdx_launch_pme_spread(mdx_system *s, hipStream_t st, unsigned bonded_flags = 0);
struct BondedArgs;
void mdx_bonded_args(mdx_system *s, unsigned flags, BondedArgs &A);
struct PullArgs;
void mdx_pull_args(mdx_system *s, bool log_now, bool energy, PullArgs &P);
int mdx_launch_bonded(mdx_system *s, bool energy, unsigned flags);
int mdx_launch_pull(mdx_system *s, bool log_now, bool energy);
int mdx_launch_update(mdx_system *s, bool do_tc, bool do_pr, bool virial);
int mdx_launch_constrain(mdx_system *s, float inv_dt, bool virial, int lincs_iter);
int mdx_fused_plan(mdx_system *s);
int mdx_launch_update_fused(mdx_system *s, bool do_tc, bool do_pr, bool write_sorted, bool fork_signal, bool with_ke = false);
int mdx_launch_tcouple_fold(mdx_system *s, hipStream_t st, long long step);
void mdx_frames_free(mdx_system *s);
int mdx_frames_flush(mdx_system *s, hipStream_t st, bool *any = nullptr);      // launch the kernels of captured trajectory frames on `st` (behind what it holds)
int mdx_launch_ke(mdx_system *s, bool old_slot, bool tensor, hipStream_t st = nullptr, bool tc_slot = false);
int mdx_launch_comm_phase(mdx_system *s, int phase);
int mdx_launch_clear(mdx_system *s, bool energies, bool ekin_new, bool ekin_old);
int mdx_launch_tcouple(mdx_system *s, hipStream_t st = nullptr, bool tc_slot = false);
int mdx_launch_clear(mdx_system *s, bool energies, bool ekin_new, bool ekin_old);
int mdx_launch_pcouple(mdx_system *s);
int mdx_launch_comm(mdx_system *s);
int mdx_launch_finish_energy(mdx_system *s, double *frame = nullptr, bool with_pcouple = false);
int mdx_launch_merge_forces(mdx_system *s, float *out3n);
int mdx_pme_init(mdx_system *s);
void mdx_pme_free(mdx_system *s);
