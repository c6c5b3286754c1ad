/*
 * mdx.h — C ABI of libmdx.so, the MI355X-native MD engine behind the pipeline's
 * `gmx mdrun` stages.
 *
 * The reference pipeline has no in-process API: every stage is the process pair
 *     gmx grompp -f F.mdp -c C.gro [-r R.gro] [-t T.cpt] -p P.top [-n N.ndx] -o O.tpr
 *     gmx mdrun -v -deffnm PREFIX -ntmpi 1 -ntomp 4
 * (/root/reference/Snakefile:705-707, 738-740, 311-315, 994-998;
 *  /root/reference/scripts/run_smd_with_distance_monitoring.py:66-75, 89-99).
 * This header is the boundary a host (the Python `gmx` shim in molecular-dynamics_amd/cli.py,
 * or any FFI) binds instead: plain pointers and sizes, int return codes, no exceptions, no
 * torch types. Each group of setters names the .mdp keys / topology sections whose GROMACS
 * behaviour it replaces.
 *
 * Conventions: units nm, ps, amu, kJ/mol, e, K, bar. Arrays are host pointers unless the
 * name ends in _device. Indices are 0-based. One handle = one HIP device + one stream; a
 * handle is not thread-safe. All calls return MDX_OK (0) or a negative error code;
 * mdx_last_error() gives the text.
 */
#ifndef MDX_H
#define MDX_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mdx_system mdx_system;

enum {
    MDX_OK = 0,
    MDX_ERR_ARG = -1,
    MDX_ERR_HIP = -2,
    MDX_ERR_STATE = -3,
    MDX_ERR_OVERFLOW = -4,
    MDX_ERR_UNSUPPORTED = -5
};

/* energy term indices for mdx_get_energies() */
enum {
    MDX_E_LJ = 0,        /* LJ (SR), potential-shifted */
    MDX_E_COUL_SR,       /* Ewald real space, potential-shifted */
    MDX_E_COUL_RECIP,    /* PME mesh */
    MDX_E_COUL_EXCL,     /* excluded-pair correction + Ewald self term */
    MDX_E_BOND,
    MDX_E_ANGLE,
    MDX_E_DIH,
    MDX_E_LJ14,
    MDX_E_COUL14,
    MDX_E_POSRES,
    MDX_E_PULL,
    MDX_E_DISPCORR,
    MDX_E_KIN,           /* leap-frog average of the two half-step kinetic energies */
    MDX_E_POT,           /* sum of all potential terms */
    MDX_E_TEMP,
    MDX_E_PRES,          /* scalar pressure, bar */
    MDX_E_VOLUME,
    MDX_E_CONSERVED,     /* total energy minus thermostat/barostat work */
    MDX_E_N
};

/* force-term selection bits for mdx_compute_forces() */
enum {
    MDX_F_NONBONDED = 1,  /* LJ + Ewald real space on the cluster pair list */
    MDX_F_PME = 2,        /* PME mesh part */
    MDX_F_EXCL = 4,       /* excluded-pair Ewald correction (+ self energy) */
    MDX_F_BONDED = 8,     /* bonds, angles, dihedrals, 1-4 pairs */
    MDX_F_POSRES = 16,
    MDX_F_PULL = 32,
    MDX_F_ALL = 63
};

int mdx_version(void);
/* Text of the last error on this handle (or of the last failed mdx_create when sys==NULL). */
const char *mdx_last_error(const mdx_system *sys);

/* Create a system of natoms atoms on HIP device `device`. Fails (returns NULL) when no HIP
 * device is usable: there is no CPU fallback. */
mdx_system *mdx_create(int device, int natoms);
void mdx_destroy(mdx_system *sys);

/* ---- topology: what `gmx grompp` bakes into the .tpr from the .top/.itp files
 * (/root/reference/scripts/create_complex_topology.py:65-122, fix_ligand_topology.py:29-96). */
/* Per-atom mass, charge, Lorentz-Berthelot sigma/epsilon (AMBER comb-rule 2), molecule index
 * and T-coupling group (`tc-grps`, /root/reference/mdp/production.mdp:32). */
int mdx_set_atoms(mdx_system *sys, const float *mass, const float *charge, const float *sigma,
                  const float *epsilon, const int *molid, const int *tc_group);
/* Exclusions as CSR over atoms (`nrexcl`, `[ exclusions ]`, `[ settles ]` implied pairs). */
int mdx_set_exclusions(mdx_system *sys, const int *excl_start, const int *excl_list);
/* `[ bonds ]` funct 1: idx 2 ints each; params (b0, kb). */
int mdx_set_bonds(mdx_system *sys, int n, const int *idx, const float *b0, const float *kb);
/* `[ angles ]` funct 1: idx 3 ints each; theta0 in degrees. */
int mdx_set_angles(mdx_system *sys, int n, const int *idx, const float *theta0_deg, const float *k);
/* `[ dihedrals ]` funct 1/4/9 (periodic): idx 4 ints each. */
int mdx_set_dihedrals(mdx_system *sys, int n, const int *idx, const float *phi0_deg, const float *k,
                      const int *mult);
/* `[ pairs ]` funct 1 with fudgeLJ/fudgeQQ already applied: c6, c12, qq (= fudgeQQ*qi*qj). */
int mdx_set_pairs14(mdx_system *sys, int n, const int *idx, const float *c6, const float *c12,
                    const float *qq);
/* `[ position_restraints ]` under `define = -DPOSRES` (/root/reference/mdp/nvt_complex.mdp:2);
 * ref = the `grompp -r` coordinates; refcoord_scaling 0 no / 1 all (/root/reference/mdp/npt_complex.mdp:72:
 * every reference position scales with the box) / 2 com (/root/reference/mdp/npt.mdp:40: the references move
 * rigidly with the scaled plain average of their positions). The mode also sets the arm of the restraint
 * virial: displacement only (all), plus the reference relative to that average (com), plus the whole
 * reference (no). */
int mdx_set_posres(mdx_system *sys, int n, const int *idx, const float *ref_xyz, const float *k_xyz,
                   int refcoord_scaling);
/* Current restraint reference positions (n x 3, in the order given to mdx_set_posres), as pressure coupling
 * has moved them; what a checkpoint has to carry for `refcoord_scaling` (same keys as above). */
int mdx_get_posres_ref(mdx_system *sys, float *ref_xyz);
/* `[ settles ]` (tip3p.itp): triples (O,H,H). */
int mdx_set_settles(mdx_system *sys, int n, const int *idx, float d_oh, float d_hh);
/* `constraints = h-bonds`, `constraint_algorithm = lincs`, `lincs_order`, `lincs_iter`
 * (/root/reference/mdp/production.mdp:16-19; pull_short.mdp:12-15). */
int mdx_set_constraints(mdx_system *sys, int n, const int *idx, const float *length, int lincs_order,
                        int lincs_iter);

/* ---- run parameters (the .mdp contract) */
/* `rcoulomb`/`rvdw` (equal), pair-list radius, Ewald beta from `ewald-rtol`, `nstlist`. */
int mdx_set_nonbonded(mdx_system *sys, float rc, float rlist, float ewald_beta, int nstlist);
/* Dual pair list with a rolling prune: what mdrun does with `verlet-buffer-tolerance` when the list lives long
 * (`nstlist = 200`, /root/reference/mdp/npt_complex.mdp:29-32). The list of mdx_set_nonbonded (radius `rlist`, rebuilt
 * every `nstlist` steps) becomes the outer list; every `nstprune` steps the engine derives from it, beside the running
 * step, the list of the pairs of clusters that are within `rlist_inner` at that moment, and the nonbonded kernel of
 * the following steps walks that one. `rlist_inner` has to hold the buffer of nstprune + 1 steps, `rlist` that of
 * `nstlist` steps. nstprune = 0 switches the scheme off (one list, as set by mdx_set_nonbonded). Call after
 * mdx_set_nonbonded; results are those of the single list up to the pairs either buffer lets slip. */
int mdx_set_pairlist_prune(mdx_system *sys, float rlist_inner, int nstprune);
/* `coulombtype = PME`, `pme_order = 4`, grid from `fourierspacing`. nx=0 disables the mesh. */
int mdx_set_pme(mdx_system *sys, int nx, int ny, int nz, int order);
/* `dt`, `nstcalcenergy`, `nstcomm`. */
int mdx_set_integrator(mdx_system *sys, float dt, int nstcalcenergy, int nstcomm);
/* `tcoupl = V-rescale`: per group ref_t, tau_t, degrees of freedom; `nsttcouple`; `ld-seed`.
 * ngroups = 0 switches the thermostat off (NVE). */
int mdx_set_tcoupl(mdx_system *sys, int ngroups, const double *ref_t, const double *tau_t,
                   const double *ndf, int nsttcouple, unsigned long long seed);
/* `pcoupl`: 0 no, 1 C-rescale, 2 Parrinello-Rahman; isotropic; `tau_p`, `ref_p`,
 * `compressibility`, `nstpcouple` (/root/reference/mdp/production.mdp:36-40). */
int mdx_set_pcoupl(mdx_system *sys, int kind, double tau_p, double ref_p, double compressibility,
                   int nstpcouple);
/* `DispCorr`: 0 no, 1 Ener, 2 EnerPres; c6_avg = <C6> over all atom pairs. */
int mdx_set_dispcorr(mdx_system *sys, int mode, double c6_avg);
/* `pull = yes`, one `distance` coordinate between two groups
 * (/root/reference/mdp/pull_short.mdp:36-46). type 0 constant-force (force = -k along the
 * distance), 1 umbrella (reference = init + rate*t). start!=0 adds the initial distance to init.
 * pbcatom: index into each group of the PBC reference atom (-1 = number-wise middle atom). */
int mdx_set_pull(mdx_system *sys, int n1, const int *group1, int n2, const int *group2, int pbcatom1,
                 int pbcatom2, int type, double k, double rate, double init, int start,
                 const int *dim_xyz, int nstout);

/* ---- state */
/* x, v: natoms*3 floats (v may be NULL = zero); box: 9 floats, rows a,b,c (lower triangular). */
int mdx_set_state(mdx_system *sys, const float *x, const float *v, const float *box9);
/* Same, from device pointers (e.g. torch-ROCm tensors); copied device-to-device on the
 * handle's stream. box9 is a host pointer. */
int mdx_set_state_device(mdx_system *sys, const void *x_device, const void *v_device, const float *box9);
int mdx_get_state(mdx_system *sys, float *x, float *v, float *box9);
long long mdx_get_step(const mdx_system *sys);
/* Continue from a checkpoint: the step counter drives output times, coupling cadence and the
 * counter-based random streams (`gmx mdrun -cpi`,
 * /root/reference/scripts/run_smd_with_recovery.py:149-156). */
int mdx_set_step(mdx_system *sys, long long step);
/* Run state beside x, v and box: the scalars that stochastic coupling, Parrinello-Rahman and the pull code carry
 * from step to step on the device. A checkpoint written for `gmx mdrun -cpi`
 * (/root/reference/scripts/run_smd_with_recovery.py:144-156; stage chaining with `grompp -t`,
 * /root/reference/Snakefile:738,994) stores these MDX_RS_N doubles; mdx_set_run_state (after mdx_set_state,
 * mdx_set_step and mdx_set_pull) makes the continuation use the scaling factors, box velocity, pull reference and
 * work integral of the interrupted run instead of re-seeding them. EKINH, EKINH_OLD, MU and COM_V are read-only
 * (the half-step kinetic energies per T-group, the last coupling step's scaling matrix and the last centre-of-mass
 * momentum sum px, py, pz, mass). */
enum {
    MDX_RS_LAMBDA = 0,            /* 8: V-rescale factor per T-group for the coming coupling step */
    MDX_RS_THERM_INTEGRAL = 8,    /* 8: thermostat work per T-group */
    MDX_RS_EKINH = 16,            /* 8 */
    MDX_RS_EKINH_OLD = 24,        /* 8 */
    MDX_RS_PR_BOXV = 32,          /* 9: Parrinello-Rahman box velocity, nm/ps */
    MDX_RS_MU = 41,               /* 9 */
    MDX_RS_BARO_INTEGRAL = 50,
    MDX_RS_PULL_INIT = 51,        /* umbrella reference at t = 0 (init + initial distance with start = yes) */
    MDX_RS_PULL_WORK = 52,
    MDX_RS_PULL_VALUE = 53,
    MDX_RS_PULL_HAVE_INIT = 54,
    MDX_RS_COM_V = 55,            /* 4 */
    MDX_RS_N = 64
};
int mdx_get_run_state(mdx_system *sys, double *rs);
int mdx_set_run_state(mdx_system *sys, const double *rs);
/* `gen_vel = yes`, `gen_temp`, `gen_seed` (/root/reference/mdp/nvt_complex.mdp:39-41):
 * Maxwell-Boltzmann velocities, centre-of-mass motion removed, scaled to exactly `temperature`
 * over 3N degrees of freedom. */
int mdx_generate_velocities(mdx_system *sys, double temperature, unsigned long long seed);
/* `continuation = no` (/root/reference/mdp/nvt_complex.mdp:13): project the start coordinates
 * onto the constraints and remove the velocity components along them. */
int mdx_constrain_start(mdx_system *sys);
/* `integrator = steep`, `emtol`, `emstep`, `nsteps` (/root/reference/mdp/em_complex.mdp:3-6):
 * steepest descent with the force kernels of the MD path; step size grows by 1.2 on an accepted
 * move and halves on a rejected one. result6: steps done, status (1 converged to emtol, 0 step
 * limit, 2 step size underflow), final Epot, final Fmax, first Epot, first Fmax. `progress` may
 * be NULL. */
int mdx_minimize(mdx_system *sys, long long nsteps, double emtol, double emstep, double *result6,
                 void (*progress)(long long step, double epot, double fmax, double stepsize));
/* Trajectory output cadence (`nstxout`, `nstvout`, `nstxout-compressed`,
 * /root/reference/mdp/production.mdp:7-12): a ring of pinned host slots. mdx_capture_frame
 * takes the state behind the steps already enqueued (x, v if wanted, box) and returns at once: the
 * kernels that write it into the slot's pinned memory are launched by the next mdx_step, beside
 * that step's force kernels (or by mdx_synchronize, by any call that replaces the state, or by a
 * wait that no step has served within 2 ms). mdx_frame_wait blocks until the frame has landed and
 * may be called from a writer thread while the stepping thread keeps enqueuing. Pointers stay valid
 * until the slot is captured into again; a slot cannot be captured into twice without a step, a
 * synchronisation or a wait in between (MDX_ERR_ARG). */
int mdx_frame_ring(mdx_system *sys, int nslots);
int mdx_capture_frame(mdx_system *sys, int slot, int want_v);
int mdx_frame_wait(mdx_system *sys, int slot, const float **x, const float **v, double *box9,
                   long long *step);
/* The capture for `nstxout-compressed` frames (/root/reference/mdp/production.mdp:12-13): the positions arrive as the
 * integers of the xtc coordinate block, l = (int)(x * precision +- 0.5) in single precision as the host encoder
 * rounds them (`compressed-x-precision`, default 1000), with the frame's statistics stats8 = extent minimum (3),
 * extent maximum (3), smallest sum of absolute differences between neighbouring atoms, flag (1: a coordinate is
 * not finite or does not fit an int). The host is left with the bit packing. Velocities, box and step as above. */
int mdx_capture_frame_xtc(mdx_system *sys, int slot, int want_v, float precision);
int mdx_frame_wait_xtc(mdx_system *sys, int slot, const int **ix, const float **v, double *box9,
                       long long *step, int *stats8);

/* ---- compute */
/* Single-point forces and energies of the selected terms at the current positions (rebuilds
 * the pair list). Results via mdx_get_forces / mdx_get_energies / mdx_get_virial. */
int mdx_compute_forces(mdx_system *sys, unsigned terms);
int mdx_get_forces(mdx_system *sys, float *f);          /* natoms*3 */
/* mdx_get_forces for callers that cannot allocate device memory (tight serving loops, harnesses that forbid
 * hipMalloc). No counterpart in the reference. Same result as mdx_get_forces, which stages the merged forces in a
 * buffer it allocates and frees per call; this one stages them in a fixed array that is part of the library's code
 * object (12 MiB per GPU, held until the process ends) and calls no hipMalloc, hipFree or other allocating routine.
 * It synchronises the handle's stream and copies to the host, so it is not a call to make while a stream is being
 * captured into a graph. Limits: 1 048 576 atoms (MDX_ERR_OVERFLOW above that); device indices 0 .. 63
 * (MDX_ERR_UNSUPPORTED above that). `f` is a host buffer of natoms*3 floats; pass pinned memory where the runtime's own
 * staging of pageable copies is unwanted. Every handle of a process on one GPU shares the array: a per-device lock
 * inside the call makes calls from several threads take turns; processes have separate copies. */
int mdx_get_forces_noalloc(mdx_system *sys, float *f);
int mdx_get_energies(mdx_system *sys, double *e);       /* MDX_E_N doubles, last energy step */
int mdx_get_virial(mdx_system *sys, double *vir9, double *pres9);
/* Advance nsteps leap-frog steps (replaces `gmx mdrun` for `integrator = md`). Enqueues work
 * on the handle's stream and returns; any getter synchronises. */
int mdx_step(mdx_system *sys, long long nsteps);
int mdx_synchronize(mdx_system *sys);
/* Pair-list statistics (8 values): of the last search [0] super-clusters, [1] j-cluster
 * entries, [2] work units, [3] exclusion masks, [4] cluster pairs (set bits of the i-masks);
 * of the last energy step [5] non-excluded atom pairs with r < rc (each once), [6] those of
 * them with a Lennard-Jones term; [7] streams the handle uses (3; 1 in the diagnostic one-stream mode). */
int mdx_get_list_stats(mdx_system *sys, long long *stats8);
/* Energy frames recorded every nstcalcenergy steps since the last call: each frame is
 * (step, time, MDX_E_N terms). Returns the number of frames copied (<= max_frames). */
long long mdx_pop_energy_frames(mdx_system *sys, double *out, long long max_frames);
/* Pull records (time, value, force, work) written every pull nstout steps since the last call.
 * These are what PREFIX_pullx.xvg / PREFIX_pullf.xvg carry
 * (/root/reference/scripts/analyze_smd_trajectory.py:18-37). */
long long mdx_pop_pull_records(mdx_system *sys, double *out, long long max_records);
/* ---- Spatial domain decomposition over several devices: one process and one handle per GPU.
 * No counterpart in the reference, which runs every gmx mdrun with a single rank
 * (/root/reference/Snakefile:232-233: -ntmpi 1); serves the scaled synthetic water box.
 * The handle is created over `capacity` atom slots filled with whole molecules of one kind (the per-slot
 * pattern of mass, charge, LJ, exclusions and settles given through the setters above, no bonded terms).
 * mdx_dd_load installs a partition: slots [0, n_home) are atoms this domain owns, [n_home, n_total) copies of
 * atoms of other domains within the pair-list radius (halo: inverse mass zero, not spread onto the charge
 * grid), the rest unused. x4 / v4 are device arrays of n_total float4 (xyz + anything). n_settle = home molecules,
 * ndf = degrees of freedom of the home atoms, density = atoms per nm^3 of the whole system.
 * A step then runs as: mdx_dd_pack (positions of listed home atoms into a send buffer) -> exchange by the
 * caller -> mdx_dd_unpack (into halo slots) -> mdx_dd_step_begin -> mdx_dd_wait_spread -> caller sums the
 * charge grid (mdx_dd_grid) over all domains -> mdx_dd_step_end. Pack/unpack/step calls enqueue on the handle's
 * main stream; mdx_dd_stream_sync waits for it. */
int mdx_dd_load(mdx_system *sys, int n_home, int n_total, const void *x4_dev, const void *v4_dev, int n_settle,
                double ndf, double density);
int mdx_dd_pack(mdx_system *sys, int n, const int *idx_dev, void *out4_dev);
int mdx_dd_unpack(mdx_system *sys, int first_slot, int n, const void *in4_dev);
int mdx_dd_step_begin(mdx_system *sys);
int mdx_dd_wait_spread(mdx_system *sys);
int mdx_dd_step_end(mdx_system *sys);
int mdx_dd_grid(mdx_system *sys, void **grid_dev, long long *nelem);
int mdx_dd_stream_sync(mdx_system *sys);
int mdx_dd_state_pointers(mdx_system *sys, void **x4_dev, void **v4_dev);
/* The handle's main stream and PME stream (hipStream_t) for callers that order their collectives on the device instead
 * of on the host: the sum of the charge grid enqueued on (or made to wait for and be waited for by) the PME stream
 * between mdx_dd_step_begin and mdx_dd_step_end needs no mdx_dd_wait_spread and no host synchronisation; exchanges
 * and scalar sums go on the main stream the same way. No counterpart in the reference (single-rank mdrun). */
int mdx_dd_streams(mdx_system *sys, void **main_stream, void **pme_stream);
/* The mesh transform shared among the domains instead of solved by every one of them. No counterpart in the reference
 * (single-rank mdrun); what a multi-rank gmx mdrun does with its PME decomposition. After mdx_dd_mesh_slabs(sys, world, rank)
 * a step runs as: mdx_dd_step_begin (now through the x forward pass of this domain's own charge grid) -> the caller
 * reduce-scatters the half-complex grid of mdx_dd_cgrid (`world` equal chunks of `planes_per_domain` k_x planes of
 * `plane_elems` float2 each, chunk r summed onto domain r) -> mdx_dd_step_mid (y and z transforms, influence function and
 * back on this domain's planes) -> the caller all-gathers the chunks -> mdx_dd_step_end (x inverse pass, gather, update).
 * mdx_dd_wait_spread blocks until the stage begun last has finished; callers that order their collectives on the PME stream
 * (mdx_dd_streams) need no wait. Every domain accumulates the reciprocal-space energy and virial of its own planes
 * (mesh_energy = 1 on all of them): they add up like the other terms. MDX_ERR_UNSUPPORTED for a PME grid whose planes
 * the fused plane kernel cannot take (the caller keeps mdx_dd_grid's scheme); world = 0 returns to that scheme. */
int mdx_dd_mesh_slabs(mdx_system *sys, int world, int rank);
int mdx_dd_cgrid(mdx_system *sys, void **cgrid_dev, int *planes_per_domain, long long *plane_elems);
int mdx_dd_step_mid(mdx_system *sys);
/* Quantities the domains have to agree on. Energy steps: every domain's energy terms, virial and pressure are
 * partial sums over the atoms it owns (a pair with a halo atom counts half on each side, the Ewald self term covers
 * the home atoms) and add up over the domains; the reciprocal-space energy and virial are the same on every domain,
 * so mdx_dd_set_roles(sys, mesh_energy, own_tcouple) lets one domain report them (mesh_energy = 0 elsewhere).
 * own_tcouple = 0: mdx_dd_step_end computes the half-step kinetic energy of the home atoms on thermostat steps but
 * leaves the V-rescale factor alone; the caller sums the MDX_MAX_TC doubles at `ekinh_dev` (mdx_dd_scalars) over the
 * domains and calls mdx_dd_tcouple, so that all domains (same seed, ndf of mdx_dd_load = that of the whole system)
 * scale by one factor. Centre-of-mass removal: mdx_dd_com(sys, 0) sums m v and m of the home atoms into the 4 doubles
 * at `com_dev`, the caller sums them over the domains, mdx_dd_com(sys, 1) subtracts the common velocity.
 * mdx_dd_kinetic puts the current half-step kinetic energy of the home atoms into the scalars (with own_tcouple = 0 the
 * caller draws the factor of the first steps from it: mdx_dd_kinetic, sum, mdx_dd_tcouple(sys, 1)). */
int mdx_dd_set_roles(mdx_system *sys, int mesh_energy, int own_tcouple);
int mdx_dd_scalars(mdx_system *sys, void **ekinh_dev, void **com_dev);
int mdx_dd_tcouple(mdx_system *sys, int before_first_step);
int mdx_dd_kinetic(mdx_system *sys);
int mdx_dd_com(mdx_system *sys, int phase);
/* Pressure coupling of a decomposed run (`pcoupl`/`tau_p`/`ref_p`/`compressibility` of
 * /root/reference/mdp/production.mdp:36-40, Parrinello-Rahman there, or C-rescale): a coupling step leaves the domain's
 * share of the pressure tensor at `pres_dev` (9 doubles); the caller sums it over the domains and calls mdx_dd_pcouple,
 * which scales box, atoms and velocities by the factor every domain draws alike and returns the new box. */
int mdx_dd_pressure(mdx_system *sys, void **pres_dev);
int mdx_dd_pcouple(mdx_system *sys, double *box9);
/* Time the nonbonded kernel with HIP events on the handle's stream (bench only): on = n > 0 samples every n-th
 * force-only launch (a stride such as 7 does not lock onto the 10- and 20-step cadences of energy and search steps). */
int mdx_enable_kernel_timing(mdx_system *sys, int on);
int mdx_get_kernel_timing(mdx_system *sys, double *nb_ms_total, long long *nb_launches);
#ifdef __cplusplus
}
#endif
#endif
