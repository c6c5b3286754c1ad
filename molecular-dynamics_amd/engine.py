"""ctypes binding of libmdx.so (include/mdx.h) and the `Engine` convenience wrapper.

The HIP library is the only compute path: if it cannot be loaded, or no HIP device is
present, construction raises — there is no CPU fallback in the product.
"""
from __future__ import annotations

import ctypes
import os
import numpy as np

from . import units as U
from .system import MDSystem

E_NAMES = ["LJ", "COUL_SR", "COUL_RECIP", "COUL_EXCL", "BOND", "ANGLE", "DIH", "LJ14", "COUL14", "POSRES",
           "PULL", "DISPCORR", "KIN", "POT", "TEMP", "PRES", "VOLUME", "CONSERVED"]
E_N = len(E_NAMES)
E = {n: i for i, n in enumerate(E_NAMES)}
F_NONBONDED, F_PME, F_EXCL, F_BONDED, F_POSRES, F_PULL, F_ALL = 1, 2, 4, 8, 16, 32, 63
# layout of the run state (MDX_RS_* in include/mdx.h)
RS = dict(LAMBDA=0, THERM_INTEGRAL=8, EKINH=16, EKINH_OLD=24, PR_BOXV=32, MU=41, BARO_INTEGRAL=50, PULL_INIT=51, PULL_WORK=52,
          PULL_VALUE=53, PULL_HAVE_INIT=54, COM_V=55)
RS_N = 64

_LIB = None
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmdx.so")

SYMBOLS = [
    "mdx_version", "mdx_last_error", "mdx_create", "mdx_destroy", "mdx_set_atoms", "mdx_set_exclusions",
    "mdx_set_bonds", "mdx_set_angles", "mdx_set_dihedrals", "mdx_set_pairs14", "mdx_set_posres", "mdx_get_posres_ref", "mdx_set_settles",
    "mdx_set_constraints", "mdx_set_nonbonded", "mdx_set_pairlist_prune", "mdx_set_pme", "mdx_set_integrator", "mdx_set_tcoupl",
    "mdx_set_pcoupl", "mdx_set_dispcorr", "mdx_set_pull", "mdx_set_state", "mdx_set_state_device", "mdx_get_state",
    "mdx_get_step", "mdx_get_run_state", "mdx_set_run_state", "mdx_compute_forces", "mdx_get_forces", "mdx_get_forces_noalloc", "mdx_get_energies", "mdx_get_virial", "mdx_step",
    "mdx_synchronize", "mdx_get_list_stats", "mdx_pop_energy_frames", "mdx_pop_pull_records",
    "mdx_enable_kernel_timing", "mdx_get_kernel_timing", "mdx_debug_time_nonbonded", "mdx_debug_set", "mdx_debug_wave_times", "mdx_debug_stamps",
    "mdx_dd_load", "mdx_dd_pack", "mdx_dd_unpack", "mdx_dd_step_begin", "mdx_dd_wait_spread", "mdx_dd_step_end", "mdx_dd_grid",
    "mdx_dd_mesh_slabs", "mdx_dd_cgrid", "mdx_dd_step_mid",
    "mdx_dd_stream_sync", "mdx_dd_state_pointers", "mdx_dd_streams", "mdx_dd_set_roles", "mdx_dd_scalars", "mdx_dd_tcouple", "mdx_dd_kinetic", "mdx_dd_com", "mdx_dd_pressure", "mdx_dd_pcouple",
    "mdx_debug_time_stage", "mdx_set_step", "mdx_generate_velocities", "mdx_constrain_start", "mdx_minimize",
    "mdx_frame_ring", "mdx_capture_frame", "mdx_frame_wait", "mdx_capture_frame_xtc", "mdx_frame_wait_xtc",
]
PROGRESS_FN = ctypes.CFUNCTYPE(None, ctypes.c_longlong, ctypes.c_double, ctypes.c_double, ctypes.c_double)


class MdxError(RuntimeError):
    pass


def _share_torch_hip_runtime():
    """One HIP runtime per process. A PyTorch-ROCm wheel ships its own libamdhip64.so; libmdx.so names the system one.
    Both have the same SONAME, so whichever is loaded first serves both — unless libmdx.so comes first, in which case
    torch still loads its bundled copy by path, the process holds two runtimes and torch then reports "No HIP GPUs are
    available". Loading torch's copy (when a torch installation has one) ahead of libmdx.so makes the order irrelevant.
    Nothing is imported from torch; without torch the system runtime is used."""
    try:
        with open("/proc/self/maps") as f:
            if "libamdhip64" in f.read():
                return                      # a runtime is already in the process: libmdx.so binds to it
    except OSError:
        pass
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except (ImportError, OSError, ValueError):
        pass


def load_library(path: str | None = None):
    """Load libmdx.so and declare the C signatures. Raises if the library is missing."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or os.environ.get("MDX_LIB") or LIB_PATH
    if not os.path.exists(p):
        raise MdxError(f"{p} not found: build it with `make -C molecular-dynamics_amd/csrc` "
                       "(the engine has no CPU fallback)")
    _share_torch_hip_runtime()
    L = ctypes.CDLL(p)
    c_int, c_f, c_d, vp = ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_void_p
    fp, ip, dp = ctypes.POINTER(c_f), ctypes.POINTER(c_int), ctypes.POINTER(c_d)
    llp = ctypes.POINTER(ctypes.c_longlong)
    L.mdx_version.restype = c_int
    L.mdx_last_error.restype = ctypes.c_char_p; L.mdx_last_error.argtypes = [vp]
    L.mdx_create.restype = vp; L.mdx_create.argtypes = [c_int, c_int]
    L.mdx_destroy.restype = None; L.mdx_destroy.argtypes = [vp]
    L.mdx_set_atoms.argtypes = [vp, fp, fp, fp, fp, ip, ip]
    L.mdx_set_exclusions.argtypes = [vp, ip, ip]
    L.mdx_set_bonds.argtypes = [vp, c_int, ip, fp, fp]
    L.mdx_set_angles.argtypes = [vp, c_int, ip, fp, fp]
    L.mdx_set_dihedrals.argtypes = [vp, c_int, ip, fp, fp, ip]
    L.mdx_set_pairs14.argtypes = [vp, c_int, ip, fp, fp, fp]
    L.mdx_set_posres.argtypes = [vp, c_int, ip, fp, fp, c_int]
    L.mdx_get_posres_ref.argtypes = [vp, fp]
    L.mdx_set_settles.argtypes = [vp, c_int, ip, c_f, c_f]
    L.mdx_set_constraints.argtypes = [vp, c_int, ip, fp, c_int, c_int]
    L.mdx_set_nonbonded.argtypes = [vp, c_f, c_f, c_f, c_int]
    L.mdx_set_pairlist_prune.argtypes = [vp, c_f, c_int]
    L.mdx_set_pme.argtypes = [vp, c_int, c_int, c_int, c_int]
    L.mdx_set_integrator.argtypes = [vp, c_f, c_int, c_int]
    L.mdx_set_tcoupl.argtypes = [vp, c_int, dp, dp, dp, c_int, ctypes.c_ulonglong]
    L.mdx_set_pcoupl.argtypes = [vp, c_int, c_d, c_d, c_d, c_int]
    L.mdx_set_dispcorr.argtypes = [vp, c_int, c_d]
    L.mdx_set_pull.argtypes = [vp, c_int, ip, c_int, ip, c_int, c_int, c_int, c_d, c_d, c_d, c_int, ip, c_int]
    L.mdx_set_state.argtypes = [vp, fp, fp, fp]
    L.mdx_set_state_device.argtypes = [vp, vp, vp, fp]
    L.mdx_get_state.argtypes = [vp, fp, fp, fp]
    L.mdx_get_step.restype = ctypes.c_longlong; L.mdx_get_step.argtypes = [vp]
    L.mdx_get_run_state.argtypes = [vp, dp]
    L.mdx_set_run_state.argtypes = [vp, dp]
    L.mdx_compute_forces.argtypes = [vp, ctypes.c_uint]
    L.mdx_get_forces.argtypes = [vp, fp]
    L.mdx_get_forces_noalloc.argtypes = [vp, fp]
    L.mdx_get_energies.argtypes = [vp, dp]
    L.mdx_get_virial.argtypes = [vp, dp, dp]
    L.mdx_step.argtypes = [vp, ctypes.c_longlong]
    L.mdx_synchronize.argtypes = [vp]
    L.mdx_get_list_stats.argtypes = [vp, llp]
    L.mdx_pop_energy_frames.restype = ctypes.c_longlong; L.mdx_pop_energy_frames.argtypes = [vp, dp, ctypes.c_longlong]
    L.mdx_pop_pull_records.restype = ctypes.c_longlong; L.mdx_pop_pull_records.argtypes = [vp, dp, ctypes.c_longlong]
    L.mdx_enable_kernel_timing.argtypes = [vp, c_int]
    L.mdx_get_kernel_timing.argtypes = [vp, dp, llp]
    L.mdx_debug_time_nonbonded.argtypes = [vp, c_int, c_int, dp]
    L.mdx_debug_set.argtypes = [vp, c_int, c_int]
    L.mdx_debug_wave_times.argtypes = [vp, ctypes.POINTER(ctypes.c_longlong), c_int, ctypes.POINTER(c_int)]
    L.mdx_dd_load.argtypes = [vp, c_int, c_int, vp, vp, c_int, c_d, c_d]
    L.mdx_dd_pack.argtypes = [vp, c_int, vp, vp]
    L.mdx_dd_unpack.argtypes = [vp, c_int, c_int, vp]
    L.mdx_dd_step_begin.argtypes = [vp]; L.mdx_dd_wait_spread.argtypes = [vp]; L.mdx_dd_step_end.argtypes = [vp]
    L.mdx_dd_grid.argtypes = [vp, ctypes.POINTER(vp), llp]
    L.mdx_dd_mesh_slabs.argtypes = [vp, c_int, c_int]; L.mdx_dd_step_mid.argtypes = [vp]
    L.mdx_dd_cgrid.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(c_int), llp]
    L.mdx_dd_stream_sync.argtypes = [vp]
    L.mdx_dd_state_pointers.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp)]
    L.mdx_dd_streams.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp)]
    L.mdx_dd_set_roles.argtypes = [vp, c_int, c_int]
    L.mdx_dd_scalars.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp)]
    L.mdx_dd_tcouple.argtypes = [vp, c_int]; L.mdx_dd_kinetic.argtypes = [vp]; L.mdx_dd_com.argtypes = [vp, c_int]
    L.mdx_dd_pressure.argtypes = [vp, ctypes.POINTER(ctypes.c_void_p)]; L.mdx_dd_pcouple.argtypes = [vp, ctypes.POINTER(c_d)]
    L.mdx_debug_stamps.argtypes = [vp, c_int, ctypes.POINTER(ctypes.c_longlong)]
    L.mdx_debug_time_stage.argtypes = [vp, c_int, c_int, dp]
    L.mdx_set_step.argtypes = [vp, ctypes.c_longlong]
    L.mdx_generate_velocities.argtypes = [vp, c_d, ctypes.c_ulonglong]
    L.mdx_constrain_start.argtypes = [vp]
    L.mdx_minimize.argtypes = [vp, ctypes.c_longlong, c_d, c_d, dp, PROGRESS_FN]
    L.mdx_frame_ring.argtypes = [vp, c_int]
    L.mdx_capture_frame.argtypes = [vp, c_int, c_int]
    L.mdx_frame_wait.argtypes = [vp, c_int, ctypes.POINTER(fp), ctypes.POINTER(fp), dp, llp]
    L.mdx_capture_frame_xtc.argtypes = [vp, c_int, c_int, c_f]
    L.mdx_frame_wait_xtc.argtypes = [vp, c_int, ctypes.POINTER(ip), ctypes.POINTER(fp), dp, llp, ip]
    for name in SYMBOLS:
        getattr(L, name)        # every declared entry point must be exported
    if path is None:
        _LIB = L
    return L


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def c6_average(sigma, eps):
    """<C6> over all distinct atom pairs for Lorentz-Berthelot LJ (`DispCorr`)."""
    key = np.stack([sigma, eps], 1)
    uniq, cnt = np.unique(key, axis=0, return_counts=True)
    s = 0.5 * (uniq[:, None, 0] + uniq[None, :, 0]); e = np.sqrt(uniq[:, None, 1] * uniq[None, :, 1])
    c6 = 4 * e * s ** 6
    n = len(sigma)
    tot = 0.5 * ((cnt[:, None] * cnt[None, :] * c6).sum() - (cnt * np.diag(c6)).sum())
    return float(tot / (0.5 * n * (n - 1)))


def verlet_rlist(rc, nstlist, dt, temperature=300.0, min_mass=1.008, water_constrained=True):
    """Pair-list radius without knowledge of the system: rc plus a buffer that covers the relative
    displacement of two light particles over the list lifetime (6 sigma of the ballistic Gaussian
    estimate, floored at 0.05 nm and capped at 0.25 nm). `verlet_buffer_rlist` is the estimate the
    engine uses when it has the topology."""
    m_eff = 4.0 * min_mass if water_constrained else min_mass   # constrained H moves with its heavy atom
    sig1 = np.sqrt(U.BOLTZ * temperature / m_eff) * (nstlist - 1) * dt
    buf = 6.0 * np.sqrt(2.0) * sig1 * 0.5
    return float(rc + min(0.25, max(0.05, buf)))


def _displacement_variance(sysm, kT_t2):
    """Per atom <dx^2> per dimension after a list lifetime t (kT_t2 = kT t^2). Free atoms: kT t^2 / m.
    An atom held by constraints moves with its rigid group of mass M (kT t^2 / M, three directions)
    and swings about the group's centre on an arm a = b (M - m) / M of the constraint length b, which
    displaces it in two directions only and by no more than the arm allows: with x = kT t^2 (M - m) /
    (m M a^2) the swing variance is a^2 x / (1 + x/3 + 2 x^2/45), entering with weight 2/3."""
    m = np.maximum(np.asarray(sysm.mass, dtype=np.float64), 1e-3)
    M = m.copy()
    arm_len = np.zeros_like(m)
    if len(sysm.constraints):
        c = np.asarray(sysm.constraints)
        b = np.asarray(sysm.constraint_len, dtype=np.float64)
        np.add.at(M, c[:, 0], m[c[:, 1]])
        np.add.at(M, c[:, 1], m[c[:, 0]])
        np.maximum.at(arm_len, c[:, 0], b)
        np.maximum.at(arm_len, c[:, 1], b)
    if len(sysm.settles):
        w3 = np.asarray(sysm.settles)
        tot = m[w3].sum(1)
        for k in range(3):
            M[w3[:, k]] = tot
        arm_len[w3.ravel()] = float(sysm.settle_doh)
    frac = (M - m) / M
    var = kT_t2 / M
    con = (M > m) & (arm_len > 0)
    arm2 = (arm_len * frac) ** 2
    x = np.where(con, kT_t2 * frac / m / np.maximum(arm2, 1e-30), 0.0)
    var = var + np.where(con, (2.0 / 3.0) * arm2 * x / (1.0 + x / 3.0 + 2.0 * x * x / 45.0), 0.0)
    return var


def _drift_model(sysm, rc, beta, nstlist, dt, temperature, volume=None):
    """Everything of the drift estimate that does not depend on the list radius; returns f(rlist)."""
    from math import erfc as _erfc
    from scipy.special import erfc as _verfc
    n = sysm.natoms
    if volume is None:
        b = np.asarray(sysm.box, dtype=np.float64)
        volume = abs(np.linalg.det(b.reshape(3, 3))) if b.size == 9 else float(np.prod(b[:3]))
    t = max(nstlist - 1, 0) * dt
    w = _displacement_variance(sysm, U.BOLTZ * temperature * t * t)
    key = np.stack([np.round(np.asarray(sysm.charge, dtype=np.float64), 2), np.round(np.asarray(sysm.sigma, dtype=np.float64), 3),
                    np.round(np.asarray(sysm.eps, dtype=np.float64), 3), np.round(w, 5)], 1)
    uniq, cnt = np.unique(key, axis=0, return_counts=True)
    q, sg, ep, wt = uniq[:, 0], uniq[:, 1], uniq[:, 2], uniq[:, 3]
    sij = 0.5 * (sg[:, None] + sg[None, :]); eij = np.sqrt(ep[:, None] * ep[None, :])
    A = 4.0 * eij * sij ** 12; B = 4.0 * eij * sij ** 6
    qq = U.ONE_4PI_EPS0 * q[:, None] * q[None, :]
    r = float(rc)
    d1 = -12.0 * A / r ** 13 + 6.0 * B / r ** 7
    d2 = 156.0 * A / r ** 14 - 42.0 * B / r ** 8
    d3 = -2184.0 * A / r ** 15 + 336.0 * B / r ** 9
    if beta > 0.0:
        ex = 2.0 * beta / np.sqrt(np.pi) * np.exp(-beta * beta * r * r); ec = _erfc(beta * r)
        g1 = -ec / r ** 2 - ex / r
        g2 = 2.0 * ec / r ** 3 + ex * (2.0 / r ** 2 + 2.0 * beta ** 2)
        g3 = -6.0 * ec / r ** 4 - ex * (6.0 / r ** 3 + 4.0 * beta ** 2 / r + 4.0 * beta ** 4 * r)
    else:                       # plain shifted Coulomb
        g1, g2, g3 = -1.0 / r ** 2, 2.0 / r ** 3, -6.0 / r ** 4
    d1 = d1 + qq * g1; d2 = d2 + qq * g2; d3 = d3 + qq * g3
    s2 = np.maximum(wt[:, None] + wt[None, :], 1e-30)
    s = np.sqrt(s2)
    weight = 0.5 * cnt[:, None] * (cnt[None, :] / volume) * 4.0 * np.pi / n / (max(nstlist, 1) * dt)

    def drift(rlist):
        rb = max(float(rlist) - r, 0.0)
        x = rb / s
        G = np.exp(-0.5 * x * x) / np.sqrt(2.0 * np.pi)
        E = 0.5 * _verfc(x / np.sqrt(2.0))
        c1 = rb * s * G - (rb * rb + s2) * E
        c2 = s * (rb * rb + 2.0 * s2) * G - rb * (rb * rb + 3.0 * s2) * E
        c3 = rb * s * (rb * rb + 5.0 * s2) * G - (rb ** 4 + 6.0 * rb * rb * s2 + 3.0 * s2 * s2) * E
        per_pair = np.abs(d1 * c1) / 2.0 + np.abs(d2 * c2) / 6.0 + np.abs(d3 * c3) / 24.0
        return float((weight * float(rlist) ** 2 * per_pair).sum())
    return drift


def verlet_buffer_drift(sysm, rc, rlist, beta, nstlist, dt, temperature, volume=None):
    """Estimated energy drift (kJ/mol/ps per atom) caused by pairs that start beyond `rlist` and
    come within `rc` during a list lifetime: the Gaussian-displacement estimate behind
    `verlet-buffer-tolerance` (/root/reference/mdp/production.mdp:23, npt_complex.mdp:32; GROMACS
    manual, "Energy drift and pair-list buffering"). Per pair type the shifted potential is expanded
    to third order around rc and averaged over the displacement distribution after nstlist - 1
    steps; absolute values of the three terms are summed, and the error at the end of the lifetime
    is charged to the whole lifetime, so the estimate errs on the large side. (GROMACS further
    lowers the radius for the pairs a cluster list holds beyond it; that bonus is not taken here.)"""
    return _drift_model(sysm, rc, beta, nstlist, dt, temperature, volume)(rlist)


def verlet_buffer_rlist(sysm, rc, beta, nstlist, dt, temperature=300.0, tolerance=0.005, volume=None):
    """Smallest pair-list radius (1e-3 nm steps, at most rc + 0.5) whose estimated drift stays within
    `verlet-buffer-tolerance` (kJ/mol/ps per atom, GROMACS' default 0.005)."""
    drift = _drift_model(sysm, rc, beta, nstlist, dt, temperature, volume)
    lo, hi = 0.0, 0.5
    if drift(rc + hi) > tolerance:
        return float(rc + hi)
    for _ in range(24):
        mid = 0.5 * (lo + hi)
        if drift(rc + mid) > tolerance:
            lo = mid
        else:
            hi = mid
    return float(round(rc + np.ceil(hi * 1000.0) / 1000.0, 4))


class Engine:
    """One MD system resident on one MI355X, driven through the C ABI."""

    def __init__(self, sysm: MDSystem, rc=1.0, rlist=None, nstlist=10, ewald_rtol=1e-5, pme=True,
                 fourierspacing=0.16, grid=None, dt=0.002, nstcalcenergy=100, nstcomm=100, device=0,
                 tc_groups=None, ref_t=None, tau_t=None, nsttcouple=10, seed=1,
                 pcoupl=0, tau_p=2.0, ref_p=1.0, compressibility=4.5e-5, nstpcouple=10,
                 dispcorr=0, lincs_order=4, lincs_iter=1, refcoord_scaling=0, use_posres=False, beta=None,
                 verlet_buffer_tolerance=0.005, nstprune=0, rlist_inner=None):
        self.lib = load_library()
        self.sys = sysm
        n = sysm.natoms
        self.n = n
        self.h = self.lib.mdx_create(int(device), int(n))
        if not self.h:
            raise MdxError("mdx_create failed: " + self.lib.mdx_last_error(None).decode())
        self.rc = float(rc)
        self.beta = float(beta) if beta is not None else U.ewald_beta(rc, ewald_rtol)
        self.dt = float(dt)
        if rlist is None:
            # `verlet-buffer-tolerance`: the list radius follows from the drift estimate, as in grompp/mdrun
            t_ref = max([float(x) for x in ref_t]) if ref_t is not None and len(ref_t) else 300.0
            rlist = verlet_buffer_rlist(sysm, rc, self.beta if pme else 0.0, nstlist, dt, max(t_ref, 1.0),
                                        float(verlet_buffer_tolerance))
        self.rlist = float(rlist)
        # dual pair list: the outer list lives `nstlist` steps, the pruned inner list serves nstprune + 1 steps at most
        self.nstprune = int(nstprune) if nstprune and int(nstprune) < int(nstlist) else 0
        self.rlist_inner = None
        if self.nstprune > 0:
            if rlist_inner is None:
                t_ref = max([float(x) for x in ref_t]) if ref_t is not None and len(ref_t) else 300.0
                rlist_inner = verlet_buffer_rlist(sysm, rc, self.beta if pme else 0.0, self.nstprune + 1, dt, max(t_ref, 1.0),
                                                  float(verlet_buffer_tolerance))
            self.rlist_inner = float(min(max(float(rlist_inner), self.rc), self.rlist))
        # T-coupling groups
        tcg = np.zeros(n, dtype=np.int32)
        ndf = []
        nc_atom = sysm.n_constraints_per_atom()
        if tc_groups:
            assigned = np.zeros(n, bool)
            for gi, name in enumerate(tc_groups):
                idx = sysm.groups[name]
                tcg[idx] = gi
                assigned[idx] = True
            if not assigned.all():
                raise MdxError("tc-grps do not cover every atom")
            total_ndf = 3.0 * n - nc_atom.sum()
            for gi, name in enumerate(tc_groups):
                idx = sysm.groups[name]
                nd = 3.0 * len(idx) - nc_atom[idx].sum()
                # the 3 COM degrees of freedom are removed proportionally
                ndf.append(nd - 3.0 * nd / total_ndf if nstcomm > 0 else nd)
        self.ntc = len(tc_groups) if tc_groups else 0
        self.ndf = [float(d) for d in ndf]          # degrees of freedom per T-coupling group, as handed to the library
        self.ndf_total = float(3.0 * n - nc_atom.sum() - (3.0 if nstcomm > 0 else 0.0))
        self._call("mdx_set_nonbonded", ctypes.c_float(self.rc), ctypes.c_float(self.rlist), ctypes.c_float(self.beta if pme else 0.0),
                   int(nstlist))
        if self.nstprune > 0:
            self._call("mdx_set_pairlist_prune", ctypes.c_float(self.rlist_inner), self.nstprune)
        m, q, sg, ep = _f32(sysm.mass), _f32(sysm.charge), _f32(sysm.sigma), _f32(sysm.eps)
        mol = _i32(sysm.molid)
        self._call("mdx_set_atoms", _p(m, ctypes.c_float), _p(q, ctypes.c_float), _p(sg, ctypes.c_float), _p(ep, ctypes.c_float),
                   _p(mol, ctypes.c_int), _p(tcg, ctypes.c_int))
        es, el = _i32(sysm.excl_start), _i32(sysm.excl_list if len(sysm.excl_list) else np.zeros(1, np.int32))
        self._call("mdx_set_exclusions", _p(es, ctypes.c_int), _p(el, ctypes.c_int))
        if len(sysm.bonds):
            a, b0, kb = _i32(sysm.bonds), _f32(sysm.bond_b0), _f32(sysm.bond_kb)
            self._call("mdx_set_bonds", len(a), _p(a, ctypes.c_int), _p(b0, ctypes.c_float), _p(kb, ctypes.c_float))
        if len(sysm.angles):
            a, t0, k = _i32(sysm.angles), _f32(sysm.angle_th0), _f32(sysm.angle_k)
            self._call("mdx_set_angles", len(a), _p(a, ctypes.c_int), _p(t0, ctypes.c_float), _p(k, ctypes.c_float))
        if len(sysm.dihedrals):
            a, p0, k, mu = _i32(sysm.dihedrals), _f32(sysm.dih_phi0), _f32(sysm.dih_k), _i32(sysm.dih_mult)
            self._call("mdx_set_dihedrals", len(a), _p(a, ctypes.c_int), _p(p0, ctypes.c_float), _p(k, ctypes.c_float), _p(mu, ctypes.c_int))
        if len(sysm.pairs14):
            a, c6, c12, qq = _i32(sysm.pairs14), _f32(sysm.pair_c6), _f32(sysm.pair_c12), _f32(sysm.pair_qq)
            self._call("mdx_set_pairs14", len(a), _p(a, ctypes.c_int), _p(c6, ctypes.c_float), _p(c12, ctypes.c_float), _p(qq, ctypes.c_float))
        if use_posres and len(sysm.posres_idx):
            a, r, k = _i32(sysm.posres_idx), _f32(sysm.posres_ref), _f32(sysm.posres_k)
            self._call("mdx_set_posres", len(a), _p(a, ctypes.c_int), _p(r, ctypes.c_float), _p(k, ctypes.c_float), int(refcoord_scaling))
        if len(sysm.settles):
            a = _i32(sysm.settles)
            self._call("mdx_set_settles", len(a), _p(a, ctypes.c_int), ctypes.c_float(sysm.settle_doh), ctypes.c_float(sysm.settle_dhh))
        if len(sysm.constraints):
            a, l = _i32(sysm.constraints), _f32(sysm.constraint_len)
            self._call("mdx_set_constraints", len(a), _p(a, ctypes.c_int), _p(l, ctypes.c_float), int(lincs_order), int(lincs_iter))
        if pme:
            if grid is None:
                grid = U.pme_grid(sysm.box, fourierspacing)
            self.grid = tuple(int(g) for g in grid)
            self._call("mdx_set_pme", self.grid[0], self.grid[1], self.grid[2], 4)
        else:
            self.grid = None
        self._call("mdx_set_integrator", ctypes.c_float(self.dt), int(nstcalcenergy), int(nstcomm))
        if self.ntc:
            rt = np.ascontiguousarray(np.broadcast_to(np.asarray(ref_t, dtype=np.float64), (self.ntc,)))
            tt = np.ascontiguousarray(np.broadcast_to(np.asarray(tau_t, dtype=np.float64), (self.ntc,)))
            nd = np.ascontiguousarray(ndf, dtype=np.float64)
            self._call("mdx_set_tcoupl", self.ntc, _p(rt, ctypes.c_double), _p(tt, ctypes.c_double), _p(nd, ctypes.c_double),
                       int(nsttcouple), ctypes.c_ulonglong(int(seed)))
        else:
            nd = np.array([self.ndf_total], dtype=np.float64)
            self._call("mdx_set_tcoupl", 0, None, None, _p(nd, ctypes.c_double), 1, ctypes.c_ulonglong(int(seed)))
        self._call("mdx_set_pcoupl", int(pcoupl), float(tau_p), float(ref_p), float(compressibility), int(nstpcouple))
        if dispcorr:
            self._call("mdx_set_dispcorr", int(dispcorr), c6_average(sysm.sigma, sysm.eps))
        self.set_state(sysm.x, sysm.v, sysm.box)

    # ------------------------------------------------------------------ plumbing
    def _call(self, name, *args):
        rc = getattr(self.lib, name)(self.h, *args)
        if rc != 0:
            raise MdxError(f"{name} failed ({rc}): " + self.lib.mdx_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.mdx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ state
    def set_state(self, x, v, box):
        x = _f32(x); box = _f32(np.asarray(box).reshape(9))
        vv = _f32(v) if v is not None else None
        self._call("mdx_set_state", _p(x, ctypes.c_float), _p(vv, ctypes.c_float) if vv is not None else None, _p(box, ctypes.c_float))

    def get_state(self):
        x = np.zeros((self.n, 3), np.float32); v = np.zeros((self.n, 3), np.float32); box = np.zeros(9, np.float32)
        self._call("mdx_get_state", _p(x, ctypes.c_float), _p(v, ctypes.c_float), _p(box, ctypes.c_float))
        return x, v, box.reshape(3, 3)

    def posres_ref(self):
        """Restraint reference positions as pressure coupling has moved them (`refcoord_scaling`)."""
        r = np.zeros((len(self.sys.posres_idx), 3), np.float32)
        self._call("mdx_get_posres_ref", _p(r, ctypes.c_float))
        return r

    def set_pull(self, group1, group2, type="constant-force", k=500.0, rate=0.0, init=0.0, start=True,
                 dim=(1, 1, 1), nstout=50, pbcatom1=-1, pbcatom2=-1):
        g1, g2 = _i32(group1), _i32(group2)
        d = _i32(dim)
        t = {"constant-force": 0, "umbrella": 1}[type]
        self._call("mdx_set_pull", len(g1), _p(g1, ctypes.c_int), len(g2), _p(g2, ctypes.c_int), int(pbcatom1), int(pbcatom2),
                   t, float(k), float(rate), float(init), 1 if start else 0, _p(d, ctypes.c_int), int(nstout))

    # ------------------------------------------------------------------ compute
    def compute(self, terms=F_ALL & ~F_PULL):
        self._call("mdx_compute_forces", ctypes.c_uint(int(terms)))
        f = np.zeros((self.n, 3), np.float32)
        self._call("mdx_get_forces", _p(f, ctypes.c_float))
        return f, self.energies()

    def forces(self, noalloc=False):
        """Forces of the last compute(); noalloc=True goes through the entry point that stages in fixed scratch."""
        f = np.zeros((self.n, 3), np.float32)
        self._call("mdx_get_forces_noalloc" if noalloc else "mdx_get_forces", _p(f, ctypes.c_float))
        return f

    def energies(self):
        e = np.zeros(E_N, np.float64)
        self._call("mdx_get_energies", _p(e, ctypes.c_double))
        return {n: float(e[i]) for i, n in enumerate(E_NAMES)}

    def virial(self):
        v = np.zeros(9); p = np.zeros(9)
        self._call("mdx_get_virial", _p(v, ctypes.c_double), _p(p, ctypes.c_double))
        return v.reshape(3, 3), p.reshape(3, 3)

    def step(self, nsteps):
        self._call("mdx_step", ctypes.c_longlong(int(nsteps)))

    def synchronize(self):
        self._call("mdx_synchronize")

    def get_step(self):
        return int(self.lib.mdx_get_step(self.h))

    def set_step(self, step):
        self._call("mdx_set_step", ctypes.c_longlong(int(step)))

    def get_run_state(self):
        """The MDX_RS_N coupling / pull scalars a checkpoint carries beside x, v and box (layout in include/mdx.h)."""
        rs = np.zeros(RS_N, np.float64)
        self._call("mdx_get_run_state", _p(rs, ctypes.c_double))
        return rs

    def set_run_state(self, rs):
        rs = np.ascontiguousarray(rs, dtype=np.float64)
        if rs.shape != (RS_N,):
            raise MdxError(f"run state must hold {RS_N} values")
        self._call("mdx_set_run_state", _p(rs, ctypes.c_double))

    def generate_velocities(self, temperature, seed):
        self._call("mdx_generate_velocities", float(temperature), ctypes.c_ulonglong(int(seed) & (2 ** 64 - 1)))

    def constrain_start(self):
        self._call("mdx_constrain_start")

    def minimize(self, nsteps, emtol=1000.0, emstep=0.01, progress=None):
        """Steepest descent; returns dict(steps, status, epot, fmax, epot0, fmax0)."""
        res = np.zeros(6, np.float64)
        cb = PROGRESS_FN(progress) if progress is not None else ctypes.cast(None, PROGRESS_FN)
        self._call("mdx_minimize", ctypes.c_longlong(int(nsteps)), float(emtol), float(emstep), _p(res, ctypes.c_double), cb)
        return dict(steps=int(res[0]), status=int(res[1]), epot=float(res[2]), fmax=float(res[3]),
                    epot0=float(res[4]), fmax0=float(res[5]))

    def frame_ring(self, nslots=4):
        self._call("mdx_frame_ring", int(nslots))
        self._nslots = int(nslots)

    def capture_frame(self, slot, want_v=False):
        self._call("mdx_capture_frame", int(slot), 1 if want_v else 0)

    def capture_frame_xtc(self, slot, precision=1000.0, want_v=False):
        """Capture with the xtc rounding done on the GPU (frame_wait_xtc fetches it)."""
        self._call("mdx_capture_frame_xtc", int(slot), 1 if want_v else 0, ctypes.c_float(float(precision)))

    def frame_wait_xtc(self, slot):
        """-> (step, ix (N,3) int32 view of the pinned slot, v or None, box (3,3) f64, stats (8,) int32)."""
        px = ctypes.POINTER(ctypes.c_int)(); pv = ctypes.POINTER(ctypes.c_float)()
        box = np.zeros(9, np.float64); stats = np.zeros(8, np.int32)
        st = ctypes.c_longlong(0)
        rc = self.lib.mdx_frame_wait_xtc(self.h, int(slot), ctypes.byref(px), ctypes.byref(pv), _p(box, ctypes.c_double), ctypes.byref(st),
                                         _p(stats, ctypes.c_int))
        if rc != 0:
            raise MdxError(f"mdx_frame_wait_xtc failed ({rc}): " + self.lib.mdx_last_error(self.h).decode())
        ix = np.ctypeslib.as_array(px, shape=(self.n, 3))
        v = np.ctypeslib.as_array(pv, shape=(self.n, 3)) if pv else None
        return int(st.value), ix, v, box.reshape(3, 3), stats

    def frame_wait(self, slot):
        """-> (step, x (N,3) f32 view of the pinned slot, v or None, box (3,3) f64). The views
        are valid until the slot is captured into again."""
        fp = ctypes.POINTER(ctypes.c_float)
        px, pv = fp(), fp()
        box = np.zeros(9, np.float64)
        st = ctypes.c_longlong(0)
        rc = self.lib.mdx_frame_wait(self.h, int(slot), ctypes.byref(px), ctypes.byref(pv), _p(box, ctypes.c_double), ctypes.byref(st))
        if rc != 0:
            raise MdxError(f"mdx_frame_wait failed ({rc})")
        x = np.ctypeslib.as_array(px, shape=(self.n, 3))
        v = np.ctypeslib.as_array(pv, shape=(self.n, 3)) if pv else None
        return int(st.value), x, v, box.reshape(3, 3)

    def list_stats(self):
        st = (ctypes.c_longlong * 8)()
        self._call("mdx_get_list_stats", st)
        return dict(superclusters=st[0], cj_entries=st[1], work_units=st[2], excl_masks=st[3], cluster_pairs=st[4],
                    pairs_within_rc=st[5], pairs_with_lj=st[6], streams=st[7])

    def pop_energy_frames(self, max_frames=8192):
        buf = np.zeros((max_frames, 2 + E_N), np.float64)
        n = self.lib.mdx_pop_energy_frames(self.h, _p(buf, ctypes.c_double), ctypes.c_longlong(max_frames))
        if n < 0:
            raise MdxError("mdx_pop_energy_frames failed: " + self.lib.mdx_last_error(self.h).decode())
        return buf[:n]

    def pop_pull_records(self, max_records=65536):
        buf = np.zeros((max_records, 4), np.float64)
        n = self.lib.mdx_pop_pull_records(self.h, _p(buf, ctypes.c_double), ctypes.c_longlong(max_records))
        if n < 0:
            raise MdxError("mdx_pop_pull_records failed: " + self.lib.mdx_last_error(self.h).decode())
        return buf[:n]

    def enable_kernel_timing(self, on=True, stride=7):
        self._call("mdx_enable_kernel_timing", int(stride) if on else 0)

    def debug_time_nonbonded(self, energy=False, iters=50):
        ms = ctypes.c_double(0)
        self._call("mdx_debug_time_nonbonded", 1 if energy else 0, int(iters), ctypes.byref(ms))
        return ms.value

    def debug_time_stage(self, stage, iters=20):
        ms = ctypes.c_double(0)
        self._call("mdx_debug_time_stage", int(stage), int(iters), ctypes.byref(ms))
        return ms.value

    def debug_wave_times(self):
        buf = np.zeros((4096, 10), dtype=np.int64); nw = ctypes.c_int(0)
        self._call("mdx_debug_wave_times", buf.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), 4096, ctypes.byref(nw))
        return buf[:nw.value]

    # ------------------------------------------------------------------ domain decomposition (device pointers as ints)
    def dd_load(self, n_home, n_total, x4_ptr, v4_ptr, n_settle, ndf, density):
        self._call("mdx_dd_load", int(n_home), int(n_total), ctypes.c_void_p(x4_ptr), ctypes.c_void_p(v4_ptr), int(n_settle),
                   float(ndf), float(density))

    def dd_pack(self, n, idx_ptr, out_ptr):
        self._call("mdx_dd_pack", int(n), ctypes.c_void_p(idx_ptr), ctypes.c_void_p(out_ptr))

    def dd_unpack(self, first, n, in_ptr):
        self._call("mdx_dd_unpack", int(first), int(n), ctypes.c_void_p(in_ptr))

    def dd_step_begin(self):
        self._call("mdx_dd_step_begin")

    def dd_wait_spread(self):
        self._call("mdx_dd_wait_spread")

    def dd_step_end(self):
        self._call("mdx_dd_step_end")

    def dd_stream_sync(self):
        self._call("mdx_dd_stream_sync")

    def dd_grid(self):
        p = ctypes.c_void_p(0); n = ctypes.c_longlong(0)
        self._call("mdx_dd_grid", ctypes.byref(p), ctypes.byref(n))
        return p.value, n.value

    def dd_mesh_slabs(self, world, rank):
        """Share the mesh transform among `world` domains (this one transforms chunk `rank` of the half-complex grid's
        planes). False when the PME grid has no plane-wise transform: the caller keeps the sum of the real-space grids."""
        rc = self.lib.mdx_dd_mesh_slabs(self.h, int(world), int(rank))
        if rc == -5:        # MDX_ERR_UNSUPPORTED
            return False
        if rc != 0:
            raise MdxError(f"mdx_dd_mesh_slabs failed ({rc}): " + self.lib.mdx_last_error(self.h).decode())
        return True

    def dd_cgrid(self):
        """(device pointer of the half-complex grid, planes per domain, complex values per plane)."""
        p = ctypes.c_void_p(0); c = ctypes.c_int(0); n = ctypes.c_longlong(0)
        self._call("mdx_dd_cgrid", ctypes.byref(p), ctypes.byref(c), ctypes.byref(n))
        return p.value, c.value, n.value

    def dd_step_mid(self):
        self._call("mdx_dd_step_mid")

    def dd_streams(self):
        """(main stream, PME stream) as raw hipStream_t values (for torch.cuda.ExternalStream)."""
        a = ctypes.c_void_p(0); b = ctypes.c_void_p(0)
        self._call("mdx_dd_streams", ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def dd_state_pointers(self):
        x = ctypes.c_void_p(0); v = ctypes.c_void_p(0)
        self._call("mdx_dd_state_pointers", ctypes.byref(x), ctypes.byref(v))
        return x.value, v.value

    def dd_set_roles(self, mesh_energy, own_tcouple):
        self._call("mdx_dd_set_roles", int(bool(mesh_energy)), int(bool(own_tcouple)))

    def dd_scalars(self):
        """Device pointers of the half-step kinetic energies (8 doubles) and of the momentum sums (4 doubles)."""
        a = ctypes.c_void_p(0); b = ctypes.c_void_p(0)
        self._call("mdx_dd_scalars", ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def dd_tcouple(self, before_first_step=False):
        self._call("mdx_dd_tcouple", int(bool(before_first_step)))

    def dd_kinetic(self):
        self._call("mdx_dd_kinetic")

    def dd_pressure(self):
        """Device pointer of the domain's share of the pressure tensor (9 doubles) after a coupling step."""
        a = ctypes.c_void_p(0)
        self._call("mdx_dd_pressure", ctypes.byref(a))
        return a.value

    def dd_pcouple(self):
        """Scale by the C-rescale factor drawn from the (summed) pressure in the scalars; returns the new box (3, 3)."""
        b = np.zeros(9, np.float64)
        self._call("mdx_dd_pcouple", _p(b, ctypes.c_double))
        return b.reshape(3, 3)

    def dd_com(self, phase):
        self._call("mdx_dd_com", int(phase))

    def debug_stamps(self, reset=False):
        buf = np.zeros((2, 24), dtype=np.int64)
        self._call("mdx_debug_stamps", 1 if reset else 0, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)))
        return buf

    def debug_set(self, key, value):
        self._call("mdx_debug_set", int(key), int(value))

    def kernel_timing(self):
        ms = ctypes.c_double(0); nl = ctypes.c_longlong(0)
        self._call("mdx_get_kernel_timing", ctypes.byref(ms), ctypes.byref(nl))
        return ms.value, nl.value
