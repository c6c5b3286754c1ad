"""CPU BASELINE — TEST / BENCH INFRASTRUCTURE ONLY (driver of oracle/cpu_md.c).

"Own CPU restatement — not GROMACS": one leap-frog step of the bench workload on the host cores in single precision,
organised like a CPU MD code (buffered Verlet list reused for `nstlist` steps, list-driven LJ + Ewald real-space
kernel, B-spline spread / gather in C, 3D transforms by pocketfft, SETTLE in C, LINCS and the bonded terms of the few
thousand solute atoms vectorised in numpy). bench.py times it at 4 threads (the pipeline's `-ntomp 4`,
/root/reference/Snakefile:232-233) and at all cores; tests/ check its forces against the float64 oracle.
Never imported by the product package.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import mdref as R

_here = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_here, "libcpumd.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/libcpumd.so missing: run `make -C oracle`")
        L = ctypes.CDLL(path)
        fp, ip, dp, vp = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double), ctypes.c_void_p
        ci, cf = ctypes.c_int, ctypes.c_float
        L.cmd_list_new.restype = vp
        L.cmd_list_free.argtypes = [vp]
        L.cmd_list_build.restype = ctypes.c_long
        L.cmd_list_build.argtypes = [vp, ci, fp, fp, cf, ip, ip, fp, ip, ci]
        L.cmd_rewrap.argtypes = [ci, fp, fp, ip, fp]
        L.cmd_nonbonded.restype = ctypes.c_long
        L.cmd_nonbonded.argtypes = [vp, fp, fp, fp, fp, cf, cf, fp, dp, ci, ci, fp]
        L.cmd_excl_correction.restype = ctypes.c_double
        L.cmd_excl_correction.argtypes = [ci, fp, fp, cf, ip, ip, fp]
        L.cmd_pme_spread.argtypes = [ci, fp, fp, fp, ci, ci, ci, fp, ci]
        L.cmd_pme_gather.argtypes = [ci, fp, fp, fp, ci, ci, ci, fp, fp, ci]
        L.cmd_leapfrog.argtypes = [ci, fp, fp, fp, fp, cf, fp, ci]
        L.cmd_settle.argtypes = [ci, ip, fp, fp, fp, fp, cf, cf, cf, ci]
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


class CpuMD:
    """NVE leap-frog of one MDSystem on `threads` host threads (no coupling: the thermostat and barostat are a few
    scalar operations per step and do not change what a step costs)."""

    def __init__(self, s, rc, beta, grid, dt, rlist, nstlist=20, threads=4):
        self.L = lib()
        self.s, self.rc, self.beta, self.grid, self.dt, self.rlist, self.nstlist, self.T = s, float(rc), float(beta), tuple(grid), float(dt), float(rlist), int(nstlist), int(threads)
        n = self.n = s.natoms
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        self.x, self.v = f32(s.x).copy(), f32(s.v if s.v is not None else np.zeros((n, 3))).copy()
        self.box = f32(np.asarray(s.box).reshape(9))
        self.recip = f32(np.linalg.inv(np.asarray(s.box, dtype=np.float64)).reshape(9))
        self.q, self.invm = f32(s.charge), f32(1.0 / s.mass)
        self.hs, self.se = f32(0.5 * s.sigma), f32(2.0 * np.sqrt(s.eps))
        self.es = np.ascontiguousarray(s.excl_start, dtype=np.int32)
        self.el = np.ascontiguousarray(s.excl_list if len(s.excl_list) else np.zeros(1), dtype=np.int32)
        self.settles = np.ascontiguousarray(s.settles, dtype=np.int32)
        self.list = self.L.cmd_list_new()
        self.xw = np.zeros((n, 3), np.float32); self.shift_of = np.zeros((n, 3), np.int32)
        self.f = np.zeros((n, 3), np.float32); self.fp = np.zeros((n, 3), np.float32); self.xp = np.zeros((n, 3), np.float32)
        self.scratch = np.zeros(self.T * n * 3, np.float32)
        self.qgrid = np.zeros(self.grid, np.float32)
        self.since_list = 1 << 30
        self.npairs_listed = 0
        self._influence()
        self._lincs_setup()
        self.timers = {}

    def close(self):
        if self.list:
            self.L.cmd_list_free(self.list); self.list = None

    # ---- reciprocal space: influence function on the half-complex grid of rfftn over (x, y, z), z halved
    def _influence(self):
        nx, ny, nz = self.grid
        box = np.asarray(self.s.box, dtype=np.float64)
        V = abs(np.linalg.det(box)); recip = np.linalg.inv(box)
        sg = lambda k, K: np.where(k < (K + 1) // 2, k, k - K)
        kx = sg(np.arange(nx), nx)[:, None, None]; ky = sg(np.arange(ny), ny)[None, :, None]; kz = np.arange(nz // 2 + 1)[None, None, :]
        M = np.stack(np.broadcast_arrays(kx, ky, kz), -1).astype(np.float64)
        mv = M @ recip.T
        m2 = (mv * mv).sum(-1); m2[0, 0, 0] = 1.0
        B = 1.0 / (R._bspline_moduli(nx)[np.arange(nx)][:, None, None] * R._bspline_moduli(ny)[np.arange(ny)][None, :, None]
                   * R._bspline_moduli(nz)[np.arange(nz // 2 + 1)][None, None, :])
        G = R.ONE_4PI_EPS0 / (np.pi * V) * np.exp(-np.pi ** 2 * m2 / self.beta ** 2) / m2 * B
        G[0, 0, 0] = 0.0
        self.G = (G * (nx * ny * nz)).astype(np.float32)          # the inverse transform below is normalised

    def _lincs_setup(self):
        c = np.asarray(self.s.constraints, dtype=np.int64).reshape(-1, 2)
        self.cons = c
        if len(c) == 0:
            return
        invm = 1.0 / self.s.mass
        self.clen = np.asarray(self.s.constraint_len, dtype=np.float64)
        self.cS = 1.0 / np.sqrt(invm[c[:, 0]] + invm[c[:, 1]])
        from collections import defaultdict
        by = defaultdict(list)
        for k, (a, b) in enumerate(c):
            by[int(a)].append(k); by[int(b)].append(k)
        ks, ms, cf = [], [], []
        for atom, lst in by.items():
            for k in lst:
                for m in lst:
                    if k != m:
                        same = (c[k, 0] == atom) == (c[m, 0] == atom)
                        ks.append(k); ms.append(m); cf.append((-1.0 if same else 1.0) * invm[atom] * self.cS[k] * self.cS[m])
        self.ck, self.cm, self.ccoef = np.array(ks, dtype=np.int64), np.array(ms, dtype=np.int64), np.array(cf)
        self.cima, self.cimb = invm[c[:, 0]], invm[c[:, 1]]

    def _lincs(self, x_old, xp, order=4, niter=1):
        c = self.cons
        a, b = c[:, 0], c[:, 1]
        xo = x_old.astype(np.float64); x = xp.astype(np.float64)
        d0 = xo[a] - xo[b]
        B = d0 / np.linalg.norm(d0, axis=1)[:, None]
        coef = self.ccoef * (B[self.ck] * B[self.cm]).sum(1)
        K = len(c)

        def solve(rhs):
            sol = rhs.copy()
            for _ in range(order):
                rhs = np.bincount(self.ck, weights=coef * rhs[self.cm], minlength=K)
                sol += rhs
            return sol

        def apply(lam):
            np.add.at(x, a, -(self.cima * lam)[:, None] * B)
            np.add.at(x, b, (self.cimb * lam)[:, None] * B)
        d = x[a] - x[b]
        apply(self.cS * solve(self.cS * ((B * d).sum(1) - self.clen)))
        for _ in range(niter):
            d = x[a] - x[b]
            p = np.sqrt(np.maximum(2 * self.clen ** 2 - (d * d).sum(1), 0.0))
            apply(self.cS * solve(self.cS * (self.clen - p)))
        return x

    # ---- forces at the current positions
    def forces(self, want_energy=False):
        import time
        from scipy import fft as sfft
        L, n, T = self.L, self.n, self.T
        t0 = time.perf_counter()
        if self.since_list >= self.nstlist:
            self.npairs_listed = L.cmd_list_build(self.list, n, _f(self.x), _f(self.box), self.rlist, _i(self.es), _i(self.el), _f(self.xw),
                                                  _i(self.shift_of), T)
            self.since_list = 0
        else:
            L.cmd_rewrap(n, _f(self.x), _f(self.box), _i(self.shift_of), _f(self.xw))
        t1 = time.perf_counter()
        ener = np.zeros(2)
        self.npairs = L.cmd_nonbonded(self.list, _f(self.xw), _f(self.q), _f(self.hs), _f(self.se), self.rc, self.beta, _f(self.f),
                                      ener.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 1 if want_energy else 0, T, _f(self.scratch))
        e_excl = L.cmd_excl_correction(n, _f(self.x), _f(self.q), self.beta, _i(self.es), _i(self.el), _f(self.f))
        t2 = time.perf_counter()
        nx, ny, nz = self.grid
        L.cmd_pme_spread(n, _f(self.x), _f(self.q), _f(self.recip), nx, ny, nz, _f(self.qgrid), T)
        t3 = time.perf_counter()
        Fq = sfft.rfftn(self.qgrid, workers=T)
        e_rec = None
        if want_energy:
            mult = np.where((np.arange(nz // 2 + 1) == 0) | ((nz % 2 == 0) & (np.arange(nz // 2 + 1) == nz // 2)), 1.0, 2.0)[None, None, :]
            e_rec = float(0.5 * (mult * self.G.astype(np.float64) / (nx * ny * nz) * np.abs(Fq) ** 2).sum())
        Fq *= self.G
        phi = np.ascontiguousarray(sfft.irfftn(Fq, s=(nx, ny, nz), workers=T), dtype=np.float32)
        t4 = time.perf_counter()
        L.cmd_pme_gather(n, _f(self.x), _f(self.q), _f(self.recip), nx, ny, nz, _f(phi), _f(self.f), T)
        t5 = time.perf_counter()
        s = self.s
        x64 = None
        eb = 0.0
        if len(s.bonds) or len(s.angles) or len(s.dihedrals) or len(s.pairs14):
            x64 = self.x.astype(np.float64)
            fb = np.zeros((n, 3))
            for e, f, _ in (R.bonds(x64, s.bonds, s.bond_b0, s.bond_kb), R.angles(x64, s.angles, s.angle_th0, s.angle_k),
                            R.dihedrals(x64, s.dihedrals, s.dih_phi0, s.dih_k, s.dih_mult)):
                fb += f; eb += e
            el, ec, f, _ = R.pairs14(x64, s.pairs14, s.pair_c6, s.pair_c12, s.pair_qq)
            fb += f; eb += el + ec
            self.f += fb.astype(np.float32)
        t6 = time.perf_counter()
        for k, v in (("list", t1 - t0), ("nonbonded", t2 - t1), ("spread", t3 - t2), ("fft_solve", t4 - t3), ("gather", t5 - t4), ("bonded", t6 - t5)):
            self.timers[k] = self.timers.get(k, 0.0) + v
        self.energies = dict(lj=ener[0], coul_sr=ener[1], excl=e_excl, recip=e_rec, bonded=eb)
        return self.f

    def step(self, nsteps=1):
        import time
        L, n, T = self.L, self.n, self.T
        for _ in range(int(nsteps)):
            self.forces()
            t0 = time.perf_counter()
            L.cmd_leapfrog(n, _f(self.x), _f(self.v), _f(self.f), _f(self.invm), self.dt, _f(self.xp), T)
            if len(self.settles):
                L.cmd_settle(len(self.settles), _i(self.settles), _f(self.x), _f(self.xp), _f(self.v), _f(self.invm), float(self.s.settle_doh),
                             float(self.s.settle_dhh), self.dt, T)
            if len(self.cons):
                before = self.xp.astype(np.float64)
                xc = self._lincs(self.x, self.xp)
                self.v += ((xc - before) / self.dt).astype(np.float32)
                self.xp[:] = xc.astype(np.float32)
            self.x, self.xp = self.xp, self.x
            self.since_list += 1
            self.timers["update"] = self.timers.get("update", 0.0) + time.perf_counter() - t0
