"""`.mdp` reader: the configuration surface of the pipeline's MD stages.

Grammar as used by /root/reference/mdp/*.mdp: `key = value ; comment`, `-` and `_`
interchangeable in keys (`pull_coord1_type` in pull_short.mdp:41 vs `cutoff-scheme` in :16),
case-insensitive keys. Unknown keys (e.g. `nstcheckpoint`, pull_short.mdp:10) only warn, as
`grompp -maxwarn 1000` tolerates them (/root/reference/Snakefile:705).
"""
from __future__ import annotations

import re
import warnings

# GROMACS defaults for the keys the engine consumes (documented defaults; the pipeline's files
# override a subset)
DEFAULTS = {
    "integrator": "md", "dt": "0.001", "nsteps": "0", "tinit": "0",
    "nstxout": "0", "nstvout": "0", "nstfout": "0", "nstlog": "1000", "nstenergy": "1000",
    "nstcalcenergy": "100", "nstxout_compressed": "0", "compressed_x_precision": "1000",
    "compressed_x_grps": "", "continuation": "no", "constraints": "none",
    "constraint_algorithm": "lincs", "lincs_order": "4", "lincs_iter": "1",
    "cutoff_scheme": "Verlet", "nstlist": "10", "rlist": "1", "verlet_buffer_tolerance": "0.005",
    "pbc": "xyz", "coulombtype": "cut-off", "rcoulomb": "1", "rvdw": "1", "vdwtype": "cut-off",
    "vdw_modifier": "potential-shift", "coulomb_modifier": "potential-shift",
    "dispcorr": "no", "fourierspacing": "0.12", "pme_order": "4", "ewald_rtol": "1e-5",
    "tcoupl": "no", "tc_grps": "", "tau_t": "", "ref_t": "", "nsttcouple": "-1",
    "pcoupl": "no", "pcoupltype": "isotropic", "tau_p": "1", "ref_p": "1", "compressibility": "4.5e-5",
    "nstpcouple": "-1", "refcoord_scaling": "no",
    "gen_vel": "no", "gen_temp": "300", "gen_seed": "-1", "ld_seed": "-1",
    "nstcomm": "100", "comm_mode": "linear", "comm_grps": "",
    "define": "", "include": "", "title": "", "ns_type": "grid",
    "pull": "no", "pull_ncoords": "1", "pull_ngroups": "0", "pull_nstxout": "50", "pull_nstfout": "50",
    "pull_print_components": "no",
    # energy minimisation (/root/reference/mdp/em.mdp:3-5, em_complex.mdp:3-5) and further documented keys that the
    # engine reads or that only shape GROMACS-side behaviour the engine has no counterpart for
    "emtol": "10.0", "emstep": "0.01", "nstcgsteep": "1000", "nbfgscorr": "10",
    "epsilon_r": "1", "epsilon_rf": "0", "rvdw_switch": "0", "rcoulomb_switch": "0", "ewald_geometry": "3d",
    "fourier_nx": "0", "fourier_ny": "0", "fourier_nz": "0", "lincs_warnangle": "30", "shake_tol": "0.0001",
}
_PULL_GROUP = re.compile(r"pull_group(\d+)_(name|pbcatom)$")
_PULL_COORD = re.compile(r"pull_coord(\d+)_(type|geometry|groups|dim|start|k|rate|init|vec|origin)$")


def _norm(key: str) -> str:
    return key.strip().lower().replace("-", "_")


class Mdp(dict):
    """Normalised key -> raw string value, with typed getters and a list of warnings."""

    def __init__(self):
        super().__init__()
        self.warnings: list[str] = []

    def s(self, key, default=None):
        k = _norm(key)
        if k in self:
            return self[k]
        return DEFAULTS.get(k, default)

    def i(self, key, default=None):
        v = self.s(key, default)
        return int(float(v)) if v not in (None, "") else default

    def f(self, key, default=None):
        v = self.s(key, default)
        return float(v) if v not in (None, "") else default

    def b(self, key):
        return str(self.s(key, "no")).strip().lower() in ("yes", "true", "on", "1")

    def words(self, key):
        v = self.s(key, "")
        return v.split() if v else []

    def floats(self, key):
        return [float(w) for w in self.words(key)]

    def defines(self):
        """`define = -DPOSRES -DFLEXIBLE` -> {'POSRES': '', 'FLEXIBLE': ''}."""
        out = {}
        for w in self.words("define"):
            if w.startswith("-D"):
                name, _, val = w[2:].partition("=")
                out[name] = val
        return out


def parse_mdp(text: str) -> Mdp:
    m = Mdp()
    for ln, raw in enumerate(text.splitlines(), 1):
        line = raw.split(";", 1)[0].strip()
        if not line:
            continue
        if "=" not in line:
            m.warnings.append(f"line {ln}: no '=' in '{line}'")
            continue
        key, val = line.split("=", 1)
        k = _norm(key)
        if k in m:
            m.warnings.append(f"line {ln}: parameter '{key.strip()}' set twice, keeping the last")
        if k not in DEFAULTS and not _PULL_GROUP.match(k) and not _PULL_COORD.match(k):
            m.warnings.append(f"line {ln}: unknown left-hand '{key.strip()}' in parameter file")
        m[k] = val.strip()
    return m


def read_mdp(path: str, warn: bool = False) -> Mdp:
    with open(path) as f:
        m = parse_mdp(f.read())
    if warn:
        for w in m.warnings:
            warnings.warn(f"{path}: {w}")
    return m


def effective_coupling_steps(mdp: Mdp):
    """nsttcouple / nstpcouple when left at -1: 10 steps, but never fewer than 5 coupling steps
    per tau for the thermostat and 10 per tau_p for the barostat (GROMACS' documented rule)."""
    dt = mdp.f("dt")
    nst = mdp.i("nsttcouple")
    if nst is None or nst < 0:
        nst = 10
        taus = [t for t in mdp.floats("tau_t") if t > 0]
        if taus:
            nst = max(1, min(nst, int(min(taus) / (5.0 * dt) + 1e-9)))
    nsp = mdp.i("nstpcouple")
    if nsp is None or nsp < 0:
        nsp = 10
        tp = mdp.f("tau_p")
        if tp and tp > 0:
            nsp = max(1, min(nsp, int(tp / (10.0 * dt) + 1e-9)))
    return nst, nsp
