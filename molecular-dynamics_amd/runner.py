"""`grompp` and `mdrun` as the pipeline calls them, on top of the HIP engine.

Reference call sites: `gmx grompp -f F.mdp -c C.gro [-r R.gro] [-t T.cpt] -p P.top [-n N.ndx] -o O.tpr
[-maxwarn k]` (/root/reference/Snakefile:705,738,994;
/root/reference/scripts/run_smd_with_distance_monitoring.py:66-75) and `gmx mdrun -v -deffnm PREFIX
-ntmpi 1 -ntomp 4` (/root/reference/Snakefile:707,740,998;
run_smd_with_distance_monitoring.py:89-95), plus `-cpi CPT -nsteps 1`
(/root/reference/scripts/run_smd_with_recovery.py:149-156).

mdrun leaves PREFIX.gro, .cpt, .log, .edr, .xtc (nstxout-compressed > 0), .trr (nstxout > 0) and,
with `pull = yes`, PREFIX_pullx.xvg / PREFIX_pullf.xvg; it prints progress lines while running,
flushes the pull files at least once a second, and on SIGTERM/SIGINT stops at the next chunk
boundary, writes .gro and .cpt and exits 0.
"""
from __future__ import annotations

import argparse
import math
import os
import queue
import signal
import sys
import threading
import time

import numpy as np

from . import units as U
from .io import mdp as mdpio
from .io import gro as groio
from .io import ndx as ndxio
from .io import top as topio
from .io import tpr as tprio
from .io import xvg as xvgio
from .io import energy as enio

PROGRAM = "mdx gmx shim (MI355X engine)"


class InputError(Exception):
    pass


# ------------------------------------------------------------------------------------------
# grompp
# ------------------------------------------------------------------------------------------
def _grompp_parser():
    p = argparse.ArgumentParser(prog="gmx grompp", add_help=True, allow_abbrev=False)
    p.add_argument("-f", default="grompp.mdp")
    p.add_argument("-c", default="conf.gro")
    p.add_argument("-r", default=None)
    p.add_argument("-rb", default=None)
    p.add_argument("-t", default=None)
    p.add_argument("-p", default="topol.top")
    p.add_argument("-n", default=None)
    p.add_argument("-o", default="topol.tpr")
    p.add_argument("-po", default="mdout.mdp")
    p.add_argument("-pp", default=None)
    p.add_argument("-maxwarn", type=int, default=0)
    p.add_argument("-v", action="store_true")
    p.add_argument("-nov", dest="v", action="store_false")
    return p


SUPPORTED = {
    "integrator": ("md", "steep"),
    "cutoff_scheme": ("verlet",),
    "pbc": ("xyz",),
    "coulombtype": ("pme",),
    "vdwtype": ("cut-off", "cutoff"),
    "tcoupl": ("no", "v-rescale", "berendsen"),
    "pcoupl": ("no", "parrinello-rahman", "c-rescale", "berendsen"),
    "pcoupltype": ("isotropic",),
    "constraints": ("none", "h-bonds", "all-bonds"),
    "constraint_algorithm": ("lincs",),
    "comm_mode": ("linear", "none"),
    "refcoord_scaling": ("no", "all", "com"),
    "dispcorr": ("no", "ener", "enerpres"),
}


def check_mdp(m: mdpio.Mdp, groups: ndxio.IndexGroups, natoms: int):
    """-> (errors, warnings). Errors abort grompp; warnings count against -maxwarn."""
    errors, warns = [], list(m.warnings)
    for key, allowed in SUPPORTED.items():
        val = str(m.s(key)).strip().lower()
        if val not in allowed:
            errors.append(f"{key} = {m.s(key)} is not supported by this engine (supported: {', '.join(allowed)})")
    if m.s("tcoupl").lower() == "berendsen":
        warns.append("tcoupl = berendsen is replaced by V-rescale (same relaxation, correct ensemble)")
    if m.s("pcoupl").lower() == "berendsen":
        warns.append("pcoupl = berendsen is replaced by C-rescale (same relaxation, correct ensemble)")
    if abs(m.f("rcoulomb") - m.f("rvdw")) > 1e-6:
        errors.append(f"rcoulomb ({m.f('rcoulomb')}) and rvdw ({m.f('rvdw')}) must be equal")
    if m.i("pme_order") != 4:
        errors.append("pme_order must be 4")
    if m.s("integrator").lower() == "md" and m.f("dt") <= 0:
        errors.append("dt must be positive")
    tc = m.s("tcoupl").lower() != "no"
    if tc:
        grps = m.words("tc_grps")
        if not grps:
            errors.append("tcoupl is on but tc-grps is empty")
        if len(m.floats("tau_t")) != len(grps) or len(m.floats("ref_t")) != len(grps):
            errors.append(f"tc-grps has {len(grps)} groups but tau_t/ref_t have {len(m.floats('tau_t'))}/{len(m.floats('ref_t'))} values")
        cover = np.zeros(natoms, dtype=np.int32)
        for g in grps:
            if g not in groups:
                errors.append(f"tc-grps group '{g}' not found in the index groups ({', '.join(groups.names)})")
            else:
                np.add.at(cover, groups[g], 1)
        if not errors and (cover != 1).any():
            errors.append(f"{int((cover == 0).sum())} atoms are not part of any T-coupling group and "
                          f"{int((cover > 1).sum())} are in more than one")
        if len(grps) > 8:
            errors.append("at most 8 T-coupling groups are supported")
    if m.b("pull"):
        if m.i("pull_ncoords") != 1 or m.i("pull_ngroups") != 2:
            errors.append("pull: exactly one coordinate between two groups is supported")
        geom = str(m.s("pull_coord1_geometry", "distance")).lower()
        if geom != "distance":
            errors.append(f"pull_coord1_geometry = {geom} is not supported (distance only)")
        typ = str(m.s("pull_coord1_type", "umbrella")).lower().replace("_", "-")
        if typ not in ("umbrella", "constant-force"):
            errors.append(f"pull_coord1_type = {typ} is not supported (umbrella, constant-force)")
        for gi in (1, 2):
            name = m.s(f"pull_group{gi}_name")
            if not name:
                errors.append(f"pull_group{gi}_name is missing")
            elif name not in groups:
                errors.append(f"pull group '{name}' not found in the index groups ({', '.join(groups.names)})")
            elif len(groups[name]) == 0:
                errors.append(f"pull group '{name}' is empty")
    for key in ("compressed_x_grps",):
        w = m.words(key)
        if w and w[0] not in groups:
            errors.append(f"{key} group '{w[0]}' not found in the index groups")
    cg = m.words("comm_grps")
    if cg and (len(cg) > 1 or len(groups[cg[0]] if cg[0] in groups else []) != natoms):
        warns.append("comm-grps other than the whole system: centre-of-mass motion is removed for the whole system")
    return errors, warns


def find_real_gmx():
    """A GROMACS installed next to this shim (for .tpr files the analysis tools can read)."""
    cand = os.environ.get("MDX_REAL_GMX")
    if cand:
        return cand if os.path.isfile(cand) and os.access(cand, os.X_OK) else None
    me = os.path.realpath(sys.argv[0]) if sys.argv and sys.argv[0] else ""
    for d in os.environ.get("PATH", "").split(os.pathsep):
        for name in ("gmx", "gmx_mpi"):
            p = os.path.join(d, name)
            if os.path.isfile(p) and os.access(p, os.X_OK) and os.path.realpath(p) != me:
                try:
                    with open(p, "rb") as f:
                        head = f.read(256)
                    if b"molecular-dynamics_amd" in head or b"mdx" in head[:64]:
                        continue
                except OSError:
                    continue
                return p
    return None


def grompp(argv, out=sys.stdout, err=sys.stderr) -> int:
    a, extra = _grompp_parser().parse_known_args(argv)
    if extra:
        err.write(f"NOTE: ignoring unrecognised grompp options: {' '.join(extra)}\n")
    try:
        for opt, path in (("-f", a.f), ("-c", a.c), ("-p", a.p), ("-r", a.r), ("-t", a.t), ("-n", a.n)):
            if path is not None and not os.path.isfile(path):
                raise InputError(f"File '{path}' does not exist or is not accessible (option {opt})")
        m = mdpio.read_mdp(a.f)
        g = groio.read_gro(a.c)
        defines = m.defines()
        inc = [w[2:] for w in m.words("include") if w.startswith("-I")]
        top = topio.read_top(a.p, defines=defines, include_dirs=inc)
        use_posres = "POSRES" in defines
        ref = None
        if use_posres:
            if a.r is None:
                raise InputError("position restraints are switched on (define = -DPOSRES) but no reference "
                                 "coordinates were given: use -r")
            ref = groio.read_gro(a.r)
        groups = ndxio.read_ndx(a.n) if a.n else None
        constraints = m.s("constraints").lower()
        if m.s("integrator").lower() == "steep" and constraints == "none":
            pass
        sysm = topio.build_system(top, g, constraints=constraints, posres_ref=ref, use_posres=use_posres, groups=groups)
        errors, warns = check_mdp(m, sysm.groups, sysm.natoms)
        warns += top.warnings
        qtot = float(sysm.charge.sum())
        if abs(qtot) > 1e-3:
            warns.append(f"System has non-zero total charge: {qtot:.6f}")
        state = dict(step=0, time=m.f("tinit"))
        if a.t:
            cp = tprio.load_checkpoint(a.t)
            if len(cp["x"]) != sysm.natoms:
                raise InputError(f"checkpoint {a.t} has {len(cp['x'])} atoms, the topology {sysm.natoms}")
            # full-precision continuation: coordinates, velocities and box from the checkpoint;
            # positions are already whole there
            sysm.x = cp["x"]; sysm.v = cp["v"]; sysm.box = cp["box"]
            if use_posres and ref is not None:
                pass
        if sysm.v is None:
            sysm.v = np.zeros_like(sysm.x)
            if not m.b("gen_vel") and m.s("integrator").lower() == "md":
                warns.append("no velocities in the coordinate file and gen_vel = no: starting from zero velocities")
        for w in warns:
            err.write(f"WARNING: {w}\n")
        if errors:
            for e in errors:
                err.write(f"ERROR: {e}\n")
            raise InputError(f"There were {len(errors)} errors in input file(s)")
        if len(warns) > a.maxwarn:
            raise InputError(f"Too many warnings ({len(warns)}). If you are sure all warnings are harmless, "
                             "use the -maxwarn option.")
        meta = dict(files=dict(mdp=os.path.abspath(a.f), gro=os.path.abspath(a.c), top=os.path.abspath(a.p),
                               ref=os.path.abspath(a.r) if a.r else None, cpt=os.path.abspath(a.t) if a.t else None,
                               ndx=os.path.abspath(a.n) if a.n else None),
                    use_posres=use_posres, state=state, created=time.strftime("%Y-%m-%d %H:%M:%S"))
        real = find_real_gmx()
        target = a.o
        if real is not None:
            import subprocess
            r = subprocess.run([real, "grompp"] + list(argv), capture_output=True, text=True)
            err.write(r.stderr[-4000:])
            if r.returncode == 0:
                target = a.o + ".mdx"
            else:
                err.write("NOTE: the installed GROMACS grompp failed; writing the shim's run input only\n")
        os.makedirs(os.path.dirname(os.path.abspath(target)), exist_ok=True)
        tprio.save_run_input(target, sysm, dict(m), meta)
        if a.po:
            try:
                with open(a.po, "w") as f:
                    f.write("; parameters as read by the shim's grompp\n")
                    for k in sorted(set(mdpio.DEFAULTS) | set(m)):
                        f.write("%-28s = %s\n" % (k.replace("_", "-"), m.s(k, "")))
            except OSError:
                pass
        out.write(f"{PROGRAM}: grompp\n  {sysm.natoms} atoms, {len(top.molecules)} molecule blocks, "
                  f"{len(sysm.settles)} rigid waters, {len(sysm.constraints)} constraints, "
                  f"{len(sysm.posres_idx)} position restraints\n  run input written to {target}\n")
        return 0
    except (InputError, topio.TopologyError, ValueError, KeyError, OSError) as e:
        err.write(f"\n-------------------------------------------------------\nProgram:     gmx grompp ({PROGRAM})\n\n"
                  f"Fatal error:\n{e}\n-------------------------------------------------------\n")
        return 1


# ------------------------------------------------------------------------------------------
# make_ndx (the non-interactive uses of the pipeline)
# ------------------------------------------------------------------------------------------
def make_ndx(argv, stdin=None, out=sys.stdout, err=sys.stderr) -> int:
    """`gmx make_ndx -f conf.gro -o index.ndx` fed with `q`
    (/root/reference/scripts/add_active_site_group.py:73-77): default groups from residue and
    atom names. The editing commands the pipeline pipes in are handled too: `A|B`, `A&B`,
    `name N text`, `del N`, `q` (/root/reference/Snakefile:1051)."""
    p = argparse.ArgumentParser(prog="gmx make_ndx", allow_abbrev=False)
    p.add_argument("-f", default=None)
    p.add_argument("-n", default=None)
    p.add_argument("-o", default="index.ndx")
    a, _ = p.parse_known_args(argv)
    try:
        if a.f is None and a.n is None:
            raise InputError("make_ndx needs -f or -n")
        if a.n:
            groups = ndxio.read_ndx(a.n)
        else:
            if tprio.is_run_input(a.f) or os.path.isfile(a.f + ".mdx"):
                sysm, _, _ = tprio.load_run_input(a.f + ".mdx" if os.path.isfile(a.f + ".mdx") else a.f)
                groups = ndxio.default_groups(sysm.resname, sysm.atomname)
            else:
                g = groio.read_gro(a.f)
                groups = ndxio.default_groups(g.resname, g.atomname)
        text = (stdin if stdin is not None else sys.stdin).read() if (stdin is not None or not sys.stdin.isatty()) else "q\n"
        for line in text.splitlines():
            cmd = line.strip()
            if not cmd:
                continue
            if cmd == "q":
                break
            w = cmd.split()
            if w[0] == "name" and len(w) >= 3:
                groups.names[int(w[1])] = w[2]
            elif w[0] == "del" and len(w) >= 2:
                k = int(w[1]); del groups.names[k]; del groups.atoms[k]
            elif "|" in cmd or "&" in cmd:
                op = "|" if "|" in cmd else "&"
                lhs, rhs = (t.strip() for t in cmd.split(op, 1))

                def pick(tok):
                    return groups.atoms[int(tok)] if tok.lstrip("-").isdigit() else groups[tok.strip('"')]
                A, B = pick(lhs), pick(rhs)
                res = np.union1d(A, B) if op == "|" else np.intersect1d(A, B)
                nm = lambda t: groups.names[int(t)] if t.lstrip("-").isdigit() else t.strip('"')
                groups.add(f"{nm(lhs)}_{nm(rhs)}" if op == "|" else f"{nm(lhs)}_&_{nm(rhs)}", res)
            else:
                err.write(f"make_ndx: command '{cmd}' is not supported and was skipped\n")
        ndxio.write_ndx(a.o, groups)
        for i, (n, at) in enumerate(zip(groups.names, groups.atoms)):
            out.write("%3d %-20s: %5d atoms\n" % (i, n, len(at)))
        return 0
    except (InputError, ValueError, KeyError, IndexError, OSError) as e:
        err.write(f"\nProgram:     gmx make_ndx ({PROGRAM})\n\nFatal error:\n{e}\n")
        return 1


# ------------------------------------------------------------------------------------------
# trjconv: the one form the pipeline needs from the shim's own containers
# ------------------------------------------------------------------------------------------
def trjconv_can_serve(argv) -> bool:
    """True for `gmx trjconv -s P.tpr -f P.cpt -o P.gro [-dump t]` on files this shim wrote: the recovery step of
    the SMD drivers (/root/reference/scripts/run_smd_with_distance_monitoring.py:231-244,
    /root/reference/scripts/run_smd_with_recovery.py:107-127). Everything else belongs to a real GROMACS."""
    try:
        a, _ = _trjconv_parser().parse_known_args(argv)
    except SystemExit:
        return False
    if not (a.s and a.f and a.o) or not a.f.endswith(".cpt") or not a.o.endswith(".gro"):
        return False
    src = a.s + ".mdx" if os.path.isfile(a.s + ".mdx") else a.s
    return tprio.is_run_input(src) and tprio.is_run_input(a.f)


def _trjconv_parser():
    p = argparse.ArgumentParser(prog="gmx trjconv", add_help=False, allow_abbrev=False)
    p.add_argument("-s", default=None)
    p.add_argument("-f", default=None)
    p.add_argument("-o", default=None)
    p.add_argument("-n", default=None)
    p.add_argument("-dump", type=float, default=None)
    return p


def trjconv(argv, stdin=None, out=sys.stdout, err=sys.stderr) -> int:
    """Writes the checkpoint's state as a `.gro` (the whole system, velocities included). The group number the
    drivers pipe in (`0` = System) is read and must select the whole system; `-dump` is accepted for any value: a
    checkpoint holds exactly one frame."""
    a, _ = _trjconv_parser().parse_known_args(argv)
    try:
        src = a.s + ".mdx" if os.path.isfile(a.s + ".mdx") else a.s
        sysm, _mdp, _meta = tprio.load_run_input(src)
        cp = tprio.load_checkpoint(a.f)
        if len(cp["x"]) != sysm.natoms:
            raise InputError(f"checkpoint {a.f} has {len(cp['x'])} atoms, the run input {sysm.natoms}")
        text = ""
        if stdin is not None or not sys.stdin.isatty():
            text = (stdin if stdin is not None else sys.stdin).read()
        sel = text.split()[0] if text.split() else "0"
        groups = sysm.groups
        names = list(groups.names) if hasattr(groups, "names") else list(groups.keys())
        if sel.lstrip("-").isdigit():
            name = names[int(sel)] if 0 <= int(sel) < len(names) else None
        else:
            name = sel if sel in names else None
        if name is None or len(groups[name]) != sysm.natoms:
            raise InputError(f"trjconv: output group '{sel}' is not the whole system; only System (0) is served by the shim")
        out.write(f"Select group for output\nSelected {sel}: '{name}'\n")
        groio.write_system_gro(a.o, sysm, x=cp["x"], v=cp["v"], box=cp["box"],
                               title=f"{sysm.title}, t= {cp['time']:.5f} step= {cp['step']}")
        out.write(f"Dumping frame at t= {cp['time']:g} ps ({PROGRAM}: state of {a.f})\n")
        return 0
    except (InputError, ValueError, KeyError, IndexError, OSError) as e:
        err.write(f"\nProgram:     gmx trjconv ({PROGRAM})\n\nFatal error:\n{e}\n")
        return 1


# ------------------------------------------------------------------------------------------
# mdrun
# ------------------------------------------------------------------------------------------
def _mdrun_parser():
    p = argparse.ArgumentParser(prog="gmx mdrun", add_help=True, allow_abbrev=False)
    p.add_argument("-deffnm", default=None)
    p.add_argument("-s", default=None)
    p.add_argument("-o", default=None)
    p.add_argument("-x", default=None)
    p.add_argument("-c", default=None)
    p.add_argument("-e", default=None)
    p.add_argument("-g", default=None)
    p.add_argument("-cpi", default=None, nargs="?", const="")
    p.add_argument("-cpo", default=None)
    p.add_argument("-px", default=None)
    p.add_argument("-pf", default=None)
    p.add_argument("-nsteps", type=int, default=None)
    p.add_argument("-cpt", type=float, default=15.0)
    p.add_argument("-v", action="store_true")
    p.add_argument("-append", action="store_true", default=True)
    p.add_argument("-noappend", dest="append", action="store_false")
    p.add_argument("-gpu_id", default=None)
    p.add_argument("-resethway", action="store_true")
    for opt in ("-ntmpi", "-ntomp", "-nt", "-nb", "-pme", "-bonded", "-update", "-pin", "-pinoffset", "-maxh", "-dlb",
                "-npme", "-ntomp_pme", "-pinstride", "-nstlist", "-tunepme"):
        p.add_argument(opt, default=None)
    return p


def _backup(path):
    if os.path.exists(path):
        d, b = os.path.split(path)
        k = 1
        while os.path.exists(os.path.join(d, f"#{b}.{k}#")):
            k += 1
        os.replace(path, os.path.join(d, f"#{b}.{k}#"))


def _gcd_pos(*vals):
    g = 0
    for v in vals:
        if v and v > 0:
            g = math.gcd(g, int(v))
    return g


def _seed(val):
    if val is None or int(val) < 0:
        return int.from_bytes(os.urandom(6), "little")
    return int(val)


def mdrun(argv, out=sys.stdout, err=sys.stderr) -> int:
    a, extra = _mdrun_parser().parse_known_args(argv)
    t_start = time.time()
    prefix = a.deffnm
    tpr = a.s or ((prefix + ".tpr") if prefix else "topol.tpr")

    def name(opt, suffix, default):
        v = getattr(a, opt)
        if v:
            return v
        return (prefix + suffix) if prefix else default

    f_gro, f_cpt = name("c", ".gro", "confout.gro"), name("cpo", ".cpt", "state.cpt")
    f_log, f_edr = name("g", ".log", "md.log"), name("e", ".edr", "ener.edr")
    f_xtc, f_trr = name("x", ".xtc", "traj_comp.xtc"), name("o", ".trr", "traj.trr")
    f_px, f_pf = name("px", "_pullx.xvg", "pullx.xvg"), name("pf", "_pullf.xvg", "pullf.xvg")
    try:
        src = tpr + ".mdx" if os.path.isfile(tpr + ".mdx") else tpr
        if not os.path.isfile(src):
            raise InputError(f"File '{tpr}' does not exist or is not accessible")
        sysm, mdpd, meta = tprio.load_run_input(src)
    except (InputError, ValueError, OSError) as e:
        err.write(f"\nProgram:     gmx mdrun ({PROGRAM})\n\nFatal error:\n{e}\n")
        return 1
    m = mdpio.Mdp()
    m.update(mdpd)
    integrator = m.s("integrator").lower()
    dt = m.f("dt")
    nsteps = m.i("nsteps") if a.nsteps is None else a.nsteps
    tinit = m.f("tinit")

    # ---- continuation from a checkpoint
    step0 = 0
    run_state0 = None
    appending = False
    cpi = a.cpi
    if cpi == "":
        cpi = f_cpt
    if cpi is not None and os.path.isfile(cpi):
        cp = tprio.load_checkpoint(cpi)
        if len(cp["x"]) != sysm.natoms:
            err.write(f"Fatal error:\ncheckpoint {cpi} has {len(cp['x'])} atoms, the run input {sysm.natoms}\n")
            return 1
        # restraint references are fractional in the run input's box (`refcoord_scaling`): carry them into the
        # checkpoint's box the way pressure coupling has moved them since
        mode = m.s("refcoord_scaling").lower()
        if meta.get("use_posres") and len(sysm.posres_idx) and mode != "no":
            carry = np.linalg.solve(np.asarray(sysm.box, float), np.asarray(cp["box"], float))
            ref = np.asarray(sysm.posres_ref, float)
            sysm.posres_ref = ref @ carry if mode == "all" else ref + (ref.mean(0) @ carry - ref.mean(0))
        sysm.x, sysm.v, sysm.box = cp["x"], cp["v"], cp["box"]
        step0 = int(cp["step"])
        run_state0 = cp.get("extra", {}).get("run_state")
        appending = a.append
    elif cpi is not None:
        out.write(f"NOTE: checkpoint {cpi} not found, starting from the run input\n")

    # ---- engine
    from .engine import Engine, MdxError, E_NAMES
    device = 0
    if a.gpu_id:
        try:
            device = int(str(a.gpu_id)[0])
        except ValueError:
            device = 0
    rc_ = m.f("rcoulomb")
    nstcalc = _gcd_pos(m.i("nstenergy"), m.i("nstlog"), m.i("nstcalcenergy")) or 100
    nsttc, nstpc = mdpio.effective_coupling_steps(m)
    tcoupl = m.s("tcoupl").lower()
    pcoupl = {"no": 0, "c-rescale": 1, "berendsen": 1, "parrinello-rahman": 2}[m.s("pcoupl").lower()]
    if integrator != "md":
        tcoupl, pcoupl = "no", 0
    tc_groups = m.words("tc_grps") if tcoupl != "no" else None
    # Like mdrun on a GPU, dynamics treats the file's nstlist as a lower bound (`-nstlist` overrides): the outer pair list
    # lives max(nstlist, 100) steps at the radius the buffer estimate gives for that time, and every 10 steps the engine
    # prunes it to the pairs within the radius of an 11-step list (the dual list of `verlet-buffer-tolerance`,
    # /root/reference/mdp/npt_complex.mdp:29-32). A cell too small for the outer radius gets one list of 20 steps.
    plans = [(max(1, min(m.i("nstlist"), 20)), 0)]
    if integrator == "md":
        plans = [(int(a.nstlist), 10 if int(a.nstlist) >= 40 else 0)] if a.nstlist else [(max(m.i("nstlist"), 100), 10), (20, 0)]
    eng = None
    for k, (nstlist, nstprune) in enumerate(plans):
        try:
            eng = Engine(
                sysm, rc=rc_, nstlist=nstlist, nstprune=nstprune, ewald_rtol=m.f("ewald_rtol"), pme=True, fourierspacing=m.f("fourierspacing"),
                dt=dt, nstcalcenergy=nstcalc, nstcomm=(m.i("nstcomm") if m.s("comm_mode").lower() != "none" else 0),
                device=device, tc_groups=tc_groups, ref_t=m.floats("ref_t") or None, tau_t=m.floats("tau_t") or None,
                nsttcouple=nsttc, seed=_seed(m.i("ld_seed")), pcoupl=pcoupl, tau_p=m.f("tau_p"),
                ref_p=(m.floats("ref_p") or [1.0])[0], compressibility=(m.floats("compressibility") or [4.5e-5])[0],
                nstpcouple=nstpc, dispcorr={"no": 0, "ener": 1, "enerpres": 2}[m.s("dispcorr").lower()],
                lincs_order=m.i("lincs_order"), lincs_iter=m.i("lincs_iter"),
                refcoord_scaling={"no": 0, "all": 1, "com": 2}[m.s("refcoord_scaling").lower()],
                use_posres=bool(meta.get("use_posres")),
                rlist=(max(m.f("rlist"), rc_) if m.f("verlet_buffer_tolerance") <= 0 else None),
                verlet_buffer_tolerance=m.f("verlet_buffer_tolerance"))
            break
        except (MdxError, OSError) as e:
            if k + 1 < len(plans) and "shorter than twice" in str(e):
                continue
            err.write(f"\nProgram:     gmx mdrun ({PROGRAM})\n\nFatal error:\n{e}\n")
            return 1
    total_mass = float(np.sum(sysm.mass))

    # ---- log header
    if not appending:
        for p in (f_log, f_edr, f_gro):
            _backup(p)
    log = enio.LogWriter(f_log, append=appending)
    log.header([f"{PROGRAM}", f"Log file opened on {time.strftime('%a %b %d %H:%M:%S %Y')}",
                f"Command line: gmx mdrun {' '.join(argv)}", f"Run input: {src}", f"System: {sysm.title}",
                f"Atoms: {sysm.natoms}   rigid waters: {len(sysm.settles)}   constraints: {len(sysm.constraints)}",
                f"PME grid: {eng.grid}   rc: {rc_}   pair-list radius: {eng.rlist:.3f}   nstlist: {nstlist}   "
                + (f"pruned every {eng.nstprune} steps to radius {eng.rlist_inner:.3f}   " if eng.nstprune else "")
                + f"Ewald beta: {eng.beta:.5f}",
                f"integrator: {integrator}   dt: {dt}   nsteps: {nsteps}   nstcalcenergy: {nstcalc}",
                "Input Parameters:"] + ["   %-28s = %s" % (k.replace("_", "-"), v) for k, v in sorted(m.items())])

    stop = {"flag": False, "sig": None}

    def on_signal(signum, _frame):
        stop["flag"] = True
        stop["sig"] = signum

    old_handlers = {}
    if threading.current_thread() is threading.main_thread():
        for sg in (signal.SIGTERM, signal.SIGINT):
            old_handlers[sg] = signal.signal(sg, on_signal)

    def final_files(step_now, with_cpt=True):
        x, v, box = eng.get_state()
        groio.write_system_gro(f_gro, sysm, x=x, v=v, box=box, title=f"{sysm.title}, t= {tinit + step_now * dt:.5f} step= {step_now}")
        if with_cpt:
            tprio.save_checkpoint(f_cpt, step_now, tinit + step_now * dt, x, v, box, extra=dict(run_state=[float(r) for r in eng.get_run_state()]))

    rc = 0
    try:
        if integrator == "steep":
            rc = _run_minimizer(eng, m, nsteps, a, out, log, f_trr, f_edr, sysm, total_mass, stop)
            final_files(0, with_cpt=False)
            log.note("\nFinished minimisation")
            log.finish(0, 0.0, time.time() - t_start, 0)
            return rc
        # ---- dynamics start-up
        if step0 == 0:
            if m.b("gen_vel"):
                eng.generate_velocities(m.f("gen_temp"), _seed(m.i("gen_seed")))
                log.note(f"Generated velocities at {m.f('gen_temp')} K")
            if not m.b("continuation"):
                eng.constrain_start()
                log.note("Constrained the start configuration and velocities")
        else:
            eng.set_step(step0)
        # ---- pull
        pull = m.b("pull")
        px = pf = None
        nst_px = nst_pf = 0
        if pull:
            g1, g2 = sysm.groups[m.s("pull_group1_name")], sysm.groups[m.s("pull_group2_name")]
            dims = [1 if w.upper().startswith("Y") else 0 for w in (m.words("pull_coord1_dim") or ["Y", "Y", "Y"])]
            nst_px, nst_pf = m.i("pull_nstxout"), m.i("pull_nstfout")
            nst_pull = _gcd_pos(nst_px, nst_pf) or 50
            pb1 = m.i("pull_group1_pbcatom", 0) or 0
            pb2 = m.i("pull_group2_pbcatom", 0) or 0

            def local_pbc(group, atom1):
                if atom1 <= 0:
                    return -1
                hit = np.nonzero(np.asarray(group) == atom1 - 1)[0]
                return int(hit[0]) if len(hit) else -1

            eng.set_pull(g1, g2, type=str(m.s("pull_coord1_type", "umbrella")).lower().replace("_", "-"),
                         k=m.f("pull_coord1_k", 0.0), rate=m.f("pull_coord1_rate", 0.0) or 0.0,
                         init=m.f("pull_coord1_init", 0.0) or 0.0, start=m.b("pull_coord1_start"), dim=dims,
                         nstout=nst_pull, pbcatom1=local_pbc(g1, pb1), pbcatom2=local_pbc(g2, pb2))
            if step0:
                eng.set_step(step0)
            if not appending:
                _backup(f_px); _backup(f_pf)
            px = xvgio.pullx_writer(f_px, append=appending) if nst_px > 0 else None
            pf = xvgio.pullf_writer(f_pf, append=appending) if nst_pf > 0 else None
        if step0 and run_state0 is not None:
            # thermostat factors, barostat box velocity, pull reference and work of the interrupted run
            eng.set_run_state(np.asarray(run_state0, dtype=np.float64))
            log.note("Restored the coupling and pull state from the checkpoint")
        # ---- trajectory output
        nst_xtc, nst_x, nst_v = m.i("nstxout_compressed"), m.i("nstxout"), m.i("nstvout")
        if m.i("nstfout"):
            log.note("NOTE: nstfout is not written by this engine (forces are not kept per frame)")
        xtc = trr = None
        sel = None
        if nst_xtc > 0:
            grp = m.words("compressed_x_grps")
            if grp and len(sysm.groups[grp[0]]) != sysm.natoms:
                sel = np.asarray(sysm.groups[grp[0]])
            from .io import traj
            if not appending:
                _backup(f_xtc)
            # `-ntomp` host threads are what the pipeline grants mdrun (/root/reference/Snakefile:232-233): up to four of them
            # share the bit packing of one frame (a whole-system frame of the production cadence is due every 1.7 ms and
            # takes one core about 3 ms), further ones pack further frames at the same time
            try:
                ntomp = max(1, min(int(a.ntomp), 64)) if a.ntomp else 4
            except ValueError:
                ntomp = 4
            xtc = traj.XtcWriter(f_xtc, sysm.natoms if sel is None else len(sel), precision=m.f("compressed_x_precision"),
                                 append=appending, threads=min(ntomp, 4))
        if nst_x > 0 or nst_v > 0:
            from .io import traj
            if not appending:
                _backup(f_trr)
            trr = traj.TrrWriter(f_trr, sysm.natoms, append=appending)
        nslots = 8
        writer = None
        if xtc is not None or trr is not None:
            eng.frame_ring(nslots)
            from .output import FramePipeline
            writer = FramePipeline(eng, nslots, xtc, trr, sel, dt, tinit, nthreads=max(1, min(ntomp // 4, nslots)) if xtc is not None else 1)
        if not appending:
            _backup(f_edr)
        edr = enio.EdrWriter(f_edr, append=appending, dt_ps=dt)
        nst_e, nst_log = m.i("nstenergy"), m.i("nstlog")

        def due(n, s):
            return n > 0 and s % n == 0

        def maybe_capture(s):
            if appending and s == step0:
                return          # the part that ended here already wrote this step's frame
            wx, tx, tv = due(nst_xtc, s) and xtc is not None, due(nst_x, s), due(nst_v, s)
            if writer is not None and (wx or tx or tv):
                writer.capture(wx, tx, tv)

        def fetch():
            """Energy frames and pull records recorded since the last call (synchronises with the device)."""
            return eng.pop_energy_frames(), (eng.pop_pull_records() if pull else [])

        def write(fetched):
            frames, records = fetched
            for fr in frames:
                s = int(fr[0])
                if appending and s == step0:
                    continue
                e = {n: fr[2 + i] for i, n in enumerate(E_NAMES)}
                terms = enio.frame_terms(e, total_mass)
                if due(nst_e, s):
                    edr.frame(s, tinit + s * dt, terms)
                if due(nst_log, s):
                    log.energies(s, tinit + s * dt, terms)
            if pull:
                for (t, val, force, _work) in records:
                    s = int(round(t / dt))
                    if px is not None and due(nst_px, s):
                        px.row(tinit + t, val)
                    if pf is not None and due(nst_pf, s):
                        pf.row(tinit + t, force)
                if px is not None:
                    px.flush()
                if pf is not None:
                    pf.flush()

        cadences = [c for c in (nst_xtc if xtc is not None else 0, nst_x, nst_v) if c > 0]
        step = step0
        chunk_cap = 250
        last_print = last_cpt = last_fetch = time.time()
        t_loop0 = time.time()
        steps_run = 0
        fetched = None
        prof = {"capture": 0.0, "step": 0.0, "write": 0.0, "fetch": 0.0}     # where this thread's time goes (MDX_RUNNER_TIMES=1 prints it)
        out.write(f"starting mdrun '{sysm.title}'\n{nsteps} steps, {nsteps * dt:8.1f} ps"
                  + (f" (continuing from step {step0}, {step0 * dt:8.1f} ps).\n" if step0 else ".\n"))
        out.flush()
        reset_pending = bool(a.resethway) and nsteps - step0 >= 2
        while step < nsteps and not stop["flag"]:
            if reset_pending and step - step0 >= (nsteps - step0) // 2:
                # `-resethway`: the performance figure of the log covers the second half of the run only
                eng.synchronize()
                if writer is not None:
                    writer.wait_idle()
                t_loop0 = time.time(); steps_run = 0; reset_pending = False
                prof = dict.fromkeys(prof, 0.0)
                log.note(f"\nStep {step}: resetting all time and cycle counters")
            tp0 = time.perf_counter()
            maybe_capture(step)
            tp1 = time.perf_counter()
            nxt = min(nsteps, step + chunk_cap)
            for c in cadences:
                nxt = min(nxt, (step // c + 1) * c)
            eng.step(nxt - step)
            tp2 = time.perf_counter()
            prof["capture"] += tp1 - tp0; prof["step"] += tp2 - tp1
            steps_run += nxt - step
            step = nxt
            if fetched is not None:
                # what the last fetch brought is written while the device works on the steps enqueued since (a second of
                # the production cadence is 650 energy frames for .edr and .log: 20 ms of this thread)
                write(fetched); fetched = None
                prof["write"] += time.perf_counter() - tp2
            now = time.time()
            if now - last_fetch >= 0.2 and step < nsteps:
                tp3 = time.perf_counter()
                # (synchronises: keeps the host at most a fifth of a second of steps ahead of the device, and what is left to
                # write behind the last step short)
                fetched = fetch(); last_fetch = now
                prof["fetch"] += time.perf_counter() - tp3
                if writer is not None and writer.error is not None:
                    raise writer.error
            if now - last_print >= 1.0 or step >= nsteps:
                if a.v or True:
                    rate = steps_run / max(now - t_loop0, 1e-9)
                    remain = (nsteps - step) / max(rate, 1e-9)
                    out.write(f"step {step}, remaining wall clock time: {remain:5.0f} s          \n")
                    out.flush()
                last_print = now
                if a.cpt > 0 and now - last_cpt >= a.cpt * 60.0:
                    x_, v_, b_ = eng.get_state()
                    tprio.save_checkpoint(f_cpt, step, tinit + step * dt, x_, v_, b_, extra=dict(run_state=[float(r) for r in eng.get_run_state()]))
                    last_cpt = now
        # the frame of the final step
        t_tail = time.perf_counter()
        if not stop["flag"]:
            maybe_capture(step)
        eng.synchronize()
        prof["tail: last steps on the device"] = time.perf_counter() - t_tail
        if fetched is not None:
            write(fetched)
        write(fetch())
        if not stop["flag"] and (due(nst_e, step) or due(nst_log, step)) and step > step0:
            # potential terms at the final coordinates (mdrun evaluates one more force call than steps)
            _, e = eng.compute()
            terms = enio.frame_terms(e, total_mass)
            if due(nst_e, step):
                edr.frame(step, tinit + step * dt, terms)
            if due(nst_log, step):
                log.energies(step, tinit + step * dt, terms)
        if writer is not None:
            writer.finish()
            if writer.error is not None:
                raise writer.error
        wall = time.time() - t_loop0
        prof["tail: all of it"] = time.perf_counter() - t_tail
        if os.environ.get("MDX_RUNNER_TIMES"):
            err.write("mdrun loop, seconds of the stepping thread: " + ", ".join("%s %.4f" % kv for kv in prof.items()) + "; wall %.4f\n" % wall)
        final_files(step)
        for w in (xtc, trr, px, pf, edr):
            if w is not None:
                w.close()
        if stop["flag"]:
            msg = f"\nReceived the {signal.Signals(stop['sig']).name} signal, stopped at step {step}\n"
            out.write(msg); log.note(msg)
        log.finish(steps_run, dt, wall, step)
        out.write(f"\nWriting final coordinates.\n" + enio.performance_block(steps_run, dt, wall) + "\n")
        out.flush()
        return 0
    except (InputError, KeyError, ValueError, OSError) as e:
        err.write(f"\nProgram:     gmx mdrun ({PROGRAM})\n\nFatal error:\n{e}\n")
        return 1
    except Exception as e:      # engine errors: try to leave the last good state behind
        err.write(f"\nProgram:     gmx mdrun ({PROGRAM})\n\nFatal error:\n{type(e).__name__}: {e}\n")
        return 1
    finally:
        log.close()
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
        eng.close()


def _run_minimizer(eng, m, nsteps, a, out, log, f_trr, f_edr, sysm, total_mass, stop) -> int:
    emtol, emstep = m.f("emtol", 10.0) or 10.0, m.f("emstep", 0.01) or 0.01
    out.write(f"\nSteepest Descents:\n   Tolerance (Fmax)   = {emtol:12.5e}\n   Number of steps    = {nsteps:12d}\n")
    last = {"t": 0.0}

    def progress(step, epot, fmax, stepsize):
        now = time.time()
        if a.v and (now - last["t"] > 0.5 or step == 0):
            out.write(f"Step={step:5d}, Dmax= {stepsize:.1e} nm, Epot= {epot:12.5e} Fmax= {fmax:11.5e}\n")
            out.flush()
            last["t"] = now

    res = eng.minimize(nsteps, emtol=emtol, emstep=emstep, progress=progress)
    e = eng.energies()
    terms = enio.frame_terms(e, total_mass)
    log.energies(res["steps"], float(res["steps"]), terms)
    edr = enio.EdrWriter(f_edr)
    edr.frame(res["steps"], float(res["steps"]), terms)
    edr.close()
    if res["status"] == 1:
        head = f"Steepest Descents converged to Fmax < {emtol:g} in {res['steps'] + 1} steps"
    elif res["status"] == 2:
        head = (f"Steepest Descents converged to machine precision in {res['steps'] + 1} steps,\n"
                f"but did not reach the requested Fmax < {emtol:g}.")
    else:
        head = f"Steepest Descents did not converge to Fmax < {emtol:g} in {res['steps'] + 1} steps."
    msg = (f"\n{head}\nPotential Energy  = {res['epot']:14.7e}\nMaximum force     = {res['fmax']:14.7e}\n"
           f"(started from Epot = {res['epot0']:.7e}, Fmax = {res['fmax0']:.7e})\n")
    out.write(msg); out.flush()
    log.note(msg)
    return 0
