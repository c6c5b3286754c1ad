"""`.top` / `.itp` front-end: preprocessor, force-field tables, molecule expansion -> MDSystem.

The dialect is what the pipeline feeds to `gmx grompp -p` (/root/reference/Snakefile:705,738,994):
a pdb2gmx topology for amber99sb-ildn + tip3p that `#include`s the force field, carries the
protein `[ moleculetype ]` inline, has `[ atomtypes ]` injected for the ligand
(/root/reference/scripts/create_complex_topology.py:65-74), includes the acpype ligand `.itp`
with its `#ifdef POSRES` block (/root/reference/scripts/fix_ligand_topology.py:88-96) and ends in
`[ system ]` / `[ molecules ]` with `MOL 1`, `SOL n`, `NA n` rows
(/root/reference/scripts/create_complex_topology.py:94-122).

Supported: `#include/#define/#undef/#ifdef/#ifndef/#else/#endif`, macro substitution, `\\`
continuation, `[ defaults ]` (combination rule 2 only: per-atom sigma/epsilon with
Lorentz-Berthelot mixing is what the device kernels implement), `[ atomtypes ]` in all column
layouts, `[ bondtypes | constrainttypes | angletypes | dihedraltypes | pairtypes ]` with
wildcard `X` and multi-line type-9 entries, and per molecule `[ atoms | bonds | pairs | angles |
dihedrals | position_restraints | settles | exclusions | constraints ]`.
Function types: bonds 1, angles 1, dihedrals 1/4/9 (periodic) and 3 (Ryckaert-Bellemans,
rewritten as a cosine series), pairs 1, constraints 1/2, settles 1, position restraints 1.
"""
from __future__ import annotations

import math
import os
import numpy as np

from ..system import MDSystem, build_exclusions


class TopologyError(Exception):
    pass


# ------------------------------------------------------------------------------------------
# preprocessor
# ------------------------------------------------------------------------------------------
def _default_include_dirs():
    dirs = []
    for env in ("GMXLIB",):
        for p in os.environ.get(env, "").split(os.pathsep):
            if p:
                dirs.append(p)
    gd = os.environ.get("GMXDATA")
    if gd:
        dirs.append(os.path.join(gd, "top"))
    for p in ("/usr/share/gromacs/top", "/usr/local/gromacs/share/gromacs/top"):
        if os.path.isdir(p):
            dirs.append(p)
    return dirs


class Preprocessor:
    def __init__(self, defines=None, include_dirs=None):
        self.defines = dict(defines or {})
        self.include_dirs = list(include_dirs or []) + _default_include_dirs()
        self.files: list[str] = []

    def _find(self, name, cur_dir):
        cands = [name] if os.path.isabs(name) else [os.path.join(cur_dir, name), name] + [
            os.path.join(d, name) for d in self.include_dirs]
        for c in cands:
            if os.path.isfile(c):
                return c
        raise TopologyError(
            f"include file '{name}' not found (searched {cur_dir}, the working directory and "
            f"{self.include_dirs or 'no library directories'}; set GMXLIB to the force-field tree)")

    def lines(self, path):
        """Yield (file, lineno, text) for every active, comment-free logical line."""
        path = os.path.abspath(path)
        self.files.append(path)
        cur_dir = os.path.dirname(path)
        stack = []          # (this_branch_active, any_branch_taken, parent_active)
        active = True
        pending = ""
        with open(path) as f:
            raw_lines = f.readlines()
        for ln, raw in enumerate(raw_lines, 1):
            line = raw.split(";", 1)[0].rstrip()
            if pending:
                line = pending + " " + line.lstrip()
                pending = ""
            if line.endswith("\\"):
                pending = line[:-1]
                continue
            s = line.strip()
            if not s:
                continue
            if s[0] == "#":
                p = s[1:].strip().split(None, 2)
                d = p[0] if p else ""
                if d in ("ifdef", "ifndef"):
                    if len(p) < 2:
                        raise TopologyError(f"{path}:{ln}: #{d} without a name")
                    cond = (p[1] in self.defines) == (d == "ifdef")
                    stack.append((active and cond, cond, active))
                    active = active and cond
                elif d == "else":
                    if not stack:
                        raise TopologyError(f"{path}:{ln}: #else without #ifdef")
                    _, taken, parent = stack.pop()
                    stack.append((parent and not taken, True, parent))
                    active = parent and not taken
                elif d == "endif":
                    if not stack:
                        raise TopologyError(f"{path}:{ln}: #endif without #ifdef")
                    _, _, parent = stack.pop()
                    active = parent
                elif not active:
                    continue
                elif d == "include":
                    rest = s[1:].strip()[len("include"):].strip()
                    name = rest.strip('"<>')
                    yield from self.lines(self._find(name, cur_dir))
                elif d == "define":
                    if len(p) < 2:
                        raise TopologyError(f"{path}:{ln}: #define without a name")
                    self.defines[p[1]] = p[2].strip() if len(p) > 2 else ""
                elif d == "undef":
                    if len(p) > 1:
                        self.defines.pop(p[1], None)
                else:
                    raise TopologyError(f"{path}:{ln}: unknown directive '#{d}'")
                continue
            if not active:
                continue
            if self.defines and not s.startswith("["):
                toks = s.split()
                if any(t in self.defines and self.defines[t] != "" for t in toks):
                    s = " ".join(self.defines[t] if (t in self.defines and self.defines[t] != "") else t
                                 for t in toks)
            yield path, ln, s
        if stack:
            raise TopologyError(f"{path}: unterminated #ifdef")


# ------------------------------------------------------------------------------------------
# parsed topology
# ------------------------------------------------------------------------------------------
class MoleculeType:
    def __init__(self, name, nrexcl):
        self.name = name
        self.nrexcl = nrexcl
        self.atoms = []          # dicts
        self.sections = {}       # name -> list of (tokens, where)

    def add(self, section, toks, where):
        self.sections.setdefault(section, []).append((toks, where))


class Topology:
    def __init__(self):
        self.defaults = dict(nbfunc=1, comb_rule=2, gen_pairs=False, fudge_lj=1.0, fudge_qq=1.0)
        self.have_defaults = False
        self.atomtypes = {}      # name -> dict(btype, mass, charge, ptype, sigma, eps, atnum)
        self.bondtypes = {}      # (a,b) -> (funct, b0, kb)
        self.constrainttypes = {}
        self.angletypes = {}     # (a,b,c) -> (funct, th0, k)
        self.dihedraltypes = []  # [types4, funct, [param tuples]]
        self.pairtypes = {}      # (a,b) -> (sigma/c6, eps/c12)
        self.nonbond_params = {}
        self.moleculetypes = {}
        self.system = ""
        self.molecules = []      # (name, count)
        self.warnings = []
        self.files = []


_MOL_SECTIONS = {"atoms", "bonds", "pairs", "pairs_nb", "angles", "dihedrals", "position_restraints",
                 "settles", "exclusions", "constraints", "cmap", "virtual_sites2", "virtual_sites3",
                 "virtual_sites4", "virtual_sitesn", "dummies2", "dummies3", "distance_restraints",
                 "dihedral_restraints", "orientation_restraints", "angle_restraints", "polarization"}
_IGNORED_GLOBAL = {"implicit_genborn_params", "cmaptypes"}


def _atomtype_row(toks, where):
    """name [bonded_type] [atomic_number] mass charge ptype V W - columns located from ptype."""
    pi = None
    for i, t in enumerate(toks):
        if len(t) == 1 and t.upper() in "ASVD" and i >= 3:
            pi = i
            break
    if pi is None or len(toks) < pi + 3:
        raise TopologyError(f"{where}: cannot read [ atomtypes ] line '{' '.join(toks)}'")
    name = toks[0]
    mass, charge = float(toks[pi - 2]), float(toks[pi - 1])
    extra = toks[1:pi - 2]
    btype, atnum = name, 0
    if len(extra) == 1:
        try:
            atnum = int(extra[0])
        except ValueError:
            btype = extra[0]
    elif len(extra) >= 2:
        btype = extra[0]
        try:
            atnum = int(extra[1])
        except ValueError:
            pass
    return name, dict(btype=btype, atnum=atnum, mass=mass, charge=charge, ptype=toks[pi].upper(),
                      v=float(toks[pi + 1]), w=float(toks[pi + 2]))


def read_top(path, defines=None, include_dirs=None) -> Topology:
    pp = Preprocessor(defines, include_dirs)
    top = Topology()
    section = None
    mol = None
    for file, ln, s in pp.lines(path):
        where = f"{file}:{ln}"
        if s.startswith("["):
            if "]" not in s:
                raise TopologyError(f"{where}: malformed section header '{s}'")
            section = s[1:s.index("]")].strip().lower()
            continue
        if section is None:
            raise TopologyError(f"{where}: data before the first section header")
        toks = s.split()
        if section == "defaults":
            top.have_defaults = True
            d = top.defaults
            d["nbfunc"] = int(toks[0])
            d["comb_rule"] = int(toks[1])
            if len(toks) > 2:
                d["gen_pairs"] = toks[2].lower() in ("yes", "y", "1", "true")
            if len(toks) > 3:
                d["fudge_lj"] = float(toks[3])
            if len(toks) > 4:
                d["fudge_qq"] = float(toks[4])
        elif section == "atomtypes":
            name, row = _atomtype_row(toks, where)
            if name in top.atomtypes:
                top.warnings.append(f"{where}: atomtype {name} redefined, keeping the last")
            top.atomtypes[name] = row
        elif section == "bondtypes":
            top.bondtypes[(toks[0], toks[1])] = (int(toks[2]),) + tuple(float(t) for t in toks[3:])
        elif section == "constrainttypes":
            top.constrainttypes[(toks[0], toks[1])] = (int(toks[2]),) + tuple(float(t) for t in toks[3:])
        elif section == "angletypes":
            top.angletypes[(toks[0], toks[1], toks[2])] = (int(toks[3]),) + tuple(float(t) for t in toks[4:])
        elif section == "dihedraltypes":
            # four type names, or the two-name (central atoms) form
            try:
                int(toks[2])
                types = ("X", toks[0], toks[1], "X")
                funct = int(toks[2])
                par = tuple(float(t) for t in toks[3:])
            except ValueError:
                types = tuple(toks[0:4])
                funct = int(toks[4])
                par = tuple(float(t) for t in toks[5:])
            last = top.dihedraltypes[-1] if top.dihedraltypes else None
            if funct == 9 and last is not None and last[0] == types and last[1] == 9:
                last[2].append(par)
            else:
                top.dihedraltypes.append([types, funct, [par]])
        elif section == "pairtypes":
            top.pairtypes[(toks[0], toks[1])] = (int(toks[2]),) + tuple(float(t) for t in toks[3:])
        elif section == "nonbond_params":
            top.nonbond_params[(toks[0], toks[1])] = tuple(float(t) for t in toks[3:])
        elif section in _IGNORED_GLOBAL:
            continue
        elif section == "moleculetype":
            mol = MoleculeType(toks[0], int(toks[1]) if len(toks) > 1 else 3)
            if mol.name in top.moleculetypes:
                raise TopologyError(f"{where}: moleculetype {mol.name} is defined twice")
            top.moleculetypes[mol.name] = mol
        elif section == "atoms":
            if mol is None:
                raise TopologyError(f"{where}: [ atoms ] outside a moleculetype")
            if len(toks) < 5:
                raise TopologyError(f"{where}: [ atoms ] line needs at least nr type resnr residue atom")
            a = dict(type=toks[1], resnr=int(toks[2]), resname=toks[3], name=toks[4],
                     cgnr=int(toks[5]) if len(toks) > 5 else len(mol.atoms) + 1,
                     charge=float(toks[6]) if len(toks) > 6 else None,
                     mass=float(toks[7]) if len(toks) > 7 else None)
            if int(toks[0]) != len(mol.atoms) + 1:
                raise TopologyError(f"{where}: atoms of {mol.name} are not numbered consecutively from 1")
            mol.atoms.append(a)
        elif section in _MOL_SECTIONS:
            if mol is None:
                raise TopologyError(f"{where}: [ {section} ] outside a moleculetype")
            mol.add(section, toks, where)
        elif section == "system":
            top.system = (top.system + " " + s).strip()
        elif section == "molecules":
            top.molecules.append((toks[0], int(toks[1])))
        else:
            top.warnings.append(f"{where}: section [ {section} ] is not used and was skipped")
    top.files = pp.files
    if not top.have_defaults:
        raise TopologyError(f"{path}: no [ defaults ] section (is the force field #include missing?)")
    if not top.molecules:
        raise TopologyError(f"{path}: no [ molecules ] section")
    return top


# ------------------------------------------------------------------------------------------
# parameter lookup
# ------------------------------------------------------------------------------------------
def _lookup2(table, a, b):
    return table.get((a, b)) or table.get((b, a))


def _lookup3(table, a, b, c):
    return table.get((a, b, c)) or table.get((c, b, a))


def _match_dihedral(top: Topology, types, funct):
    """Most specific entry wins (fewest wildcards), the first such entry in file order on ties;
    type 9 and type 4 share nothing, type 1 lines also serve type 9 requests."""
    ok_funct = {9: (9, 1), 1: (1, 9), 4: (4,), 2: (2,), 3: (3,)}.get(funct, (funct,))
    best, best_score = None, -1
    for ent in top.dihedraltypes:
        t, f, _ = ent
        if f not in ok_funct:
            continue
        for order in (types, types[::-1]):
            score = 0
            for want, have in zip(t, order):
                if want == "X":
                    continue
                if want != have:
                    score = -1
                    break
                score += 1
            if score > best_score:
                best, best_score = ent, score
    return best


def rb_to_periodic(c):
    """Ryckaert-Bellemans sum_n C_n cos^n(psi), psi = phi - 180 deg, as terms k(1+cos(m phi - phi0)).
    cos^n(phi-180) = (-1)^n cos^n(phi); the power-to-multiple-angle identities give the Fourier
    amplitudes a_m; each becomes |a_m|(1+cos(m phi - phi0)) and the constants are collected in
    an m = 0 term so that the energy, not only the force, is reproduced."""
    c = list(c) + [0.0] * (6 - len(c))
    s = [c[n] * (-1) ** n for n in range(6)]            # coefficients of cos^n(phi)
    a = [0.0] * 6
    a[0] = s[0] + s[2] / 2 + 3 * s[4] / 8
    a[1] = s[1] + 3 * s[3] / 4 + 10 * s[5] / 16
    a[2] = s[2] / 2 + s[4] / 2
    a[3] = s[3] / 4 + 5 * s[5] / 16
    a[4] = s[4] / 8
    a[5] = s[5] / 16
    terms = []
    const = a[0]
    for m in range(1, 6):
        if a[m] == 0.0:
            continue
        k = abs(a[m])
        phi0 = 0.0 if a[m] > 0 else 180.0
        terms.append((phi0, k, m))
        const -= k
    if const != 0.0:
        terms.append((0.0, 0.5 * const, 0))
    return terms


# ------------------------------------------------------------------------------------------
# per-moleculetype compilation
# ------------------------------------------------------------------------------------------
class _Compiled:
    pass


def _compile(top: Topology, mol: MoleculeType, constraints_mode: str, use_posres: bool) -> _Compiled:
    c = _Compiled()
    na = len(mol.atoms)
    if na == 0:
        raise TopologyError(f"moleculetype {mol.name} has no atoms")
    c.n = na
    tnames = [a["type"] for a in mol.atoms]
    for t in tnames:
        if t not in top.atomtypes:
            raise TopologyError(f"moleculetype {mol.name}: atomtype {t} not found")
    at = [top.atomtypes[t] for t in tnames]
    bt = [a["btype"] for a in at]
    c.mass = np.array([a["mass"] if a["mass"] is not None else t["mass"] for a, t in zip(mol.atoms, at)])
    c.charge = np.array([a["charge"] if a["charge"] is not None else t["charge"] for a, t in zip(mol.atoms, at)])
    c.sigma = np.array([t["v"] for t in at])
    c.eps = np.array([t["w"] for t in at])
    c.resnr = np.array([a["resnr"] for a in mol.atoms])
    c.resname = [a["resname"] for a in mol.atoms]
    c.atomname = [a["name"] for a in mol.atoms]
    c.types = tnames
    is_h = np.array([n[:1].upper() == "H" for n in c.atomname])

    def idx(tok, where):
        i = int(tok) - 1
        if not 0 <= i < na:
            raise TopologyError(f"{where}: atom index {tok} outside moleculetype {mol.name} (1..{na})")
        return i

    bonds, b0, kb, cons, clen, edges = [], [], [], [], [], []
    for toks, where in mol.sections.get("bonds", []):
        i, j = idx(toks[0], where), idx(toks[1], where)
        funct = int(toks[2]) if len(toks) > 2 else 1
        if funct == 5:          # connection only
            edges.append((i, j))
            continue
        if funct != 1:
            raise TopologyError(f"{where}: bond function {funct} is not supported (harmonic, 1, only)")
        if len(toks) >= 5:
            r0, k = float(toks[3]), float(toks[4])
        else:
            p = _lookup2(top.bondtypes, bt[i], bt[j])
            if p is None:
                raise TopologyError(f"{where}: no default bond type for {bt[i]}-{bt[j]}")
            r0, k = p[1], p[2]
        edges.append((i, j))
        as_constraint = constraints_mode in ("all-bonds", "h-angles", "all-angles") or (
            constraints_mode == "h-bonds" and (is_h[i] or is_h[j]))
        if as_constraint:
            ct = _lookup2(top.constrainttypes, bt[i], bt[j])
            cons.append((i, j)); clen.append(ct[1] if ct else r0)
        else:
            bonds.append((i, j)); b0.append(r0); kb.append(k)
    for toks, where in mol.sections.get("constraints", []):
        i, j = idx(toks[0], where), idx(toks[1], where)
        funct = int(toks[2]) if len(toks) > 2 else 1
        if len(toks) >= 4:
            r0 = float(toks[3])
        else:
            ct = _lookup2(top.constrainttypes, bt[i], bt[j])
            if ct is None:
                raise TopologyError(f"{where}: no constraint length for {bt[i]}-{bt[j]}")
            r0 = ct[1]
        cons.append((i, j)); clen.append(r0)
        if funct == 1:
            edges.append((i, j))
    c.bonds = np.array(bonds, dtype=np.int32).reshape(-1, 2)
    c.bond_b0, c.bond_kb = np.array(b0), np.array(kb)
    c.constraints = np.array(cons, dtype=np.int32).reshape(-1, 2)
    c.constraint_len = np.array(clen)

    settles = []
    c.settle_doh = c.settle_dhh = None
    for toks, where in mol.sections.get("settles", []):
        o = idx(toks[0], where)
        if o + 2 >= na:
            raise TopologyError(f"{where}: settle needs the two hydrogens to follow the oxygen")
        settles.append((o, o + 1, o + 2))
        c.settle_doh, c.settle_dhh = float(toks[2]), float(toks[3])
        edges += [(o, o + 1), (o, o + 2), (o + 1, o + 2)]
    c.settles = np.array(settles, dtype=np.int32).reshape(-1, 3)

    ang, th0, ka = [], [], []
    for toks, where in mol.sections.get("angles", []):
        i, j, k_ = idx(toks[0], where), idx(toks[1], where), idx(toks[2], where)
        funct = int(toks[3]) if len(toks) > 3 else 1
        if funct not in (1, 5):
            raise TopologyError(f"{where}: angle function {funct} is not supported (harmonic, 1, only)")
        if len(toks) >= 6:
            t0, kk = float(toks[4]), float(toks[5])
        else:
            p = _lookup3(top.angletypes, bt[i], bt[j], bt[k_])
            if p is None:
                raise TopologyError(f"{where}: no default angle type for {bt[i]}-{bt[j]}-{bt[k_]}")
            t0, kk = p[1], p[2]
        if funct == 5 and ((len(toks) >= 8 and float(toks[7]) != 0.0)):
            raise TopologyError(f"{where}: Urey-Bradley term with non-zero force constant is not supported")
        ang.append((i, j, k_)); th0.append(t0); ka.append(kk)
    c.angles = np.array(ang, dtype=np.int32).reshape(-1, 3)
    c.angle_th0, c.angle_k = np.array(th0), np.array(ka)

    dih, ph, dk, dm = [], [], [], []
    for toks, where in mol.sections.get("dihedrals", []):
        ijkl = tuple(idx(t, where) for t in toks[0:4])
        funct = int(toks[4]) if len(toks) > 4 else 1
        par = toks[5:]
        if funct in (1, 4, 9):
            if len(par) >= 3:
                sets = [(float(par[0]), float(par[1]), int(float(par[2])))]
            else:
                ent = _match_dihedral(top, tuple(bt[a] for a in ijkl), funct)
                if ent is None:
                    raise TopologyError(f"{where}: no default dihedral type for "
                                        f"{'-'.join(bt[a] for a in ijkl)} (function {funct})")
                sets = [(p[0], p[1], int(p[2])) for p in ent[2]]
            for (p0, kk, mm) in sets:
                dih.append(ijkl); ph.append(p0); dk.append(kk); dm.append(mm)
        elif funct == 3:
            if len(par) >= 6:
                cs = [float(t) for t in par[:6]]
            else:
                ent = _match_dihedral(top, tuple(bt[a] for a in ijkl), 3)
                if ent is None:
                    raise TopologyError(f"{where}: no default RB dihedral for {'-'.join(bt[a] for a in ijkl)}")
                cs = list(ent[2][0])
            for (p0, kk, mm) in rb_to_periodic(cs):
                dih.append(ijkl); ph.append(p0); dk.append(kk); dm.append(mm)
        else:
            raise TopologyError(f"{where}: dihedral function {funct} is not supported (1, 3, 4, 9)")
    c.dihedrals = np.array(dih, dtype=np.int32).reshape(-1, 4)
    c.dih_phi0, c.dih_k, c.dih_mult = np.array(ph), np.array(dk), np.array(dm, dtype=np.int32)

    d = top.defaults
    prs, c6, c12, qq = [], [], [], []
    for toks, where in mol.sections.get("pairs", []):
        i, j = idx(toks[0], where), idx(toks[1], where)
        funct = int(toks[2]) if len(toks) > 2 else 1
        if funct != 1:
            raise TopologyError(f"{where}: pair function {funct} is not supported (1 only)")
        if len(toks) >= 5:
            s_ij, e_ij = float(toks[3]), float(toks[4])
        else:
            p = _lookup2(top.pairtypes, tnames[i], tnames[j]) or _lookup2(top.pairtypes, bt[i], bt[j])
            if p is not None:
                s_ij, e_ij = p[1], p[2]
            elif d["gen_pairs"]:
                s_ij = 0.5 * (c.sigma[i] + c.sigma[j])
                e_ij = d["fudge_lj"] * math.sqrt(c.eps[i] * c.eps[j])
            else:
                raise TopologyError(f"{where}: no pair type for {tnames[i]}-{tnames[j]} and gen-pairs is off")
        prs.append((i, j))
        c6.append(4.0 * e_ij * s_ij ** 6); c12.append(4.0 * e_ij * s_ij ** 12)
        qq.append(d["fudge_qq"] * c.charge[i] * c.charge[j])
    c.pairs14 = np.array(prs, dtype=np.int32).reshape(-1, 2)
    c.pair_c6, c.pair_c12, c.pair_qq = np.array(c6), np.array(c12), np.array(qq)

    pidx, pk = [], []
    if use_posres:
        for toks, where in mol.sections.get("position_restraints", []):
            i = idx(toks[0], where)
            funct = int(toks[1]) if len(toks) > 1 else 1
            if funct != 1:
                raise TopologyError(f"{where}: position restraint function {funct} is not supported")
            pidx.append(i); pk.append([float(t) for t in toks[2:5]])
    c.posres_idx = np.array(pidx, dtype=np.int32)
    c.posres_k = np.array(pk, dtype=np.float64).reshape(-1, 3)

    extra = []
    for toks, where in mol.sections.get("exclusions", []):
        i = idx(toks[0], where)
        for t in toks[1:]:
            extra.append((i, idx(t, where)))
    for sec in ("cmap", "virtual_sites2", "virtual_sites3", "virtual_sites4", "virtual_sitesn",
                "dummies2", "dummies3", "polarization"):
        if mol.sections.get(sec):
            raise TopologyError(f"moleculetype {mol.name}: [ {sec} ] is not supported")
    c.edges = np.array(edges, dtype=np.int32).reshape(-1, 2)
    c.excl_start, c.excl_list = build_exclusions(
        na, c.edges, np.full(na, mol.nrexcl), np.array(extra).reshape(-1, 2) if extra else None)
    # BFS tree used to make each instance whole
    nbr = [[] for _ in range(na)]
    for a, b in c.edges:
        nbr[a].append(int(b)); nbr[b].append(int(a))
    seen = np.zeros(na, bool)
    tree = []
    for s0 in range(na):
        if seen[s0]:
            continue
        seen[s0] = True
        queue = [s0]
        while queue:
            a = queue.pop(0)
            for b in nbr[a]:
                if not seen[b]:
                    seen[b] = True
                    tree.append((b, a))
                    queue.append(b)
    c.tree = tree
    return c


def _unwrap(d, box):
    """Shift difference vectors (.., 3) to their nearest image in a lower-triangular cell."""
    for k in (2, 1, 0):
        s = np.rint(d[..., k] / box[k, k])
        d -= s[..., None] * box[k]
    return d


# ------------------------------------------------------------------------------------------
# expansion
# ------------------------------------------------------------------------------------------
def build_system(top: Topology, gro, constraints: str = "none", posres_ref=None,
                 use_posres: bool | None = None, groups=None, make_whole: bool = True) -> MDSystem:
    """Expand `[ molecules ]` over the coordinate file's atoms.

    gro        GroFrame with the coordinates (and velocities) in topology order
    posres_ref GroFrame or (N,3) array of restraint reference positions (`grompp -r`)
    groups     IndexGroups / dict to attach; default groups from residue names otherwise
    """
    d = top.defaults
    if d["nbfunc"] != 1:
        raise TopologyError("only nbfunc 1 (Lennard-Jones) is supported")
    if d["comb_rule"] != 2:
        raise TopologyError(f"combination rule {d['comb_rule']} is not supported: the nonbonded kernels "
                            "mix per-atom sigma/epsilon by Lorentz-Berthelot (rule 2, AMBER/GAFF)")
    if top.nonbond_params:
        raise TopologyError("[ nonbond_params ] overrides are not supported")
    constraints = constraints.lower().replace("_", "-")
    if constraints not in ("none", "h-bonds", "all-bonds"):
        raise TopologyError(f"constraints = {constraints} is not supported (none, h-bonds, all-bonds)")
    if use_posres is None:
        use_posres = posres_ref is not None
    compiled = {}
    total = 0
    for name, count in top.molecules:
        if name not in top.moleculetypes:
            raise TopologyError(f"[ molecules ] names '{name}' but no such moleculetype is defined")
        if name not in compiled:
            compiled[name] = _compile(top, top.moleculetypes[name], constraints, use_posres)
        total += compiled[name].n * count
    if total != gro.natoms:
        raise TopologyError(f"number of coordinates in the coordinate file ({gro.natoms}) does not match "
                            f"the topology ({total})")
    f = {k: [] for k in ("mass", "charge", "sigma", "eps", "molid", "bonds", "bond_b0", "bond_kb", "angles",
                         "angle_th0", "angle_k", "dihedrals", "dih_phi0", "dih_k", "dih_mult", "pairs14",
                         "pair_c6", "pair_c12", "pair_qq", "posres_idx", "posres_k", "settles",
                         "constraints", "constraint_len", "excl_counts", "excl_list")}
    x = np.array(gro.x, dtype=np.float64)
    x_read = x.copy()
    box = np.array(gro.box, dtype=np.float64)
    if abs(box[0, 1]) + abs(box[0, 2]) + abs(box[1, 2]) > 1e-9:
        raise TopologyError("the box must have v1(y) = v1(z) = v2(z) = 0")
    off, nmol = 0, 0
    doh = dhh = None
    mismatch = 0
    for name, count in top.molecules:
        if count == 0:
            continue
        c = compiled[name]
        offs = off + c.n * np.arange(count, dtype=np.int64)

        def tile_idx(a):
            return (a[None, :, :] + offs[:, None, None]).reshape(-1, a.shape[1]).astype(np.int32)

        for key in ("mass", "charge", "sigma", "eps"):
            f[key].append(np.tile(getattr(c, key), count))
        f["molid"].append(np.repeat(nmol + np.arange(count), c.n))
        for key, pars in (("bonds", ("bond_b0", "bond_kb")), ("angles", ("angle_th0", "angle_k")),
                          ("dihedrals", ("dih_phi0", "dih_k", "dih_mult")),
                          ("pairs14", ("pair_c6", "pair_c12", "pair_qq")),
                          ("constraints", ("constraint_len",)), ("settles", ())):
            a = getattr(c, key)
            if len(a):
                f[key].append(tile_idx(a))
                for p in pars:
                    f[p].append(np.tile(getattr(c, p), count))
        if len(c.posres_idx):
            f["posres_idx"].append((c.posres_idx[None, :] + offs[:, None]).reshape(-1).astype(np.int32))
            f["posres_k"].append(np.tile(c.posres_k, (count, 1)))
        if len(c.settles):
            if doh is not None and (abs(doh - c.settle_doh) > 1e-9 or abs(dhh - c.settle_dhh) > 1e-9):
                raise TopologyError("more than one SETTLE geometry in the system is not supported")
            doh, dhh = c.settle_doh, c.settle_dhh
        f["excl_counts"].append(np.tile(np.diff(c.excl_start), count))
        if len(c.excl_list):
            f["excl_list"].append((c.excl_list[None, :] + offs[:, None]).reshape(-1).astype(np.int32))
        if make_whole and c.tree:
            xm = x[off:off + c.n * count].reshape(count, c.n, 3)
            for child, parent in c.tree:
                dvec = xm[:, child] - xm[:, parent]
                xm[:, child] = xm[:, parent] + _unwrap(dvec, box)
        if gro.atomname is not None:
            first = gro.atomname[off:off + c.n]
            mismatch += sum(1 for a, b in zip(first, c.atomname) if a != b)
        off += c.n * count
        nmol += count
    if mismatch:
        top.warnings.append(f"{mismatch} atom names in the coordinate file differ from the topology "
                            "(coordinate-file names are ignored)")

    def cat(key, shape, dt):
        return (np.concatenate(f[key]).astype(dt) if f[key] else np.zeros(shape, dt))

    n = total
    sysm = MDSystem(x=x, box=box, mass=cat("mass", 0, np.float64), charge=cat("charge", 0, np.float64),
                    sigma=cat("sigma", 0, np.float64), eps=cat("eps", 0, np.float64),
                    molid=cat("molid", 0, np.int32))
    sysm.v = np.array(gro.v, dtype=np.float64) if gro.v is not None else None
    sysm.bonds = cat("bonds", (0, 2), np.int32); sysm.bond_b0 = cat("bond_b0", 0, np.float64)
    sysm.bond_kb = cat("bond_kb", 0, np.float64)
    sysm.angles = cat("angles", (0, 3), np.int32); sysm.angle_th0 = cat("angle_th0", 0, np.float64)
    sysm.angle_k = cat("angle_k", 0, np.float64)
    sysm.dihedrals = cat("dihedrals", (0, 4), np.int32); sysm.dih_phi0 = cat("dih_phi0", 0, np.float64)
    sysm.dih_k = cat("dih_k", 0, np.float64); sysm.dih_mult = cat("dih_mult", 0, np.int32)
    sysm.pairs14 = cat("pairs14", (0, 2), np.int32); sysm.pair_c6 = cat("pair_c6", 0, np.float64)
    sysm.pair_c12 = cat("pair_c12", 0, np.float64); sysm.pair_qq = cat("pair_qq", 0, np.float64)
    sysm.settles = cat("settles", (0, 3), np.int32)
    if doh is not None:
        sysm.settle_doh, sysm.settle_dhh = doh, dhh
    sysm.constraints = cat("constraints", (0, 2), np.int32)
    sysm.constraint_len = cat("constraint_len", 0, np.float64)
    counts = cat("excl_counts", 0, np.int64)
    sysm.excl_start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    sysm.excl_list = cat("excl_list", 0, np.int32)
    sysm.posres_idx = cat("posres_idx", 0, np.int32)
    sysm.posres_k = cat("posres_k", (0, 3), np.float64)
    if len(sysm.posres_idx):
        if posres_ref is None:
            raise TopologyError("position restraints are active but no reference coordinates were given (-r)")
        ref = np.asarray(posres_ref.x if hasattr(posres_ref, "x") else posres_ref, dtype=np.float64)
        if len(ref) != n:
            raise TopologyError(f"restraint reference file has {len(ref)} atoms, the topology {n}")
        # reference positions follow the lattice shifts that made the molecules whole
        sysm.posres_ref = (ref + (x - x_read))[sysm.posres_idx]
    sysm.resnr = np.asarray(gro.resnr)
    sysm.resname = list(gro.resname)
    sysm.atomname = list(gro.atomname)
    sysm.title = top.system or gro.title
    if groups is None:
        from .ndx import default_groups
        groups = default_groups(sysm.resname, sysm.atomname)
    sysm.groups = groups
    sysm.validate()
    return sysm


# ------------------------------------------------------------------------------------------
# writer: an MDSystem as a self-contained topology tree in the pipeline's dialect
# ------------------------------------------------------------------------------------------
def _type_table(sysm):
    """Unique (sigma, eps, mass) triples -> generated atomtype names."""
    key = np.stack([np.round(sysm.sigma, 9), np.round(sysm.eps, 9), np.round(sysm.mass, 6)], 1)
    uniq, inv = np.unique(key, axis=0, return_inverse=True)
    names = []
    for i, (s, e, m) in enumerate(uniq):
        el = "h" if m < 1.5 else ("c" if m < 12.5 else ("n" if m < 14.5 else ("o" if m < 16.5 else "m")))
        names.append(f"{el}x{i}")
    return names, uniq, np.asarray(inv).reshape(-1)


def _molecule_blocks(sysm):
    """Consecutive molecules with the same (size, residue names, atom names) form one block."""
    molid = np.asarray(sysm.molid)
    n = len(molid)
    starts = np.concatenate([[0], np.nonzero(np.diff(molid))[0] + 1, [n]])
    blocks = []
    for m in range(len(starts) - 1):
        a, b = int(starts[m]), int(starts[m + 1])
        sig = (b - a, sysm.resname[a], tuple(sysm.atomname[a:a + min(b - a, 4)]))
        if blocks and blocks[-1]["sig"] == sig and b - a <= 4:
            blocks[-1]["count"] += 1
            blocks[-1]["end"] = b
        else:
            blocks.append(dict(sig=sig, start=a, na=b - a, count=1, end=b))
    return blocks


def write_top_tree(dirpath: str, sysm: MDSystem, name: str = "topol.top", posres_files: bool = True,
                   merge_protein: bool = True):
    """Write topol.top (+ included .itp files) that read_top/build_system turn back into `sysm`.

    Layout mirrors the pipeline's: force-field include with [ defaults ] and [ atomtypes ], the
    protein moleculetype inline with `#ifdef POSRES / #include "posre.itp"`, the ligand as an
    included `.itp` with its own `#ifdef POSRES` block, water and ion includes, [ system ],
    [ molecules ]. Returns the path of the .top file."""
    os.makedirs(dirpath, exist_ok=True)
    ffdir = os.path.join(dirpath, "mdx.ff")
    os.makedirs(ffdir, exist_ok=True)
    tnames, tvals, tinv = _type_table(sysm)
    with open(os.path.join(ffdir, "forcefield.itp"), "w") as f:
        f.write("; force field of a generated system: AMBER conventions\n[ defaults ]\n"
                "; nbfunc comb-rule gen-pairs fudgeLJ fudgeQQ\n1 2 yes 0.5 0.8333333333\n\n[ atomtypes ]\n"
                "; name at.num mass charge ptype sigma epsilon\n")
        for nm, (s, e, m) in zip(tnames, tvals):
            f.write(f"{nm:<6s} {nm:<6s} 0 {m:.5f} 0.0000 A {s:.9e} {e:.9e}\n")
    blocks = _molecule_blocks(sysm)
    # merge consecutive multi-atom "protein" molecules (chains) into one moleculetype, as pdb2gmx
    # writes one Protein_chain per file
    merged = []
    for b in blocks:
        if (merge_protein and merged and merged[-1]["na_total"] > 4 and b["na"] > 4
                and sysm.resname[b["start"]] == sysm.resname[merged[-1]["start"]]):
            merged[-1]["end"] = b["end"]; merged[-1]["na_total"] = merged[-1]["end"] - merged[-1]["start"]
        else:
            merged.append(dict(start=b["start"], end=b["end"], count=b["count"], na=b["na"],
                               na_total=b["end"] - b["start"] if b["count"] == 1 else b["na"]))
    posres_of = {int(i): k for i, k in zip(sysm.posres_idx, sysm.posres_k)}

    def in_range(arr, a, b):
        if not len(arr):
            return np.zeros(0, dtype=np.int64)
        return np.nonzero((arr[:, 0] >= a) & (arr[:, 0] < b))[0]

    def write_moleculetype(f, mname, a, b, posres_include):
        na = b - a
        lines = []
        edges = []
        lines.append("[ atoms ]\n; nr type resnr residue atom cgnr charge mass")
        r0 = int(sysm.resnr[a])
        for i in range(a, b):
            lines.append(f"{i - a + 1:6d} {tnames[tinv[i]]:<6s} {int(sysm.resnr[i]) - r0 + 1:6d} {sysm.resname[i]:<5s} "
                         f"{sysm.atomname[i]:<5s} {i - a + 1:6d} {sysm.charge[i]: .10f} {sysm.mass[i]:.5f}")
        sel = in_range(sysm.bonds, a, b)
        if len(sel):
            lines.append("\n[ bonds ]")
            for t in sel:
                i, j = sysm.bonds[t] - a
                lines.append(f"{i + 1} {j + 1} 1 {sysm.bond_b0[t]:.9e} {sysm.bond_kb[t]:.9e}")
                edges.append((i, j))
        sel = in_range(sysm.constraints, a, b)
        if len(sel):
            lines.append("\n[ constraints ]")
            for t in sel:
                i, j = sysm.constraints[t] - a
                lines.append(f"{i + 1} {j + 1} 1 {sysm.constraint_len[t]:.9e}")
                edges.append((i, j))
        sel = in_range(sysm.pairs14, a, b)
        if len(sel):
            lines.append("\n[ pairs ]")
            for t in sel:
                i, j = sysm.pairs14[t] - a
                # explicit sigma/epsilon reproduce c6/c12 exactly
                c6, c12 = sysm.pair_c6[t], sysm.pair_c12[t]
                if c6 > 0 and c12 > 0:
                    s_ij = (c12 / c6) ** (1.0 / 6.0); e_ij = c6 * c6 / (4.0 * c12)
                else:
                    s_ij, e_ij = 0.0, 0.0
                lines.append(f"{i + 1} {j + 1} 1 {s_ij:.12e} {e_ij:.12e}")
        sel = in_range(sysm.angles, a, b)
        if len(sel):
            lines.append("\n[ angles ]")
            for t in sel:
                i, j, k = sysm.angles[t] - a
                lines.append(f"{i + 1} {j + 1} {k + 1} 1 {sysm.angle_th0[t]:.9e} {sysm.angle_k[t]:.9e}")
        sel = in_range(sysm.dihedrals, a, b)
        if len(sel):
            lines.append("\n[ dihedrals ]")
            for t in sel:
                i, j, k, l = sysm.dihedrals[t] - a
                lines.append(f"{i + 1} {j + 1} {k + 1} {l + 1} 9 {sysm.dih_phi0[t]:.9e} {sysm.dih_k[t]:.9e} {int(sysm.dih_mult[t])}")
        sel = in_range(sysm.settles, a, b)
        if len(sel):
            lines.append("\n[ settles ]\n; OW funct doh dhh")
            for t in sel:
                o = sysm.settles[t][0] - a
                lines.append(f"{o + 1} 1 {sysm.settle_doh:.6f} {sysm.settle_dhh:.6f}")
                edges += [(o, o + 1), (o, o + 2), (o + 1, o + 2)]
        # exclusions: nrexcl 3 when the bond graph reproduces them, explicit lists otherwise
        es, el = sysm.excl_start, sysm.excl_list
        want = [set(int(t) - a for t in el[es[i]:es[i + 1]]) for i in range(a, b)]
        nrexcl = 3
        if na > 1:
            gs, gl = build_exclusions(na, np.array(edges, dtype=np.int32).reshape(-1, 2), np.full(na, 3))
            got = [set(int(t) for t in gl[gs[i]:gs[i + 1]]) for i in range(na)]
            if got != want:
                nrexcl = 0
        if nrexcl == 0:
            lines.append("\n[ exclusions ]")
            for i in range(na):
                if want[i]:
                    lines.append(" ".join(str(t + 1) for t in [i] + sorted(want[i])))
        pr = [(i - a, posres_of[i]) for i in range(a, b) if i in posres_of]
        f.write(f"[ moleculetype ]\n; name nrexcl\n{mname} {nrexcl}\n\n" + "\n".join(lines) + "\n")
        if pr:
            if posres_include:
                with open(os.path.join(dirpath, posres_include), "w") as g:
                    g.write("[ position_restraints ]\n; atom type fx fy fz\n")
                    for i, k in pr:
                        g.write(f"{i + 1:6d} 1 {k[0]:g} {k[1]:g} {k[2]:g}\n")
                f.write(f'\n; Include Position restraint file\n#ifdef POSRES\n#include "{posres_include}"\n#endif\n')
            else:
                f.write("\n#ifdef POSRES\n[ position_restraints ]\n")
                for i, k in pr:
                    f.write(f"{i + 1:6d} 1 {k[0]:g} {k[1]:g} {k[2]:g}\n")
                f.write("#endif\n")
        f.write("\n")

    molecules = []
    top_path = os.path.join(dirpath, name)
    used_names = {}
    with open(top_path, "w") as f:
        f.write(f"; topology of a generated system ({sysm.title})\n")
        f.write('#include "mdx.ff/forcefield.itp"\n\n')
        for m in merged:
            a = m["start"]
            rn = sysm.resname[a]
            if m["count"] > 1 or m["na"] <= 4:
                mname = rn
                na = m["na"]
                fn = os.path.join("mdx.ff", f"{rn.lower()}.itp")
                if mname not in used_names:
                    with open(os.path.join(dirpath, fn), "w") as g:
                        write_moleculetype(g, mname, a, a + na, None)
                    f.write(f'#include "{fn}"\n\n')
                    used_names[mname] = True
                molecules.append((mname, m["count"]))
            elif rn == "MOL" or m["na_total"] < 200:
                mname = rn
                fn = f"{rn.lower()}_ligand.itp"
                with open(os.path.join(dirpath, fn), "w") as g:
                    write_moleculetype(g, mname, a, m["end"], None)
                f.write(f'; Include ligand topology\n#include "{fn}"\n\n')
                molecules.append((mname, 1))
            else:
                mname = "Protein_chain_A" if "Protein_chain_A" not in used_names else f"Protein_chain_{len(used_names)}"
                used_names[mname] = True
                write_moleculetype(f, mname, a, m["end"], "posre.itp" if posres_files else None)
                molecules.append((mname, 1))
        f.write(f"[ system ]\n; Name\n{sysm.title}\n\n[ molecules ]\n; Compound #mols\n")
        for mname, cnt in molecules:
            f.write(f"{mname:<20s} {cnt}\n")
    return top_path
