"""Synthetic but physically faithful systems for parity tests and benchmarks.

No `.gro/.top` of the solvated ERa-estradiol complex is committed in the pipeline (its
`output/` is git-ignored, /root/reference/.gitignore:3) and the force-field tree ships with
GROMACS, so the engine is exercised on systems generated here with the same cell
(rhombic dodecahedron, /root/reference/config.yaml:16), the same water model (TIP3P,
config.yaml:10), the same kind of topology (AMBER-style bonds/angles/type-9 dihedrals,
h-bond constraints, 1-4 pairs with fudge 0.5/0.8333) and the same index groups the `.mdp`
files name (Protein, Non-Protein, MOL, SOL, NA, ActiveSite, System).
"""
from __future__ import annotations

import json
import os
import numpy as np

from . import units as U
from .system import MDSystem, build_exclusions


def dodecahedron_box(d: float) -> np.ndarray:
    """Rhombic dodecahedron (xy-square) with image distance d, GROMACS convention."""
    return np.array([[d, 0.0, 0.0], [0.0, d, 0.0], [0.5 * d, 0.5 * d, 0.5 * np.sqrt(2.0) * d]])


def _random_rotations(rng, n):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    w, x, y, z = q.T
    R = np.empty((n, 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - z * w); R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w); R[:, 2, 1] = 2 * (y * z + x * w); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def _water_geometry():
    h = np.sqrt(U.TIP3P_DOH ** 2 - (0.5 * U.TIP3P_DHH) ** 2)
    return np.array([[0.0, 0.0, 0.0], [0.5 * U.TIP3P_DHH, h, 0.0], [-0.5 * U.TIP3P_DHH, h, 0.0]])


def water_sites(box: np.ndarray, density: float = 33.3):
    """Skew lattice of molecule centres that is seamless under the triclinic PBC."""
    vol = abs(np.linalg.det(box))
    lens = np.linalg.norm(box, axis=1)
    s = (density * vol) ** (1.0 / 3.0) / np.prod(lens) ** (1.0 / 3.0)
    n = np.maximum(1, np.rint(lens * s).astype(int))
    g = np.stack(np.meshgrid(*[(np.arange(k) + 0.5) / k for k in n], indexing="ij"), axis=-1).reshape(-1, 3)
    return g @ box, n


def maxwell_velocities(mass, T, rng, ndf=None):
    v = rng.normal(size=(len(mass), 3)) * np.sqrt(U.BOLTZ * T / mass)[:, None]
    v -= (mass[:, None] * v).sum(0) / mass.sum()
    if ndf is not None:
        ke = 0.5 * (mass[:, None] * v * v).sum()
        v *= np.sqrt(0.5 * ndf * U.BOLTZ * T / ke)
    return v


def _assemble_water(centres, rng, jitter):
    nw = len(centres)
    geom = _water_geometry()
    R = _random_rotations(rng, nw)
    xw = centres[:, None, :] + np.einsum("nij,kj->nki", R, geom)
    if jitter > 0:
        xw += rng.normal(scale=jitter, size=(nw, 1, 3))
    return xw.reshape(-1, 3)


def _water_arrays(nw):
    mass = np.tile([U.MASS_O, U.MASS_H, U.MASS_H], nw)
    q = np.tile([U.TIP3P_QO, U.TIP3P_QH, U.TIP3P_QH], nw)
    sig = np.tile([U.TIP3P_SIGMA_O, 0.0, 0.0], nw)
    eps = np.tile([U.TIP3P_EPS_O, 0.0, 0.0], nw)
    return mass, q, sig, eps


def water_box(d: float = 12.5, density: float = 33.3, seed: int = 20250829,
              jitter: float = 0.01, temperature: float = 300.0, box: np.ndarray | None = None,
              rigid: bool = True) -> MDSystem:
    """TIP3P box in a rhombic dodecahedron (or any lower-triangular `box`)."""
    rng = np.random.default_rng(seed)
    if box is None:
        box = dodecahedron_box(d)
    centres, _ = water_sites(box, density)
    nw = len(centres)
    x = _assemble_water(centres, rng, jitter)
    mass, q, sig, eps = _water_arrays(nw)
    n = 3 * nw
    idx = np.arange(n, dtype=np.int32).reshape(nw, 3)
    # exclusions: all three atoms of a molecule exclude each other
    excl_start = (2 * np.arange(n + 1)).astype(np.int32)
    o, h1, h2 = idx[:, 0], idx[:, 1], idx[:, 2]
    excl = np.stack([np.stack([h1, h2], 1), np.stack([o, h2], 1), np.stack([o, h1], 1)], 1).reshape(-1)
    sysm = MDSystem(x=x, box=box, mass=mass, charge=q, sigma=sig, eps=eps,
                    molid=np.repeat(np.arange(nw, dtype=np.int32), 3),
                    excl_start=excl_start, excl_list=excl.astype(np.int32))
    if rigid:
        sysm.settles = idx.copy()
    else:
        kb, ka = 502416.0, 628.02
        sysm.bonds = np.concatenate([idx[:, [0, 1]], idx[:, [0, 2]]]).astype(np.int32)
        sysm.bond_b0 = np.full(2 * nw, U.TIP3P_DOH)
        sysm.bond_kb = np.full(2 * nw, kb)
        sysm.angles = idx[:, [1, 0, 2]].astype(np.int32)
        sysm.angle_th0 = np.full(nw, 104.52)
        sysm.angle_k = np.full(nw, ka)
    sysm.resnr = np.repeat(np.arange(1, nw + 1), 3)
    sysm.resname = ["SOL"] * n
    sysm.atomname = ["OW", "HW1", "HW2"] * nw
    ndf = 3 * n - 3 - (3 * nw if rigid else 0)
    sysm.v = maxwell_velocities(mass, temperature, np.random.default_rng(1), ndf)
    allidx = np.arange(n, dtype=np.int32)
    sysm.groups = {"System": allidx, "SOL": allidx, "Water": allidx, "Non-Protein": allidx,
                   "Protein": np.zeros(0, np.int32)}
    sysm.title = f"TIP3P water, {nw} molecules"
    return sysm


# ----------------------------------------------------------------------------------------
# pseudo-complex: alkane-bundle "protein" + 44-atom ligand + Na+ + TIP3P
# ----------------------------------------------------------------------------------------
_C = dict(sigma=0.339967, eps=0.457730, mass=12.011)
_HC = dict(sigma=0.264953, eps=0.0656888, mass=1.008)
_OH = dict(sigma=0.306647, eps=0.880314, mass=15.9994)
_HO = dict(sigma=0.0, eps=0.0, mass=1.008)
_NA = dict(sigma=0.332840, eps=0.0115897, mass=22.98977)
KB_CC, B0_CC = 259408.0, 0.1526
KB_CH, B0_CH = 284512.0, 0.1090
KA_CCC, KA_HCC, KA_HCH = 334.72, 418.40, 292.88
TH0 = 109.5


def _alkane_bundle(radius: float, spacing: float = 0.46):
    """All-trans CH2 chains along z packed on a square lattice inside a sphere."""
    half = np.deg2rad(TH0 / 2)
    dz = B0_CC * np.sin(half)
    amp = 0.5 * B0_CC * np.cos(half)
    hx, hy = B0_CH * np.cos(half), B0_CH * np.sin(half)
    X, el, bonds, chains = [], [], [], []
    m = int(radius / spacing) + 1
    for ix in range(-m, m + 1):
        for iy in range(-m, m + 1):
            cx, cy = ix * spacing, iy * spacing
            r2 = radius ** 2 - cx ** 2 - cy ** 2
            if r2 <= 0:
                continue
            nC = int(2 * np.sqrt(r2) / dz)
            if nC < 4:
                continue
            z0 = -0.5 * (nC - 1) * dz
            cidx = []
            for k in range(nC):
                s = 1.0 if k % 2 == 0 else -1.0
                c = np.array([cx + s * amp, cy, z0 + k * dz])
                ci = len(X)
                X.append(c); el.append("C"); cidx.append(ci)
                if k > 0:
                    bonds.append((cidx[k - 1], ci))
                for sy in (1.0, -1.0):
                    X.append(c + np.array([s * hx, sy * hy, 0.0])); el.append("H")
                    bonds.append((ci, len(X) - 1))
                if k == 0 or k == nC - 1:
                    # methyl cap along the direction the next carbon would take
                    sgn = -1.0 if k == 0 else 1.0
                    dirv = np.array([-s * 2 * amp, 0.0, sgn * dz])
                    dirv /= np.linalg.norm(dirv)
                    X.append(c + B0_CH * dirv); el.append("H")
                    bonds.append((ci, len(X) - 1))
            chains.append(cidx)
    return np.array(X), el, np.array(bonds, dtype=np.int32), chains


def _bonded_from_graph(n, bonds):
    nbr = [[] for _ in range(n)]
    for a, b in bonds:
        nbr[a].append(int(b)); nbr[b].append(int(a))
    angles, dihs = [], []
    for j in range(n):
        nb = sorted(nbr[j])
        for a in range(len(nb)):
            for b in range(a + 1, len(nb)):
                angles.append((nb[a], j, nb[b]))
    for j, k in bonds:
        j, k = int(j), int(k)
        for i in nbr[j]:
            if i == k:
                continue
            for l in nbr[k]:
                if l == j or l == i:
                    continue
                dihs.append((i, j, k, l))
    return nbr, np.array(angles, dtype=np.int32).reshape(-1, 3), np.array(dihs, dtype=np.int32).reshape(-1, 4)


def _angle_deg(x, a):
    u = x[a[:, 0]] - x[a[:, 1]]
    w = x[a[:, 2]] - x[a[:, 1]]
    c = (u * w).sum(1) / np.linalg.norm(u, axis=1) / np.linalg.norm(w, axis=1)
    return np.rad2deg(np.arccos(np.clip(c, -1, 1)))


def _load_ligand():
    p = os.path.join(os.path.dirname(__file__), "data", "ligand44.json")
    with open(p) as f:
        d = json.load(f)
    return d["elements"], np.array(d["xyz_nm"]), np.array([b[:2] for b in d["bonds"]], dtype=np.int32)


def _solute_topology(x, el, bonds, charges, flexible_from_geometry):
    """Bonded terms for a C/H/O solute. H-bonds become constraints (`constraints = h-bonds`,
    /root/reference/mdp/production.mdp:17)."""
    n = len(el)
    nbr, angles, dihs = _bonded_from_graph(n, bonds)
    isH = np.array([e == "H" for e in el])
    hb = isH[bonds[:, 0]] | isH[bonds[:, 1]]
    blen = np.linalg.norm(x[bonds[:, 0]] - x[bonds[:, 1]], axis=1)
    out = {}
    flex = bonds[~hb]
    out["bonds"] = flex
    if flexible_from_geometry:
        out["bond_b0"] = blen[~hb]
        out["bond_kb"] = np.full(len(flex), 280000.0)
        out["constraint_len"] = blen[hb]
    else:
        out["bond_b0"] = np.full(len(flex), B0_CC)
        out["bond_kb"] = np.full(len(flex), KB_CC)
        out["constraint_len"] = np.full(int(hb.sum()), B0_CH)
    out["constraints"] = bonds[hb]
    out["angles"] = angles
    nh = isH[angles[:, 0]].astype(int) + isH[angles[:, 2]].astype(int)
    out["angle_k"] = np.choose(nh, [KA_CCC, KA_HCC, KA_HCH]).astype(float)
    out["angle_th0"] = _angle_deg(x, angles) if flexible_from_geometry else np.full(len(angles), TH0)
    # dihedrals: type-9 style multi-term on heavy-atom quadruples, single term otherwise
    deg = np.array([len(v) for v in nbr])
    D, P, K, M = [], [], [], []
    for (i, j, k, l) in dihs:
        sp2 = (deg[j] == 3 and deg[k] == 3 and el[j] == "C" and el[k] == "C")
        if sp2:
            D.append((i, j, k, l)); P.append(180.0); K.append(15.167); M.append(2)
        elif not isH[i] and not isH[l]:
            for (ph, kk, mm) in ((0.0, 0.75312, 3), (180.0, 1.046, 2), (180.0, 0.8368, 1)):
                D.append((i, j, k, l)); P.append(ph); K.append(kk); M.append(mm)
        else:
            D.append((i, j, k, l)); P.append(0.0); K.append(0.6508), M.append(3)
    out["dihedrals"] = np.array(D, dtype=np.int32).reshape(-1, 4)
    out["dih_phi0"] = np.array(P); out["dih_k"] = np.array(K); out["dih_mult"] = np.array(M, dtype=np.int32)
    # 1-4 pairs (unique, not also 1-2/1-3)
    excl_start, excl_list = build_exclusions(n, bonds, np.full(n, 3))
    ex2_start, ex2_list = build_exclusions(n, bonds, np.full(n, 2))
    p14 = set()
    for (i, _, _, l) in dihs:
        a, b = (int(i), int(l)) if i < l else (int(l), int(i))
        if b in ex2_list[ex2_start[a]:ex2_start[a + 1]] or a == b:
            continue
        p14.add((a, b))
    out["pairs14"] = np.array(sorted(p14), dtype=np.int32).reshape(-1, 2)
    out["excl"] = (excl_start, excl_list)
    return out


def _lj_of(el, nbr_el=None):
    tab = {"C": _C, "H": _HC, "O": _OH, "NA": _NA}
    return tab[el]


def pseudo_complex(d: float = 12.5, protein_radius: float = 2.1, density: float = 33.3,
                   seed: int = 20250829, temperature: float = 300.0, n_ions: int = 7,
                   ligand_offset: float = 3.0, n_active: int = 300, posres_fc: float = 1000.0,
                   ligand_posres_fc: float = 10000.0) -> MDSystem:
    """Pseudo-complex: ~3.9 k-atom alkane-bundle "protein" (net charge -n_ions), the 44-atom
    ligand `ligand_offset` nm outside it, n_ions Na+, TIP3P, rhombic dodecahedron."""
    rng = np.random.default_rng(seed)
    box = dodecahedron_box(d)
    centre = 0.5 * box.sum(0)

    # ---- protein
    xp, elp, bp, chains = _alkane_bundle(protein_radius)
    xp = xp + centre
    npz = len(elp)
    qp = np.where(np.array(elp) == "C", -0.12, 0.06).astype(float)
    # polar sites: alternate +-0.3 on a subset of chain carbons, then shift to net -n_ions
    carb = np.array([c for ch in chains for c in ch[2:-2:7]])
    qp[carb[0::2]] += 0.3
    qp[carb[1::2]] -= 0.3
    qp -= qp.sum() / npz
    surf = np.argsort(-np.linalg.norm(xp - centre, axis=1))[:max(1, n_ions) * 10]
    qp[surf] -= float(n_ions) / len(surf)
    top_p = _solute_topology(xp, elp, bp, qp, flexible_from_geometry=False)

    # ---- ligand
    ell, xl, bl = _load_ligand()
    xl = xl - xl.mean(0) + centre + np.array([protein_radius + ligand_offset, 0.0, 0.0])
    ql = np.zeros(len(ell))
    for a, b in bl:
        ea, eb = ell[a], ell[b]
        if {ea, eb} == {"O", "H"}:
            ql[a if ea == "H" else b] += 0.41; ql[a if ea == "O" else b] -= 0.41
        elif {ea, eb} == {"C", "O"}:
            ql[a if ea == "C" else b] += 0.25; ql[a if ea == "O" else b] -= 0.25
        elif {ea, eb} == {"C", "H"}:
            ql[a if ea == "H" else b] += 0.06; ql[a if ea == "C" else b] -= 0.06
    ql -= ql.sum() / len(ql)
    top_l = _solute_topology(xl, ell, bl, ql, flexible_from_geometry=True)

    # ---- water, carve out solute
    centres, _ = water_sites(box, density)
    from scipy.spatial import cKDTree
    xs = np.concatenate([xp, xl])
    # include the periodic images of the solute so that waters are also carved across the cell faces
    imgs = [i * box[0] + j * box[1] + k * box[2] for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (-1, 0, 1)]
    xs = np.concatenate([xs + s_ for s_ in imgs])
    tree = cKDTree(xs)
    dmin, _ = tree.query(centres, k=1)
    keep = dmin > 0.33
    centres = centres[keep]
    # ions replace waters far from the solute
    far = np.where(dmin[keep] > 1.0)[0]
    ion_sites = rng.choice(far, size=n_ions, replace=False) if n_ions else np.zeros(0, int)
    ion_mask = np.zeros(len(centres), bool)
    ion_mask[ion_sites] = True
    xion = centres[ion_mask]
    xw = _assemble_water(centres[~ion_mask], rng, 0.01)
    nw = len(xw) // 3

    # ---- concatenate: protein, ligand, water, ions (pipeline order: Snakefile:617,645)
    nl = len(ell)
    off_l, off_w, off_i = npz, npz + nl, npz + nl + 3 * nw
    n = off_i + n_ions
    x = np.concatenate([xp, xl, xw, xion]) if n_ions else np.concatenate([xp, xl, xw])
    mw, qw, sw, ew = _water_arrays(nw)

    def lj(els, key):
        return np.array([_lj_of(e)[key] for e in els], dtype=float)
    ell_lj = list(ell)
    # hydroxyl hydrogens carry no LJ
    for a, b in bl:
        if {ell[a], ell[b]} == {"O", "H"}:
            ell_lj[a if ell[a] == "H" else b] = "HO"
    tabl = {"C": _C, "H": _HC, "O": _OH, "HO": _HO}
    mass = np.concatenate([lj(elp, "mass"), np.array([tabl[e]["mass"] for e in ell_lj]), mw, np.full(n_ions, _NA["mass"])])
    sig = np.concatenate([lj(elp, "sigma"), np.array([tabl[e]["sigma"] for e in ell_lj]), sw, np.full(n_ions, _NA["sigma"])])
    eps = np.concatenate([lj(elp, "eps"), np.array([tabl[e]["eps"] for e in ell_lj]), ew, np.full(n_ions, _NA["eps"])])
    q = np.concatenate([qp, ql, qw, np.full(n_ions, 1.0)])
    q[:npz] -= (q.sum()) / npz     # exact neutrality
    # molecule ids: each protein chain is part of one "Protein" molecule block per chain
    molid = np.zeros(n, dtype=np.int32)
    # connected components of the protein bond graph
    comp = -np.ones(npz, dtype=np.int64)
    nbrp = [[] for _ in range(npz)]
    for a, b in bp:
        nbrp[a].append(int(b)); nbrp[b].append(int(a))
    nc = 0
    for s0 in range(npz):
        if comp[s0] >= 0:
            continue
        stack = [s0]; comp[s0] = nc
        while stack:
            a = stack.pop()
            for b in nbrp[a]:
                if comp[b] < 0:
                    comp[b] = nc; stack.append(b)
        nc += 1
    molid[:npz] = comp
    molid[off_l:off_w] = nc
    molid[off_w:off_i] = nc + 1 + np.repeat(np.arange(nw), 3)
    molid[off_i:] = nc + 1 + nw + np.arange(n_ions)

    def cat_idx(a, b, ob):
        return np.concatenate([a, b + ob]).astype(np.int32)

    sysm = MDSystem(x=x, box=box, mass=mass, charge=q, sigma=sig, eps=eps, molid=molid)
    sysm.bonds = cat_idx(top_p["bonds"], top_l["bonds"], off_l)
    sysm.bond_b0 = np.concatenate([top_p["bond_b0"], top_l["bond_b0"]])
    sysm.bond_kb = np.concatenate([top_p["bond_kb"], top_l["bond_kb"]])
    sysm.angles = cat_idx(top_p["angles"], top_l["angles"], off_l)
    sysm.angle_th0 = np.concatenate([top_p["angle_th0"], top_l["angle_th0"]])
    sysm.angle_k = np.concatenate([top_p["angle_k"], top_l["angle_k"]])
    sysm.dihedrals = cat_idx(top_p["dihedrals"], top_l["dihedrals"], off_l)
    sysm.dih_phi0 = np.concatenate([top_p["dih_phi0"], top_l["dih_phi0"]])
    sysm.dih_k = np.concatenate([top_p["dih_k"], top_l["dih_k"]])
    sysm.dih_mult = np.concatenate([top_p["dih_mult"], top_l["dih_mult"]]).astype(np.int32)
    sysm.pairs14 = cat_idx(top_p["pairs14"], top_l["pairs14"], off_l)
    sysm.constraints = cat_idx(top_p["constraints"], top_l["constraints"], off_l)
    sysm.constraint_len = np.concatenate([top_p["constraint_len"], top_l["constraint_len"]])
    pi, pj = sysm.pairs14[:, 0], sysm.pairs14[:, 1]
    sij = 0.5 * (sig[pi] + sig[pj]); eij = np.sqrt(eps[pi] * eps[pj])
    sysm.pair_c6 = 0.5 * 4 * eij * sij ** 6
    sysm.pair_c12 = 0.5 * 4 * eij * sij ** 12
    sysm.pair_qq = (1.0 / 1.2) * q[pi] * q[pj]
    # exclusions CSR
    es_p, el_p = top_p["excl"]; es_l, el_l = top_l["excl"]
    widx = np.arange(3 * nw, dtype=np.int32).reshape(nw, 3) + off_w
    o, h1, h2 = widx[:, 0], widx[:, 1], widx[:, 2]
    exw = np.stack([np.stack([h1, h2], 1), np.stack([o, h2], 1), np.stack([o, h1], 1)], 1).reshape(-1)
    counts = np.concatenate([np.diff(es_p), np.diff(es_l), np.full(3 * nw, 2), np.zeros(n_ions, int)])
    sysm.excl_start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    sysm.excl_list = np.concatenate([el_p, el_l + off_l, exw]).astype(np.int32)
    sysm.settles = widx.astype(np.int32)
    # position restraints (-DPOSRES: protein heavy atoms fc 1000, every ligand atom fc 10000,
    # /root/reference/scripts/fix_ligand_topology.py:88-96)
    heavy_p = np.array([i for i, e in enumerate(elp) if e != "H"], dtype=np.int32)
    lig = np.arange(off_l, off_w, dtype=np.int32)
    sysm.posres_idx = np.concatenate([heavy_p, lig]).astype(np.int32)
    sysm.posres_ref = x[sysm.posres_idx].copy()
    sysm.posres_k = np.concatenate([np.full((len(heavy_p), 3), posres_fc), np.full((len(lig), 3), ligand_posres_fc)])
    # names
    sysm.resname = ["ALA"] * npz + ["MOL"] * nl + ["SOL"] * (3 * nw) + ["NA"] * n_ions
    sysm.atomname = [("C" if e == "C" else "H") for e in elp] + [e for e in ell] + ["OW", "HW1", "HW2"] * nw + ["NA"] * n_ions
    resnr_p = comp + 1
    sysm.resnr = np.concatenate([resnr_p, np.full(nl, nc + 1), nc + 2 + np.repeat(np.arange(nw), 3),
                                 nc + 2 + nw + np.arange(n_ions)]).astype(np.int64)
    prot = np.arange(npz, dtype=np.int32)
    sol = np.arange(off_w, off_i, dtype=np.int32)
    na = np.arange(off_i, n, dtype=np.int32)
    dcen = np.linalg.norm(xp - centre, axis=1)
    # "active site": protein atoms of the chains nearest the centre, whole chains first
    active = np.sort(np.argsort(dcen)[:n_active]).astype(np.int32)
    sysm.groups = {"System": np.arange(n, dtype=np.int32), "Protein": prot, "MOL": lig, "SOL": sol,
                   "NA": na, "Non-Protein": np.arange(npz, n, dtype=np.int32),
                   "Water": sol, "Water_and_ions": np.concatenate([sol, na]), "ActiveSite": active}
    ndf = 3 * n - 3 - 3 * nw - len(sysm.constraints)
    sysm.v = maxwell_velocities(mass, temperature, np.random.default_rng(1), ndf)
    sysm.title = f"pseudo-complex: {npz} protein atoms, {nl} ligand atoms, {nw} waters, {n_ions} NA"
    return sysm


def tiled(sysm: MDSystem, reps=(2, 2, 2)) -> MDSystem:
    """Tile a water-only system `reps` times along each box vector (the ~1 M-atom case)."""
    assert len(sysm.bonds) == 0 and len(sysm.constraints) == 0
    n = sysm.natoms
    shifts = [i * sysm.box[0] + j * sysm.box[1] + k * sysm.box[2]
              for i in range(reps[0]) for j in range(reps[1]) for k in range(reps[2])]
    m = len(shifts)
    x = np.concatenate([sysm.x + s for s in shifts])
    def tile(a):
        return np.tile(a, m)
    offs = np.repeat(np.arange(m) * n, n)
    out = MDSystem(x=x, box=sysm.box * np.array(reps)[:, None], mass=tile(sysm.mass), charge=tile(sysm.charge),
                   sigma=tile(sysm.sigma), eps=tile(sysm.eps),
                   molid=(tile(sysm.molid) + np.repeat(np.arange(m) * (int(sysm.molid.max()) + 1), n)).astype(np.int32))
    out.v = np.tile(sysm.v, (m, 1)) if sysm.v is not None else None
    ne = len(sysm.excl_list)
    out.excl_start = np.concatenate([[0], np.cumsum(np.tile(np.diff(sysm.excl_start), m))]).astype(np.int32)
    out.excl_list = (np.tile(sysm.excl_list, m) + np.repeat(np.arange(m) * n, ne)).astype(np.int32)
    ns = len(sysm.settles)
    out.settles = (np.tile(sysm.settles, (m, 1)) + np.repeat(np.arange(m) * n, ns)[:, None]).astype(np.int32)
    out.resname = sysm.resname * m if sysm.resname else None
    out.atomname = sysm.atomname * m if sysm.atomname else None
    out.resnr = np.tile(sysm.resnr, m) if sysm.resnr is not None else None
    allidx = np.arange(n * m, dtype=np.int32)
    out.groups = {"System": allidx, "SOL": allidx, "Water": allidx, "Non-Protein": allidx,
                  "Protein": np.zeros(0, np.int32)}
    out.title = f"{sysm.title} x{m}"
    return out
