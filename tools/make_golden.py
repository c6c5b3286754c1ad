#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ by running the pipeline's own Python
helpers (importable in the build container only) on inputs authored by this repository.

What is pinned: file formats, not physics - the `.xvg` dialect the analysis scripts read, the
`.gro` box line convention, the index-file layout of the `[ ActiveSite ]` group, and the topology
dialect that the ligand clean-up and the complex-topology merge produce.

Run from the repository root:  python tools/make_golden.py
Needs /root/reference/scripts; the tests only read the files written here.
"""
import importlib.util
import io
import json
import os
import shutil
import sys
from contextlib import redirect_stdout

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/scripts"
GOLD = os.path.join(ROOT, "tests", "golden")

import mdx  # noqa: E402,F401
from mdx import synth  # noqa: E402
from mdx.io import gro, xvg  # noqa: E402
from mdx.system import MDSystem  # noqa: E402


def ref_module(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    with redirect_stdout(io.StringIO()):
        spec.loader.exec_module(mod)
    return mod


def main():
    os.environ.setdefault("MPLBACKEND", "Agg")
    out = {}
    # ---- (1) xvg dialect: files written by this package, read by the pipeline's readers
    d = os.path.join(GOLD, "xvg")
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(7)
    t = np.arange(0, 2.0001, 0.05)
    dist = 3.0 - 0.4 * t + 0.01 * rng.normal(size=len(t))
    w = xvg.pullx_writer(os.path.join(d, "demo_pullx.xvg"))
    for a, b in zip(t, dist):
        w.row(a, b)
    w.close()
    w = xvg.pullf_writer(os.path.join(d, "demo_pullf.xvg"))
    force = -500.0 + 25.0 * rng.normal(size=len(t))
    force[7] = 400.0      # one spike
    for a, b in zip(t, force):
        w.row(a, b)
    w.close()
    # the header carries a date: normalise it so that regenerating does not churn the fixture
    for fn in ("demo_pullx.xvg", "demo_pullf.xvg"):
        p = os.path.join(d, fn)
        lines = open(p).read().split("\n")
        lines[1] = "# (date)"
        open(p, "w").write("\n".join(lines))
    smd = ref_module("analyze_smd_trajectory")
    spikes = ref_module("analyze_force_spikes")
    tx, vx = smd.read_xvg_file(os.path.join(d, "demo_pullx.xvg"))
    tf, vf = spikes.read_pullf_xvg(os.path.join(d, "demo_pullf.xvg"))
    out["xvg"] = dict(pullx_times=tx.tolist(), pullx_values=vx.tolist(), pullf_times=tf.tolist(), pullf_values=vf.tolist())
    cases = {"const": np.full(5000, -500.0), "empty": np.zeros(0), "demo": vf,
             "ramp": np.linspace(-600, -400, 101), "two_spikes": np.r_[np.full(200, -500.0), [900.0, -2000.0]]}
    out["spike_score"] = {k: [float(v) for v in spikes.spike_score(a)] for k, a in cases.items()}
    out["spike_inputs"] = {k: a.tolist() for k, a in cases.items() if k in ("ramp", "two_spikes")}
    an = smd.analyze_smd_quality(tf, vf, vx)
    out["smd_quality"] = {k: (bool(v) if isinstance(v, (bool, np.bool_)) else float(v)) for k, v in an.items()}

    # ---- (2) .gro box line and pose metrics on a .gro written by this package
    pdm = ref_module("pose_distance_metrics")
    lines = ["9.52 9.52 6.73172 0 0 0 0 4.76 4.76", "3.0 4.0 5.0", "   7.50000   7.50000   5.30330   0.00000   0.00000   0.00000   0.00000   3.75000   3.75000"]
    out["box_lines"] = [dict(line=l, matrix=pdm._parse_box_matrix_from_line(l).tolist()) for l in lines]
    d = os.path.join(GOLD, "gro")
    os.makedirs(d, exist_ok=True)
    s = synth.pseudo_complex(d=5.0, protein_radius=0.8, n_ions=2, n_active=40, ligand_offset=1.0)
    # keep the fixture small: solute, ions and the 60 waters nearest the ligand
    lig = s.groups["MOL"]
    sol = s.groups["SOL"].reshape(-1, 3)
    dl = np.linalg.norm(s.x[sol[:, 0]] - s.x[lig].mean(0), axis=1)
    keep = np.sort(np.concatenate([s.groups["Protein"], lig, sol[np.argsort(dl)[:60]].ravel(), s.groups["NA"]]))
    gro.write_gro(os.path.join(d, "pose.gro"), "pose fixture", s.resnr[keep], [s.resname[i] for i in keep],
                  [s.atomname[i] for i in keep], s.x[keep], s.box, s.v[keep])
    resid_active = sorted(set(int(s.resnr[i]) for i in s.groups["ActiveSite"]))[:3]
    with open(os.path.join(d, "active_site_residues.txt"), "w") as f:
        f.write(",".join(str(r) for r in resid_active) + "\n")
    met = pdm.compute_metrics(os.path.join(d, "pose.gro"), os.path.join(d, "active_site_residues.txt"))
    out["pose_metrics"] = {k: float(v) for k, v in met.items()}
    atoms, B = pdm.read_gro_atoms_and_box(os.path.join(d, "pose.gro"))
    out["pose_gro"] = dict(natoms=len(atoms), box_matrix=B.tolist(), first=dict(resid=atoms[0]["resid"], resname=atoms[0]["resname"],
                                                                              atom=atoms[0]["atom"], coords=atoms[0]["coords"].tolist()))

    # ---- (3) index file: the pipeline's index script run against this package's `gmx make_ndx`
    d = os.path.join(GOLD, "ndx")
    os.makedirs(d, exist_ok=True)
    env_path = os.environ.get("PATH", "")
    os.environ["PATH"] = os.path.join(ROOT, "bin") + os.pathsep + env_path
    asg = ref_module("add_active_site_group")
    with redirect_stdout(io.StringIO()):
        res = asg.read_active_site_residues(os.path.join(GOLD, "gro", "active_site_residues.txt"))
        asg.create_index_file(os.path.join(GOLD, "gro", "pose.gro"), "unused.top", res, d)
    os.environ["PATH"] = env_path
    out["active_site"] = dict(residues=res)

    # ---- (4) topology dialect: ligand clean-up and complex merge run on the hand-written inputs
    td = os.path.join(GOLD, "topdialect")
    flt = ref_module("fix_ligand_topology")
    cct = ref_module("create_complex_topology")
    with redirect_stdout(io.StringIO()):
        flt.fix_ligand_topology(os.path.join(td, "ligand_GMX.itp"), os.path.join(td, "clean_ligand.itp"), add_posres=True)
        cct.create_complex_topology(os.path.join(td, "protein.top"), os.path.join(td, "ligand_GMX.itp"), os.path.join(td, "complex.top"))
    # what `gmx solvate` and `gmx genion` do to the topology: append molecule counts
    nw = build_complex_gro(os.path.join(td, "complex.gro"))
    shutil.copy(os.path.join(td, "complex.top"), os.path.join(td, "ion.top"))
    with open(os.path.join(td, "ion.top"), "a") as f:
        f.write("SOL         %d\nNA               1\nCL               1\n" % nw)
    out["topdialect"] = dict(n_water=nw)
    with open(os.path.join(GOLD, "reference_outputs.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("fixtures written under", GOLD)


def build_complex_gro(path):
    """N-methylacetamide + methanol + TIP3P + NA + CL in a 2.4 nm cube, in topology order."""
    # approximate planar trans-NMA (nm)
    nma = np.array([
        [-0.2390, 0.1020, 0.0000],   # HH31
        [-0.1940, 0.0030, 0.0000],   # CH3
        [-0.2260, -0.0520, 0.0890],  # HH32
        [-0.2260, -0.0520, -0.0890],  # HH33
        [-0.0420, 0.0100, 0.0000],   # C
        [0.0190, 0.1170, 0.0000],    # O
        [0.0210, -0.1080, 0.0000],   # N
        [-0.0330, -0.1930, 0.0000],  # H
        [0.1660, -0.1190, 0.0000],   # CH3
        [0.2060, -0.0180, 0.0000],   # HH31
        [0.1990, -0.1720, 0.0890],   # HH32
        [0.1990, -0.1720, -0.0890]]) # HH33
    meoh = np.array([
        [0.0000, 0.0000, 0.0000],    # C1
        [0.1426, 0.0000, 0.0000],    # O1
        [-0.0372, 0.1028, 0.0000],   # H1
        [-0.0372, -0.0514, 0.0890],  # H2
        [-0.0372, -0.0514, -0.0890],  # H3
        [0.1730, 0.0925, 0.0000]])   # H4
    L = 2.4
    box = np.diag([L, L, L])
    xs = np.concatenate([nma + np.array([0.8, 1.2, 1.2]), meoh + np.array([1.55, 1.2, 1.25])])
    w = synth.water_box(box=box, seed=5)
    from scipy.spatial import cKDTree
    ow = w.x[0::3]
    imgs = [np.array([i, j, k]) * L for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (-1, 0, 1)]
    tree = cKDTree(np.concatenate([xs + s_ for s_ in imgs]))
    dmin, _ = tree.query(ow, k=1)
    keep = np.nonzero(dmin > 0.30)[0]
    far = keep[np.argsort(-dmin[keep])[:2]]
    wat = np.array([k for k in keep if k not in far])
    xw = np.concatenate([w.x[3 * k:3 * k + 3] for k in wat])
    x = np.concatenate([xs, xw, ow[far]])
    nw = len(wat)
    resname = ["ACE"] * 6 + ["NME"] * 6 + ["MOL"] * 6 + ["SOL"] * (3 * nw) + ["NA", "CL"]
    atomname = (["HH31", "CH3", "HH32", "HH33", "C", "O", "N", "H", "CH3", "HH31", "HH32", "HH33"] +
                ["C1", "O1", "H1", "H2", "H3", "H4"] + ["OW", "HW1", "HW2"] * nw + ["NA", "CL"])
    resnr = [1] * 6 + [2] * 6 + [3] * 6 + list(np.repeat(np.arange(4, 4 + nw), 3)) + [4 + nw, 5 + nw]
    gro.write_gro(path, "NMA + methanol in water", np.array(resnr), resname, atomname, x, box)
    return nw


if __name__ == "__main__":
    main()
