"""Host-side file formats against the fixtures under tests/golden/ (written by
tools/make_golden.py from the pipeline's own helper scripts) and against hand-derived expectations."""
import io
import json
import os

import numpy as np
import pytest

import mdx  # noqa: F401
from mdx import synth, runner
from mdx.io import mdp, gro, ndx, xvg, top, traj, tpr, energy

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
with open(os.path.join(GOLD, "reference_outputs.json")) as _f:
    REFOUT = json.load(_f)


# ------------------------------------------------------------------------------------------ mdp
def test_mdp_grammar_and_defaults():
    text = """
; comment line
integrator  = md        ; trailing comment
dt = 0.002
tc-grps     = Protein Non-Protein
tau_t       = 0.1	0.1
ref_t = 300 300
define = -DPOSRES -DFLEXIBLE=1
nstcheckpoint = 1000
Pull_Coord1_K = 500
cutoff-scheme = Verlet
"""
    m = mdp.parse_mdp(text)
    assert m.s("integrator") == "md" and m.f("dt") == 0.002
    assert m.words("tc_grps") == ["Protein", "Non-Protein"] and m.words("tc-grps") == m.words("tc_grps")
    assert m.floats("tau_t") == [0.1, 0.1]
    assert m.defines() == {"POSRES": "", "FLEXIBLE": "1"}
    assert m.f("pull_coord1_k") == 500.0
    # unknown keys warn and never abort
    assert len(m.warnings) == 1 and "nstcheckpoint" in m.warnings[0]
    # documented defaults for keys the files leave out
    assert m.i("nstcalcenergy") == 100 and m.i("nstcomm") == 100 and m.f("ewald_rtol") == 1e-5
    assert m.i("pull_nstxout") == 50 and m.f("compressed_x_precision") == 1000.0
    assert m.s("cutoff-scheme") == "Verlet"


def test_mdp_coupling_cadence():
    m = mdp.parse_mdp("dt = 0.002\ntau_t = 0.1 0.1\ntau_p = 2.0\n")
    assert mdp.effective_coupling_steps(m) == (10, 10)
    m = mdp.parse_mdp("dt = 0.002\ntau_t = 1.0\nnsttcouple = 100\nnstpcouple = 100\ntau_p = 5.0\n")
    assert mdp.effective_coupling_steps(m) == (100, 100)
    m = mdp.parse_mdp("dt = 0.002\ntau_t = 0.05\ntau_p = 0.1\n")
    assert mdp.effective_coupling_steps(m) == (5, 5)


# ------------------------------------------------------------------------------------------ gro
def test_box_line_convention_matches_pipeline_reader():
    for case in REFOUT["box_lines"]:
        mine = gro.box_from_line(case["line"])
        # the pipeline's matrix holds the box vectors as columns, this package as rows
        assert np.allclose(mine.T, np.array(case["matrix"]))
    b = synth.dodecahedron_box(12.5)
    assert np.allclose(gro.box_from_line(gro.box_to_line(b)), b, atol=1e-5)
    assert len(gro.box_to_line(np.diag([3.0, 4.0, 5.0])).split()) == 3


def test_gro_fixture_read_like_the_pipeline():
    g = gro.read_gro(os.path.join(GOLD, "gro", "pose.gro"))
    ref = REFOUT["pose_gro"]
    assert g.natoms == ref["natoms"]
    assert np.allclose(g.box.T, np.array(ref["box_matrix"]))
    assert g.resnr[0] == ref["first"]["resid"] and g.resname[0] == ref["first"]["resname"] and g.atomname[0] == ref["first"]["atom"]
    assert np.allclose(g.x[0], ref["first"]["coords"])
    assert g.v is not None and g.v.shape == g.x.shape


def test_gro_roundtrip_wraps_numbers_and_keeps_velocities(tmp_path):
    n = 100005
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 9, size=(n, 3)); v = rng.normal(size=(n, 3))
    resnr = np.arange(1, n + 1)
    p = str(tmp_path / "big.gro")
    gro.write_gro(p, "big", resnr, ["SOL"] * n, ["OW"] * n, x, synth.dodecahedron_box(9.0), v)
    g = gro.read_gro(p)
    assert g.natoms == n and g.resnr[99999] == 0 and g.resnr[100000] == 1      # five-column fields wrap
    assert np.abs(g.x - x).max() <= 5.01e-4 and np.abs(g.v - v).max() <= 5.01e-5
    assert g.residue_index()[-1] == n - 1
    with pytest.raises(ValueError):
        gro.parse_gro(["t\n", "5\n", "x\n"])


# ------------------------------------------------------------------------------------------ xvg
def test_xvg_written_here_is_read_identically_by_pipeline_readers():
    X, leg = xvg.read_xvg(os.path.join(GOLD, "xvg", "demo_pullx.xvg"))
    F, _ = xvg.read_xvg(os.path.join(GOLD, "xvg", "demo_pullf.xvg"))
    assert np.array_equal(X[:, 0], REFOUT["xvg"]["pullx_times"]) and np.array_equal(X[:, 1], REFOUT["xvg"]["pullx_values"])
    assert np.array_equal(F[:, 0], REFOUT["xvg"]["pullf_times"]) and np.array_equal(F[:, 1], REFOUT["xvg"]["pullf_values"])
    assert leg == ["1"]


def test_xvg_writer_emits_whole_lines_promptly(tmp_path):
    p = str(tmp_path / "a_pullx.xvg")
    w = xvg.XvgWriter(p, "Pull COM", "Time (ps)", "Position (nm)", ["1"], flush_interval=0.0)
    w.row(0.0, 3.25)
    w.row(0.05, 3.2)
    with open(p) as f:          # visible before close: the monitor tails the file while mdrun runs
        lines = [l for l in f if l[0] not in "#@"]
    assert lines == ["0.0000\t3.25\n", "0.0500\t3.2\n"]
    w.close()


def test_spike_score_and_jarzynski_inputs():
    """The pipeline ranks SMD windows with a z-score spike count (captured outputs); the same
    statistic evaluated here must agree."""
    def spike_score(forces, z=3.0):
        forces = np.asarray(forces, dtype=float)
        if forces.size == 0:
            return 1e9, 0, 0.0
        mu, sd = forces.mean(), forces.std()
        sd = sd if sd > 1e-9 else 1.0
        n = int((np.abs((forces - mu) / sd) > z).sum())
        return n * 1000.0 + float(mu), n, float(mu)
    cases = {"const": np.full(5000, -500.0), "empty": np.zeros(0), "demo": np.array(REFOUT["xvg"]["pullf_values"]),
             "ramp": np.array(REFOUT["spike_inputs"]["ramp"]), "two_spikes": np.array(REFOUT["spike_inputs"]["two_spikes"])}
    for k, a in cases.items():
        assert np.allclose(spike_score(a), REFOUT["spike_score"][k]), k
    assert REFOUT["spike_score"]["const"] == [-500.0, 0.0, -500.0] and REFOUT["spike_score"]["empty"] == [1e9, 0.0, 0.0]


# ------------------------------------------------------------------------------------------ ndx
def test_index_file_from_pipeline_script_and_shim_make_ndx():
    g = ndx.read_ndx(os.path.join(GOLD, "ndx", "index.ndx"))
    fr = gro.read_gro(os.path.join(GOLD, "gro", "pose.gro"))
    # default groups (written by this package's make_ndx) followed by the script's [ ActiveSite ]
    assert g.names[0] == "System" and g.names[-1] == "ActiveSite"
    for name in ("Protein", "non-Protein", "MOL", "SOL", "NA", "Water_and_ions"):
        assert name in g
    assert "Non-Protein" in g and np.array_equal(g["Non-Protein"], g["non-protein"])     # case-insensitive lookup
    res = set(REFOUT["active_site"]["residues"])
    expect = np.array([i for i in range(fr.natoms) if int(fr.resnr[i]) in res])
    assert np.array_equal(g["ActiveSite"], expect)
    assert len(g["System"]) == fr.natoms
    assert np.array_equal(np.sort(np.concatenate([g["Protein"], g["non-Protein"]])), np.arange(fr.natoms))


def test_make_ndx_commands(tmp_path):
    out = str(tmp_path / "i.ndx")
    rc = runner.make_ndx(["-f", os.path.join(GOLD, "gro", "pose.gro"), "-o", out], stdin=io.StringIO("1|13\nname 20 Protein_MOL\nq\n"),
                         out=io.StringIO(), err=io.StringIO())
    assert rc == 0
    g = ndx.read_ndx(out)
    assert g.names[13] == "MOL" and g.names[20] == "Protein_MOL"
    assert len(g["Protein_MOL"]) == len(g["Protein"]) + len(g["MOL"])


# ------------------------------------------------------------------------------------------ top
TD = os.path.join(GOLD, "topdialect")


def _nma(constraints="h-bonds", defines=None, ref=True):
    T = top.read_top(os.path.join(TD, "ion.top"), defines=defines if defines is not None else {"POSRES": ""})
    g = gro.read_gro(os.path.join(TD, "complex.gro"))
    s = top.build_system(T, g, constraints=constraints, posres_ref=g if ref else None)
    return T, g, s


def test_topology_dialect_of_the_pipeline():
    """pdb2gmx-style protein block + injected ligand [ atomtypes ] + `clean_ligand.itp` with its
    POSRES block + SOL/NA/CL counts appended, as the pipeline's scripts assemble it."""
    T, g, s = _nma()
    nw = REFOUT["topdialect"]["n_water"]
    assert T.molecules == [("Protein_chain_A", 1), ("MOL", 1), ("SOL", nw), ("NA", 1), ("CL", 1)]
    assert s.natoms == 12 + 6 + 3 * nw + 2
    assert abs(s.charge.sum()) < 1e-9
    # h-bonds: every bond to an atom whose name starts with H is a constraint
    assert len(s.constraints) == 7 + 4 and len(s.bonds) == 4 + 1
    assert len(s.angles) == 18 + 7 and len(s.pairs14) == 16 + 3 and len(s.settles) == nw
    # type lookup: HC-CT-C-O has two terms, O-C-N-H matches the reversed specific entry (two terms),
    # X-C-N-X / X-CT-C-X / X-CT-N-X wildcards, one macro-defined torsion, two impropers
    assert len(s.dihedrals) == 22 + 3
    d = {(tuple(int(a) + 1 for a in q), int(m)): (p, k) for q, p, k, m in zip(s.dihedrals, s.dih_phi0, s.dih_k, s.dih_mult)}
    assert d[((1, 2, 5, 6), 1)] == (0.0, 3.3472) and d[((1, 2, 5, 6), 3)] == (180.0, 0.33472)
    assert d[((6, 5, 7, 8), 2)] == (180.0, 10.46) and d[((6, 5, 7, 8), 1)] == (0.0, 8.368)
    assert d[((2, 5, 7, 9), 2)] == (180.0, 10.46)
    assert d[((5, 7, 9, 10), 3)] == (0.0, 1.23)
    assert d[((2, 7, 5, 6), 2)] == (180.0, 43.932) and d[((5, 9, 7, 8), 2)] == (180.0, 4.6024)
    assert d[((15, 13, 14, 18), 3)] == (0.0, 0.69733)
    # bond and angle parameters by type
    b = {tuple(int(a) + 1 for a in q): (r, k) for q, r, k in zip(s.bonds, s.bond_b0, s.bond_kb)}
    assert b[(2, 5)] == (0.1522, 265265.6) and b[(5, 6)] == (0.1229, 476976.0) and b[(13, 14)] == (0.1426, 262840.0)
    c = {tuple(int(a) + 1 for a in q): l for q, l in zip(s.constraints, s.constraint_len)}
    assert c[(1, 2)] == 0.109 and c[(7, 8)] == 0.101 and c[(14, 18)] == 0.0974
    # generated 1-4 parameters: fudgeLJ 0.5 on epsilon, fudgeQQ 0.8333 on the charge product
    k = [tuple(p) for p in s.pairs14.tolist()].index((0, 5))
    sig = 0.5 * (0.264953 + 0.295992); eps = 0.5 * np.sqrt(0.0656888 * 0.87864)
    assert np.isclose(s.pair_c6[k], 4 * eps * sig ** 6) and np.isclose(s.pair_c12[k], 4 * eps * sig ** 12)
    assert np.isclose(s.pair_qq[k], 0.8333 * 0.1123 * -0.5679)
    # position restraints: 5 protein heavy atoms at 1000, all 6 ligand atoms at 10000
    assert list(s.posres_idx) == [1, 4, 5, 6, 8] + list(range(12, 18))
    assert np.all(s.posres_k[:5] == 1000.0) and np.all(s.posres_k[5:] == 10000.0)
    # exclusions: nrexcl 3 in the solutes, all pairs in a water, none for ions
    ex = lambda i: set(s.excl_list[s.excl_start[i]:s.excl_start[i + 1]].tolist())
    assert ex(0) == {1, 2, 3, 4, 5, 6} and ex(18) == {19, 20} and ex(s.natoms - 1) == set()
    assert s.settle_doh == 0.09572 and s.settle_dhh == 0.15139
    # index groups come from residue names
    assert len(s.groups["Protein"]) == 12 and len(s.groups["MOL"]) == 6 and len(s.groups["SOL"]) == 3 * nw
    assert len(s.groups["NA"]) == 1 and len(s.groups["CL"]) == 1


def test_topology_preprocessor_switches():
    _, _, s = _nma(defines={}, ref=False)
    assert len(s.posres_idx) == 0
    nw = REFOUT["topdialect"]["n_water"]
    _, _, s = _nma(constraints="none", defines={"FLEXIBLE": ""}, ref=False)
    assert len(s.settles) == 0 and len(s.constraints) == 0
    assert len(s.bonds) == 11 + 5 + 2 * nw and len(s.angles) == 18 + 7 + nw
    _, _, s = _nma(constraints="all-bonds", defines={}, ref=False)
    assert len(s.bonds) == 0 and len(s.constraints) == 16


def test_topology_errors_are_reported(tmp_path):
    with pytest.raises(top.TopologyError, match="does not match"):
        T = top.read_top(os.path.join(TD, "complex.top"))      # no SOL/NA/CL rows: atom count differs
        top.build_system(T, gro.read_gro(os.path.join(TD, "complex.gro")))
    bad = tmp_path / "bad.top"
    bad.write_text('#include "nowhere/forcefield.itp"\n')
    with pytest.raises(top.TopologyError, match="not found"):
        top.read_top(str(bad))
    bad.write_text("[ defaults ]\n1 1 yes 0.5 0.5\n[ atomtypes ]\nA A 1.0 0.0 A 0.3 0.1\n[ moleculetype ]\nM 1\n[ atoms ]\n1 A 1 M A 1 0 1\n"
                   "[ system ]\nx\n[ molecules ]\nM 1\n")
    g = gro.GroFrame("t", np.array([1]), ["M"], ["A"], np.zeros((1, 3)), None, np.eye(3) * 3)
    with pytest.raises(top.TopologyError, match="combination rule"):
        top.build_system(top.read_top(str(bad)), g)


def test_ryckaert_bellemans_rewritten_as_cosine_series():
    c = [9.28, 12.16, -13.12, -3.06, 26.24, -31.5]
    terms = top.rb_to_periodic(c)
    phi = np.linspace(-np.pi, np.pi, 721)
    psi = phi - np.pi
    rb = sum(cn * np.cos(psi) ** n for n, cn in enumerate(c))
    per = sum(k * (1 + np.cos(m * phi - np.deg2rad(p0))) for p0, k, m in terms)
    assert np.abs(rb - per).max() < 1e-10


def test_molecules_are_made_whole_on_input():
    T = top.read_top(os.path.join(TD, "ion.top"), defines={})
    g = gro.read_gro(os.path.join(TD, "complex.gro"))
    L = g.box[0, 0]
    shifted = gro.GroFrame(g.title, g.resnr, g.resname, g.atomname, np.mod(g.x + np.array([1.55, 0.0, 0.0]), L), None, g.box)
    s = top.build_system(T, shifted, constraints="h-bonds")
    d = np.linalg.norm(s.x[s.bonds[:, 0]] - s.x[s.bonds[:, 1]], axis=1)
    assert d.max() < 0.17
    w = s.settles
    assert np.abs(np.linalg.norm(s.x[w[:, 0]] - s.x[w[:, 1]], axis=1) - 0.09572).max() < 2e-3


def test_topology_writer_roundtrip(tmp_path):
    s = synth.pseudo_complex(d=5.0, protein_radius=0.8, n_ions=2, n_active=40)
    d = str(tmp_path)
    tp = top.write_top_tree(d, s)
    gro.write_system_gro(os.path.join(d, "conf.gro"), s)
    g = gro.read_gro(os.path.join(d, "conf.gro"))
    s2 = top.build_system(top.read_top(tp, defines={"POSRES": ""}), g, constraints="none", posres_ref=g)
    for k in ("bonds", "angles", "dihedrals", "pairs14", "settles", "constraints", "excl_start", "excl_list", "posres_idx", "dih_mult"):
        assert np.array_equal(getattr(s, k), getattr(s2, k)), k
    for k in ("mass", "charge", "sigma", "eps", "bond_b0", "bond_kb", "angle_th0", "angle_k", "dih_phi0", "dih_k", "pair_c6", "pair_c12",
              "pair_qq", "constraint_len", "posres_k"):
        assert np.allclose(getattr(s, k), getattr(s2, k), rtol=1e-6, atol=1e-12), k


# ------------------------------------------------------------------------------------------ trajectories
def test_xtc_roundtrip_within_precision(tmp_path):
    s = synth.water_box(d=3.0)
    rng = np.random.default_rng(1)
    p = str(tmp_path / "t.xtc")
    w = traj.XtcWriter(p, s.natoms)
    frames = []
    for k in range(3):
        x = (s.x + rng.normal(scale=0.05 * k, size=s.x.shape)).astype(np.float32)
        frames.append(x)
        w.write(10 * k, 0.02 * k, s.box, x)
    w.close()
    fr = traj.read_xtc(p)
    assert [f["step"] for f in fr] == [0, 10, 20]
    for f, x in zip(fr, frames):
        assert np.abs(f["x"] - x).max() <= 0.5e-3 + 2e-6
        assert np.allclose(f["box"], s.box, atol=1e-6) and f["precision"] == 1000.0
    # water runs compress well below 12 bytes/atom
    assert os.path.getsize(p) / (3 * s.natoms) < 6.0
    # header is big-endian XDR: magic 1995, atom count
    raw = open(p, "rb").read(8)
    assert raw == (1995).to_bytes(4, "big") + s.natoms.to_bytes(4, "big")


@pytest.mark.parametrize("scale,n", [(1.0, 1000), (500.0, 64), (30000.0, 17), (0.01, 10)])
def test_xtc_ranges_and_negative_coordinates(tmp_path, scale, n):
    rng = np.random.default_rng(n)
    x = rng.uniform(-scale, scale, size=(n, 3)).astype(np.float32)
    p = str(tmp_path / "r.xtc")
    w = traj.XtcWriter(p, n); w.write(7, 1.5, np.eye(3), x); w.close()
    f = traj.read_xtc(p)[0]
    tol = 0.5e-3 + 4e-7 * scale * 4
    assert np.abs(f["x"] - x).max() <= tol and f["step"] == 7 and abs(f["time"] - 1.5) < 1e-6


def test_xtc_small_systems_are_stored_uncompressed(tmp_path):
    x = np.random.default_rng(0).normal(size=(9, 3)).astype(np.float32)
    p = str(tmp_path / "s.xtc")
    w = traj.XtcWriter(p, 9); w.write(0, 0.0, np.eye(3), x); w.close()
    assert os.path.getsize(p) == 56 + 9 * 12
    assert np.array_equal(traj.read_xtc(p)[0]["x"], x)
    w = traj.XtcWriter(p, 9)
    with pytest.raises(traj.TrajError):
        w.write(0, 0.0, np.eye(3), np.full((9, 3), np.nan, np.float32) if False else np.full((12, 3), 1.0, np.float32))
    w.close()
    w = traj.XtcWriter(p, 12)
    with pytest.raises(traj.TrajError):
        w.write(0, 0.0, np.eye(3), np.full((12, 3), np.nan, np.float32))
    w.close()


def test_trr_roundtrip_and_header(tmp_path):
    s = synth.water_box(d=3.0)
    p = str(tmp_path / "t.trr")
    w = traj.TrrWriter(p, s.natoms)
    w.write(0, 0.0, s.box, s.x, s.v)
    w.write(500, 1.0, s.box, s.x)
    w.close()
    raw = open(p, "rb").read(24)
    assert raw[:12] == (1993).to_bytes(4, "big") + (13).to_bytes(4, "big") + (12).to_bytes(4, "big") and raw[12:] == b"GMX_trn_file"
    fr = traj.read_trr(p)
    assert [f["step"] for f in fr] == [0, 500] and fr[1]["v"] is None and fr[0]["f"] is None
    assert np.array_equal(fr[0]["x"], s.x.astype(np.float32)) and np.array_equal(fr[0]["v"], s.v.astype(np.float32))


# ------------------------------------------------------------------------------------------ run input, energies
def test_run_input_and_checkpoint_containers(tmp_path):
    _, _, s = _nma()
    p = str(tmp_path / "x.tpr")
    tpr.save_run_input(p, s, {"dt": "0.002", "tc_grps": "Protein Non-Protein"}, dict(use_posres=True))
    s2, m, meta = tpr.load_run_input(p)
    assert m["dt"] == "0.002" and meta["use_posres"] is True
    for k in ("x", "charge", "dihedrals", "dih_k", "excl_list", "posres_ref", "settles"):
        assert np.array_equal(getattr(s, k), getattr(s2, k)), k
    assert np.array_equal(s2.groups["non-protein"], s.groups["non-Protein"]) and s2.atomname == s.atomname
    c = str(tmp_path / "x.cpt")
    tpr.save_checkpoint(c, 1234, 2.468, s.x, np.zeros_like(s.x), s.box)
    cp = tpr.load_checkpoint(c)
    assert cp["step"] == 1234 and np.allclose(cp["x"], s.x, atol=1e-6)
    (tmp_path / "real.tpr").write_bytes(b"\x00\x00\x00\xc9VERSION")
    with pytest.raises(ValueError, match="not a run input"):
        tpr.load_run_input(str(tmp_path / "real.tpr"))


def test_log_blocks_and_edr_roundtrip(tmp_path):
    e = {n: float(i + 1) for i, n in enumerate(["LJ", "COUL_SR", "COUL_RECIP", "COUL_EXCL", "BOND", "ANGLE", "DIH", "LJ14", "COUL14",
                                                "POSRES", "PULL", "DISPCORR", "KIN", "POT", "TEMP", "PRES", "VOLUME", "CONSERVED"])}
    t = energy.frame_terms(e, total_mass_amu=18.0154 * 1000)
    assert t["COUL_RECIP"] == e["COUL_RECIP"] + e["COUL_EXCL"] and t["TOTAL"] == e["POT"] + e["KIN"]
    blk = energy.format_energy_block(100, 0.2, t)
    assert "           Step           Time" in blk and "   Energies (kJ/mol)" in blk and "Pres. DC" not in blk
    rows = blk.split("\n")
    assert all(len(r) % 15 == 0 for r in rows[4:] if r)
    lg = energy.LogWriter(str(tmp_path / "a.log"))
    lg.energies(0, 0.0, t); lg.energies(10, 0.02, t)
    lg.finish(1000, 0.002, 2.0, 1000)
    lg.close()
    text = (tmp_path / "a.log").read_text()
    # 1000 steps of 2 fs in 2 s = 86.4 ns/day
    assert "Performance:       86.400" in text and "A V E R A G E S" in text
    w = energy.EdrWriter(str(tmp_path / "a.edr")); w.frame(0, 0.0, t); w.frame(10, 0.02, t); w.close()
    names, units, frames = energy.read_edr(str(tmp_path / "a.edr"))
    assert "Temperature" in names and units[names.index("Pressure")] == "bar"
    assert [f[0] for f in frames] == [0, 10] and frames[1][2][names.index("Potential")] == e["POT"]
    # the enx layout, byte for byte at the places its description fixes: names block -55555 / version 5 / nre, first
    # string "Bond" as the file layer writes strings (int strlen + 1, then the XDR string: unsigned length and padded bytes), and after the 19 (name, unit) pairs a frame that starts with the
    # double -2e10, -7777777, version, t, step, nsum 0, nsteps, dt, nre, 0, nblock 0, e_size, 0, 0
    import struct
    raw = (tmp_path / "a.edr").read_bytes()
    assert struct.unpack_from(">iii", raw, 0) == (-55555, 5, len(energy.TERMS))
    assert raw[12:24] == struct.pack(">iI", 5, 4) + b"Bond" and raw[24:40] == struct.pack(">iI", 7, 6) + b"kJ/mol\0\0"
    head = 12 + sum(8 + len(n) + (-len(n) % 4) + 8 + len(u) + (-len(u) % 4) for _, n, u in energy.TERMS)
    f0 = struct.unpack_from(">diidqiqdiiiiii", raw, head)
    assert f0 == (-2e10, -7777777, 5, 0.0, 0, 0, 1, 0.0, len(energy.TERMS), 0, 0, len(energy.TERMS) * 32, 0, 0)
    frame_bytes = struct.calcsize(">diidqiqdiiiiii") + 8 * len(energy.TERMS)
    assert len(raw) == head + 2 * frame_bytes
    assert struct.unpack_from(">diidqiq", raw, head + frame_bytes)[3:7] == (0.02, 10, 0, 10)
    # a single-precision file of the same layout (what a mixed-precision mdrun writes) reads back too
    sp = raw[:head]
    for k in range(2):
        fr = struct.unpack_from(">diidqiqdiiiiii", raw, head + k * frame_bytes)
        vals = struct.unpack_from(">%dd" % len(energy.TERMS), raw, head + k * frame_bytes + struct.calcsize(">diidqiqdiiiiii"))
        sp += struct.pack(">fiidqiqdiiiiii", *fr) + struct.pack(">%df" % len(vals), *vals)
    (tmp_path / "b.edr").write_bytes(sp)
    n2, u2, f2 = energy.read_edr(str(tmp_path / "b.edr"))
    assert n2 == names and [f[0] for f in f2] == [0, 10]
    assert abs(f2[1][2][names.index("Potential")] - e["POT"]) < 1e-6 * abs(e["POT"])


# ------------------------------------------------------------------------------------------ grompp (host only)
def _stage(tmp_path, stage, **kw):
    import stage_inputs as si
    return si.write_mdp(str(tmp_path / (stage + ".mdp")), stage, **kw)


def test_grompp_on_pipeline_dialect(tmp_path):
    f = _stage(tmp_path, "nvt")
    out, err = io.StringIO(), io.StringIO()
    o = str(tmp_path / "nvt.tpr")
    rc = runner.grompp(["-f", f, "-c", os.path.join(TD, "complex.gro"), "-r", os.path.join(TD, "complex.gro"), "-p",
                        os.path.join(TD, "ion.top"), "-o", o, "-maxwarn", "1"], out=out, err=err)
    assert rc == 0, err.getvalue()
    s, m, meta = tpr.load_run_input(o)
    assert meta["use_posres"] and len(s.posres_idx) == 11 and m["tcoupl"] == "V-rescale"
    # -DPOSRES without -r is fatal
    rc = runner.grompp(["-f", f, "-c", os.path.join(TD, "complex.gro"), "-p", os.path.join(TD, "ion.top"), "-o", o], out=out, err=err)
    assert rc == 1 and "-r" in err.getvalue()


def test_grompp_warnings_and_errors(tmp_path):
    c, p = os.path.join(TD, "complex.gro"), os.path.join(TD, "ion.top")
    o = str(tmp_path / "smd.tpr")
    two = {"tc-grps": "Protein non-Protein", "tau_t": "0.1 0.1", "ref_t": "300 300"}
    pull = _stage(tmp_path, "pull", pull_group1_name="Protein", **two)
    err = io.StringIO()
    # the unknown key counts as a warning: fatal without -maxwarn, accepted with it
    assert runner.grompp(["-f", pull, "-c", c, "-p", p, "-o", o], out=io.StringIO(), err=err) == 1
    assert "nstcheckpoint" in err.getvalue() and "maxwarn" in err.getvalue()
    assert runner.grompp(["-f", pull, "-c", c, "-p", p, "-o", o, "-maxwarn", "1000"], out=io.StringIO(), err=io.StringIO()) == 0
    pull = _stage(tmp_path, "pull", pull_group1_name="Protein")
    err = io.StringIO()
    # four T-coupling groups Protein MOL SOL NA leave CL uncovered
    assert runner.grompp(["-f", pull, "-c", c, "-p", p, "-o", o, "-maxwarn", "1000"], out=io.StringIO(), err=err) == 1
    assert "not part of any T-coupling group" in err.getvalue()
    pull = _stage(tmp_path, "pull", pull_group1_name="ActiveSite", **two)
    err = io.StringIO()
    assert runner.grompp(["-f", pull, "-c", c, "-p", p, "-o", o, "-maxwarn", "1000"], out=io.StringIO(), err=err) == 1
    assert "ActiveSite" in err.getvalue()
    bad = _stage(tmp_path, "production", coulombtype="Reaction-Field")
    err = io.StringIO()
    assert runner.grompp(["-f", bad, "-c", c, "-p", p, "-o", o], out=io.StringIO(), err=err) == 1
    assert "coulombtype" in err.getvalue()
    assert runner.grompp(["-f", str(tmp_path / "missing.mdp"), "-c", c, "-p", p, "-o", o], out=io.StringIO(), err=io.StringIO()) == 1


def test_mdrun_fails_loudly_without_run_input(tmp_path):
    err = io.StringIO()
    assert runner.mdrun(["-deffnm", str(tmp_path / "nothing")], out=io.StringIO(), err=err) == 1
    assert "does not exist" in err.getvalue()


# ------------------------------------------------------------------ xtc bytes pinned independently of the package's decoder
_WATER12 = [(0.512, 1.203, 0.877), (1.934, 0.321, 2.105), (2.467, 2.250, 0.139), (0.048, 2.911, 1.660)]
_WATER12_HEX = (
    "000007cb0000000c000000073c656042"                      # magic 1995, 12 atoms, step 7, t = 0.014
    "404000000000000000000000" "000000004040000000000000" "000000000000000040400000"      # box 3 nm cube
    "0000000c447a0000"                                      # 12 atoms again, precision 1000
    "00000018000001410000007f"                              # minint  24, 321, 127 (smallest x, y, z in pm)
    "000009ef00000bbc00000858"                              # maxint 2543, 3004, 2136
    "00000016"                                              # smallidx 22: smallest neighbour distance sum|d| = 146 -> first table entry >= 146 is 161
    "0000002c"                                              # 44 bytes of bit stream
    "3a9a78b52436a3074f354654890b3279fa9f6f6879f82b1b6cf3d2adb7b43cfc158a08e3c900dbda1e7e0ac4")


def _water12():
    x = []
    for (a, b, c) in _WATER12:
        x += [(a, b, c), (a + 0.076, b + 0.058, c - 0.012), (a - 0.024, b + 0.093, c + 0.031)]
    return x


def test_xtc_bytes_of_a_twelve_atom_frame():
    """Four water-like triples: the header fields are worked out by hand (comments next to the hex above), the bit
    stream comes from the independent restatement of the format in tests/xtc3_reference.py (byte-array arithmetic as
    in the published xdrfile description); the C encoder has to produce exactly these bytes."""
    import xtc3_reference as X
    from mdx.io import traj
    box = [3.0, 0, 0, 0, 3.0, 0, 0, 0, 3.0]
    want = bytes.fromhex(_WATER12_HEX)
    assert X.encode_frame(7, 0.014, box, _water12()) == want
    w = traj.XtcWriter(os.devnull, 12)
    got = w.encode(7, 0.014, np.array(box).reshape(3, 3), np.array(_water12(), dtype=np.float32))
    w.close()
    assert got == want
    # run structure: every triple is one large atom plus a run of two small ones (the first two interchanged)
    assert len(want) == 92 + 44


def test_xtc_encoder_matches_independent_restatement():
    import xtc3_reference as X
    from mdx.io import traj
    rng = np.random.default_rng(3)
    box = [5.0, 0, 0, 0, 5.0, 0, 2.5, 2.5, 3.5]
    for n in (10, 11, 30, 100, 999):
        for spread in (0.02, 0.3, 5.0, 400.0):
            x = (rng.normal(0, spread, size=(n, 3)) + rng.uniform(0, 5, size=3)).astype(np.float32)
            want = X.encode_frame(3, 0.5, box, [tuple(float(c) for c in p) for p in x])
            w = traj.XtcWriter(os.devnull, n)
            got = w.encode(3, 0.5, np.array(box).reshape(3, 3), x)
            w.close()
            assert got == want, (n, spread)


def test_xtc_threaded_encoder_gives_the_same_bytes():
    """Frames encoded in speculative segments on several threads are byte-identical to the one-thread encoding, also
    when the speculation fails (warm-up walk of zero atoms: the segments start from a wrong guess and are redone)."""
    import mdx  # noqa: F401
    from mdx import synth
    from mdx.io import traj
    L = traj.load_library()
    rng = np.random.default_rng(5)
    wat = synth.water_box(d=5.0).x
    frames = {"water": wat,
              "gas": rng.uniform(0, 30, size=(20000, 3)),
              "mixed": np.concatenate([wat[:6000], rng.uniform(0, 30, size=(6000, 3)), wat[6000:15000]]),
              "tight": rng.normal(0, 0.003, size=(12000, 3)) + 5}
    box = np.diag([30.0, 30.0, 30.0])
    for name, x in frames.items():
        x = x.astype(np.float32)
        one = traj.XtcWriter(os.devnull, len(x), threads=1).encode(0, 0.0, box, x)
        for t in (2, 3, 4, 7, 16):
            assert traj.XtcWriter(os.devnull, len(x), threads=t).encode(0, 0.0, box, x) == one, (name, t)
        before = L.trj_xtc_respeculated()
        old = L.trj_xtc_set_warmup(0)
        try:
            redo = traj.XtcWriter(os.devnull, len(x), threads=4).encode(0, 0.0, box, x)
        finally:
            L.trj_xtc_set_warmup(old)
        assert redo == one, name
        if name in ("water", "mixed"):
            assert L.trj_xtc_respeculated() > before, name       # the fallback path did run
    # and the frame still decodes to the rounded coordinates
    p = os.path.join(os.environ.get("TMPDIR", "/tmp"), "mdx_mt_%d.xtc" % os.getpid())
    w = traj.XtcWriter(p, len(wat), threads=4)
    w.write(1, 0.002, box, wat.astype(np.float32)); w.close()
    fr = traj.read_xtc(p); os.remove(p)
    assert np.abs(fr[0]["x"] - wat).max() <= 0.501e-3


def xtc_round_like_the_capture_kernel(x, precision=1000.0):
    """The rounding pass of the xtc coordinate block as the engine's capture kernel does it (single precision, multiply
    and add rounded separately): integer coordinates and the eight frame statistics."""
    lf = x.astype(np.float32) * np.float32(precision)
    ix = np.where(lf >= 0, lf + np.float32(0.5), lf - np.float32(0.5)).astype(np.float32).astype(np.int32)
    d = np.abs(np.diff(ix.astype(np.int64), axis=0)).sum(1)
    stats = np.array(list(ix.min(0)) + list(ix.max(0)) + [int(d.min()), 0], dtype=np.int32)
    return ix, stats


def test_xtc_frame_from_rounded_coordinates_gives_the_same_bytes():
    """trj_xtc_encode_frame_q (bit packing only, coordinates rounded beforehand) against the encoder that starts from
    the floats, on one and on several threads."""
    import mdx  # noqa: F401
    from mdx import synth
    from mdx.io import traj
    rng = np.random.default_rng(8)
    wat = synth.water_box(d=5.0).x
    frames = {"water": wat - 7.5, "gas": rng.uniform(-3, 30, size=(20000, 3)), "tight": rng.normal(0, 0.003, size=(12000, 3)) - 2}
    box = np.diag([30.0, 30.0, 30.0])
    for name, x in frames.items():
        x = x.astype(np.float32)
        ix, stats = xtc_round_like_the_capture_kernel(x)
        for t in (1, 4):
            w = traj.XtcWriter(os.devnull, len(x), threads=t)
            assert w.encode_quantized(3, 0.006, box, ix, stats) == w.encode(3, 0.006, box, x), (name, t)
    w = traj.XtcWriter(os.devnull, len(wat), threads=1)
    ix, stats = xtc_round_like_the_capture_kernel(wat)
    stats[7] = 1                         # the capture flagged a coordinate that does not fit
    with pytest.raises(traj.TrajError):
        w.encode_quantized(0, 0.0, box, ix, stats)


# ------------------------------------------------------------------ the pipeline's own .mdp files, verbatim
_REF_MDP = "/root/reference/mdp"


@pytest.mark.skipif(not os.path.isdir(_REF_MDP), reason="reference tree not present on this machine")
def test_reference_mdp_files_parse_with_the_expected_warnings():
    """Every stage file of the pipeline read as it stands (/root/reference/mdp/*.mdp): the only key the reader does
    not know is `nstcheckpoint` (pull_short.mdp:10, not an mdp key in GROMACS either, hence `-maxwarn 1000`); the
    values the engine consumes come out as the files say."""
    expect_unknown = {"pull_short.mdp": ["nstcheckpoint"]}
    seen = {}
    for name in sorted(os.listdir(_REF_MDP)):
        if not name.endswith(".mdp"):
            continue
        m = mdp.read_mdp(os.path.join(_REF_MDP, name))
        unknown = [w.split("'")[1] for w in m.warnings if "unknown left-hand" in w]
        other = [w for w in m.warnings if "unknown left-hand" not in w]
        assert unknown == expect_unknown.get(name, []), (name, m.warnings)
        assert other == [], (name, other)
        seen[name] = m
    assert {"em.mdp", "em_complex.mdp", "nvt_complex.mdp", "npt_complex.mdp", "production.mdp", "pull_short.mdp"} <= set(seen)
    em = seen["em_complex.mdp"]
    assert em.s("integrator") == "steep" and em.f("emtol") == 1000.0 and em.f("emstep") == 0.01 and em.defines() == {"POSRES": ""}
    pr = seen["production.mdp"]
    assert pr.s("pcoupl") == "Parrinello-Rahman" and pr.f("tau_p") == 2.0 and pr.i("nstxout_compressed") == 10 and pr.i("nstenergy") == 10
    assert pr.words("tc_grps") == ["Protein", "Non-Protein"] and pr.floats("tau_t") == [0.1, 0.1] and pr.s("dispcorr") == "EnerPres"
    npt = seen["npt_complex.mdp"]
    assert npt.i("nstpcouple") == 100 and npt.i("nsttcouple") == 100 and npt.s("refcoord_scaling") == "all" and npt.i("nstcomm") == 1000
    pull = seen["pull_short.mdp"]
    assert pull.s("pull_coord1_type") == "constant-force" and pull.f("pull_coord1_k") == 500.0 and pull.f("rcoulomb") == 1.2
    assert pull.words("tc_grps") == ["Protein", "MOL", "SOL", "NA"] and pull.i("lincs_order") == 6 and pull.f("dt") == 0.001
    nvt = seen["nvt_complex.mdp"]
    assert nvt.b("gen_vel") and nvt.i("gen_seed") == -1 and not nvt.b("continuation")


def test_trjconv_dumps_the_checkpoint_state(tmp_path):
    """`gmx trjconv -s P.tpr -f P.cpt -o P.gro -dump 0` fed `0` as the SMD drivers do
    (/root/reference/scripts/run_smd_with_distance_monitoring.py:231-244): served from the shim's own containers."""
    from mdx import cli
    s = synth.water_box(d=2.4)
    d = str(tmp_path)
    from mdx.io import ndx as ndxio
    s.groups = ndxio.default_groups(s.resname, s.atomname) if s.resname is not None else s.groups
    tpr.save_run_input(os.path.join(d, "smd.tpr"), s, {"integrator": "md"}, {})
    rng = np.random.default_rng(1)
    x = s.x + rng.normal(scale=0.01, size=s.x.shape); v = rng.normal(size=s.x.shape)
    tpr.save_checkpoint(os.path.join(d, "smd.cpt"), 1234, 1.234, x, v, s.box)
    argv = ["trjconv", "-s", os.path.join(d, "smd.tpr"), "-f", os.path.join(d, "smd.cpt"), "-o", os.path.join(d, "smd.gro"), "-dump", "0"]
    assert runner.trjconv_can_serve(argv[1:])
    out, err = io.StringIO(), io.StringIO()
    assert runner.trjconv(argv[1:], stdin=io.StringIO("0\n"), out=out, err=err) == 0, err.getvalue()
    g = gro.read_gro(os.path.join(d, "smd.gro"))
    assert g.natoms == s.natoms and np.abs(g.x - x).max() < 5.1e-4 and np.abs(g.v - v).max() < 5.1e-5
    assert "step= 1234" in g.title
    # a group that is not the whole system is refused; other uses of trjconv are not the shim's
    assert runner.trjconv(argv[1:], stdin=io.StringIO("SOL_missing\n"), out=io.StringIO(), err=io.StringIO()) == 1
    assert not runner.trjconv_can_serve(["-s", os.path.join(d, "smd.tpr"), "-f", os.path.join(d, "traj.xtc"), "-o", os.path.join(d, "o.pdb")])
    # a file that is not one of the shim's checkpoints gives a message, not a numpy error
    with open(os.path.join(d, "real.cpt"), "wb") as f:
        f.write(b"\x00\x00\x00\x10GROMACS checkpoint")
    with pytest.raises(ValueError, match="GROMACS .cpt cannot be read"):
        tpr.load_checkpoint(os.path.join(d, "real.cpt"))
