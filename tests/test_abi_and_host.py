"""CPU-side checks: the C-ABI library loads and exports every symbol include/mdx.h declares,
fails loudly without a GPU, and the synthetic generators produce consistent topologies."""
import ctypes
import os
import re

import numpy as np
import pytest

import mdx  # noqa: F401
from mdx import synth, engine as eng

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for h in sorted(os.listdir(inc)):
        if h.endswith(".h"):
            txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(inc, h)).read(), flags=re.S)
            names |= set(re.findall(r"\b(mdx_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(eng.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"libmdx.so does not export {n}"
    assert sorted(eng.SYMBOLS) == names


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = eng.load_library()
    h = lib.mdx_create(0, 100)
    assert not h
    assert b"no HIP device" in lib.mdx_last_error(None)
    with pytest.raises(eng.MdxError):
        eng.Engine(synth.water_box(d=3.2))


def test_water_box_geometry_and_neutrality():
    s = synth.water_box(d=3.4)
    s.validate()
    assert s.natoms % 3 == 0 and abs(s.charge.sum()) < 1e-9
    d = np.linalg.norm(s.x[0::3] - s.x[1::3], axis=1)
    assert np.allclose(d, 0.09572, atol=1e-12)
    dens = (s.natoms / 3) / abs(np.linalg.det(s.box))
    assert 30 < dens < 36
    # exclusions are symmetric
    for i in (0, 1, 2, 100):
        for j in s.excl_list[s.excl_start[i]:s.excl_start[i + 1]]:
            assert i in s.excl_list[s.excl_start[j]:s.excl_start[j + 1]]


def test_pseudo_complex_topology(small_complex):
    s = small_complex
    s.validate()
    assert abs(s.charge.sum()) < 1e-9
    assert len(s.groups["MOL"]) == 44
    assert set(s.groups) >= {"System", "Protein", "Non-Protein", "MOL", "SOL", "NA", "ActiveSite"}
    assert len(s.groups["Protein"]) + len(s.groups["Non-Protein"]) == s.natoms
    # every constraint involves a hydrogen; no flexible bond does
    isH = s.mass < 1.1
    assert (isH[s.constraints[:, 0]] | isH[s.constraints[:, 1]]).all()
    assert not (isH[s.bonds[:, 0]] | isH[s.bonds[:, 1]]).any()
    # no non-excluded LJ overlap, also across the periodic faces
    from oracle import mdref as R
    o = R.nonbonded(s.x, s.charge * 0, s.sigma, s.eps, s.box, 1.0, 3.0, s.excl_start, s.excl_list, flags=1)
    assert o["e_lj"] < 0 and np.abs(o["f"]).max() < 5e3


def test_verlet_buffer_estimate():
    """Pair-list radius from `verlet-buffer-tolerance` (/root/reference/mdp/production.mdp:23): grows with the
    list lifetime and the temperature, shrinks with a looser tolerance, and meets the tolerance it was asked for."""
    from mdx import synth, units as U
    s = synth.pseudo_complex(d=6.0)
    beta = U.ewald_beta(1.0, 1e-5)
    r10 = eng.verlet_buffer_rlist(s, 1.0, beta, 10, 0.002, 300.0)
    r20 = eng.verlet_buffer_rlist(s, 1.0, beta, 20, 0.002, 300.0)
    r40 = eng.verlet_buffer_rlist(s, 1.0, beta, 40, 0.002, 300.0)
    assert 1.0 < r10 < r20 < r40 <= 1.5
    assert 1.02 < r10 < 1.06 and 1.06 < r20 < 1.13          # water at 300 K, 2 fs
    assert eng.verlet_buffer_rlist(s, 1.0, beta, 20, 0.002, 300.0, tolerance=0.05) < r20
    assert eng.verlet_buffer_rlist(s, 1.0, beta, 20, 0.002, 400.0) > r20
    assert eng.verlet_buffer_rlist(s, 1.0, beta, 1, 0.002, 300.0) <= 1.001     # a list rebuilt every step needs no buffer
    assert eng.verlet_buffer_drift(s, 1.0, r20, beta, 20, 0.002, 300.0) <= 0.005
    assert eng.verlet_buffer_drift(s, 1.0, r20 - 0.005, beta, 20, 0.002, 300.0) > 0.005
    # constrained hydrogens swing on a short arm: their displacement saturates, a free atom's does not
    kt2 = U.BOLTZ * 300.0 * 0.1 ** 2
    var = eng._displacement_variance(s, kt2)
    h = s.settles[0, 1]
    assert var[h] < kt2 / s.mass[h] and var[h] > kt2 / s.mass[s.settles[0]].sum()


def test_rlist_and_grid_helpers():
    import importlib
    U = importlib.import_module("molecular-dynamics_amd.units")
    assert eng.verlet_rlist(1.0, 10, 0.002) > 1.0
    assert eng.verlet_rlist(1.0, 40, 0.002) > eng.verlet_rlist(1.0, 10, 0.002)
    g = U.pme_grid(synth.dodecahedron_box(12.5), 0.16)
    assert g == (80, 80, 56)
    for k in g:
        m = k
        for p in (2, 3, 5, 7):
            while m % p == 0:
                m //= p
        assert m == 1


def test_gromacs_comparator_without_gromacs():
    """tools/gmx_compare.py is the script that would pin the oracle against a real GROMACS; where none is installed (every
    machine of this build) it says so and exits 0, and its mapping of `gmx energy` terms covers every energy term the
    engine reports for the interactions of production.mdp."""
    import subprocess
    import sys
    import importlib.util
    path = os.path.join(ROOT, "tools", "gmx_compare.py")
    spec = importlib.util.spec_from_file_location("gmx_compare", path)
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    ours = {t for _, names in mod.TERMS for t in names} | set(mod.COULOMB[1])
    assert {"BOND", "ANGLE", "DIH", "LJ14", "COUL14", "LJ", "DISPCORR", "POT", "COUL_SR", "COUL_EXCL", "COUL_RECIP"} <= ours
    if mod.find_gromacs() is None:
        p = subprocess.run([sys.executable, path], capture_output=True, text=True, timeout=120)
        assert p.returncode == 0 and "reference unavailable" in p.stdout
