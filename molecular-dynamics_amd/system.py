"""Flat, fully expanded description of one simulation system.

This is what the `.gro/.top/.ndx` front-end (io/top.py) and the synthetic generators
(synth.py) both produce and what the engine binding (engine.py) uploads through the C ABI.
All arrays are numpy; indices are 0-based; units are nm / ps / amu / kJ mol^-1 / e.
"""
from __future__ import annotations

from dataclasses import dataclass, field
import numpy as np


def _e(shape, dt):
    return np.zeros(shape, dtype=dt)


@dataclass
class MDSystem:
    x: np.ndarray                       # (N,3) f64
    box: np.ndarray                     # (3,3) rows = a, b, c (lower triangular)
    mass: np.ndarray                    # (N,)
    charge: np.ndarray                  # (N,)
    sigma: np.ndarray                   # (N,) per-atom LJ sigma (Lorentz-Berthelot)
    eps: np.ndarray                     # (N,) per-atom LJ epsilon
    molid: np.ndarray                   # (N,) int32 molecule index (contiguous blocks)
    v: np.ndarray | None = None         # (N,3)
    resnr: np.ndarray | None = None     # (N,) int
    resname: list | None = None         # N strings
    atomname: list | None = None        # N strings
    # exclusions as CSR over atoms (symmetric, no self)
    excl_start: np.ndarray = field(default_factory=lambda: _e(1, np.int32))
    excl_list: np.ndarray = field(default_factory=lambda: _e(0, np.int32))
    # bonded interactions
    bonds: np.ndarray = field(default_factory=lambda: _e((0, 2), np.int32))
    bond_b0: np.ndarray = field(default_factory=lambda: _e(0, np.float64))
    bond_kb: np.ndarray = field(default_factory=lambda: _e(0, np.float64))
    angles: np.ndarray = field(default_factory=lambda: _e((0, 3), np.int32))
    angle_th0: np.ndarray = field(default_factory=lambda: _e(0, np.float64))   # degrees
    angle_k: np.ndarray = field(default_factory=lambda: _e(0, np.float64))
    dihedrals: np.ndarray = field(default_factory=lambda: _e((0, 4), np.int32))  # periodic (type 1/4/9)
    dih_phi0: np.ndarray = field(default_factory=lambda: _e(0, np.float64))    # degrees
    dih_k: np.ndarray = field(default_factory=lambda: _e(0, np.float64))
    dih_mult: np.ndarray = field(default_factory=lambda: _e(0, np.int32))
    pairs14: np.ndarray = field(default_factory=lambda: _e((0, 2), np.int32))
    pair_c6: np.ndarray = field(default_factory=lambda: _e(0, np.float64))     # already fudged
    pair_c12: np.ndarray = field(default_factory=lambda: _e(0, np.float64))
    pair_qq: np.ndarray = field(default_factory=lambda: _e(0, np.float64))     # fudgeQQ*qi*qj (no 1/4pi eps0)
    # position restraints
    posres_idx: np.ndarray = field(default_factory=lambda: _e(0, np.int32))
    posres_ref: np.ndarray = field(default_factory=lambda: _e((0, 3), np.float64))
    posres_k: np.ndarray = field(default_factory=lambda: _e((0, 3), np.float64))
    # constraints
    settles: np.ndarray = field(default_factory=lambda: _e((0, 3), np.int32))   # O, H, H
    settle_doh: float = 0.09572
    settle_dhh: float = 0.15139
    constraints: np.ndarray = field(default_factory=lambda: _e((0, 2), np.int32))
    constraint_len: np.ndarray = field(default_factory=lambda: _e(0, np.float64))
    # named index groups (name -> int32 array)
    groups: dict = field(default_factory=dict)
    title: str = "mdx system"

    @property
    def natoms(self) -> int:
        return int(self.x.shape[0])

    def n_constraints_per_atom(self) -> np.ndarray:
        """Half a degree of freedom is removed from each atom of each constraint
        (SETTLE removes 3 constraints per water)."""
        nc = np.zeros(self.natoms, dtype=np.float64)
        if len(self.constraints):
            np.add.at(nc, self.constraints[:, 0], 0.5)
            np.add.at(nc, self.constraints[:, 1], 0.5)
        if len(self.settles):
            np.add.at(nc, self.settles.ravel(), 1.0)
        return nc

    def validate(self) -> None:
        n = self.natoms
        assert self.box.shape == (3, 3)
        assert abs(self.box[0, 1]) + abs(self.box[0, 2]) + abs(self.box[1, 2]) < 1e-9, \
            "box must be lower triangular (a along x, b in the xy plane)"
        for name in ("mass", "charge", "sigma", "eps", "molid"):
            assert getattr(self, name).shape == (n,), name
        assert self.excl_start.shape == (n + 1,)
        for arr in (self.bonds, self.angles, self.dihedrals, self.pairs14, self.settles,
                    self.constraints):
            if arr.size:
                assert arr.min() >= 0 and arr.max() < n


def build_exclusions(n: int, bond_like: np.ndarray, nrexcl_of_atom: np.ndarray,
                     extra_pairs: np.ndarray | None = None):
    """Exclusion CSR from the bond graph: atoms within `nrexcl` bonds exclude each other
    (the `[ moleculetype ] nrexcl` rule of the .top dialect,
    /root/reference/scripts/fix_ligand_topology.py keeps that section verbatim).
    `bond_like` holds every chemical bond, constrained or not, and SETTLE edges."""
    nbr = [[] for _ in range(n)]
    for a, b in bond_like:
        nbr[a].append(int(b))
        nbr[b].append(int(a))
    ex = [set() for _ in range(n)]
    for i in range(n):
        depth = int(nrexcl_of_atom[i])
        if depth <= 0 or not nbr[i]:
            continue
        frontier = {i}
        seen = {i}
        for _ in range(depth):
            nxt = set()
            for a in frontier:
                for b in nbr[a]:
                    if b not in seen:
                        seen.add(b)
                        nxt.add(b)
            frontier = nxt
        seen.discard(i)
        ex[i] = seen
    if extra_pairs is not None:
        for a, b in extra_pairs:
            ex[int(a)].add(int(b))
            ex[int(b)].add(int(a))
    start = np.zeros(n + 1, dtype=np.int32)
    for i in range(n):
        start[i + 1] = start[i] + len(ex[i])
    lst = np.zeros(int(start[-1]), dtype=np.int32)
    for i in range(n):
        if ex[i]:
            lst[start[i]:start[i + 1]] = sorted(ex[i])
    return start, lst
