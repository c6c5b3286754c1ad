"""Index (`.ndx`) files and the default groups the `.mdp` files refer to by name.

Format: `[ name ]` followed by 1-based atom numbers with arbitrary whitespace and line breaks
(/root/reference/scripts/add_active_site_group.py:80-88 appends `[ ActiveSite ]`, 15 numbers
per line, width 6). Group names are matched case-insensitively, which is what lets
`tc-grps = Protein Non-Protein` (/root/reference/mdp/nvt_complex.mdp:31) find the group the
index generator spells `non-Protein`.
"""
from __future__ import annotations

import numpy as np

PROTEIN_RESIDUES = {
    "ALA", "ARG", "ASN", "ASP", "ASPH", "ASH", "CYS", "CYS2", "CYX", "CYM", "CYSH", "GLN", "GLU",
    "GLUH", "GLH", "GLY", "HIS", "HIE", "HID", "HIP", "HISA", "HISB", "HISH", "HIS1", "HSD", "HSE",
    "HSP", "ILE", "LEU", "LYS", "LYN", "LYSH", "MET", "MSE", "PHE", "PRO", "SER", "THR", "TRP",
    "TYR", "VAL", "ACE", "NME", "NH2", "NAC", "ORN", "DAB", "HYP", "PGLU", "QLN",
}
# amber terminal residues carry an N/C prefix (NALA, CVAL ...)
PROTEIN_RESIDUES |= {p + r for p in "NC" for r in list(PROTEIN_RESIDUES) if len(r) == 3}
WATER_RESIDUES = {"SOL", "WAT", "HOH", "HO4", "TIP3", "TIP4", "TIP5", "SPC", "T3P", "T4P", "T5P"}
ION_RESIDUES = {"NA", "NA+", "CL", "CL-", "K", "K+", "CA", "CA2+", "MG", "ZN", "LI", "RB", "CS",
                "F", "BR", "I", "CU", "CU1", "CU2", "SOD", "CLA", "POT", "CAL"}
BACKBONE_ATOMS = {"N", "CA", "C"}
MAINCHAIN_ATOMS = {"N", "CA", "C", "O", "O1", "O2", "OC1", "OC2", "OT", "OXT"}


class IndexGroups:
    """Ordered name -> 0-based int32 atom indices, with case-insensitive lookup."""

    def __init__(self):
        self.names: list[str] = []
        self.atoms: list[np.ndarray] = []

    def add(self, name: str, idx) -> None:
        self.names.append(name)
        self.atoms.append(np.asarray(idx, dtype=np.int32))

    def find(self, name: str):
        low = name.lower()
        hit = [i for i, n in enumerate(self.names) if n.lower() == low]
        return hit[0] if hit else None

    def __contains__(self, name):
        return self.find(name) is not None

    def __getitem__(self, name) -> np.ndarray:
        i = self.find(name)
        if i is None:
            raise KeyError(f"index group '{name}' not found; available: {', '.join(self.names)}")
        return self.atoms[i]

    def as_dict(self) -> dict:
        return {n: a for n, a in zip(self.names, self.atoms)}

    def __len__(self):
        return len(self.names)


def parse_ndx(text: str) -> IndexGroups:
    g = IndexGroups()
    name, cur = None, []
    for raw in text.splitlines():
        line = raw.split(";", 1)[0].strip()
        if not line:
            continue
        if line.startswith("["):
            if name is not None:
                g.add(name, np.asarray(cur, dtype=np.int64) - 1)
            name = line.strip("[]").strip()
            cur = []
        else:
            if name is None:
                raise ValueError("index file has atom numbers before the first [ group ] header")
            cur.extend(int(t) for t in line.split())
    if name is not None:
        g.add(name, np.asarray(cur, dtype=np.int64) - 1)
    return g


def read_ndx(path: str) -> IndexGroups:
    with open(path) as f:
        return parse_ndx(f.read())


def format_group(name: str, idx0) -> str:
    """One group in the generator's layout: 15 numbers per line, each right-aligned."""
    idx = np.asarray(idx0, dtype=np.int64) + 1
    rows = ["[ %s ]" % name]
    for s in range(0, len(idx), 15):
        rows.append(" ".join("%4d" % a for a in idx[s:s + 15]))
    return "\n".join(rows) + "\n"


def write_ndx(path: str, groups) -> None:
    items = zip(groups.names, groups.atoms) if isinstance(groups, IndexGroups) else groups.items()
    with open(path, "w") as f:
        for name, idx in items:
            f.write(format_group(name, idx))


def default_groups(resname, atomname) -> IndexGroups:
    """The groups an index generator derives from residue and atom names alone, in its order:
    System, the protein family, non-Protein, one group per non-protein residue name ('Other'
    molecules such as MOL, then ions), Water/SOL, non-Water, Ion, Water_and_ions."""
    rn = np.asarray([r.upper() for r in resname])
    an = np.asarray([a.upper() for a in atomname])
    n = len(rn)
    allidx = np.arange(n, dtype=np.int32)
    is_prot = np.isin(rn, list(PROTEIN_RESIDUES))
    is_wat = np.isin(rn, list(WATER_RESIDUES))
    is_ion = np.isin(rn, list(ION_RESIDUES)) & ~is_prot
    is_other = ~(is_prot | is_wat | is_ion)
    is_h = np.asarray([a[:1] == "H" or (a[:1].isdigit() and a[1:2] == "H") for a in an], dtype=bool) if n else np.zeros(0, bool)
    g = IndexGroups()
    g.add("System", allidx)
    if is_prot.any():
        bb = is_prot & np.isin(an, list(BACKBONE_ATOMS))
        mc = is_prot & np.isin(an, list(MAINCHAIN_ATOMS))
        cb = is_prot & (an == "CB")
        mch = mc | (is_prot & np.isin(an, ["H", "H1", "H2", "H3", "HN"]))
        g.add("Protein", allidx[is_prot])
        g.add("Protein-H", allidx[is_prot & ~is_h])
        g.add("C-alpha", allidx[is_prot & (an == "CA")])
        g.add("Backbone", allidx[bb])
        g.add("MainChain", allidx[mc])
        g.add("MainChain+Cb", allidx[mc | cb])
        g.add("MainChain+H", allidx[mch])
        g.add("SideChain", allidx[is_prot & ~mch])
        g.add("SideChain-H", allidx[is_prot & ~mch & ~is_h])
        g.add("Prot-Masses", allidx[is_prot])
        g.add("non-Protein", allidx[~is_prot])
    if is_other.any():
        g.add("Other", allidx[is_other])
        for name in _unique_in_order(rn[is_other]):
            g.add(name, allidx[rn == name])
    for name in _unique_in_order(rn[is_ion]):
        g.add(name, allidx[rn == name])
    if is_wat.any():
        g.add("Water", allidx[is_wat])
        for name in _unique_in_order(rn[is_wat]):
            if name != "WATER":
                g.add(name, allidx[rn == name])
        g.add("non-Water", allidx[~is_wat])
    if is_ion.any():
        g.add("Ion", allidx[is_ion])
    if is_wat.any() and is_ion.any():
        g.add("Water_and_ions", allidx[is_wat | is_ion])
    return g


def _unique_in_order(arr):
    seen, out = set(), []
    for a in arr:
        if a not in seen:
            seen.add(a)
            out.append(str(a))
    return out


def merge(default: IndexGroups, explicit: IndexGroups | None) -> IndexGroups:
    """With `-n` only the file's groups exist; without it the defaults do."""
    return explicit if explicit is not None and len(explicit) else default
