"""`.gro` coordinate files: reader and writer.

Dialect as consumed and produced by the pipeline's own scripts: title line, atom count, fixed
columns resid[0:5] resname[5:10] atomname[10:15] atomnr[15:20] x[20:28] y[28:36] z[36:44]
plus three optional velocity columns, then a box line of 3 or 9 floats in the order
`xx yy zz xy xz yx yz zx zy` (/root/reference/scripts/place_pose_at_tunnel.py:27-35,77-79,
/root/reference/scripts/pose_distance_metrics.py:39-62, /root/reference/scripts/add_active_site_group.py:46-49).
Residue and atom numbers wrap at 100 000, as the five-column fields force.
"""
from __future__ import annotations

from dataclasses import dataclass
import numpy as np


@dataclass
class GroFrame:
    title: str
    resnr: np.ndarray        # (N,) int, as written (wrapped)
    resname: list
    atomname: list
    x: np.ndarray            # (N,3) nm
    v: np.ndarray | None     # (N,3) nm/ps or None
    box: np.ndarray          # (3,3) rows a, b, c

    @property
    def natoms(self):
        return len(self.x)

    def residue_index(self) -> np.ndarray:
        """0-based running residue index that survives the 100 000 wrap: a new residue
        starts wherever the (number, name) pair changes."""
        rn = np.asarray(self.resnr)
        names = np.asarray(self.resname)
        new = np.ones(len(rn), dtype=bool)
        if len(rn) > 1:
            new[1:] = (rn[1:] != rn[:-1]) | (names[1:] != names[:-1])
        return np.cumsum(new) - 1


def box_from_line(line: str) -> np.ndarray:
    """3 floats -> rectangular; 9 floats `v1x v2y v3z v1y v1z v2x v2z v3x v3y` -> rows a,b,c."""
    p = [float(t) for t in line.split()]
    if len(p) == 3:
        return np.diag(p)
    if len(p) != 9:
        raise ValueError(f"box line needs 3 or 9 numbers, got {len(p)}: '{line.strip()}'")
    xx, yy, zz, xy, xz, yx, yz, zx, zy = p
    return np.array([[xx, xy, xz], [yx, yy, yz], [zx, zy, zz]])


def box_to_line(box: np.ndarray) -> str:
    b = np.asarray(box, dtype=np.float64)
    off = [b[0, 1], b[0, 2], b[1, 0], b[1, 2], b[2, 0], b[2, 1]]
    if all(abs(t) < 5e-6 for t in off):
        return "%10.5f%10.5f%10.5f" % (b[0, 0], b[1, 1], b[2, 2])
    return "%10.5f%10.5f%10.5f%10.5f%10.5f%10.5f%10.5f%10.5f%10.5f" % (
        b[0, 0], b[1, 1], b[2, 2], b[0, 1], b[0, 2], b[1, 0], b[1, 2], b[2, 0], b[2, 1])


def parse_gro(text_lines) -> GroFrame:
    lines = text_lines
    if len(lines) < 3:
        raise ValueError("not a .gro file: fewer than three lines")
    title = lines[0].rstrip("\n")
    try:
        n = int(lines[1].split()[0])
    except (ValueError, IndexError):
        raise ValueError("could not read the atom count on line 2 of the .gro file")
    if len(lines) < n + 3:
        raise ValueError(f".gro file announces {n} atoms but holds {len(lines) - 3} atom lines")
    resnr = np.empty(n, dtype=np.int64)
    resname, atomname = [None] * n, [None] * n
    x = np.empty((n, 3))
    v = None
    if n:
        # the width of one coordinate field follows from the distance between decimal points
        first = lines[2]
        p1 = first.find(".", 20)
        p2 = first.find(".", p1 + 1)
        w = p2 - p1 if p1 > 0 and p2 > 0 else 8
        if len(first.rstrip()) >= 20 + 6 * w - 2:
            v = np.empty((n, 3))
    for i in range(n):
        ln = lines[2 + i]
        resnr[i] = int(ln[0:5])
        resname[i] = ln[5:10].strip()
        atomname[i] = ln[10:15].strip()
        for d in range(3):
            x[i, d] = float(ln[20 + d * w:20 + (d + 1) * w])
        if v is not None:
            vw = w  # velocities are written one digit more precise in the same width
            try:
                for d in range(3):
                    v[i, d] = float(ln[20 + 3 * w + d * vw:20 + 3 * w + (d + 1) * vw])
            except ValueError:
                raise ValueError(f".gro line {3 + i}: velocity columns are incomplete")
    box = box_from_line(lines[2 + n])
    return GroFrame(title, resnr, resname, atomname, x, v, box)


def read_gro(path: str) -> GroFrame:
    with open(path) as f:
        return parse_gro(f.readlines())


def format_gro(title, resnr, resname, atomname, x, box, v=None) -> str:
    n = len(x)
    x = np.asarray(x, dtype=np.float64)
    out = [title.strip() or "mdx", "%5d" % n]
    rn = np.asarray(resnr) % 100000
    an = (np.arange(n) + 1) % 100000
    if v is None:
        for i in range(n):
            out.append("%5d%-5.5s%5.5s%5d%8.3f%8.3f%8.3f" % (
                rn[i], resname[i], atomname[i], an[i], x[i, 0], x[i, 1], x[i, 2]))
    else:
        v = np.asarray(v, dtype=np.float64)
        for i in range(n):
            out.append("%5d%-5.5s%5.5s%5d%8.3f%8.3f%8.3f%8.4f%8.4f%8.4f" % (
                rn[i], resname[i], atomname[i], an[i], x[i, 0], x[i, 1], x[i, 2],
                v[i, 0], v[i, 1], v[i, 2]))
    out.append(box_to_line(box))
    return "\n".join(out) + "\n"


def write_gro(path: str, title, resnr, resname, atomname, x, box, v=None) -> None:
    tmp = path + ".tmp"
    with open(tmp, "w") as f:
        f.write(format_gro(title, resnr, resname, atomname, x, box, v))
    import os
    os.replace(tmp, path)


def write_system_gro(path: str, sysm, x=None, v=None, box=None, title=None) -> None:
    """Final-frame writer for an MDSystem (what `PREFIX.gro` holds after mdrun)."""
    n = sysm.natoms
    resnr = sysm.resnr if sysm.resnr is not None else np.ones(n, dtype=int)
    resname = sysm.resname if sysm.resname is not None else ["MOL"] * n
    atomname = sysm.atomname if sysm.atomname is not None else ["X"] * n
    write_gro(path, title or sysm.title, resnr, resname, atomname,
              sysm.x if x is None else x, sysm.box if box is None else box,
              sysm.v if v is None and x is None else v)
