"""Energy output: the `.log` text blocks and the `.edr` energy file of an mdrun.

`nstlog` / `nstenergy` set the cadence (/root/reference/mdp/production.mdp:10-11,
nvt_complex.mdp:10-11); the pipeline declares `PREFIX.log` and `PREFIX.edr` as rule outputs
(/root/reference/Snakefile:218,258,299-300). The `.log` follows the familiar layout (step/time
header, five right-aligned 15-column terms per row, closing `Performance:` line in ns/day and
hour/ns). The `.edr` follows the enx layout of GROMACS energy files (see the comment at `EdrWriter`).
"""
from __future__ import annotations

import struct
import time as _time

TERMS = [  # (engine key, printed name, unit)
    ("BOND", "Bond", "kJ/mol"), ("ANGLE", "Angle", "kJ/mol"), ("DIH", "Proper Dih.", "kJ/mol"),
    ("LJ14", "LJ-14", "kJ/mol"), ("COUL14", "Coulomb-14", "kJ/mol"), ("LJ", "LJ (SR)", "kJ/mol"),
    ("DISPCORR", "Disper. corr.", "kJ/mol"), ("COUL_SR", "Coulomb (SR)", "kJ/mol"),
    ("COUL_RECIP", "Coul. recip.", "kJ/mol"), ("POSRES", "Position Rest.", "kJ/mol"),
    ("PULL", "COM Pull En.", "kJ/mol"), ("POT", "Potential", "kJ/mol"), ("KIN", "Kinetic En.", "kJ/mol"),
    ("TOTAL", "Total Energy", "kJ/mol"), ("CONSERVED", "Conserved En.", "kJ/mol"), ("TEMP", "Temperature", "K"),
    ("PRES", "Pressure", "bar"), ("VOLUME", "Volume", "nm^3"), ("DENSITY", "Density", "kg/m^3"),
]


def frame_terms(e: dict, total_mass_amu: float | None = None) -> dict:
    """Engine energy dict -> printed terms. The exclusion correction and the Ewald self term are
    folded into `Coul. recip.`, which is where mdrun reports them."""
    out = dict(e)
    out["COUL_RECIP"] = e["COUL_RECIP"] + e["COUL_EXCL"]
    out["TOTAL"] = e["POT"] + e["KIN"]
    vol = e.get("VOLUME", 0.0)
    out["DENSITY"] = (total_mass_amu * 1.66053906660 / vol) if (total_mass_amu and vol > 0) else 0.0
    return out


def format_energy_block(step: int, time: float, terms: dict, skip_zero=("POSRES", "PULL", "DISPCORR")) -> str:
    rows = ["           Step           Time", "%15d%15.5f" % (step, time), "", "   Energies (kJ/mol)"]
    items = [(name, terms[key]) for key, name, _ in TERMS if key in terms and not (key in skip_zero and terms[key] == 0.0)]
    for s in range(0, len(items), 5):
        chunk = items[s:s + 5]
        rows.append("".join("%15s" % n for n, _ in chunk))
        rows.append("".join("%15.5e" % v for _, v in chunk))
    rows.append("")
    return "\n".join(rows) + "\n"


class LogWriter:
    def __init__(self, path: str, append: bool = False):
        self.f = open(path, "a" if append else "w")
        self.sums = {}
        self.count = 0

    def header(self, lines):
        self.f.write("\n".join(lines) + "\n\n")
        self.f.flush()

    def energies(self, step, time, terms):
        self.f.write(format_energy_block(step, time, terms))
        for k, v in terms.items():
            self.sums[k] = self.sums.get(k, 0.0) + v
        self.count += 1

    def note(self, text):
        self.f.write(text.rstrip("\n") + "\n")
        self.f.flush()

    def finish(self, nsteps_done: int, dt_ps: float, wall_s: float, last_step: int):
        if self.count:
            self.f.write("\t<======  ###############  ==>\n\t<====  A V E R A G E S  ====>\n\t<==  ###############  ======>\n\n")
            self.f.write("\tStatistics over %d steps using %d frames\n\n" % (last_step + 1, self.count))
            avg = {k: v / self.count for k, v in self.sums.items()}
            self.f.write(format_energy_block(last_step, last_step * dt_ps, avg).split("\n", 3)[3])
        self.f.write(performance_block(nsteps_done, dt_ps, wall_s))
        self.f.write("Finished mdrun on %s\n" % _time.strftime("%a %b %d %H:%M:%S %Y"))
        self.f.flush()

    def flush(self):
        self.f.flush()

    def close(self):
        if not self.f.closed:
            self.f.close()


def performance_block(nsteps: int, dt_ps: float, wall_s: float) -> str:
    wall_s = max(wall_s, 1e-9)
    ns = nsteps * dt_ps * 1e-3
    nsday = ns * 86400.0 / wall_s if nsteps > 0 else 0.0
    hourns = (wall_s / 3600.0) / ns if ns > 0 else 0.0
    return ("\n               Core t (s)   Wall t (s)        (%%)\n       Time: %12.3f %12.3f      100.0\n"
            "                 (ns/day)    (hour/ns)\nPerformance: %12.3f %12.3f\n" % (wall_s, wall_s, nsday, hourns))


# ---------------------------------------------------------------------------------------------- .edr (enx layout)
# The energy file of GROMACS ("enx", file version 5) as its source documents it (src/gromacs/fileio/enxio.cpp,
# do_enxnms / do_eheader / do_enx), written the way a double-precision build writes it (a reader tells the precision
# from the first real of a frame, -2e10, and `gmx energy` accepts either):
#   file head   int -55555 | int version = 5 | int nre | nre x (string name, string unit)
#   every frame real -2e10 | int -7777777 | int version | double t | int64 step | int nsum | int64 nsteps | double dt
#               | int nre | int 0 | int nblock = 0 | int e_size | int 0 | int 0 | nre x real e
#               (nsum = 0: no running sums, so a term is its value only; with nsum > 0 each term would be followed by
#               its average and sum)
# XDR: big endian. A string goes through the file layer's string routine (gmx_fio_do_string): first an int holding
# strlen + 1, then the XDR string itself, i.e. an unsigned length (strlen) and the bytes padded to a multiple of four.
# No GROMACS is installed where this was written: the layout follows the description above and is read back by
# `read_edr`; it has not been opened by `gmx energy`.
ENX_NAMES_MAGIC, ENX_FRAME_MAGIC, ENX_VERSION, ENX_FIRST_REAL = -55555, -7777777, 5, -2e10


def _xdr_string(s: str) -> bytes:
    b = s.encode()
    return struct.pack(">iI", len(b) + 1, len(b)) + b + b"\0" * (-len(b) % 4)


class EdrWriter:
    def __init__(self, path: str, append: bool = False, dt_ps: float = 0.0):
        self.keys = [k for k, _, _ in TERMS]
        self.dt = float(dt_ps)
        self.last_step = None
        self.f = open(path, "ab" if append else "wb")
        if not append or self.f.tell() == 0:
            self.f.write(struct.pack(">iii", ENX_NAMES_MAGIC, ENX_VERSION, len(TERMS)))
            for _, name, unit in TERMS:
                self.f.write(_xdr_string(name) + _xdr_string(unit))

    def frame(self, step, time, terms):
        nre = len(self.keys)
        nsteps = 1 if self.last_step is None else max(1, int(step) - self.last_step)
        self.last_step = int(step)
        self.f.write(struct.pack(">diidqiqdiiiiii", ENX_FIRST_REAL, ENX_FRAME_MAGIC, ENX_VERSION, float(time), int(step), 0, nsteps,
                                 self.dt, nre, 0, 0, nre * 8 * 4, 0, 0))
        self.f.write(struct.pack(">%dd" % nre, *[float(terms.get(k, 0.0)) for k in self.keys]))

    def flush(self):
        self.f.flush()

    def close(self):
        if not self.f.closed:
            self.f.close()


def read_edr(path: str):
    """-> (names, units, list of (step, time, values)). Reads the enx layout in either precision (frames without blocks)."""
    with open(path, "rb") as f:
        data = f.read()
    magic, version, n = struct.unpack_from(">iii", data, 0)
    if magic != ENX_NAMES_MAGIC:
        raise ValueError(f"{path}: not an energy file (the names block does not start with {ENX_NAMES_MAGIC})")
    if version > ENX_VERSION:
        raise ValueError(f"{path}: energy file version {version} is newer than {ENX_VERSION}")
    pos = 12
    names, units = [], []
    for _ in range(n):
        for dst in (names, units):
            slen, ln = struct.unpack_from(">iI", data, pos)
            if slen != ln + 1:
                raise ValueError(f"{path}: string at byte {pos}: size field {slen} does not match length {ln}")
            pos += 8
            dst.append(data[pos:pos + ln].decode())
            pos += ln + (-ln % 4)
    frames = []
    while pos + 8 <= len(data):
        if struct.unpack_from(">f", data, pos)[0] == struct.unpack(">f", struct.pack(">f", ENX_FIRST_REAL))[0]:
            real, rsz = "f", 4
        elif struct.unpack_from(">d", data, pos)[0] == ENX_FIRST_REAL:
            real, rsz = "d", 8
        else:
            raise ValueError(f"{path}: frame at byte {pos} does not start with {ENX_FIRST_REAL}")
        pos += rsz
        fmagic, fver, t, step, nsum, nsteps, dt, nre, _, nblock = struct.unpack_from(">iidqiqdiii", data, pos)
        if fmagic != ENX_FRAME_MAGIC:
            raise ValueError(f"{path}: frame magic {fmagic} at byte {pos}")
        pos += struct.calcsize(">iidqiqdiii")
        if nblock != 0:
            raise ValueError(f"{path}: frames with data blocks are not read here")
        pos += 12                                   # e_size and two reserved ints
        per = 3 if nsum > 0 else 1                  # value (+ average and sum when running sums are stored)
        vals = struct.unpack_from(">%d%s" % (nre * per, real), data, pos)
        pos += nre * per * rsz
        frames.append((step, t, list(vals[::per])))
    return names, units, frames
