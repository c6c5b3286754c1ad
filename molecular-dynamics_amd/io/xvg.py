"""`.xvg` time series: the pull output `PREFIX_pullx.xvg` / `PREFIX_pullf.xvg` and a reader.

Consumers (/root/reference/scripts/analyze_force_spikes.py:6-20,
/root/reference/scripts/run_smd_with_distance_monitoring.py:32-53) skip lines that start with
`#` or `@`, split the rest on whitespace and take column 0 as time in ps and column 1 as the
value. The distance monitor polls the file every 2 s by byte offset, so data lines must reach
the file system promptly and always as whole lines.
"""
from __future__ import annotations

import time
import numpy as np


def read_xvg(path: str):
    """-> (n_rows, n_cols) float array of the data lines; legends in a second return value."""
    rows, legends = [], []
    with open(path) as f:
        for line in f:
            s = line.strip()
            if not s:
                continue
            if s[0] == "#":
                continue
            if s[0] == "@":
                p = s.split(None, 3)
                if len(p) >= 4 and p[2] == "legend":
                    legends.append(p[3].strip('"'))
                continue
            try:
                rows.append([float(t) for t in s.split()])
            except ValueError:
                continue
    width = max((len(r) for r in rows), default=0)
    rows = [r for r in rows if len(r) == width]
    return np.asarray(rows, dtype=np.float64).reshape(len(rows), width), legends


class XvgWriter:
    """Append-only writer that emits whole lines and flushes on a wall-clock interval."""

    def __init__(self, path: str, title: str, xlabel: str, ylabel: str, legends,
                 program: str = "mdx mdrun", flush_interval: float = 1.0, append: bool = False):
        self.path = path
        self.flush_interval = flush_interval
        self._last = time.monotonic()
        self.f = open(path, "a" if append else "w")
        if not append or self.f.tell() == 0:
            h = ["# This file was created by %s" % program,
                 "# %s" % time.strftime("%a %b %d %H:%M:%S %Y"),
                 '@    title "%s"' % title,
                 '@    xaxis  label "%s"' % xlabel,
                 '@    yaxis  label "%s"' % ylabel,
                 "@TYPE xy"]
            if legends:
                h.append("@ view 0.15, 0.15, 0.75, 0.85")
                h.append("@ legend on")
                h.append("@ legend box on")
                h.append("@ legend loctype view")
                h.append("@ legend 0.78, 0.8")
                h.append("@ legend length 2")
                for i, l in enumerate(legends):
                    h.append('@ s%d legend "%s"' % (i, l))
            self.f.write("\n".join(h) + "\n")
            self.f.flush()

    def row(self, t: float, *values) -> None:
        self.f.write("%.4f" % t + "".join("\t%g" % v for v in values) + "\n")
        now = time.monotonic()
        if now - self._last >= self.flush_interval:
            self.f.flush()
            self._last = now

    def flush(self):
        self.f.flush()
        self._last = time.monotonic()

    def close(self):
        if not self.f.closed:
            self.f.flush()
            self.f.close()


def pullx_writer(path, append=False):
    return XvgWriter(path, "Pull COM", "Time (ps)", "Position (nm)", ["1"], append=append)


def pullf_writer(path, append=False):
    return XvgWriter(path, "Pull force", "Time (ps)", "Force (kJ/mol/nm)", ["1"], append=append)
