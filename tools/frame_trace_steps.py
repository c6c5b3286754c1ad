#!/usr/bin/env python3
"""Per-step budget from a rocprofv3 kernel trace of a run of steps (tools/frame_trace.sh): for every step the nonbonded
kernel, what follows it on its queue up to the next nonbonded kernel, and the end of the PME chain (the gather).
Prints one line per step of the last cycle of 10 and the averages of plain steps."""
import csv, glob, os, sys
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:40], r.get("Queue_Id", "?")))
rows.sort()
nb = [i for i, r in enumerate(rows) if r[2].startswith("k_nonbonded")]
mainq = rows[nb[0]][3]
steps = []
for a, b in zip(nb[:-1], nb[1:]):
    s0, e0, name, _ = rows[a]
    tail = [r for r in rows[a + 1:b] if r[3] == mainq]
    gath = [r for r in rows[a:b] if r[2].startswith("k_pme_gather")]
    steps.append(dict(name=name, nb=(e0 - s0) / 1e3, period=(rows[b][0] - s0) / 1e3, tail=[(r[2], (r[0] - e0) / 1e3, (r[1] - r[0]) / 1e3) for r in tail],
                      gather_end=(gath[-1][1] - e0) / 1e3 if gath else None, last_end=(tail[-1][1] - e0) / 1e3 if tail else 0.0,
                      next_gap=(rows[b][0] - (tail[-1][1] if tail else e0)) / 1e3))
for s in steps[-21:-1]:
    print("%-16s nb %6.1f  period %6.1f  gather ends %+6.1f  then: %s  | gap to next nb %.1f" % (
        s["name"], s["nb"], s["period"], s["gather_end"] if s["gather_end"] is not None else float("nan"),
        ", ".join("%s@%.1f(%.1f)" % t for t in s["tail"]), s["next_gap"]))
plain = [s for s in steps[50:] if s["name"] == "k_nonbonded_f" and len(s["tail"]) == 1 and s["tail"][0][0].startswith("k_update_fused<false")]
if plain:
    n = len(plain)
    print("plain steps (%d): nb %.1f, nb end -> update start %.1f, update %.1f, update end -> next nb %.1f, period %.1f, gather end relative to nb end %+.1f" % (
        n, sum(s["nb"] for s in plain) / n, sum(s["tail"][0][1] for s in plain) / n, sum(s["tail"][0][2] for s in plain) / n,
        sum(s["next_gap"] for s in plain) / n, sum(s["period"] for s in plain) / n, sum(s["gather_end"] for s in plain) / n))
