#!/usr/bin/env python3
"""Per-step critical-path summary of a rocprofv3 kernel trace of the bench: for every MD step (from one k_prepare
to the next) the times at which the nonbonded kernel, the PME chain and the bonded kernels end and the update
starts; averaged per step class (plain, thermostat, energy, search)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"]); r["n"] = r["Kernel_Name"]
rows.sort(key=lambda r: r["s"])
prep = [i for i, r in enumerate(rows) if r["n"].startswith("k_prepare")]
acc = collections.defaultdict(list)
last = int(sys.argv[2]) if len(sys.argv) > 2 else 450
for a, b in zip(prep[-last-1:-1], prep[-last:]):
    st = rows[a:b]; t0 = st[0]["s"]
    def end(pred): 
        v = [r["e"] for r in st if pred(r["n"])]
        return (max(v) - t0) / 1e3 if v else float("nan")
    def start(pred):
        v = [r["s"] for r in st if pred(r["n"])]
        return (min(v) - t0) / 1e3 if v else float("nan")
    names = " ".join(r["n"] for r in st)
    cls = "search" if "k_build_list" in names else ("energy" if "k_nonbonded_ev" in names else ("thermostat" if "k_tcouple" in names else "plain"))
    acc[cls].append((
        (rows[b]["s"] - t0) / 1e3,
        start(lambda n: n.startswith("k_nonbonded")), end(lambda n: n.startswith("k_nonbonded")),
        start(lambda n: "k_pme_spread" in n), end(lambda n: "k_pme_spread" in n), end(lambda n: "k_pme_gather" in n),
        start(lambda n: "k_bonded" in n), end(lambda n: "k_bonded" in n),
        start(lambda n: n.startswith("k_leapfrog")), end(lambda n: "k_lincs" in n or "k_settle" in n)))
print("%-10s %5s %7s | %7s %7s | %7s %7s %7s | %7s %7s | %7s %7s" % ("class", "n", "wall", "nb>", "nb<", "spread>", "spread<", "gather<", "bonded>", "bonded<", "update>", "update<"))
import numpy as np
tot = 0.0; cnt = 0
for cls, v in acc.items():
    m = np.nanmean(np.array(v), axis=0)
    print("%-10s %5d %7.1f | %7.1f %7.1f | %7.1f %7.1f %7.1f | %7.1f %7.1f | %7.1f %7.1f" % ((cls, len(v)) + tuple(m)))
    tot += np.array(v)[:, 0].sum(); cnt += len(v)
print("mean step %.1f us over %d steps" % (tot / cnt, cnt))
