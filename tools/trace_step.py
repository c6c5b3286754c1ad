#!/usr/bin/env python3
"""Parse a rocprofv3 kernel trace CSV and print the timeline of one MD step (between two k_prepare launches)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
prep = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_prepare")]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(prep) // 2
i0, i1 = prep[which], prep[which + 1]
t0 = rows[i0]["s"]
print("step wall %.1f us" % ((rows[i1]["s"] - t0) / 1e3))
for r in rows[i0:i1]:
    print("%8.1f %8.1f  q%-3s %s" % ((r["s"] - t0) / 1e3, (r["e"] - t0) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
