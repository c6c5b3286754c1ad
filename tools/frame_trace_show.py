#!/usr/bin/env python3
"""Timeline around one trajectory-frame capture from a rocprofv3 kernel + memory-copy trace (tools/frame_trace.sh):
every kernel and copy from 60 us before the rounding kernel of a late capture to 400 us after it, times in us."""
import csv, glob, os, sys
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48], "q%s" % r.get("Queue_Id", "?")))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") , ""))
rows.sort()
caps = [r for r in rows if r[2].startswith("k_frame_quantize")]
print("%d kernels and copies, %d captures" % (len(rows), len(caps)))
for cap in caps[-3:-1]:
    t0 = cap[0]
    print("---- capture at %d" % t0)
    for a, b, n, q in rows:
        if t0 - 60000 <= a <= t0 + 400000:
            print("%9.1f %9.1f %7.1f  %-4s %s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, q, n))
