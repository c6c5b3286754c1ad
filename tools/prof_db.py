#!/usr/bin/env python3
"""Per-kernel averages from a rocprofv3 results database: prof_db.py results.db [substring ...]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
keys = sys.argv[2:]
for name, n, avg in c.execute("select name, count(*), avg(end-start) from kernels group by name order by sum(end-start) desc"):
    if not keys or any(k in name for k in keys):
        print("%-60s %6d %9.1f us" % (name[:60], n, avg / 1000))
