#!/usr/bin/env python3
"""Median of one rocprofv3 --pmc counter over the dispatches of one kernel: pmc_summary.py counter_collection.csv KERNEL COUNTER"""
import csv, sys, statistics
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[3]]
vals = [float(r["Counter_Value"]) for r in rows]
print(sys.argv[3], "dispatches", len(vals), "median", statistics.median(vals) if vals else None)
