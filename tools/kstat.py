#!/usr/bin/env python3
"""Print calls / mean us / share for the kernels of a rocprofv3 kernel_stats.csv whose name contains one of the given words.
usage: kstat.py <kernel_stats.csv> word [word ...]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(w in n for w in sys.argv[2:]):
        print(f'{n.split("(")[0][-48:]:48s} n={int(r["Calls"]):5d} mean={float(r["AverageNs"]) / 1e3:9.1f} us  {float(r["Percentage"]):5.2f} %')
