#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly.  usage: tools/kstats.py <dir-or-csv>"""
import csv, glob, os, sys
p = sys.argv[1]
f = p if p.endswith(".csv") else sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True))[-1]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("ptz::(anonymous namespace)::", "").replace("void ", "")[:64]
    print(f"{n:64s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f}us min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:9.1f} tot {float(r['TotalDurationNs'])/1e6:8.2f}ms")
