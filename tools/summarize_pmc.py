#!/usr/bin/env python3
"""Condenses rocprofv3 output dirs (stats + pmc passes) into one table: per kernel, average
duration and the mean of every collected counter per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
dur = {}
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]))
cnt = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name in sorted(dur, key=lambda n: -dur[n][0] * dur[n][1]):
    calls, avg = dur[name]
    if "ita_" not in name:
        continue
    print(f"== {name[:90]}\n   calls {calls}  avg {avg/1e3:.2f} us")
    for c, vals in sorted(cnt.get(name, {}).items()):
        print(f"   {c:32s} {sum(vals)/len(vals):16.1f}   (n={len(vals)})")
