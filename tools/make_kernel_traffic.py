#!/usr/bin/env python3
"""profiles/kernel_traffic.json from a tools/profile_gpu.sh output directory.

Per kernel: HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (both reported in KB; on gfx950 FETCH_SIZE
counts a wide coalesced read at half its bytes -- MI355X_MICROARCH.md, "HBM" section -- WRITE_SIZE is exact
for 16-byte-per-lane stores).  FETCH_SIZE and WRITE_SIZE come from separate --pmc passes.

usage: python tools/make_kernel_traffic.py gpurun_out/<tag> "<source note>" > profiles/kernel_traffic.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, note = sys.argv[1], sys.argv[2]
dur = {}
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = float(r["AverageNs"]) / 1e3
cnt = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
kernels = {}
for name, c in cnt.items():
    if "ita_" not in name or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    short = name.replace("void ", "").split("(")[0]
    fk = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
    wk = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
    kernels[short] = {"avg_us": round(dur.get(name, 0.0), 2), "fetch_kb_raw": round(fk, 1), "write_kb": round(wk, 1),
                      "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
print(json.dumps({"source": note, "correction": "FETCH_SIZE x2 (gfx950 wide-read under-count), WRITE_SIZE x1; KB units",
                  "kernels": kernels}, indent=1))
