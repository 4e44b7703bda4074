#!/usr/bin/env python3
"""Per-step kernel timeline from a rocprofv3 --kernel-trace CSV: for the last full steps of a bench.py run, each
kernel's start and end relative to the first kernel of its step, and the idle gap before it.
usage: tools/kernel_timeline.py <dir with *_kernel_trace.csv> [first-kernel-substring]"""
import csv, glob, sys
d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "ita_stream_kernel<64, true, 1"
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
starts = [i for i, r in enumerate(rows) if first in r[2]]
acc = {}
n = 0
for a, b in zip(starts[-12:-1], starts[-11:]):
    step = rows[a:b]
    if len(step) > 12:
        continue
    n += 1
    t0 = step[0][0]
    prev_end = t0
    for j, (s, e, name) in enumerate(step):
        key = (j, name.split("(")[0][:60])
        v = acc.setdefault(key, [0, 0, 0])
        v[0] += s - t0; v[1] += e - t0; v[2] += s - prev_end
        prev_end = max(prev_end, e)
    v = acc.setdefault((99, "next step starts"), [0, 0, 0])
    v[0] += rows[b][0] - t0; v[2] += rows[b][0] - prev_end
print(f"{n} steps averaged; us from the start of the step's first kernel")
print(f"{'kernel':62s} {'start':>8s} {'end':>8s} {'dur':>7s} {'gap before':>10s}")
for (j, name), v in sorted(acc.items()):
    print(f"{name:62s} {v[0]/n/1e3:8.2f} {v[1]/n/1e3:8.2f} {(v[1]-v[0])/n/1e3:7.2f} {v[2]/n/1e3:10.2f}")
