"""Prints consecutive kernel intervals (start, end, name) from a rocprofv3 kernel trace, relative to the first, to see
which kernels overlap when two streams are used.  usage: python tools/kernel_overlap.py <dir> [skip] [count]"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:32],
                         r.get("Queue_Id", "?")))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 26
t0 = rows[skip][0]
for s, e, n, q in rows[skip:skip + cnt]:
    print(f"{(s - t0) / 1e3:9.2f} -> {(e - t0) / 1e3:9.2f} us  ({(e - s) / 1e3:6.2f})  q{q}  {n}")
