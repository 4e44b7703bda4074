"""Steady-state timeline of one forward step from a rocprofv3 kernel trace (kernel_trace.csv):
per kernel the median duration and the median gap to the next kernel of the same step.

usage: python tools/kernel_timeline.py <dir-with-*_kernel_trace.csv>
"""
import csv
import glob
import statistics
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda n: n.split("(")[0].split("<")[0].replace("void ", "")[:28]
# a step starts at every tokenizer launch
steps, cur = [], None
first = "tokenizer" if any("tokenizer" in n for _, _, n in rows) else "ita_stream_kernel"   # first kernel of a step
for s, e, n in rows:
    if first in n:
        cur = []
        steps.append(cur)
    if cur is not None:
        cur.append((s, e, short(n)))
steps = [st for st in steps if len(st) == len(steps[len(steps) // 2])][5:-1]
print("steps analysed:", len(steps), "kernels per step:", len(steps[0]))
tot = []
for i in range(len(steps[0])):
    dur = [st[i][1] - st[i][0] for st in steps]
    gap = [(st[i + 1][0] - st[i][1]) if i + 1 < len(st) else 0 for st in steps]
    print("%-28s dur %7.2f us   gap-after %6.2f us" % (steps[0][i][2], statistics.median(dur) / 1e3, statistics.median(gap) / 1e3))
span = [st[-1][1] - st[0][0] for st in steps]
period = [steps[j + 1][0][0] - steps[j][0][0] for j in range(len(steps) - 1)]
print("step span (first start -> last end) median %.2f us; step period median %.2f us" % (statistics.median(span) / 1e3, statistics.median(period) / 1e3))
