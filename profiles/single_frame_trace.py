#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace of profiles/single_frame.py and prints, for the pipelined phase, each kernel's average
duration and the average gap between consecutive kernels (end -> next start): where one frame per call spends its time.
usage: python3 profiles/single_frame_trace.py <dir with *kernel_trace.csv>"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows if "dbde::" in r["Kernel_Name"] and "synth" not in r["Kernel_Name"])
# the pipelined phase: the longest run of kernels whose gaps stay below 20 us
best, cur = [], []
for e in ev:
    if cur and e[0] - cur[-1][1] > 20000:
        if len(cur) > len(best): best = cur
        cur = []
    cur.append(e)
if len(cur) > len(best): best = cur
dur, gap = {}, {}
for i, e in enumerate(best):
    n = e[2].split("(")[0][-60:]
    dur.setdefault(n, []).append(e[1] - e[0])
    if i: gap.setdefault(n, []).append(e[0] - best[i - 1][1])
print(f"{len(best)} kernels in the pipelined phase, {(best[-1][1] - best[0][0]) / 1e3:.1f} us in all")
for n in dur:
    g = gap.get(n, [0])
    print(f"{n:62s} x{len(dur[n]):5d}  avg {sum(dur[n]) / len(dur[n]) / 1e3:6.2f} us  min {min(dur[n]) / 1e3:6.2f}  gap in front avg {sum(g) / len(g) / 1e3:6.2f} us")
per = (best[-1][1] - best[0][0]) / 1e3 / max(1, len(dur[next(iter(dur))]))
print(f"per round trip: {per:.2f} us")
