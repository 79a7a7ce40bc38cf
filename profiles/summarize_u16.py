#!/usr/bin/env python3
"""gpurun_out/prof_<tag>_u16/ (run_profile_u16.sh) -> profiles/<tag>_u16_summary.txt: per content, the DBDE16 kernels'
durations under rocprofv3 and the roofline fraction they give on the algorithmic bytes (raw + packed, as printed by
u16_bench.py in the same run)."""
import csv, glob, os, re, sys
from collections import defaultdict

d = sys.argv[1].rstrip("/")
tag = os.path.basename(d).replace("prof_", "")
out = []
for c in ("mixed", "full", "d12"):
    log = open(os.path.join(d, c + ".log")).read()
    m = re.search(r"DBDE16 (\d+) x (\d+)x(\d+): packed/raw ([0-9.]+)", log)
    if not m:
        out.append(f"== {c}: no bench line"); continue
    n, W, H, ratio = int(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(4))
    alg = n * W * H * 2 * (1 + ratio)
    rows = []
    for f in glob.glob(os.path.join(d, c, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    dur = defaultdict(list)
    for r in rows:
        dur[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    out.append(f"== {c}: {n} frames of {W}x{H} U16, packed/raw {ratio}, algorithmic bytes per launch {alg/1e9:.3f} GB")
    out.append("   " + [ln for ln in log.splitlines() if ln.startswith("DBDE16 ")][-1])
    out.append("   kernel, calls, avg us, min us, avg us without the first 2 (warm-up), fraction of 8 TB/s on that average")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
        if "dbde" not in k:
            continue
        t = [x[1] for x in sorted(v)]
        timed = t[2:] if len(t) > 4 else t
        avg = sum(timed) / len(timed)
        frac = f"{alg / (avg * 1e-9) / 8e12:.3f}" if ("encode_kernel" in k or "dec16_kernel" in k or "enc16_kernel" in k) else "-"
        out.append(f"   {k[:80]:80s} {len(t):4d} {sum(t)/len(t)/1e3:9.1f} {min(t)/1e3:9.1f} {avg/1e3:9.1f} {frac}")
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"{tag}_summary.txt")
open(path, "w").write("\n".join(out) + "\n")
print("\n".join(out))
