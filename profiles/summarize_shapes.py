#!/usr/bin/env python3
"""gpurun_out/prof_r03_shapes/ (r03_shapes_trace.sh) -> profiles/r03_shapes_summary.txt: per shape, each dbde kernel's
average duration under rocprofv3 (the first 2 of 12 launches = abbench's warm-up dropped) and the roofline fraction it
gives on the algorithmic bytes of the launch (raw + packed, from abbench's own JSON line of the same run)."""
import csv, glob, json, os, sys
from collections import defaultdict
d = sys.argv[1].rstrip("/")
out = []
for log in sorted(glob.glob(os.path.join(d, "*.log"))):
    tag = os.path.basename(log)[:-4]
    line = [l for l in open(log) if l.startswith("{")]
    if not line:
        out.append(f"== {tag}: no abbench line"); continue
    j = json.loads(line[-1])
    alg = j["W"] * j["H"] * j["frames"] * (1 + j["packed_over_raw"])
    rows = []
    for f in glob.glob(os.path.join(d, tag, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    dur = defaultdict(list)
    for r in rows:
        if "dbde::" in r["Kernel_Name"] and "synth" not in r["Kernel_Name"]:
            dur[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    out.append(f"== {tag}: {j['frames']} frames, packed/raw {j['packed_over_raw']}, algorithmic bytes per launch {alg/1e9:.3f} GB; "
               f"abbench (HIP events, same run): encode {j['enc_frac']}, decode {j['dec_frac']}, {j['fps']:.0f} frames/s")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
        t = [x[1] for x in sorted(v)]
        timed = t[2:] if len(t) > 4 else t
        avg = sum(timed) / len(timed)
        main = any(s in k for s in ("encode_kernel", "decode_kernel", "encode_mid", "decode_mid", "encode_small", "encode_tiny", "decode_tiny",
                                    "encode_group", "encode_frames", "decode_frames"))
        frac = f"{alg / (avg * 1e-9) / 8e12:.3f}" if main else "-"
        out.append(f"   {k.split('(')[0][-70:]:70s} x{len(t):3d}  avg {avg/1e3:9.1f} us  min {min(t)/1e3:9.1f}  frac {frac}")
# (the directory's name says which round: prof_r03_shapes -> r03_shapes_summary.txt)
name = os.path.basename(d).replace("prof_", "") + "_summary.txt"
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), name)
open(path, "w").write("\n".join(out) + "\n")
print("\n".join(out))
