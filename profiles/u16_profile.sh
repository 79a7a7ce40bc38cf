#!/bin/bash
# rocprofv3 kernel trace of profiles/u16_bench.py -> gpurun_out/r02_u16_summary.txt (run through gpurun)
set -o pipefail
O=gpurun_out/prof_r02_u16
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 profiles/u16_bench.py > $O/kt.log 2>&1 || echo "kt failed"
tail -1 $O/kt.log
python3 - <<PY
import csv, glob
from collections import defaultdict
d = defaultdict(list)
for f in glob.glob("$O/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
lines = ["== rocprofv3 --kernel-trace of profiles/u16_bench.py (128 frames 4096x3072 U16, per-tile depth 0..16): name, calls, avg us, min us, max us"]
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:8]:
    lines.append(f"{k[:80]:80s} {len(v):5d} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} {max(v)/1e3:10.2f}")
lines.append(open("$O/kt.log").read().strip().splitlines()[-1])
open("gpurun_out/r02_u16_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
