#!/bin/bash
# kt.sh TAG variant W H frames content layout steps : rocprofv3 kernel trace of one abbench run, per-kernel averages
T=$1; shift
O=gpurun_out/kt_$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- profiles/abbench profiles/variants/$1/libdbde_hip.so $2 $3 $4 $5 $6 $7 $1 > $O/run.log 2>&1
python3 - <<PY
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("$O/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[:70]:70s} n={len(v):4d} avg {sum(v)/len(v)/1e3:9.2f} us  min {min(v)/1e3:9.2f}  max {max(v)/1e3:9.2f}")
PY
tail -2 $O/run.log
