#!/bin/bash
# round 4: what the alignment of the encoder's pixel fetches costs (profiles/read_align_probe.hip: rate + FETCH_SIZE), baseline abbench lines,
# the per-workgroup in-kernel timeline of persistent-encoder launches (-DDBDE_DIAG variant: profiles/variants.sh diag="-DDBDE_DIAG")
O=gpurun_out/r04_probes; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 120 profiles/read_align_probe 1921 1081 2048 > $O/probe_1921.txt 2>&1
timeout -k 10 120 profiles/read_align_probe 1920 1080 2048 > $O/probe_1920.txt 2>&1
timeout -k 10 120 profiles/read_align_probe 1922 1081 2048 > $O/probe_1922.txt 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- profiles/read_align_probe 1921 1081 512 > $O/fetch.log 2>&1
python3 - <<PY > $O/fetch_summary.txt
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("$O/fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print(f"{k[:60]:60s} n={len(v)} FETCH_SIZE avg {sum(v)/len(v):.0f} KiB-ish -> x2x1024 = {sum(v)/len(v)*2*1024/ (1921*1081*512):.4f} of pixels")
PY
profiles/ab.sh r04p1 "base 2048 2048 1000 mixed concat 20" "base 1921 1081 2048 mixed slots 20" "base 1920 1080 2048 mixed slots 20" "base 1920 1080 512 mixed slots 20" "base 4096 3072 256 mixed slots 10" > $O/ab.txt 2>&1
for spec in "2048 2048 1000 mixed concat" "1921 1081 2048 mixed slots" "1920 1080 512 mixed slots"; do
  ABBENCH_DIAG=1 timeout -k 10 120 profiles/abbench profiles/variants/diag/libdbde_hip.so $spec 1 diag >> $O/diag.txt 2>&1
done
cat $O/probe_1921.txt $O/probe_1920.txt $O/probe_1922.txt $O/fetch_summary.txt $O/ab.txt $O/diag.txt
