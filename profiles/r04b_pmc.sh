#!/bin/bash
# round 4, second half: SQ counters of the persistent small-frame kernels on 72x72 / 64x64 (abbench, 3 timed steps):
# where their waves spend their cycles -> gpurun_out/prof_r04b_pmc/
O=gpurun_out/prof_r04b_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LIB=dbde-video-cpp_amd/libdbde_hip.so
for shape in "72 72 262144 mixed" "72 72 262144 noise8" "64 64 262144 mixed"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/a_$tag -- profiles/abbench $LIB $shape slots 3 $tag > $O/a_$tag.log 2>&1 || echo "a $tag failed"
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d $O/b_$tag -- profiles/abbench $LIB $shape slots 3 $tag > $O/b_$tag.log 2>&1 || echo "b $tag failed"
done
python3 - <<PY
import csv, glob, collections, os
for d in sorted(glob.glob("$O/[ab]_*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "dbde::" in k and "synth" not in k and "count_diff" not in k:
                acc[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", os.path.basename(d))
    for k, c in acc.items():
        print("  ", k, {n: round(sum(v) / len(v)) for n, v in c.items()})
PY
