#!/bin/bash
# rocprofv3 kernel trace of profiles/abbench on the shapes round 3 widened into (one run per shape, 10 timed steps):
# -> gpurun_out/prof_r03_shapes/<tag>/ ; summarised by profiles/summarize_shapes.py into profiles/r03_shapes_summary.txt
set -o pipefail
OUT=gpurun_out/prof_r03_shapes
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LIB=dbde-video-cpp_amd/libdbde_hip.so
while read -r W H N C; do
    tag=${W}x${H}_${C}
    rocprofv3 --kernel-trace --output-format csv -d $OUT/$tag -- profiles/abbench $LIB $W $H $N $C slots 10 $tag > $OUT/$tag.log 2>&1 || echo "$tag failed"
done <<LIST
72 72 262144 mixed
96 96 262144 mixed
128 128 262144 mixed
1366 768 4096 mixed
1366 768 4096 noise8
1440 900 2048 mixed
720 1280 4096 mixed
720 1280 4096 noise8
1080 1920 2048 mixed
1600 900 2048 mixed
1921 1081 2048 noise8
LIST
echo "profiled $OUT"
