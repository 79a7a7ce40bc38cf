#!/bin/bash
# rocprofv3 kernel trace of profiles/abbench on the small-frame shapes of round 4's second half (one run per shape, 10 timed
# steps): which kernel runs, its duration, the roofline fraction -> gpurun_out/prof_r04b_shapes/<tag>/ ; summarised by
# profiles/summarize_shapes.py into profiles/r04b_shapes_summary.txt
set -o pipefail
OUT=gpurun_out/prof_r04b_shapes
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LIB=dbde-video-cpp_amd/libdbde_hip.so
while read -r W H N C; do
    tag=${W}x${H}_${C}
    rocprofv3 --kernel-trace --output-format csv -d $OUT/$tag -- profiles/abbench $LIB $W $H $N $C slots 10 $tag > $OUT/$tag.log 2>&1 || echo "$tag failed"
done <<LIST
8 8 1048576 mixed
32 32 524288 mixed
64 64 262144 mixed
64 64 262144 noise8
72 72 262144 mixed
72 72 262144 noise8
96 96 131072 mixed
104 100 131072 mixed
128 128 65536 mixed
160 120 65536 mixed
320 240 16384 mixed
LIST
python3 profiles/summarize_shapes.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt | cut -c1-200
echo "profiled $OUT"
