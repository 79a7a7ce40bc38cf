#!/bin/bash
# rocprofv3 kernel trace of the DBDE16 workloads of profiles/u16_bench.py (128 frames of 4096x3072 U16; 12 launches of
# each kernel per content, the first two are warm-up) -> gpurun_out/prof_<tag>_u16/; summarised by summarize_u16.py
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/prof_${TAG}_u16
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in mixed full d12; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$c -- python3 profiles/u16_bench.py $c > $OUT/$c.log 2>&1 || echo "$c failed"
done
echo "profiled $OUT"
