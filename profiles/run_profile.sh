#!/bin/bash
# Collects the rocprofv3 evidence for one bench.py workload on the GPU box (run through gpurun):
#   1. --kernel-trace --stats            -> per-kernel durations
#   2. --pmc FETCH_SIZE                  -> HBM read bytes   (own pass: TCC slots)
#   3. --pmc WRITE_SIZE                  -> HBM write bytes  (own pass)
#   4. --pmc SQ counters                 -> LDS conflicts, wave cycles, waits, VALU
# usage: run_profile.sh TAG CONFIG CONTENT   (e.g. r02 2 mixed) -> gpurun_out/prof_<tag>_cfg<N>_<content>/
set -o pipefail
TAG=${1:-r02}; CFG=${2:-2}; CONTENT=${3:-noise8}
OUT=gpurun_out/prof_${TAG}_cfg${CFG}_${CONTENT}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 3 --warmup 1 --only --no-cpu --config $CFG --content $CONTENT"
# the duration pass runs 2 warm-up + 20 timed steps: with 1 + 3 (rounds 1-3) the cold first dispatch was a quarter of
# the average (3.77 ms against 3.31 ms steady on the mixed workload) and the summary read 5 % slower than bench.py's
# own timed region, which never includes the warm-up; summarize.py prints both averages
KT_ARGS="bench.py --steps 20 --warmup 2 --only --no-cpu --config $CFG --content $CONTENT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $KT_ARGS > $OUT/kt.log 2>&1 || echo "kt failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1 || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1 || echo "write failed"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1 || echo "sq failed"
grep "^{" $OUT/kt.log | tail -1 > $OUT/bench_line.json
echo "profiled $OUT"
