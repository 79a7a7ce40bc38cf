#!/bin/bash
# Round 2, first GPU pass: parity suite, the default bench line, a 2-rank rehearsal of `--gpus 2` on the one
# GPU (gloo, both ranks on cuda:0: launch contract only) and the streaming driver (config 5, reduced).
set -o pipefail
T=${1:-r02a}
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/${T}_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/${T}_tests.log
tail -3 $O/${T}_tests.log
python bench.py > $O/bench_${T}_default.json 2> $O/bench_${T}_default.err; echo "bench rc=$?"
DBDE_BENCH_REHEARSAL=1 python bench.py --gpus 2 --frames 128 --steps 5 --no-cpu > $O/bench_${T}_rehearsal2.json 2> $O/bench_${T}_rehearsal2.err; echo "rehearsal rc=$?"
python bench.py --config 5 --frames 2000 --no-cpu > $O/bench_${T}_cfg5_n2000.json 2> $O/bench_${T}_cfg5.err; echo "cfg5 rc=$?"
python bench.py --config 5 --frames 2000 --content mixed --no-cpu > $O/bench_${T}_cfg5_mixed_n2000.json 2>> $O/bench_${T}_cfg5.err; echo "cfg5 mixed rc=$?"
tail -c 1500 $O/bench_${T}_default.err
