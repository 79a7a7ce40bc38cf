#!/bin/bash
T=${1:-r02}
O=gpurun_out; mkdir -p $O
DBDE_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --frames 128 --steps 5 --no-cpu > $O/bench_${T}_rehearsal2.json 2> $O/bench_${T}_rehearsal2.err; echo "rehearsal rc=$?"
tail -4 $O/bench_${T}_rehearsal2.err; head -c 1500 $O/bench_${T}_rehearsal2.json; echo
timeout -k 10 300 python bench.py --config 5 --frames 2000 --no-cpu > $O/bench_${T}_cfg5_n2000.json 2> $O/bench_${T}_cfg5.err; echo "cfg5 rc=$?"
tail -4 $O/bench_${T}_cfg5.err; head -c 1500 $O/bench_${T}_cfg5_n2000.json; echo
timeout -k 10 300 python bench.py --config 5 --frames 2000 --content mixed --no-cpu > $O/bench_${T}_cfg5_mixed_n2000.json 2>> $O/bench_${T}_cfg5.err; echo "cfg5 mixed rc=$?"
head -c 1500 $O/bench_${T}_cfg5_mixed_n2000.json; echo
