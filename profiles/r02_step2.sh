#!/bin/bash
# Round 2, GPU pass: access-shape microbenchmark, A/B of library variants (abbench), then bench legs with a
# time limit each.
set -o pipefail
T=${1:-r02b}
O=gpurun_out
mkdir -p $O
timeout -k 10 120 profiles/mempattern > $O/${T}_mempattern.txt 2>&1; echo "mempattern rc=$?"
cat $O/${T}_mempattern.txt
: > $O/${T}_ab.jsonl
for v in r01 head ct256 ct1024 nont; do
  for c in mixed noise8 smooth; do
    timeout -k 10 120 profiles/abbench profiles/variants/$v/libdbde_hip.so 4096 3072 512 $c slots 10 $v >> $O/${T}_ab.jsonl 2>> $O/${T}_ab.err || echo "abbench $v $c rc=$?"
  done
done
for v in r01 head; do
  timeout -k 10 120 profiles/abbench profiles/variants/$v/libdbde_hip.so 1921 1081 2048 mixed slots 10 $v >> $O/${T}_ab.jsonl 2>> $O/${T}_ab.err || echo "abbench $v cfg4 rc=$?"
  timeout -k 10 120 profiles/abbench profiles/variants/$v/libdbde_hip.so 2048 2048 1000 mixed concat 10 $v >> $O/${T}_ab.jsonl 2>> $O/${T}_ab.err || echo "abbench $v cfg3 rc=$?"
  timeout -k 10 120 profiles/abbench profiles/variants/$v/libdbde_hip.so 4096 3072 512 mixed concat 10 $v >> $O/${T}_ab.jsonl 2>> $O/${T}_ab.err || echo "abbench $v concat rc=$?"
done
python3 - <<PY
import json
for ln in open("$O/${T}_ab.jsonl"):
    d = json.loads(ln)
    print(f"{d['tag']:8s} {d['W']}x{d['H']} {d['content']:7s} {d['layout']:6s} enc {d['enc_ms']:.3f} ms {d['enc_frac']:.3f}  dec {d['dec_ms']:.3f} ms {d['dec_frac']:.3f} idx {d['idx_ms']:.3f}  fps {d['fps']:.0f} diff {d['diff_dwords']}")
PY
timeout -k 10 400 python bench.py > $O/bench_${T}_default.json 2> $O/bench_${T}_default.err; echo "bench rc=$?"
tail -5 $O/bench_${T}_default.err
