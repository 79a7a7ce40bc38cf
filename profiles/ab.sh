#!/bin/bash
# ab.sh TAG "variant W H frames content layout steps" ... : runs abbench lines on the GPU box, prints a table
T=$1; shift
O=gpurun_out; mkdir -p $O; : > $O/${T}_ab.jsonl
for spec in "$@"; do
  set -- $spec
  ABBENCH_DIAG=1 timeout -k 10 120 profiles/abbench profiles/variants/$1/libdbde_hip.so $2 $3 $4 $5 $6 $7 $1 >> $O/${T}_ab.jsonl 2>> $O/${T}_ab.err || echo "abbench $spec rc=$?"
done
python3 - <<PY
import json
for ln in open("$O/${T}_ab.jsonl"):
    d = json.loads(ln)
    print(f"{d['tag']:10s} {d['W']}x{d['H']} x{d['frames']} {d['content']:7s} {d['layout']:6s} enc {d['enc_ms']:.3f} ms {d['enc_frac']:.3f}  dec {d['dec_ms']:.3f} ms {d['dec_frac']:.3f} idx {d['idx_ms']:.3f}  fps {d['fps']:.0f} diff {d['diff_dwords']}")
PY
tail -3 $O/${T}_ab.err 2>/dev/null
