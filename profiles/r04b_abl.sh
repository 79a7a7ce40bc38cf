#!/bin/bash
# round 4, second half: decode_mid_kernel, what is left -- without the payload loads (-DDBDE_MID_ABLATE_LOADS), without the
# image stores (-DDBDE_MID_ABLATE_STORES), with plain instead of non-temporal image stores (-DDBDE_MID_NT=0); the ablated
# builds decode garbage (abbench leaves with 3), their TIMES are what is read.  Libraries built in the container.
O=gpurun_out/r04b_abl; mkdir -p $O; : > $O/ab.jsonl
run() { ABBENCH_DIAG=0 timeout -k 10 120 profiles/abbench $1 $3 $4 $5 $6 slots 10 $2 >> $O/ab.jsonl 2>> $O/ab.err; }
IFS=";" read -ra LIST <<< "${SHAPES:-72 72 262144;64 64 262144;96 96 131072}"
for shape in "${LIST[@]}"; do
  for content in mixed noise8; do
    run dbde-video-cpp_amd/libdbde_hip.so new $shape $content
    for v in ${VARIANTS:-plain nold nost}; do run profiles/ab_libs/$v/libdbde_hip.so $v $shape $content; done
  done
done
python3 - <<PY
import json
for ln in open("$O/ab.jsonl"):
    d = json.loads(ln)
    print(f"{d['tag']:6s} {d['W']}x{d['H']} x{d['frames']} {d['content']:7s} enc {d['enc_ms']:.3f} ms {d['enc_frac']:.3f}  dec {d['dec_ms']:.3f} ms {d['dec_frac']:.3f} idx {d['idx_ms']:.3f}  fps {d['fps']:.0f} diff {d['diff_dwords']}")
PY
