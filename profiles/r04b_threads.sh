#!/bin/bash
# round 4, second half: where decode_mid_kernel (256-thread persistent workgroups) stops paying against the chunk decoder:
# the in-tree library against profiles/ab_libs/$ALT (built in the container with -DDBDE_MID_DECODE_TILES=n) on $SHAPES
# ("W H frames;W H frames;...")
O=gpurun_out/r04b_thr; mkdir -p $O; : > $O/ab.jsonl
run() { ABBENCH_DIAG=0 timeout -k 10 120 profiles/abbench $1 $3 $4 $5 $6 slots 10 $2 >> $O/ab.jsonl 2>> $O/ab.err || echo "abbench $* rc=$?"; }
IFS=";" read -ra LIST <<< "${SHAPES:-104 100 131072;120 120 65536;128 120 65536;128 128 65536;96 96 131072;121 100 65536}"
for shape in "${LIST[@]}"; do
  for content in mixed noise8; do
    run dbde-video-cpp_amd/libdbde_hip.so base $shape $content
    run profiles/ab_libs/${ALT:-m256}/libdbde_hip.so ${ALT:-m256} $shape $content
  done
done
python3 - <<PY
import json
for ln in open("$O/ab.jsonl"):
    d = json.loads(ln)
    print(f"{d['tag']:8s} {d['W']}x{d['H']} x{d['frames']} {d['content']:7s} enc {d['enc_ms']:.3f} ms {d['enc_frac']:.3f}  dec {d['dec_ms']:.3f} ms {d['dec_frac']:.3f} idx {d['idx_ms']:.3f} fps {d['fps']:.0f} diff {d['diff_dwords']}")
PY
tail -3 $O/ab.err 2>/dev/null
