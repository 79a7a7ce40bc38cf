#!/bin/bash
# round 4, second half: what a decode chunk's index round trip costs (probe build -DDBDE_DEC_FAKE_INDEX: frame offset, verdict
# and chunk extent computed instead of loaded; valid for noise8 in abbench's slots only) -- in-tree library against profiles/ab_libs/$ALT
O=gpurun_out/r04b_probe; mkdir -p $O; : > $O/ab.jsonl
run() { ABBENCH_DIAG=0 timeout -k 10 120 profiles/abbench $1 $3 $4 $5 $6 slots 20 $2 >> $O/ab.jsonl 2>> $O/ab.err || echo "abbench $* rc=$?"; }
IFS=";" read -ra LIST <<< "${SHAPES:-1921 1081 2048;1920 1080 2048;2048 2048 1000;4096 3072 256;1366 768 4096;640 480 8192}"
for shape in "${LIST[@]}"; do
  for content in ${CONTENTS:-noise8}; do
    run dbde-video-cpp_amd/libdbde_hip.so base $shape $content
    run profiles/ab_libs/${ALT:-fake}/libdbde_hip.so ${ALT:-fake} $shape $content
    run dbde-video-cpp_amd/libdbde_hip.so base $shape $content
    run profiles/ab_libs/${ALT:-fake}/libdbde_hip.so ${ALT:-fake} $shape $content
  done
done
python3 - <<PY
import json
for ln in open("$O/ab.jsonl"):
    d = json.loads(ln)
    print(f"{d['tag']:8s} {d['W']}x{d['H']} x{d['frames']} {d['content']:7s} enc {d['enc_ms']:.3f} ms {d['enc_frac']:.3f}  dec {d['dec_ms']:.3f} ms {d['dec_frac']:.3f} idx {d['idx_ms']:.3f} fps {d['fps']:.0f} diff {d['diff_dwords']}")
PY
tail -3 $O/ab.err 2>/dev/null
