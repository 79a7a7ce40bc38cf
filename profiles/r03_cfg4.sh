#!/bin/bash
# cfg4 A/B on the GPU box: old any-geometry encoder (DBDE_HIP_EXPERIMENT=4) vs the frame-sequential one, several shapes
L=${1:-dbde-video-cpp_amd/libdbde_hip.so}
for spec in "1921 1081 2048 mixed" "1921 1081 2048 noise8" "1921 1081 2048 smooth" "1001 1001 4096 mixed" "1928 1080 2048 mixed" "1368 768 4096 mixed"; do
  set -- $spec
  for e in 4 0; do
    DBDE_HIP_EXPERIMENT=$e timeout -k 10 120 profiles/abbench $L $1 $2 $3 $4 slots 10 exp$e 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln)
        print(f\"{d['tag']:6s} {d['W']}x{d['H']} x{d['frames']} {d['content']:7s} enc {d['enc_ms']:.3f} ms {d['enc_frac']:.3f}  dec {d['dec_ms']:.3f} ms {d['dec_frac']:.3f} idx {d['idx_ms']:.3f} fps {d['fps']:.0f} diff {d['diff_dwords']}\")
    else:
        print(ln.rstrip())
"
  done
done
