#!/bin/bash
# round 4, second half: encode_frames_kernel, 256 against 512 threads (frames_threads_for: the larger workgroup only where it
# fills its lanes DBDE_FRAMES_512_GAIN percent better): in-tree library against profiles/ab_libs/$ALT (built in the container)
O=gpurun_out/r04b_gain; mkdir -p $O; : > $O/ab.jsonl
run() { ABBENCH_DIAG=0 timeout -k 10 120 profiles/abbench $1 $3 $4 $5 $6 slots 20 $2 >> $O/ab.jsonl 2>> $O/ab.err || echo "abbench $* rc=$?"; }
IFS=";" read -ra LIST <<< "${SHAPES:-80 72 262144;88 80 131072;96 96 131072;104 88 131072;120 88 131072;112 96 131072}"
for shape in "${LIST[@]}"; do
  for content in mixed noise8; do
    run dbde-video-cpp_amd/libdbde_hip.so g115 $shape $content
    run profiles/ab_libs/${ALT:-g100}/libdbde_hip.so ${ALT:-g100} $shape $content
  done
done
python3 - <<PY
import json
for ln in open("$O/ab.jsonl"):
    d = json.loads(ln)
    T=((d['W']+7)//8)*((d['H']+7)//8)
    print(f"{d['tag']:5s} {d['W']}x{d['H']} T={T} u256={(512//T)*T} u512={(1024//T)*T} {d['content']:7s} enc {d['enc_ms']:.3f} ms {d['enc_frac']:.3f}  dec {d['dec_frac']:.3f} diff {d['diff_dwords']}")
PY
