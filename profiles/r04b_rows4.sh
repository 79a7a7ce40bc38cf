#!/bin/bash
# round 4, last hours: small frames whose rows are 4 mod 8 bytes through the staged / persistent forms (decode_mid_kernel staged,
# encode_group_kernel): in-tree library against profiles/ab_libs/prev (the commit before, built in the container), then the GPU suite
O=gpurun_out/r04b_rows4; mkdir -p $O; : > $O/ab.jsonl
run() { ABBENCH_DIAG=0 timeout -k 10 120 profiles/abbench $1 $3 $4 $5 $6 slots 20 $2 >> $O/ab.jsonl 2>> $O/ab.err || echo "abbench $* rc=$?"; }
for shape in "60 60 262144" "20 20 524288" "44 36 524288" "36 28 524288" "64 64 262144" "32 32 524288" "72 72 262144" "60 64 262144"; do
  for content in mixed noise8; do
    run dbde-video-cpp_amd/libdbde_hip.so new $shape $content
    run profiles/ab_libs/prev/libdbde_hip.so prev $shape $content
  done
done
python3 - <<PY
import json
for ln in open("$O/ab.jsonl"):
    d = json.loads(ln)
    print(f"{d['tag']:6s} {d['W']}x{d['H']} x{d['frames']} {d['content']:7s} enc {d['enc_ms']:.3f} ms {d['enc_frac']:.3f}  dec {d['dec_ms']:.3f} ms {d['dec_frac']:.3f} diff {d['diff_dwords']}")
PY
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
