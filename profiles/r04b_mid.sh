#!/bin/bash
# round 4, second half: decode_mid_kernel pipelined (persistent workgroups) + pixels staged through LDS, against its round-3
# form (-DDBDE_MID_V1) and the pipelined form without the staging (-DDBDE_MID_NO_STAGE=1): abbench lines, then the
# mid-frame parity tests.  The libraries under profiles/ab_libs/ are built in the container:
#   make -B -C dbde-video-cpp_amd/csrc OUT=$PWD/profiles/ab_libs/v1 EXTRA=-DDBDE_MID_V1 $PWD/profiles/ab_libs/v1/libdbde_hip.so
O=gpurun_out/r04b_mid; mkdir -p $O; : > $O/ab.jsonl
run() {  # lib tag W H frames content
  ABBENCH_DIAG=0 timeout -k 10 120 profiles/abbench $1 $3 $4 $5 $6 slots 10 $2 >> $O/ab.jsonl 2>> $O/ab.err || echo "abbench $* rc=$?"
}
for shape in "72 72 262144" "96 96 131072" "80 80 262144" "520 8 262144" "65 64 262144" "104 96 131072"; do
  for content in mixed noise8; do
    run dbde-video-cpp_amd/libdbde_hip.so new $shape $content
    for v in ${VARIANTS:-v1 nostage}; do run profiles/ab_libs/$v/libdbde_hip.so $v $shape $content; done
  done
done
python3 - <<PY
import json
for ln in open("$O/ab.jsonl"):
    d = json.loads(ln)
    print(f"{d['tag']:8s} {d['W']}x{d['H']} x{d['frames']} {d['content']:7s} enc {d['enc_ms']:.3f} ms {d['enc_frac']:.3f}  dec {d['dec_ms']:.3f} ms {d['dec_frac']:.3f} idx {d['idx_ms']:.3f}  fps {d['fps']:.0f} diff {d['diff_dwords']}")
PY
tail -3 $O/ab.err 2>/dev/null
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mid_frames or staged_frame_decoder or tiny or tile_api" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
