O=gpurun_out/r04b_pipe; mkdir -p $O; : > $O/ab.jsonl
run() { ABBENCH_DIAG=0 timeout -k 10 120 profiles/abbench $1 $3 $4 $5 $6 slots 20 $2 >> $O/ab.jsonl 2>> $O/ab.err || echo "abbench $* rc=$?"; }
IFS=";" read -ra LIST <<< "${SHAPES:-72 72 262144;64 64 262144;96 96 131072;128 128 65536;32 32 524288;8 8 1048576;80 80 131072;71 73 131072;104 100 131072}"
for shape in "${LIST[@]}"; do
  for content in mixed noise8; do
    run dbde-video-cpp_amd/libdbde_hip.so deep $shape $content
    run profiles/ab_libs/prev/libdbde_hip.so prev $shape $content
  done
done
python3 - <<PY
import json
for ln in open("$O/ab.jsonl"):
    d = json.loads(ln)
    print(f"{d['tag']:8s} {d['W']}x{d['H']} x{d['frames']} {d['content']:7s} enc {d['enc_ms']:.3f} ms {d['enc_frac']:.3f}  dec {d['dec_ms']:.3f} ms {d['dec_frac']:.3f} fps {d['fps']:.0f} diff {d['diff_dwords']}")
PY
tail -3 $O/ab.err 2>/dev/null
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mid_frames or staged_frame_decoder or tiny or tile_api or reads_nothing or malformed or baseline_configs" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
