O=gpurun_out/r04b_abl; mkdir -p $O; : > $O/ab.jsonl
run() { ABBENCH_DIAG=0 timeout -k 10 120 profiles/abbench $1 $3 $4 $5 $6 slots 10 $2 >> $O/ab.jsonl 2>> $O/ab.err || echo "abbench $* rc=$?"; }
for shape in "72 72 262144" "96 96 131072"; do
  for content in mixed noise8; do
    run dbde-video-cpp_amd/libdbde_hip.so new $shape $content
    run profiles/ab_libs/nold/libdbde_hip.so nold $shape $content
    run profiles/ab_libs/nost/libdbde_hip.so nost $shape $content
  done
done
for shape in "64 64 262144" "8 8 1048576" "128 128 65536" "160 120 65536" "320 240 16384"; do
  run dbde-video-cpp_amd/libdbde_hip.so new $shape mixed
done
python3 - <<PY
import json
for ln in open("$O/ab.jsonl"):
    d = json.loads(ln)
    print(f"{d['tag']:6s} {d['W']}x{d['H']} x{d['frames']} {d['content']:7s} enc {d['enc_ms']:.3f} ms {d['enc_frac']:.3f}  dec {d['dec_ms']:.3f} ms {d['dec_frac']:.3f} idx {d['idx_ms']:.3f}  fps {d['fps']:.0f} diff {d['diff_dwords']}")
PY
tail -3 $O/ab.err 2>/dev/null
