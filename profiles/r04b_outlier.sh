O=gpurun_out/r04b_outlier; mkdir -p $O; : > $O/ab.jsonl
for i in 1 2 3; do
  for shape in "64 64 262144" "32 32 524288" "72 72 262144"; do
    timeout -k 10 300 profiles/abbench dbde-video-cpp_amd/libdbde_hip.so $shape mixed slots 300 steps300 >> $O/ab.jsonl 2>> $O/ab.err || echo "rc=$?"
  done
done
python3 - <<PY
import json
for ln in open("$O/ab.jsonl"):
    d = json.loads(ln)
    print(f"{d['tag']:8s} {d['W']}x{d['H']} x{d['frames']} {d['content']:7s} enc {d['enc_ms']:.3f} ms {d['enc_frac']:.3f}  dec {d['dec_ms']:.3f} ms {d['dec_frac']:.3f} fps {d['fps']:.0f} diff {d['diff_dwords']}")
PY
mkdir -p gpurun_out/soak
for seed in 11 12; do timeout -k 10 500 python tests/soak.py --rounds 2500 --seed $seed > gpurun_out/soak/s$seed.log 2>&1; echo "soak $seed rc=$?"; tail -1 gpurun_out/soak/s$seed.log; done
