O=gpurun_out/r04b_parity; mkdir -p $O; : > $O/ab.jsonl
run() { ABBENCH_DIAG=0 timeout -k 10 120 profiles/abbench $1 $3 $4 $5 $6 slots 20 $2 >> $O/ab.jsonl 2>> $O/ab.err || echo "abbench $* rc=$?"; }
for shape in "72 72 262144" "80 64 262144" "72 64 262144" "88 72 131072" "104 72 131072" "96 64 131072" "80 72 262144" "64 64 262144" "88 56 262144"; do
  for content in mixed noise8; do run dbde-video-cpp_amd/libdbde_hip.so cur $shape $content; done
done
python3 - <<PY
import json
for ln in open("$O/ab.jsonl"):
    d = json.loads(ln)
    T=((d['W']+7)//8)*((d['H']+7)//8)
    tiles=T*d['frames']
    print(f"{d['W']}x{d['H']} T={T} fpw={256//T} fill={(256//T)*T/256:.2f} meta%16={(32+2*T)%16} {d['content']:7s} enc {d['enc_ms']:.3f} ms {d['enc_frac']:.3f} {d['enc_ms']*1e9/tiles:.1f} ps/tile  dec {d['dec_ms']:.3f} ms {d['dec_frac']:.3f} {d['dec_ms']*1e9/tiles:.1f} ps/tile")
PY
