#!/bin/bash
# round 4, second half: a sweep of everyday shapes through abbench (about 4 GB of pixels per launch, mixed and incompressible)
# to find outliers: W H -> encode / decode fraction of 8 TB/s, which plan forms they take
O=gpurun_out/r04b_sweep; mkdir -p $O; : > $O/ab.jsonl
for shape in "640 480" "641 481" "800 600" "1000 600" "1024 768" "1280 720" "1280 1024" "1366 768" "1600 1200" "1920 1080" "1920 1200" "2048 1080" "2560 1440" "2592 1944" "3840 2160" "4000 3000" "4096 2160" "5120 2880" "1936 1216" "1288 964" "2448 2048" "808 608" "1004 1002" "750 1334" "1125 2436" "480 270" "352 288" "240 320" "200 200" "150 150" "100 75"; do
  set -- $shape
  n=$(( 4000000000 / ($1 * $2) )); [ $n -gt 262144 ] && n=262144
  for content in mixed noise8; do
    ABBENCH_DIAG=0 timeout -k 10 120 profiles/abbench dbde-video-cpp_amd/libdbde_hip.so $1 $2 $n $content slots 10 sweep >> $O/ab.jsonl 2>> $O/ab.err || echo "abbench $shape $content rc=$?"
  done
done
python3 - <<PY
import json
rows = {}
for ln in open("$O/ab.jsonl"):
    d = json.loads(ln)
    rows.setdefault((d['W'], d['H'], d['frames']), {})[d['content']] = d
for (W, H, n), c in rows.items():
    T = ((W + 7) // 8) * ((H + 7) // 8)
    m, z = c.get('mixed'), c.get('noise8')
    f = lambda d: f"enc {d['enc_frac']:.3f} dec {d['dec_frac']:.3f} idx {d['idx_ms']/max(d['dec_ms'],1e-9):.2f} diff {d['diff_dwords']}" if d else "-"
    print(f"{W}x{H} x{n} T={T} T%4={T%4} W%16={W%16}: mixed {f(m)} | noise8 {f(z)}")
PY
