#!/bin/bash
# round 4, second half: what the encoder's any-alignment output path costs (tile counts that are no multiple of 4) and what
# aligned 16-byte blocks re-aligned from the LDS words give: in-tree library against profiles/ab_libs/prev (built in the
# container from the commit before), then the parity tests that exercise unaligned frames
O=gpurun_out/r04b_unaligned; mkdir -p $O; : > $O/ab.jsonl
run() { ABBENCH_DIAG=0 timeout -k 10 120 profiles/abbench $1 $3 $4 $5 $6 $7 20 $2 >> $O/ab.jsonl 2>> $O/ab.err || echo "abbench $* rc=$?"; }
for shape in "1008 1008 4096" "1008 1000 4096" "1000 1000 4096" "1016 1000 4096" "2000 1000 2048" "3000 2008 700" "1001 1003 4096"; do
  for content in mixed noise8; do
    run dbde-video-cpp_amd/libdbde_hip.so new $shape $content slots
    run profiles/ab_libs/prev/libdbde_hip.so prev $shape $content slots
  done
done
run dbde-video-cpp_amd/libdbde_hip.so new 1008 1000 4096 mixed concat
run profiles/ab_libs/prev/libdbde_hip.so prev 1008 1000 4096 mixed concat
python3 - <<PY
import json
for ln in open("$O/ab.jsonl"):
    d = json.loads(ln)
    T=((d['W']+7)//8)*((d['H']+7)//8)
    print(f"{d['tag']:5s} {d['W']}x{d['H']} T%4={T%4} {d['layout']:6s} {d['content']:7s} enc {d['enc_ms']:.3f} ms {d['enc_frac']:.3f}  dec {d['dec_ms']:.3f} ms {d['dec_frac']:.3f} diff {d['diff_dwords']}")
PY
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
