#!/bin/bash
# r02_bench.sh TAG [bench args]: GPU parity suite + one bench.py line into gpurun_out/
T=${1:-r02}; shift
O=gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/${T}_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/${T}_tests.log
timeout -k 10 500 python bench.py "$@" > $O/bench_${T}.json 2> $O/bench_${T}.err; echo "bench rc=$?"
python3 - <<PY
import json
d = json.load(open("$O/bench_${T}.json"))
r = d["roofline"]
print("value", d["value"], "enc", r.get("encode", {}).get("frac"), "dec", r.get("decode", {}).get("frac"))
for k, v in d.get("contents", {}).items(): print(" content", k, v["frames_per_s"], v["encode"]["frac"], v["decode"]["frac"])
for k, v in d.get("configs", {}).items(): print(" config", k, v["frames_per_s"], v["encode"]["frac"], v["decode"]["frac"], v["decode"].get("scan_ms"), v["round_trip_frac"])
print(" single", d.get("single_frame"))
print(" cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
PY
