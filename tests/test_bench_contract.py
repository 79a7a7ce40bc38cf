"""CPU: the bench line committed under profiles/ (produced by bench.py on the GPU box) carries
every field the measurement contract names, and bench.py still emits the same set."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TOP = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
       "vs_baseline", "dtype", "data", "config", "roofline"]
ROOF = ["bound", "achieved", "peak", "unit", "frac", "traffic"]
CPU = ["value", "unit", "cores", "kind", "sample"]


def test_committed_bench_lines_follow_the_contract():
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "bench_r*_noise8.json")))
    assert lines, "no bench line committed under profiles/"
    d = json.load(open(lines[-1]))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in TOP:
        assert k in d, k
    assert d["metric"] == base["metric"].replace("×", "x") and d["unit"] == "frames/s"
    assert d["dtype"] == "u8" and d["vs_baseline"] is None and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"]
    for k in ROOF:
        assert k in d["roofline"], k
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # measured traffic within a few per cent of the algorithmic bytes: nothing is re-read
    assert r["traffic"] is not None and 0.98 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.05
    for k in CPU:
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["mismatched_pixels"] == 0
    # value is whole-job frames/s: frames per step / time per step
    fps = d["config"]["frames_per_step_per_gpu"] * d["n_gpus"] / (d["ms_per_step"] * 1e-3)
    assert abs(fps - d["value"]) / d["value"] < 0.01


def test_bench_source_emits_the_contract_fields():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for k in TOP + ["cpu_baseline"]:
        assert re.search(r'"%s"' % k, src), k
    for k in ROOF + CPU:
        assert '"%s"' % k in src, k
    # only the cpu_baseline leg may touch the oracle
    uses = [m.start() for m in re.finditer(r"oracle_ffi|Oracle\(|Reference\(", src)]
    lo, hi = src.index("def cpu_baseline"), src.index("def main")
    assert uses and all(lo <= u < hi for u in uses), "oracle used outside cpu_baseline"
