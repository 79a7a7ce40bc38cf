"""CPU: the bench line committed under profiles/ (produced by bench.py on the GPU box) carries
every field the measurement contract names, and bench.py still emits the same set."""
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TOP = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
       "vs_baseline", "dtype", "data", "config", "roofline"]
ROOF = ["bound", "achieved", "peak", "unit", "frac", "traffic"]
CPU = ["value", "unit", "cores", "kind", "sample"]


def test_committed_bench_lines_follow_the_contract():
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "bench_r*_default.json")))   # `python bench.py`, no flags
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
    # measured traffic (replayed from the PMC passes taken on the same kernel sources, else null) within a few
    # per cent of the algorithmic bytes: nothing is re-read
    if r["traffic"] is not None:
        assert 0.98 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.05 and r["traffic_source"]
    # the path that really bit-packs and the other single-GPU configs are in the same line
    assert d["contents"]["mixed"]["identical"] and {"3", "4"} <= set(d["configs"])
    # every timed configuration that has a reference-made fixture was also checked against it (not identity alone)
    assert d["contents"]["mixed"]["packed_sha_ok"] is True
    assert all(d["configs"][k]["packed_sha_ok"] is True for k in ("3", "4"))
    if "shapes" in d:   # round 3 on: shapes whose kernel forms differ from the configs', gated on the reference's SHA-256 as well
        assert all(v.get("packed_sha_ok") is True for v in d["shapes"].values()), d["shapes"]
    if "dbde16" in d:   # the extension's record (round 3 on): reported beside the headline, round trip checked
        assert d["dbde16"]["identical"] is True and d["dbde16"]["parity"].startswith("unpinned")
    assert d["cpu_baseline"]["mixed"]["value"] > 0 and d["cpu_baseline"]["mixed"]["mismatched_pixels"] == 0
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
    lo = src.index("def cpu_baseline")
    hi = lo + re.search(r"\n(?:def|class) ", src[lo + 1:]).start()
    assert uses and all(lo <= u < hi for u in uses), "oracle used outside cpu_baseline"


def run_bench(*argv, env=None):
    e = dict(os.environ)
    e.update(env or {})
    e.pop("WORLD_SIZE", None), e.pop("RANK", None), e.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True,
                       env=e, timeout=300)
    return r


def test_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` (the driver's command form, no launcher) must itself start two ranks: the
    dry run exercises the spawn, the rendezvous on 127.0.0.1, sharding, barrier, max-over-ranks and the
    variable-length gather over gloo -- everything around the GPU work -- and reports n_gpus from the group."""
    r = run_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["steps"] == 3 and d["warmup"] == 1
    assert d["frames_of_rank0"] == [0, 16] and d["gathered_bytes"] == 100 + 107
    assert d["value"] == 0.0 and "dry run" in d["data"]


def test_world_size_must_match_gpus_flag():
    """Launched under a launcher with a different world size, bench.py refuses instead of mislabelling."""
    e = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                       capture_output=True, text=True, env=e, timeout=120)
    assert r.returncode != 0 and "--gpus 2" in r.stderr


def test_a_stuck_exchange_leaves_non_zero_with_the_line_printed():
    """A rank that never answers a collective of the exchange leg: rank 0's watchdog prints the line it has (with the
    error in it) and the run ends NON-ZERO -- a launcher must not read a hung exchange as success."""
    r = run_bench("--gpus", "2", "--steps", "2", "--warmup", "0", "--dry-run",
                  env={"DBDE_BENCH_DRY_FAIL": "skip_gather", "DBDE_BENCH_WATCHDOG_S": "8"})
    assert r.returncode != 0, (r.stdout, r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "error" in d["gather"] and d["gathered_bytes"] is None


def test_bench_source_leaves_non_zero_on_exchange_failure():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "os._exit(0)" not in src, "a process that gave up on an exchange must not leave with 0"
    assert src.count("os._exit(EXIT_EXCHANGE_FAILED)") >= 2      # the watchdog and the failed-gather path
    assert "packed_sha_ok_ranks" in src                          # every rank is held to the reference SHA
