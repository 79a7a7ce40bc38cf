#!/usr/bin/env python3
"""Summarise a gpurun_out/prof_<tag>_<content>/ directory (see run_profile.sh) into
profiles/<tag>_<content>_summary.txt and update profiles/hbm_traffic.json.

HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are collected in
separate passes, are in KiB, and on gfx950 FETCH_SIZE reports half of a wide coalesced read
stream, so it is doubled."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read_csvs(pattern):
    rows = []
    for f in glob.glob(pattern, recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    return rows


def main():
    d = sys.argv[1].rstrip("/")
    tag = os.path.basename(d).replace("prof_", "")
    content = "_".join(tag.split("_")[-2:])        # cfg<N>_<content>: the key bench.py looks up
    out = []
    # 1. kernel trace
    kt = read_csvs(os.path.join(d, "kt", "**", "*kernel_trace.csv"))
    dur = defaultdict(list)
    for r in kt:
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out.append("== kernel trace (rocprofv3 --kernel-trace): name, calls, avg us, min us, max us, total us, "
               "avg us of the timed steps (the first 2 dispatches = bench.py's warm-up dropped; '-' below 6 calls)")
    start = defaultdict(list)
    for r in kt:
        start[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        timed = [x[1] for x in sorted(start[k])[2:]] if len(v) >= 6 else []
        steady = f"{sum(timed)/len(timed)/1e3:10.2f}" if timed else "         -"
        out.append(f"{k[:90]:90s} {len(v):6d} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} {max(v)/1e3:10.2f} {sum(v)/1e3:12.1f} {steady}")
    # 2/3/4 counters: per kernel averages
    traffic = {}
    for sub in ("fetch", "write", "sq"):
        rows = read_csvs(os.path.join(d, sub, "**", "*counter_collection.csv"))
        acc = defaultdict(lambda: defaultdict(list))
        for r in rows:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        out.append(f"== counters ({sub}): kernel, counter, avg per dispatch, dispatches")
        for k in acc:
            for c, v in acc[k].items():
                avg = sum(v) / len(v)
                out.append(f"{k[:70]:70s} {c:24s} {avg:18.1f} {len(v):6d}")
                if c in ("FETCH_SIZE", "WRITE_SIZE") and ("dbde::encode_kernel" in k or "dbde::decode_kernel" in k):
                    name = "dbde::encode_kernel" if "encode_kernel" in k else "dbde::decode_kernel"
                    traffic.setdefault(name, {})[c] = avg
    for name, t in traffic.items():
        if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
            rd, wr = 2 * t["FETCH_SIZE"] * 1024, t["WRITE_SIZE"] * 1024
            out.append(f"== HBM traffic per launch {name}: read {rd/1e6:.1f} MB (2 x FETCH_SIZE KiB), "
                       f"write {wr/1e6:.1f} MB, total {(rd+wr)/1e6:.1f} MB")
            t["bytes"] = rd + wr
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, f"{tag}_summary.txt"), "w") as f:
        f.write("\n".join(out) + "\n")
    tf = os.path.join(here, "hbm_traffic.json")
    cur = json.load(open(tf)) if os.path.exists(tf) else {}
    cur[content] = {k: int(v["bytes"]) for k, v in traffic.items() if "bytes" in v}
    # the workload and the kernel sources this record belongs to (bench.py replays it only when both match)
    line = os.path.join(d, "bench_line.json")
    if os.path.exists(line):
        try:
            cur[content]["frames_per_launch"] = json.load(open(line))["config"]["frames_per_step_per_gpu"]
        except Exception:
            pass
    sys.path.insert(0, os.path.dirname(here))
    import bench
    cur["kernels_sha"] = bench.kernels_fingerprint()
    cur["tag"] = tag.split("_")[0]
    # the revision the pass was taken at (profiles/rekey.py compares later sources against it; "+dirty": uncommitted edits)
    try:
        import subprocess
        root = os.path.dirname(here)
        rev = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "dbde-video-cpp_amd/csrc"], capture_output=True, text=True).stdout.strip()
        if rev:
            cur["revision"] = rev + ("+dirty" if dirty else "")
    except Exception:
        pass
    cur.pop("rekeyed", None)
    cur.pop("isa_sha", None)
    json.dump(cur, open(tf, "w"), indent=1)
    print("\n".join(out))


if __name__ == "__main__":
    main()
