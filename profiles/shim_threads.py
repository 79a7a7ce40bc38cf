#!/usr/bin/env python3
"""The drop-in API (libdbde_util_hip.so: the reference's own signatures, host buffers in and out) from 1 / 4 / 16 caller
threads: 4096x3072 round trips per second, PCIe inclusive (tests/c_client/threads.cpp, a C++ program that sees
include/dbde_util.h only).  Never the headline -- bench.py measures device-resident batches -- but this is what the
one-line makefile swap of INTEGRATION.md gives a maintainer.  Expected bytes come from the real reference (oracle/_ref)
where it is built, else from the oracle."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_ffi import Oracle, Reference  # noqa: E402
from test_c_client import build_threads_client, write_thread_inputs  # noqa: E402

W, H = 4096, 3072
packer = Reference() if Reference.available() else Oracle()
out = []
with tempfile.TemporaryDirectory() as d:
    exe = build_threads_client(d)
    for content, mode in (("mixed", 1), ("noise8", 0)):
        frames, expected = write_thread_inputs(d, W, H, 16, mode, packer)
        for threads, reps, pinned_from in ((1, 60, 6), (2, 60, 6), (4, 60, 6), (8, 40, 6), (16, 30, 6), (4, 60, 1), (8, 40, 1), (16, 30, 1), (8, 40, 0), (16, 30, 0)):
            env = dict(os.environ, THREADS_CHECK_LAST_ONLY="1", DBDE_HIP_SHIM_PINNED_FROM=str(pinned_from))
            r = subprocess.run([exe, str(W), str(H), str(threads), str(reps), frames, expected], capture_output=True, text=True, timeout=600, env=env)
            rec = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-300:], "rc": r.returncode}
            rec.update({"content": content, "threads": threads, "checked_against": type(packer).__name__,
                        "pinned_staging_from_calls_in_flight": pinned_from})
            print(json.dumps(rec), flush=True)
            out.append(rec)
