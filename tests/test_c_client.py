"""The C-ABI from plain C: include/dbde_hip.h is C99-clean, a gcc-built client links against
libdbde_hip.so alone (CPU: compile + link; GPU: run it against the README golden bytes)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "tests", "c_client", "roundtrip.c")


def _build(tmp_path):
    import dbde_video_cpp_amd as dv
    if not os.path.exists(dv.LIB_PATH):
        dv.build()
    exe = str(tmp_path / "roundtrip")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-O1", "-D_POSIX_C_SOURCE=200809L",
                    "-I", os.path.join(ROOT, "include"), SRC, "-L", dv.PKG_DIR, "-ldbde_hip",
                    "-Wl,-rpath," + dv.PKG_DIR, "-o", exe], check=True, capture_output=True, text=True)
    return exe


def test_headers_are_c99_and_cxx14_clean():
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                    os.path.join(ROOT, "include", "dbde_hip.h")], check=True)
    subprocess.run(["g++", "-std=c++14", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++",
                    os.path.join(ROOT, "include", "dbde_util.h")], check=True)


def test_c_client_builds_with_gcc(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True, check=True).stdout
    assert "dbde_hip_pack_frame" in out and "dbde_hip_unpack_frame" in out
    assert "hip" not in out.replace("dbde_hip_", ""), "the client must not need the HIP runtime itself"


@pytest.mark.gpu
def test_c_client_runs(tmp_path):
    exe = _build(tmp_path)
    m = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))
    readme = [f for f in m["frames"] if f["name"].startswith("readme")][0]
    import numpy as np
    arrays = np.load(os.path.join(ROOT, "tests", "golden", "frames.npz"))
    assert readme["index"] == 7 and readme["packed_bytes"] == 112
    image = arrays[readme["name"] + ".image"].tobytes().hex()
    golden = arrays[readme["name"] + ".packed"].tobytes().hex()   # produced by the real reference
    r = subprocess.run([exe, image, golden], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.stdout, r.stderr)


GATHER_SRC = os.path.join(ROOT, "tests", "c_client", "gather1.cpp")


def _build_gather(tmp_path):
    import dbde_video_cpp_amd as dv
    if not os.path.exists(dv.LIB_PATH):
        dv.build()
    exe = str(tmp_path / "gather1")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-I", os.path.join(ROOT, "include"), GATHER_SRC,
                    "-L", dv.PKG_DIR, "-ldbde_hip", "-Wl,-rpath," + dv.PKG_DIR, "-o", exe], check=True, capture_output=True, text=True)
    return exe


def test_gather_client_builds(tmp_path):
    exe = _build_gather(tmp_path)
    out = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True, check=True).stdout
    assert "dbde_hip_gather_post" in out and "nccl" not in out, "the caller needs the C-ABI only, not RCCL itself"


@pytest.mark.gpu
def test_gather_client_runs_one_rank_rccl(tmp_path):
    """The native gather at nranks = 1 from a process without Python or torch: RCCL communicator, size all-gather,
    in-place root segment, and the ncclSend / ncclRecv path in loopback (tests/c_client/gather1.cpp)."""
    exe = _build_gather(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout[-500:], r.stderr[-1500:])
