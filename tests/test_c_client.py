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


THREADS_SRC = os.path.join(ROOT, "tests", "c_client", "threads.cpp")


def build_threads_client(out_dir):
    """g++ only: the client sees include/dbde_util.h and links libdbde_util_hip.so, as a user of the reference would."""
    import dbde_video_cpp_amd as dv
    if not os.path.exists(dv.LIB_PATH):
        dv.build()
    exe = os.path.join(str(out_dir), "threads")
    subprocess.run(["g++", "-std=c++14", "-O2", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"), THREADS_SRC,
                    "-L", dv.PKG_DIR, "-ldbde_util_hip", "-Wl,-rpath," + dv.PKG_DIR, "-o", exe],
                   check=True, capture_output=True, text=True)
    return exe


def write_thread_inputs(out_dir, W, H, n, mode, packer):
    """n distinct synthetic frames and what `packer` (the real reference where it is built, else the oracle) makes of them."""
    import numpy as np
    from oracle_ffi import Oracle
    ora = Oracle()
    frames, expected = os.path.join(str(out_dir), "frames.bin"), os.path.join(str(out_dir), "expected.bin")
    with open(frames, "wb") as f, open(expected, "wb") as e:
        for t in range(n):
            img = ora.synth_frame(mode, 0xDBDE2016, 77 + t, W, H)
            f.write(img.tobytes())
            packed = packer.pack_frame(1000 + t, img, W, H)
            e.write(np.uint64(len(packed)).tobytes())
            e.write(packed.tobytes())
    return frames, expected


def test_threads_client_builds(tmp_path):
    exe = build_threads_client(tmp_path)
    out = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True, check=True).stdout
    assert "_Z15dbde_pack_framemPhiiS_" in out and "_Z17dbde_unpack_framePPhiiS_" in out   # the reference's own mangled symbols
    assert "hip" not in out.lower().replace("dbde", ""), "a user of the drop-in needs neither the C-ABI nor the HIP runtime by name"


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,threads,mode", [(200, 123, 8, 1), (1921, 1081, 6, 1), (64, 64, 12, 0)])
def test_drop_in_api_from_many_threads(tmp_path, W, H, threads, mode):
    """The reference's API called from several threads at once on distinct buffers (it is re-entrant: dbde_util.h:21-37):
    every packed frame equals the REAL reference's bytes, every unpacked image its input, nothing is written behind a
    frame -- with more threads than the shim may hold contexts (DBDE_HIP_SHIM_CONTEXTS=4), so that callers also queue."""
    from oracle_ffi import Oracle, Reference
    packer = Reference() if Reference.available() else Oracle()
    exe = build_threads_client(tmp_path)
    frames, expected = write_thread_inputs(tmp_path, W, H, threads, mode, packer)
    for limit in ("4", "16"):
        r = subprocess.run([exe, str(W), str(H), str(threads), "12", frames, expected], capture_output=True, text=True,
                           timeout=300, env=dict(os.environ, DBDE_HIP_SHIM_CONTEXTS=limit))
        assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d["mismatches"] == 0 and d["threads"] == threads
