"""CPU: the C-ABI library loads and exports every symbol include/dbde_hip.h declares; the
dbde_util.h shim exports the reference's mangled C++ symbols.  No compute."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REFERENCE_MANGLED = [
    "_Z13dbde_pack_8x8PhiS_", "_Z21dbde_pack_8x8_partialPhiiiS_", "_Z15dbde_pack_imagePhiiS_",
    "_Z22dbde_pack_frame_header12frame_headerPh", "_Z15dbde_pack_framemPhiiS_",
    "_Z22dbde_pack_video_header12video_headerPh", "_Z15dbde_unpack_8x8hhPhmS_",
    "_Z23dbde_unpack_8x8_partialhhPhmiiS_", "_Z17dbde_unpack_imagePhiiS_",
    "_Z24dbde_unpack_frame_headerPPh", "_Z17dbde_unpack_framePPhiiS_", "_Z24dbde_unpack_video_headerPPh",
    "_Z20dbde_start_file_walkPKciP12video_header", "_Z16dbde_walk_a_fileP16dbde_file_walkerP12frame_headerPh",
    "_Z18dbde_end_file_walkP16dbde_file_walker", "_Z24dbde_advance_file_bufferR16dbde_file_walker",
]


def _ensure_built():
    import dbde_video_cpp_amd as dv
    if not (os.path.exists(dv.LIB_PATH) and os.path.exists(dv.SHIM_PATH)):
        dv.build()
    return dv


def test_header_and_export_list_agree():
    dv = _ensure_built()
    header = open(os.path.join(ROOT, "include", "dbde_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(dbde(?:16)?_hip_[a-z0-9_]+)\s*\(", header)))
    assert declared == sorted(dv.C_ABI_SYMBOLS), set(declared) ^ set(dv.C_ABI_SYMBOLS)
    out = subprocess.run(["nm", "-D", "--defined-only", dv.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (dbde(?:16)?_hip_[a-z0-9_]+)", out))
    assert set(declared) <= exported, set(declared) - exported
    lib = dv.lib()                          # loads (no GPU needed) and binds every symbol
    assert all(hasattr(lib, s) for s in declared)


def test_shim_exports_reference_symbols():
    dv = _ensure_built()
    out = subprocess.run(["nm", "-D", "--defined-only", dv.SHIM_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (\S+)", out))
    assert set(REFERENCE_MANGLED) <= exported, set(REFERENCE_MANGLED) - exported


def test_host_header_functions_and_sizes():
    """Header wire format is host code in the C-ABI: check it against the golden vectors."""
    import json
    dv = _ensure_built()
    m = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))
    for h in m["headers"]["frame"]:
        wire = dv.pack_frame_header(*h["in"])
        assert wire.tobytes().hex() == h["wire"]
        n, out = dv.unpack_frame_header(wire)
        assert n == h["advance"] and list(out) == h["out"]
    for h in m["headers"]["video"]:
        wire = dv.pack_video_header(*h["in"])
        assert wire.tobytes().hex() == h["wire"]
        n, out = dv.unpack_video_header(wire)
        assert n == h["advance"] and list(out) == h["out"]
    assert dv.max_frame_bytes(4096, 3072) == 12976160 and dv.max_frame_bytes(10, 10) == 296
    assert dv.max_frame_bytes(0, 5) == 0


def test_no_cpu_fallback():
    """Without a GPU the codec must refuse to exist, not compute on the CPU."""
    import pytest
    import torch
    dv = _ensure_built()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(dv.DbdeError):
        dv.Codec(0)
    # and nothing in the product tree reaches for the oracle (test infrastructure only)
    pkg = os.path.join(ROOT, "dbde-video-cpp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                for needle in ("liboracle", "dbde_oracle", "oracle_ffi", "oracle/_ref", "libdbde_ref"):
                    assert needle not in text, (f, needle)
