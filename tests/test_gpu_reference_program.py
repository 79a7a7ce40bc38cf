"""GPU: the reference's own test program (dbde_util_test.cpp), compiled unchanged and linked
against libdbde_util_hip.so instead of the reference's dbde_util.o (oracle/Makefile builds it
into oracle/_ref/ where /root/reference exists).  It must pass exactly as it does against the
reference library: tile demos, the 8x16 known-answer test, the 2536x2048 round trip, 1024
randomized round trips, and -- in the second binary -- the file walker over a 250-frame file."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "dbde_test_hip")
BIN_WALK = os.path.join(ROOT, "oracle", "_ref", "dbde_test_hip_walk")


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/dbde_test_hip not built")
def test_reference_test_program_passes_on_hip():
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = r.stdout
    lines = out.splitlines()
    # the four printed tile demos (dbde_util_test.cpp:234-299): codes as probed on the reference
    codes = [l.strip() for l in lines if re.fullmatch(r"[0-9a-f]{1,3}", l.strip())]
    assert codes[:4] == ["513", "218", "31c", "1a"], codes[:6]
    # the 2536x2048 round trip (:303-349): sizes agree and no pixel differs
    assert "5356064 5356064" in out
    assert "!! 0 of 5193728" in out
    assert "Failed iteration" not in out and "does not match" not in out


@pytest.mark.skipif(not os.path.exists(BIN_WALK), reason="oracle/_ref/dbde_test_hip_walk not built")
def test_reference_file_walk_on_hip(tmp_path):
    r = subprocess.run([BIN_WALK], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "FAILED OPEN" not in r.stdout
    # prints a progress number every 100 frames: 250 frames -> "100" and "200"
    nums = [l.strip() for l in r.stdout.splitlines() if l.strip() in ("100", "200")]
    assert nums == ["100", "200"], r.stdout[-800:]
    assert os.path.getsize("/tmp/dbde_min.dbde") == 28 + 250 * 100
    # frame 3 dumped as PGM equals the 8x16 known-answer image
    pgm = open(os.path.join(str(tmp_path), "a_frame.pgm")).read().split()
    assert pgm[:4] == ["P2", "16", "8", "255"]
    got = np.array(pgm[4:], int).reshape(8, 16)
    want = np.load(os.path.join(ROOT, "tests", "golden", "frames.npz"))["kat_8x16.image"]
    assert (got == want).all()
