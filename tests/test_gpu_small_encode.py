"""GPU: the small-launch encoder (encode_small_kernel, csrc/dbde_kernels.hip) -- launches with fewer chunks than the
device holds workgroups, one 4096x3072 frame per call above all (BASELINE configs[1] taken literally).  A workgroup's
chunk is its id; word counts travel as epoch-tagged records that nothing ever clears.  Checked byte for byte against
the oracle: every input path (aligned, any geometry, narrower than 16), both layouts, repeated launches on one context
(the epoch; records of earlier launches and of the persistent encoder's launches stay in the workspace), shrinking and
growing launches, the fallback that computes a silent workgroup's count from the pixels (forced: $DBDE_HIP_EXPERIMENT
bit 6 makes every odd chunk publish nothing), and two contexts on concurrent streams."""
import os

import numpy as np
import pytest

from oracle_ffi import Oracle   # checker only

pytestmark = pytest.mark.gpu
SEED = 0xDBDE2016


@pytest.fixture(scope="module")
def dv():
    import dbde_video_cpp_amd as m
    return m


@pytest.fixture(scope="module")
def ora():
    return Oracle()


def _codec(dv, flags=None):
    if flags is not None:
        os.environ["DBDE_HIP_EXPERIMENT"] = str(flags)
    try:
        return dv.Codec(0)
    finally:
        os.environ.pop("DBDE_HIP_EXPERIMENT", None)


def _check(codec, ora, mode, W, H, n, slots, first=7, reps=1):
    import torch
    imgs = codec.synth_frames(mode, SEED, first, n, W, H)
    maxf = int(codec.L.dbde_hip_max_frame_bytes(W, H))
    slot = ((maxf + 255) // 256) * 256 if slots else 0
    cap = (n - 1) * slot + maxf if slots else n * maxf
    buf = torch.full((32 + cap + 64,), 0xEE, dtype=torch.uint8, device="cuda")
    for _ in range(reps):
        offs, sizes = codec.encode_frames(imgs, W, H, n, buf, 32, cap, first_index=first, slot_stride=slot)
    back, res = codec.decode_frames(buf, 32, cap, offs, W, H, n)
    codec.sync()
    assert torch.equal(back, imgs), (mode, W, H, n, slots)
    host, o, s = buf.cpu().numpy(), offs.cpu().numpy(), sizes.cpu().numpy()
    ih = imgs.cpu().numpy()
    for f in range(n):
        want = ora.pack_frame(first + f, ih[f], W, H)
        got = host[32 + int(o[f]): 32 + int(o[f] + s[f])]
        assert int(s[f]) == len(want) and got.tobytes() == want.tobytes(), (mode, W, H, n, slots, f)
    if not slots:
        assert o[0] == 0 and (o[1:] == np.cumsum(s)[:-1]).all()
        assert (host[32 + int(o[-1] + s[-1]):-64] == 0xEE).all()


@pytest.mark.parametrize("flags", [None, 64])     # default; odd chunks silent: every waiter computes their counts itself
@pytest.mark.parametrize("W,H,n,mode", [(4096, 3072, 1, "mixed"), (4096, 3072, 2, "noise8"), (1921, 1081, 3, "mixed"),
                                        (1928, 1080, 2, "smooth"), (2048, 2048, 5, "mixed"), (1000, 1003, 4, "mixed"),
                                        (15, 700, 6, "mixed"), (8200, 256, 2, "noise8"), (200, 123, 9, "flat")])
@pytest.mark.parametrize("slots", [False, True])
def test_small_launches_match_the_oracle(dv, ora, flags, W, H, n, mode, slots):
    codec = _codec(dv, flags)
    try:
        _check(codec, ora, mode, W, H, n, slots, reps=3)   # three launches: the records of the first two carry older epochs
    finally:
        codec.close()


def test_small_and_large_launches_share_the_workspace(dv, ora):
    """Small launches leave epoch-tagged records behind, the persistent encoder AGG / INC records: each kind must be
    invisible to the other, whatever the order and however the workspace grows."""
    codec = _codec(dv)
    try:
        _check(codec, ora, "mixed", 2048, 2048, 3, False)        # small: 192 chunks
        _check(codec, ora, "mixed", 2048, 2048, 40, False)       # persistent: 2560 chunks (the workspace grows)
        _check(codec, ora, "noise8", 2048, 2048, 7, True)        # small again, over the persistent launch's records
        _check(codec, ora, "mixed", 1921, 1081, 2, False)
        _check(codec, ora, "mixed", 1024, 768, 300, True)        # persistent
        _check(codec, ora, "mixed", 4096, 3072, 1, False, reps=5)
        _check(codec, ora, "smooth", 64, 64, 100, False)          # tiny frames concatenated: one chunk per frame
    finally:
        codec.close()


def test_two_contexts_on_concurrent_streams(dv, ora):
    """Two contexts (own streams, own workspaces) encoding single frames at the same time: together their workgroups
    exceed nothing here, but neither may see the other's records."""
    import torch
    a, b = _codec(dv), _codec(dv, 64)
    try:
        W, H = 4096, 3072
        ia, ib = a.synth_frames("mixed", SEED, 0, 1, W, H), b.synth_frames("noise8", SEED, 1, 1, W, H)
        bufa, lead, cap = a.alloc_stream(W, H, 1)
        bufb, _, _ = b.alloc_stream(W, H, 1)
        for _ in range(50):
            oa, sa = a.encode_frames(ia, W, H, 1, bufa, lead, cap)
            ob, sb = b.encode_frames(ib, W, H, 1, bufb, lead, cap, first_index=1)
        a.sync(); b.sync()
        wa = ora.pack_frame(0, ia[0].cpu().numpy(), W, H)
        wb = ora.pack_frame(1, ib[0].cpu().numpy(), W, H)
        assert bufa[lead: lead + int(sa[0])].cpu().numpy().tobytes() == wa.tobytes()
        assert bufb[lead: lead + int(sb[0])].cpu().numpy().tobytes() == wb.tobytes()
    finally:
        a.close(); b.close()
