"""GPU: the fused index + decode launch (decode_kernel<IMG, kIdxFused>, csrc/dbde_kernels.hip) -- few LARGE frames (one
4096x3072 frame per call is BASELINE configs[1] taken literally): the decode workgroups build the chunk index among
themselves through epoch-tagged records instead of a separate index kernel.  Checked against the images the frames
were made from and the reference's validation outcomes (dbde_util.cpp:295-303, 335, 342): every image geometry class
(direct, staged, tile by tile), rejected frames (n64, nb, nm, a depth above 8, a truncated stream) beside good ones,
repeated launches on one context (the epoch), the fallback that computes a silent workgroup's record from the stream
(forced: $DBDE_HIP_EXPERIMENT bit 4 makes every odd chunk publish nothing), and two contexts whose launches together
exceed the device's workgroup slots."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0xDBDE2016


@pytest.fixture(scope="module")
def dv():
    import dbde_video_cpp_amd as m
    return m


def _codec(dv, flags=None):
    if flags is not None:
        os.environ["DBDE_HIP_EXPERIMENT"] = str(flags)
    try:
        return dv.Codec(0)
    finally:
        os.environ.pop("DBDE_HIP_EXPERIMENT", None)


def _round_trip(codec, mode, W, H, n, reps=1):
    import torch
    imgs = codec.synth_frames(mode, SEED, 3, n, W, H)
    buf, lead, cap = codec.alloc_stream(W, H, n)
    offs, sizes = codec.encode_frames(imgs, W, H, n, buf, lead, cap, first_index=3)
    for _ in range(reps):
        back, res = codec.decode_frames(buf, lead, cap, offs, W, H, n)
    codec.sync()
    assert torch.equal(back, imgs), (W, H, n, mode)
    s = sizes.cpu().numpy()
    assert codec.parse_results(res) == [(2, 3 + f, 0, int(s[f])) for f in range(n)]
    return imgs, buf, lead, cap, offs, sizes


@pytest.mark.parametrize("flags", [None, 16, 8])     # default (fused), odd chunks silent (fallback), fused off (index kernel)
@pytest.mark.parametrize("W,H,n,mode", [(4096, 3072, 1, "mixed"), (4096, 3072, 2, "noise8"), (1921, 1081, 3, "mixed"),
                                        (1928, 1080, 2, "smooth"), (2048, 2048, 4, "mixed"), (1000, 1000, 5, "mixed"),
                                        (8200, 256, 1, "mixed"), (4096, 3072, 1, "flat")])
def test_round_trip(dv, flags, W, H, n, mode):
    codec = _codec(dv, flags)
    try:
        _round_trip(codec, mode, W, H, n, reps=3)     # three launches: the records of the first two carry older epochs
    finally:
        codec.close()


@pytest.mark.parametrize("flags", [None, 16])
def test_rejected_frames_beside_good_ones(dv, flags):
    """Frame 1: n64 off by one; frame 2: a depth byte of 9 (in the LAST chunk); frame 3: nb wrong; frame 4 good; the
    stream ends inside frame 5's payload.  Rejected frames leave their image untouched and report consumed = 20."""
    import torch
    codec = _codec(dv, flags)
    try:
        W, H, n = 1024, 768, 6         # T = 12288: 24 chunks per frame, 144 in the launch
        T = (W // 8) * (H // 8)
        imgs = codec.synth_frames("mixed", SEED, 0, n, W, H)
        buf, lead, cap = codec.alloc_stream(W, H, n)
        offs, sizes = codec.encode_frames(imgs, W, H, n, buf, lead, cap)
        codec.sync()
        o, s = offs.cpu().numpy(), sizes.cpu().numpy()
        total = int(o[-1] + s[-1])
        b = buf.clone()
        b[lead + int(o[1]) + 28 + 2 * T] ^= 1                   # n64
        b[lead + int(o[2]) + 24 + T - 1] = 9                    # depth of the frame's last tile
        b[lead + int(o[3]) + 20] ^= 1                           # nb
        canvas = torch.full_like(imgs, 0xEE)
        back, res = codec.decode_frames(b, lead, total - 5, offs, W, H, n, images=canvas)
        codec.sync()
        rr = codec.parse_results(res)
        for f in range(n):
            if f in (0, 4):
                assert rr[f] == (2, f, 0, int(s[f])) and torch.equal(back[f], imgs[f]), f
            else:
                assert rr[f][0] == 0xFFFFFFFF and rr[f][3] == 20 and rr[f][1] == f, (f, rr[f])
                assert (back[f] == 0xEE).all(), f
    finally:
        codec.close()


def test_two_contexts_exceed_the_workgroup_slots(dv):
    """Two contexts on two streams, each decoding 2 frames of 4096x3072 (768 workgroups) at the same time: more than
    the device holds at once, so part of each launch waits for slots while the rest is already polling records."""
    import torch
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    codecs = [dv.Codec(0, stream=s) for s in streams]
    try:
        W, H, n = 4096, 3072, 2
        st = []
        for k, c in enumerate(codecs):
            with torch.cuda.stream(streams[k]):
                imgs = c.synth_frames("mixed", SEED, 10 * k, n, W, H)
                buf, lead, cap = c.alloc_stream(W, H, n)
                offs, sizes = c.encode_frames(imgs, W, H, n, buf, lead, cap)
                st.append((imgs, buf, lead, cap, offs, torch.empty_like(imgs)))
        torch.cuda.synchronize(dev)
        for rep in range(8):
            for k, c in enumerate(codecs):
                imgs, buf, lead, cap, offs, back = st[k]
                with torch.cuda.stream(streams[k]):
                    c.decode_frames(buf, lead, cap, offs, W, H, n, images=back)
        for c in codecs:
            c.sync()
        for imgs, buf, lead, cap, offs, back in st:
            assert torch.equal(back, imgs)
    finally:
        for c in codecs:
            c.close()
