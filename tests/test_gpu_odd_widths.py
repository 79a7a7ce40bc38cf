"""GPU: widths that are not multiples of 16, one slot per frame and concatenated -- the encoder's any-geometry input
path (kInRaw: the 16 bytes of a row's last lane run into the next image row and are replaced by the constant padding,
dbde_util.cpp:116-128; in the batch's LAST image row the fetch is moved left instead, so that nothing is read past the
caller's buffer; bottom padding, :129-132) and the decoder's staged copy-out (tile-aligned LDS image, re-alignment on the
read side).  Every frame byte for byte against the oracle, odd / even tile counts, one chunk and many, byte-granular
output (T % 4 != 0), explicit indices / elapsed_ns; BASELINE configs[3] against the reference's own SHA-256."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0xDBDE2016


@pytest.fixture(scope="module")
def dv():
    import dbde_video_cpp_amd as m
    return m


@pytest.fixture(scope="module")
def rows_codec(dv):
    c = dv.Codec(0)
    yield c
    c.close()


def _encode_slots(codec, imgs, W, H, n, first_index=0, **kw):
    import torch
    slot = ((codec.L.dbde_hip_max_frame_bytes(W, H) + 255) // 256) * 256
    buf, lead, cap = codec.alloc_stream(W, H, n, slot_stride=slot)
    buf.fill_(0xEE)
    offs, sizes = codec.encode_frames(imgs, W, H, n, buf, lead, cap, first_index=first_index, slot_stride=slot, **kw)
    codec.sync()
    o, s = offs.cpu().numpy(), sizes.cpu().numpy()
    assert all(o[f] == f * slot for f in range(n))
    return buf, lead, cap, slot, offs, sizes, o, s


@pytest.mark.parametrize("W,H,n", [(513, 17, 3), (520, 9, 2), (1001, 33, 5), (1025, 64, 2), (2047, 8, 3), (4095, 24, 2),
                                   (1921, 1081, 2), (8200, 9, 2), (777, 777, 1), (515, 1, 4), (4096, 3072, 2), (2048, 2048, 3),
                                   (1920, 1080, 2), (10, 10, 5), (15, 40, 3), (16, 16, 4), (33, 31, 7), (1, 1, 3), (200, 123, 6),
                                   # 16-byte aligned rows that are not whole cache lines: chunks of whole tile rows decode staged
                                   # (all depth 8) or with direct 16-byte stores (anything else), per chunk; low chunk fill
                                   # (1440, 1600 wide): direct 16-byte stores from plain chunks; few frames take the fused
                                   # index + decode launch, many the index kernel; heights that end inside a tile
                                   (720, 1283, 3), (720, 1283, 40), (1440, 900, 2), (1440, 900, 40), (1360, 765, 3), (1600, 20, 300),
                                   (1080, 1925, 2), (1080, 1925, 30),
                                   # whole tile rows that fill 384 tile slots better than 512: the 192-thread decode workgroup
                                   # (odd rows, 8-byte rows, 16-byte rows; one and two tile rows per chunk; fused / index kernel)
                                   (1366, 768, 2), (1366, 768, 40), (1365, 33, 50), (3000, 40, 3), (3000, 40, 60), (2704, 24, 30),
                                   (1448, 900, 3), (2999, 17, 40)])
@pytest.mark.parametrize("mode", ["noise8", "mixed", "smooth", "flat"])
def test_slots_match_oracle(rows_codec, oracle, W, H, n, mode):
    import torch
    codec = rows_codec
    imgs = codec.synth_frames(mode, SEED, 100, n, W, H)
    imgs_h = imgs.cpu().numpy()
    buf, lead, cap, slot, offs, sizes, o, s = _encode_slots(codec, imgs, W, H, n, first_index=100)
    host = buf.cpu().numpy()
    for f in range(n):
        want = oracle.pack_frame(100 + f, imgs_h[f], W, H)
        assert s[f] == len(want), (W, H, mode, f)
        assert host[lead + o[f]: lead + o[f] + s[f]].tobytes() == want.tobytes(), (W, H, mode, f)
        assert (host[lead + o[f] + s[f]: lead + min(o[f] + slot, cap)] == 0xEE).all(), "wrote past the frame"
    assert (host[:lead] == 0xEE).all()
    back, res = codec.decode_frames(buf, lead, cap, offs, W, H, n)
    codec.sync()
    assert torch.equal(back, imgs)


def test_many_small_odd_frames(rows_codec, oracle):
    """More frames than resident workgroups, explicit frame numbers and elapsed_ns (trap T1: F64 on the wire)."""
    import torch
    codec = rows_codec
    W, H, n = 513, 17, 1300
    imgs = codec.synth_frames("mixed", SEED, 0, n, W, H)
    idx = torch.arange(n, dtype=torch.int64, device=imgs.device) * 3 + 7
    el = torch.arange(n, dtype=torch.int64, device=imgs.device) * 1000000007
    buf, lead, cap, slot, offs, sizes, o, s = _encode_slots(codec, imgs, W, H, n, indices=idx, elapsed_ns=el)
    host = buf.cpu().numpy()
    imgs_h = imgs.cpu().numpy()
    for f in range(n):
        want = oracle.pack_frame(3 * f + 7, imgs_h[f], W, H)
        want[:20] = oracle.pack_frame_header(2, 3 * f + 7, 1000000007 * f)
        assert host[lead + o[f]: lead + o[f] + s[f]].tobytes() == want.tobytes(), f
    back, res = codec.decode_frames(buf, lead, cap, offs, W, H, n)
    codec.sync()
    assert torch.equal(back, imgs)
    rr = codec.parse_results(res)
    assert all(rr[f][:2] == (2, 3 * f + 7) for f in range(n))


@pytest.mark.parametrize("mode", ["noise8", "mixed"])
def test_config4_against_reference_sha(dv, golden, mode):
    """BASELINE configs[3] as bench.py runs it (slots): frames 0 and 3 against the SHA-256 of the reference's own output;
    the last frame's last image row is where the encoder's fetch is moved left."""
    import torch
    manifest, _ = golden
    codec = dv.Codec(0)
    W, H, n = 1921, 1081, 512
    imgs = codec.synth_frames(mode, SEED, 0, n, W, H)
    buf, lead, cap, slot, offs, sizes, o, s = _encode_slots(codec, imgs, W, H, n)
    for e in [e for e in manifest["big"] if e["name"] == "cfg4_1921x1081" and e["mode"] == mode]:
        f = e["frame"]
        got = buf[lead + int(o[f]): lead + int(o[f] + s[f])].cpu().numpy()
        assert len(got) == e["packed_bytes"] and hashlib.sha256(got.tobytes()).hexdigest() == e["packed_sha"], (mode, f)
    back, res = codec.decode_frames(buf, lead, cap, offs, W, H, n)
    codec.sync()
    assert torch.equal(back, imgs)
    codec.close()
