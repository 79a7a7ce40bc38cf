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


@pytest.mark.parametrize("W,H,n", [
    # one wave per segment of a tile row (input_mode 4): 121 pairs = 61 + 60, 63 = one wave, 188 = 63 + 63 + 62,
    # and an even width that takes the form only at an odd base
    (1921, 17, 3), (1921, 17, 700), (1001, 25, 3), (1001, 25, 600), (2999, 17, 2), (2999, 9, 560), (1928, 16, 3), (1928, 9, 600),
    (1921, 1081, 2),
    # pairs dealt linearly, 63 per wave (input_mode 3): segments would leave the waves half empty
    (1081, 33, 3), (1081, 9, 900), (1009, 9, 3), (4093, 9, 2), (4093, 9, 300),
    # fifteen columns in the last pair: fetched where they lie (input_mode 1)
    (1935, 17, 3)])
def test_encoder_fetch_forms_at_every_image_base(rows_codec, oracle, dv, W, H, n):
    """The any-geometry encoder moves its fetches down to dword boundaries when image rows start at odd addresses, in one
    of two lane arrangements (dbde_hip_encode_plan: input_mode 3 / 4).  Images at every base address mod 4 (and a few mod
    128), in a buffer that ENDS with the last image (the batch's last fetch is pinned to end there), few frames (one
    workgroup per chunk) and enough for the persistent kernel: every frame byte for byte against the oracle."""
    import torch
    codec = rows_codec
    rng = np.random.default_rng(W * 7 + H * 3 + n)
    few = min(n, 4)
    imgs_h = codec.synth_frames("mixed", SEED, 11, n, W, H).cpu().numpy()
    imgs_h[-1, -1, -16:] = rng.integers(0, 256, 16, dtype=np.uint8)          # the pinned fetch's bytes matter
    imgs_h[0, 0, :16] = rng.integers(0, 256, 16, dtype=np.uint8)
    check = sorted(set(list(range(few)) + list(range(n - few, n)) + [int(x) for x in rng.integers(0, n, 8)]))
    want = {f: oracle.pack_frame(40 + f, imgs_h[f], W, H) for f in check}
    slot = ((codec.L.dbde_hip_max_frame_bytes(W, H) + 255) // 256) * 256
    modes = set()
    for base in (0, 1, 2, 3, 5, 127):
        canvas = torch.full((256 + base + n * W * H,), 0xEE, dtype=torch.uint8, device="cuda")
        view = canvas[256 + base:].view(n, H, W)
        view.copy_(torch.from_numpy(imgs_h))
        modes.add(dv.encode_plan(W, H, n, image_address=view.data_ptr(), slot_stride=slot)["input_mode"])
        buf, lead, cap = codec.alloc_stream(W, H, n, slot_stride=slot)
        offs, sizes = codec.encode_frames(view, W, H, n, buf, lead, cap, first_index=40, slot_stride=slot)
        codec.sync()
        s = sizes.cpu().numpy()
        for f in check:
            got = buf[lead + f * slot: lead + f * slot + int(s[f])].cpu().numpy()
            assert got.tobytes() == want[f].tobytes(), (W, H, n, base, f)
        assert (canvas[:256 + base] == 0xEE).all()
    expect = {1921: {4}, 1001: {4}, 2999: {4}, 1928: {1, 4}, 1081: {3}, 1009: {3}, 4093: {3}, 1935: {1}}[W]
    assert modes == expect, modes
