"""GPU parity: the HIP path (through the C-ABI) against the oracle and the golden fixtures.

Everything here calls libdbde_hip.so; the oracle is only the checker.  Bit-exact is the bar:
this is byte and integer work, there is no tolerance anywhere in this file.
"""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MODES = {"noise8": 0, "mixed": 1, "flat": 2, "smooth": 3}
SEED = 0xDBDE2016


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def dv():
    import dbde_video_cpp_amd as m
    return m


@pytest.fixture(scope="module")
def codec(dv):
    c = dv.Codec(0)
    assert c.arch.startswith("gfx950")
    yield c
    c.close()


def gpu_encode(codec, imgs, W, H, n, first_index=0, slot_stride=0, misalign=0, **kw):
    """-> list of per-frame packed byte arrays (host)."""
    import torch
    buf, lead, cap = codec.alloc_stream(W, H, n, slot_stride=slot_stride, lead=32 + 16)
    lead += misalign
    buf.fill_(0xEE)
    offs, sizes = codec.encode_frames(imgs, W, H, n, buf, lead, cap, first_index=first_index,
                                      slot_stride=slot_stride, **kw)
    codec.sync()
    host = buf.cpu().numpy()
    o, s = offs.cpu().numpy(), sizes.cpu().numpy()
    frames = [host[lead + o[f]: lead + o[f] + s[f]].copy() for f in range(n)]
    # nothing outside the frames may be touched
    mask = np.ones(len(host), bool)
    for f in range(n):
        mask[lead + o[f]: lead + o[f] + s[f]] = False
    assert (host[mask] == 0xEE).all(), "encoder wrote outside the frames it reported"
    if not slot_stride:
        assert o[0] == 0 and all(o[f + 1] == o[f] + s[f] for f in range(n - 1)), "frames not concatenated"
    else:
        assert all(o[f] == f * slot_stride for f in range(n))
    return frames, (buf, lead, offs, sizes)


def test_synth_matches_oracle(codec, oracle):
    for (W, H) in [(10, 10), (64, 64), (200, 123), (1921, 9)]:
        for mname, mode in MODES.items():
            got = codec.synth_frames(mname, SEED, 5, 3, W, H).cpu().numpy()
            for f in range(3):
                assert (got[f] == oracle.synth_frame(mode, SEED, 5 + f, W, H)).all(), (W, H, mname, f)


def test_golden_frames_host_api(codec, golden):
    """dbde_pack_frame / dbde_unpack_frame semantics on every fixture the reference produced."""
    manifest, arrays = golden
    for e in manifest["frames"]:
        img, want = arrays[e["name"] + ".image"], arrays[e["name"] + ".packed"]
        got = codec.pack_frame(e["index"], img, e["W"], e["H"])
        assert got.tobytes() == want.tobytes(), e["name"]
        body = codec.pack_image(img, e["W"], e["H"])
        assert body.tobytes() == want[20:].tobytes(), e["name"]
        n, fh, back = codec.unpack_frame(want, e["W"], e["H"])
        assert n == len(want) and fh == (2, e["index"], 0) and (back == img).all(), e["name"]
        n, back = codec.unpack_image(want[20:], e["W"], e["H"])
        assert n == len(want) - 20 and (back == img).all(), e["name"]


def test_reference_known_answer_stream(codec, dv, golden):
    """The reference's own KAT (dbde_util_test.cpp:135-213), replayed through the HIP path."""
    from test_oracle_golden import KAT_STREAM
    _, arrays = golden
    img = arrays["kat_8x16.image"]
    stream = np.frombuffer(KAT_STREAM, np.uint8)
    n, vh = dv.unpack_video_header(stream)
    assert n == 28 and vh == (3, 8, 16, 1.0)
    n, fh = dv.unpack_frame_header(stream[28:])
    assert n == 20 and fh == (2, 1, 0)
    n, fh, back = codec.unpack_frame(stream[28:], 16, 8)
    assert n == 100 and fh == (2, 1, 0) and (back == img).all()
    out = np.concatenate([dv.pack_video_header(3, 8, 16, 1.0), codec.pack_frame(1, img, 16, 8)])
    assert out.tobytes() == KAT_STREAM


def test_tile_api(codec, oracle, golden):
    manifest, arrays = golden
    flat = arrays["demo_10x10.image"].reshape(-1).copy()
    for t in manifest["tile_demos"]:
        if t["rm"] == 8 and t["dm"] == 8:
            code, payload, raw = codec.pack_8x8(flat, t["off"], 10)
        else:
            code, payload, raw = codec.pack_8x8_partial(flat, t["off"], 10, t["rm"], t["dm"])
        assert code == t["code"] and payload.tobytes().hex() == t["payload"]
        assert (raw[len(payload):] == 0xEE).all()
        canvas = np.full(100, 0xEE, np.uint8)
        codec.unpack_8x8_partial(code >> 8, code & 0xFF, payload, 10, t["rm"], t["dm"], canvas, 0)
        want = np.full(100, 0xEE, np.uint8)
        oracle.unpack_8x8_partial(code >> 8, code & 0xFF, payload, 10, t["rm"], t["dm"], want, 0)
        assert (canvas == want).all()
    for rng, depth in manifest["depth_bounds"]:
        tile = np.zeros(64, np.uint8)
        tile[5] = rng
        code, payload, _ = codec.pack_8x8(tile, 0, 8)
        assert code >> 8 == depth and len(payload) == 8 * depth
    # random tiles at a stride, every depth
    rng = np.random.default_rng(7)
    for d in range(9):
        img = (rng.integers(0, 256 - (1 << d) + 1) + rng.integers(0, 1 << d, (8, 24))).astype(np.uint8).reshape(-1)
        code, payload, _ = codec.pack_8x8(img, 8, 24)
        wcode, wpayload, _ = oracle.pack_8x8(img, 8, 24)
        assert code == wcode and payload.tobytes() == wpayload.tobytes()
        canvas = np.full(8 * 24 + 8, 0xEE, np.uint8)
        codec.unpack_8x8(code >> 8, code & 0xFF, payload, 24, canvas, 8)
        want = np.full(8 * 24 + 8, 0xEE, np.uint8)
        oracle.unpack_8x8(code >> 8, code & 0xFF, payload, 24, want, 8)
        assert (canvas == want).all()


@pytest.mark.parametrize("W,H,n", [(10, 10, 3), (16, 8, 2), (64, 64, 4), (200, 123, 5), (1024, 40, 3),
                                   (4096, 24, 2), (4104, 16, 2), (8200, 9, 2), (33, 31, 7), (1, 1, 4), (7, 300, 2)])
@pytest.mark.parametrize("mode", ["noise8", "mixed", "smooth", "flat"])
def test_batch_encode_matches_oracle(codec, oracle, W, H, n, mode):
    import torch
    imgs = codec.synth_frames(mode, SEED, 100, n, W, H)
    imgs_h = imgs.cpu().numpy()
    import os
    slot_bytes = ((codec.L.dbde_hip_max_frame_bytes(W, H) + 255) // 256) * 256
    # layouts: concatenated; one slot per frame
    for slot in (0, slot_bytes):
        for misalign in (0, 3):
            frames, (buf, lead, offs, sizes) = gpu_encode(codec, imgs, W, H, n, first_index=100,
                                                          slot_stride=slot, misalign=misalign)
            for f in range(n):
                want = oracle.pack_frame(100 + f, imgs_h[f], W, H)
                assert frames[f].tobytes() == want.tobytes(), (W, H, mode, slot, misalign, f)
            total = int((offs[-1] + sizes[-1]).item())
            back, res = codec.decode_frames(buf, lead, total, offs, W, H, n)
            codec.sync()
            assert torch.equal(back, imgs), (W, H, mode, slot, misalign)
            for f, r in enumerate(codec.parse_results(res)):
                assert r == (2, 100 + f, 0, len(frames[f]))


def test_indices_and_elapsed(codec, oracle, golden):
    import torch
    manifest, _ = golden
    W, H, n = 40, 24, 6
    cases = [h["in"] for h in manifest["headers"]["frame"] if h["in"][0] == 2]
    n = len(cases)
    imgs = codec.synth_frames("mixed", SEED, 0, n, W, H)
    idx = torch.from_numpy(np.array([c[1] for c in cases], np.uint64).view(np.int64)).to(imgs.device)
    el = torch.from_numpy(np.array([c[2] for c in cases], np.uint64).view(np.int64)).to(imgs.device)
    frames, (buf, lead, offs, sizes) = gpu_encode(codec, imgs, W, H, n, indices=idx, elapsed_ns=el)
    wires = {tuple(h["in"]): h for h in manifest["headers"]["frame"]}
    for f, c in enumerate(cases):
        assert frames[f][:20].tobytes().hex() == wires[tuple(c)]["wire"]
    total = int((offs[-1] + sizes[-1]).item())
    _, res = codec.decode_frames(buf, lead, total, offs, W, H, n)
    codec.sync()
    for f, r in enumerate(codec.parse_results(res)):
        assert list(r[:3]) == wires[tuple(cases[f])]["out"]


def test_malformed_frames(codec, oracle, golden):
    import torch
    manifest, arrays = golden
    base = arrays["malformed_base.packed"]
    for m in manifest["malformed"]:
        bad = base.copy()
        bad[m["pos"]] = (int(bad[m["pos"]]) + m["delta"]) % 256
        n, fh, img = codec.unpack_frame(bad, 10, 10, fill=0xEE)
        assert n == m["advance"] and fh[0] == m["u64s"] and fh[1] == m["index"], m["label"]
        assert bool((img == 0xEE).all()) == m["image_untouched"] and sha(img) == m["image_sha"], m["label"]
        n_img, _ = codec.unpack_image(bad[20:], 10, 10)
        assert n_img == m["unpack_image_ret"], m["label"]
    # depth byte > 8: rejected (the documented deviation), image untouched
    bad = base.copy()
    bad[24] = 9
    bad[36] = (int(bad[36]) + 5) % 256
    n, fh, img = codec.unpack_frame(bad, 10, 10, fill=0xEE)
    assert n == 20 and fh[0] == 0xFFFFFFFF and (img == 0xEE).all()
    # batch: one bad frame among good ones leaves only its own image untouched
    W, H, n = 72, 40, 5
    imgs = codec.synth_frames("mixed", SEED, 0, n, W, H)
    frames, (buf, lead, offs, sizes) = gpu_encode(codec, imgs, W, H, n)
    o = offs.cpu().numpy()
    T = ((W + 7) // 8) * ((H + 7) // 8)
    buf[lead + int(o[2]) + 28 + 2 * T] += 1          # corrupt n64 of frame 2
    total = int((offs[-1] + sizes[-1]).item())
    canvas = torch.full((n, H, W), 0xEE, dtype=torch.uint8, device=imgs.device)
    back, res = codec.decode_frames(buf, lead, total, offs, W, H, n, images=canvas)
    codec.sync()
    r = codec.parse_results(res)
    for f in range(n):
        if f == 2:
            assert r[f][0] == 0xFFFFFFFF and r[f][3] == 20 and bool((back[f] == 0xEE).all())
        else:
            assert r[f][0] == 2 and torch.equal(back[f], imgs[f])
    # byte-wise wrapping add on decode: min + value > 255 in a stream that passes validation
    img = np.zeros((8, 8), np.uint8)
    img[0, 1] = 255
    packed = oracle.pack_frame(0, img, 8, 8)
    packed[20 + 4 + 1 + 4] = 200
    n1, f1, i1 = codec.unpack_frame(packed, 8, 8)
    n2, f2, i2 = oracle.unpack_frame(packed, 8, 8)
    assert n1 == n2 and f1 == f2 and (i1 == i2).all()


def test_index_stream(codec):
    W, H, n = 200, 123, 9
    imgs = codec.synth_frames("mixed", SEED, 0, n, W, H)
    frames, (buf, lead, offs, sizes) = gpu_encode(codec, imgs, W, H, n)
    total = int((offs[-1] + sizes[-1]).item())
    found, cnt = codec.index_stream(buf, lead, total, W, H, 100)
    assert cnt == n and (found.cpu().numpy() == offs.cpu().numpy()).all()
    found, cnt = codec.index_stream(buf, lead, total - 1, W, H, 100)   # truncated last frame
    assert cnt == n - 1
    # the asynchronous form: offsets and the device-side count, usable by a decode enqueued behind it
    import torch
    offs2 = torch.full((100,), -1, dtype=torch.int64, device=buf.device)
    offs2, count = codec.index_stream_async(buf, lead, total, W, H, 100, offs2)
    back, res = codec.decode_frames(buf, lead, total, offs2, W, H, n)
    codec.sync()
    assert int(count.item()) == n and torch.equal(offs2[:n], offs) and (offs2[n:] == -1).all()
    assert torch.equal(back, imgs)


def test_index_stream_speculative_walk(codec, oracle):
    """Long streams are walked in up to 16 segments at once, each from a position that LOOKS like a frame start, and
    stitched only where the exact chain arrives (scan_spec_kernel).  Offsets must equal the encoder's for honest
    streams, for tiny frames (many per segment), for frames larger than a segment, for a truncated tail, for a
    max_frames cut -- and for a stream salted with byte patterns that look exactly like frame starts."""
    import torch
    for (W, H, n, mode) in [(200, 123, 300, "mixed"), (10, 10, 5000, "smooth"), (2048, 2048, 40, "mixed"),
                            (1024, 768, 97, "noise8"), (64, 64, 1200, "flat")]:
        imgs = codec.synth_frames(mode, SEED, 0, n, W, H)
        frames, (buf, lead, offs, sizes) = gpu_encode(codec, imgs, W, H, n)
        total = int((offs[-1] + sizes[-1]).item())
        found, cnt = codec.index_stream(buf, lead, total, W, H, n + 10)
        assert cnt == n and torch.equal(found, offs), (W, H, n, mode)
        found, cnt = codec.index_stream(buf, lead, total - 5, W, H, n + 10)      # truncated last frame
        assert cnt == n - 1 and torch.equal(found, offs[:n - 1])
        found, cnt = codec.index_stream(buf, lead, total, W, H, n // 2)          # the caller's limit
        assert cnt == n // 2 and torch.equal(found, offs[:n // 2])
    # decoys: all-noise frames whose payload is overwritten, at many positions, with the three fields of a frame
    # start (2 | T | T).  The chain never lands on them, so they must change nothing.
    W, H, n = 256, 256, 400
    T = (W // 8) * (H // 8)
    imgs = codec.synth_frames("noise8", SEED, 0, n, W, H)
    frames, (buf, lead, offs, sizes) = gpu_encode(codec, imgs, W, H, n)
    total = int((offs[-1] + sizes[-1]).item())
    host = buf.cpu().numpy().copy()
    o, s = offs.cpu().numpy(), sizes.cpu().numpy()
    rng = np.random.default_rng(3)
    for f in range(n):
        pay = lead + int(o[f]) + 32 + 2 * T              # payload of frame f: 64*T bytes of noise
        for _ in range(3):
            c = pay + 8 * int(rng.integers(0, (int(s[f]) - 32 - 2 * T - 32 - 2 * T) // 8 - 8))
            host[c:c + 4] = np.frombuffer(np.uint32(2).tobytes(), np.uint8)
            host[c + 20:c + 24] = np.frombuffer(np.uint32(T).tobytes(), np.uint8)
            host[c + 24 + T:c + 28 + T] = np.frombuffer(np.uint32(T).tobytes(), np.uint8)
    salted = torch.from_numpy(host).cuda()
    found, cnt = codec.index_stream(salted, lead, total, W, H, n + 10)
    assert cnt == n and torch.equal(found, offs)


def test_config3_at_its_stated_size(codec, oracle, golden):
    """BASELINE.json configs[2] as it is stated: a 1000-frame 2048x2048 stream, mixed depths 0-8, one concatenated
    body, read the way the reference's walker reads a file (dbde_util.cpp:408-421: sizes are only in-band) -- the device
    scanner finds the frame starts, one decode launch takes its offsets from there.  Offsets == the encoder's, count ==
    1000, frames 0 / 3 against the REAL reference's SHA-256, three more frames byte for byte against the oracle, round
    trip equal -- what bench.py asserts on this config, as a test."""
    import torch
    manifest, _ = golden
    W, H, n = 2048, 2048, 1000
    imgs = codec.synth_frames("mixed", SEED, 0, n, W, H)
    buf, lead, cap = codec.alloc_stream(W, H, n)
    offs, sizes = codec.encode_frames(imgs, W, H, n, buf, lead, cap, first_index=0)
    codec.sync()
    o, s = offs.cpu().numpy(), sizes.cpu().numpy()
    assert o[0] == 0 and (o[1:] == (o + s)[:-1]).all(), "frames not concatenated"
    total = int(o[-1] + s[-1])
    # a reader's view: the bytes and their total length only
    found = torch.empty(n + 8, dtype=torch.int64, device=imgs.device)
    count = torch.zeros(1, dtype=torch.int32, device=imgs.device)
    codec.index_stream_async(buf, lead, total, W, H, n + 8, found, count)
    back, res = codec.decode_frames(buf, lead, total, found, W, H, n)
    codec.sync()
    assert int(count.item()) == n and torch.equal(found[:n], offs) and (found[n:] == -1).all()
    assert torch.equal(back, imgs)
    for f, (u64s, index, el, consumed) in enumerate(codec.parse_results(res)):
        assert (u64s, index, el, consumed) == (2, f, 0, int(s[f])), f
    checked = 0
    for e in [e for e in manifest["big"] if e["name"] == "cfg3_2048x2048" and e["mode"] == "mixed"]:
        f = e["frame"]
        got = buf[lead + int(o[f]):lead + int(o[f] + s[f])].cpu().numpy()
        assert len(got) == e["packed_bytes"] and sha(got) == e["packed_sha"], f
        checked += 1
    assert checked == 2
    for f in (1, 499, n - 1):
        want = oracle.pack_frame(f, oracle.synth_frame(MODES["mixed"], SEED, f, W, H), W, H)
        got = buf[lead + int(o[f]):lead + int(o[f] + s[f])].cpu().numpy()
        assert got.tobytes() == want.tobytes(), f


def test_scan_ahead_reader_pipeline(codec):
    """dbde_hip_scan_ahead / dbde_hip_scan_join: an un-indexed stream read a batch at a time, the walk of the next
    batch on the context's second stream beside the decode of the current one; uneven last batch, cursor carried
    across calls, a truncated tail ends the walk."""
    import torch
    W, H, n, per = 200, 123, 23, 5
    imgs = codec.synth_frames("mixed", SEED, 0, n, W, H)
    frames, (buf, lead, offs, sizes) = gpu_encode(codec, imgs, W, H, n)
    total = int((offs[-1] + sizes[-1]).item())
    dev = imgs.device
    for stream_bytes, want in ((total, n), (total - 3, n - 1)):
        cursor = torch.zeros(1, dtype=torch.int64, device=dev)
        nb = (n + per - 1) // per
        found = torch.full((nb * per,), -1, dtype=torch.int64, device=dev)
        counts = torch.zeros(nb, dtype=torch.int32, device=dev)
        out = torch.full_like(imgs, 0xEE)
        codec.scan_ahead(buf, lead, stream_bytes, W, H, per, cursor, found[0:per], counts[0:1])
        for b in range(nb):
            codec.scan_join()
            if b + 1 < nb:
                codec.scan_ahead(buf, lead, stream_bytes, W, H, per, cursor, found[(b + 1) * per:(b + 2) * per],
                                 counts[b + 1:b + 2])
            k = min(per, n - b * per)
            # a reader that expects k frames decodes k; entries the walk did not reach are rejected (offset -1 is
            # outside the stream), never decoded from garbage
            codec.decode_frames(buf, lead, stream_bytes, found[b * per:b * per + k], W, H, k, images=out[b * per:b * per + k])
        codec.sync()
        assert int(counts.sum().item()) == want
        assert torch.equal(found[:want], offs[:want]) and (found[want:] == -1).all()
        assert torch.equal(out[:want], imgs[:want]) and (out[want:] == 0xEE).all()
        assert int(cursor.item()) == int((offs[want - 1] + sizes[want - 1]).item())


@pytest.mark.parametrize("W,H,n", [(64, 64, 4), (1024, 768, 3), (640, 480, 300)])
def test_wild_frame_offsets_are_rejected(codec, W, H, n):
    """Offsets are the caller's (a stale -1 from a reused array, a cursor gone wrong): any offset outside
    [0, stream_bytes) must be rejected by every index form -- self-indexing decode (64x64), the split index kernel
    (1024x768 x 3) and one index workgroup per frame (640x480 x 300) -- with no wrap-around in `off + need`: the
    frame reports u64s = 0xFFFFFFFF, consumed = 20, index = elapsed = 0 (nothing was read), its image is untouched,
    and the frames around it decode as usual.  The stream is its own allocation (it starts at byte 0 of a block)."""
    import torch
    imgs = codec.synth_frames("mixed", SEED, 9, n, W, H)
    frames, (buf, lead, offs, sizes) = gpu_encode(codec, imgs, W, H, n, first_index=9)
    total = int((offs[-1] + sizes[-1]).item())
    stream = buf[lead:lead + total].clone()
    need = 32 + 2 * ((W + 7) // 8) * ((H + 7) // 8)
    wild = [-1, -need, -need - 8, -(1 << 63), (1 << 63) - 1, total, total - 10, -total, -(total - 20)]
    for i, v in enumerate(wild):
        bad = sorted({i % n, n - 1})
        o = offs.clone()
        for k in bad:
            o[k] = v
        canvas = torch.full_like(imgs, 0xEE)
        back, res = codec.decode_frames(stream, 0, total, o, W, H, n, images=canvas)
        codec.sync()
        rr = codec.parse_results(res)
        for f in range(n):
            if f in bad:
                assert rr[f] == (0xFFFFFFFF, 0, 0, 20), (v, f, rr[f])
                assert (back[f] == 0xEE).all(), (v, f)
            else:
                assert rr[f] == (2, 9 + f, 0, len(frames[f])), (v, f, rr[f])
                assert torch.equal(back[f], imgs[f]), (v, f)


def test_scanners_mark_unvisited_offsets(codec):
    """dbde_hip.h: entries of the offsets array past the frame count are set to 2^64 - 1 by both walkers (serial and
    speculative), so a decode bounded by max_frames reports them as failed instead of decoding stale offsets."""
    import torch
    for W, H, n in ((200, 123, 7), (512, 512, 40)):       # short stream: serial hop; long one: speculative segments
        imgs = codec.synth_frames("mixed", SEED, 0, n, W, H)
        frames, (buf, lead, offs, sizes) = gpu_encode(codec, imgs, W, H, n)
        total = int((offs[-1] + sizes[-1]).item())
        found = torch.arange(n + 9, dtype=torch.int64, device=imgs.device) * 8      # stale but plausible entries
        count = torch.zeros(1, dtype=torch.int32, device=imgs.device)
        codec.index_stream_async(buf, lead, total, W, H, n + 9, found, count)
        canvas = torch.full((n + 9, H, W), 0xEE, dtype=torch.uint8, device=imgs.device)
        back, res = codec.decode_frames(buf, lead, total, found, W, H, n + 9, images=canvas)
        codec.sync()
        assert int(count.item()) == n and torch.equal(found[:n], offs) and (found[n:] == -1).all()
        rr = codec.parse_results(res)
        assert all(r == (0xFFFFFFFF, 0, 0, 20) for r in rr[n:]) and (back[n:] == 0xEE).all()
        assert torch.equal(back[:n], imgs)


@pytest.mark.parametrize("W,H,n,mode", [(200, 123, 3, "mixed"), (8, 8, 1, "noise8"), (9, 9, 2, "mixed"),
                                        (4096, 24, 1, "noise8"), (1, 1, 1, "flat"), (24, 8, 1, "noise8"),
                                        (72, 72, 3, "mixed"), (64, 64, 2, "smooth")])   # (the small-frame decoder: 16-byte pieces from the payload's first byte on)
def test_decode_reads_nothing_past_stream_bytes(codec, W, H, n, mode):
    """dbde_hip.h: stream_bytes is the READABLE extent.  The stream is placed so that its end falls on
    every residue mod 16 (the payload DMA moves whole 16-byte slots from an aligned-down source; the slot
    that straddles the end must be fetched byte by byte), with different garbage behind it each time:
    the decode must be identical, and the index/scan must accept exactly these bytes."""
    import torch
    imgs = codec.synth_frames(mode, SEED, 3, n, W, H)
    frames, (buf, lead, offs, sizes) = gpu_encode(codec, imgs, W, H, n, first_index=3)
    total = int((offs[-1] + sizes[-1]).item())
    for pad in range(16):
        for junk in (0xA5, 0x00, 0xFF):
            t = torch.full((pad + total + 48,), junk, dtype=torch.uint8, device=imgs.device)
            t[pad:pad + total] = buf[lead:lead + total]
            back, res = codec.decode_frames(t, pad, total, offs, W, H, n)
            codec.sync()
            assert torch.equal(back, imgs), (pad, junk)
            assert all(r == (2, 3 + f, 0, len(frames[f])) for f, r in enumerate(codec.parse_results(res)))
        # one byte short: the last frame no longer fits and is rejected, the others still decode
        canvas = torch.full_like(imgs, 0xEE)
        back, res = codec.decode_frames(t, pad, total - 1, offs, W, H, n, images=canvas)
        codec.sync()
        rr = codec.parse_results(res)
        assert rr[-1][0] == 0xFFFFFFFF and (back[-1] == 0xEE).all() and torch.equal(back[:-1], imgs[:-1])


@pytest.mark.parametrize("W,H", [(17, 9), (23, 41), (31, 8), (33, 24), (63, 17), (65, 65), (100, 30), (121, 16),
                                 (250, 33), (257, 8), (1001, 25), (1921, 17), (1922, 9), (1923, 16), (4093, 9)])
def test_decode_staged_rows_off_alignment(codec, oracle, W, H):
    """Widths whose image rows are not 8-byte aligned, content whose tiles are all of depth 0 or 8 (the workgroups
    that stage their pixels in LDS, re-aligned in registers, and write whole cache lines) and bit-packed content
    (the workgroups that store tile by tile), at every image base alignment mod 4 and a few mod 128; the bytes
    around the images must stay untouched."""
    import torch
    n = 3
    rng = np.random.default_rng(W * 131 + H)
    noise = rng.integers(0, 256, (n, H, W), dtype=np.uint8)
    # tiles of depth 8 next to flat ones: rows of tiles alternate
    flat8 = noise.copy()
    for ty in range(0, (H + 7) // 8, 2):
        flat8[:, 8 * ty: 8 * ty + 8, :] = 77
    mixed = codec.synth_frames("mixed", SEED, 7, n, W, H).cpu().numpy()
    for name, imgs_h in (("noise", noise), ("flat8", flat8), ("mixed", mixed)):
        packed = [oracle.pack_frame(5 + f, imgs_h[f], W, H) for f in range(n)]
        blob = np.concatenate(packed)
        offs = torch.tensor(np.cumsum([0] + [len(x) for x in packed[:-1]]), dtype=torch.int64, device="cuda")
        stream = torch.zeros(32 + len(blob) + 64, dtype=torch.uint8, device="cuda")
        stream[32: 32 + len(blob)] = torch.from_numpy(blob).cuda()
        for base in (0, 1, 2, 3, 5, 64, 127):
            canvas = torch.full((256 + base + n * W * H + 256,), 0xEE, dtype=torch.uint8, device="cuda")
            view = canvas[256 + base: 256 + base + n * W * H].view(n, H, W)
            back, res = codec.decode_frames(stream, 32, len(blob), offs, W, H, n, images=view)
            codec.sync()
            got = canvas.cpu().numpy()
            assert (got[:256 + base] == 0xEE).all() and (got[256 + base + n * W * H:] == 0xEE).all(), (W, H, name, base)
            assert np.array_equal(got[256 + base: 256 + base + n * W * H].reshape(n, H, W), imgs_h), (W, H, name, base)
            for f, r in enumerate(codec.parse_results(res)):
                assert r == (2, 5 + f, 0, len(packed[f]))


@pytest.mark.parametrize("d", range(9))
def test_uniform_depth_content(codec, oracle, d):
    """Every tile of one depth (synth modes 4..12): the regular lane strides the LDS swizzles exist for, and the only
    content that reaches the decoder's swizzled general unpack (chunks whose word count is 4 or 7 per tile).  Encoder
    bytes against the oracle, depth array as promised, decode back, misaligned stream included."""
    import torch
    for (W, H, n) in [(4096, 16, 3), (1024, 40, 2), (1921, 17, 2), (200, 123, 2)]:
        imgs = codec.synth_frames(4 + d, SEED, 11, n, W, H)
        imgs_h = imgs.cpu().numpy()
        T = ((W + 7) // 8) * ((H + 7) // 8)
        for misalign in (0, 3):
            frames, (buf, lead, offs, sizes) = gpu_encode(codec, imgs, W, H, n, first_index=11, misalign=misalign)
            for f in range(n):
                want = oracle.pack_frame(11 + f, imgs_h[f], W, H)
                assert frames[f].tobytes() == want.tobytes(), (d, W, H, misalign, f)
                if W % 8 == 0 and H % 8 == 0:   # (partial tiles may miss the two pixels that pin the range)
                    assert (frames[f][24: 24 + T] == d).all(), (d, W, H, "depth array")
            total = int((offs[-1] + sizes[-1]).item())
            back, res = codec.decode_frames(buf, lead, total, offs, W, H, n)
            codec.sync()
            assert torch.equal(back, imgs), (d, W, H, misalign)


@pytest.mark.parametrize("name", ["cfg2_4096x3072", "cfg3_2048x2048", "cfg4_1921x1081", "shape_1080x1920", "shape_1366x768",
                                  "shape_1440x900", "shape_720x1280", "shape_72x72", "shape_96x96",
                                  "shape_160x120", "shape_176x144", "shape_320x240", "shape_64x64", "shape_128x128", "cfg2_rank_frames"])
def test_baseline_configs_full_size(codec, golden, name):
    """BASELINE.json configs 2-4 and the shapes whose kernel forms differ from theirs (portrait HD, 1366x768, 16-byte
    rows that are not whole cache lines, frames of 81 / 144 tiles): packed-frame hashes equal the REAL reference's
    (fixtures)."""
    import torch
    manifest, _ = golden
    for e in [e for e in manifest["big"] if e["name"] == name]:
        W, H = e["W"], e["H"]
        imgs = codec.synth_frames(e["mode"], manifest["seed"], e["frame"], 1, W, H)
        assert sha(imgs.cpu().numpy()) == e["image_sha"]
        frames, (buf, lead, offs, sizes) = gpu_encode(codec, imgs, W, H, 1, first_index=e["frame"])
        assert len(frames[0]) == e["packed_bytes"] and sha(frames[0]) == e["packed_sha"], e
        back, res = codec.decode_frames(buf, lead, len(frames[0]), offs, W, H, 1)
        codec.sync()
        assert torch.equal(back, imgs)


def test_large_batch_properties(codec, oracle):
    """Size-independent checks at bench scale: 48 distinct 4096x3072 frames (604 MB raw,
    beyond the 256 MiB Infinity Cache): concatenation, exact sizes, round trip, and three
    frames spot-checked byte-for-byte against the oracle."""
    import torch
    W, H, n = 4096, 3072, 48
    T = (W // 8) * (H // 8)
    for mode in ("mixed", "noise8"):
        imgs = codec.synth_frames(mode, SEED, 0, n, W, H)
        buf, lead, cap = codec.alloc_stream(W, H, n)
        offs, sizes = codec.encode_frames(imgs, W, H, n, buf, lead, cap)
        codec.sync()
        o, s = offs.cpu().numpy(), sizes.cpu().numpy()
        assert o[0] == 0 and (o[1:] == np.cumsum(s)[:-1]).all()
        host = buf.cpu().numpy()
        for f in range(n):
            fr = host[lead + o[f]: lead + o[f] + s[f]]
            n64 = int(fr[28 + 2 * T: 32 + 2 * T].view("<u4")[0])
            assert s[f] == 32 + 2 * T + 8 * n64 and n64 == int(fr[24:24 + T].astype(np.int64).sum())
            if mode == "noise8":
                assert n64 == 8 * T
        for f in (0, 17, n - 1):
            want = oracle.pack_frame(f, oracle.synth_frame(MODES[mode], SEED, f, W, H), W, H)
            assert host[lead + o[f]: lead + o[f] + s[f]].tobytes() == want.tobytes(), (mode, f)
        back, res = codec.decode_frames(buf, lead, int(o[-1] + s[-1]), offs, W, H, n)
        codec.sync()
        assert torch.equal(back, imgs)
        assert all(r == (2, f, 0, int(s[f])) for f, r in enumerate(codec.parse_results(res)))
        del imgs, buf, back


def test_slots_many_frames(codec, oracle):
    """Slot layout with more frames than resident workgroups (every persistent encoder workgroup walks
    many frames; the decode index takes its one-workgroup-per-frame form from 256 frames on), against
    the oracle and round trip."""
    import torch
    for (W, H, n, mode) in [(1024, 768, 1300, "mixed"), (333, 200, 700, "smooth"), (2048, 2048, 520, "noise8")]:
        imgs = codec.synth_frames(mode, SEED, 7, n, W, H)
        slot = ((codec.L.dbde_hip_max_frame_bytes(W, H) + 255) // 256) * 256
        buf, lead, cap = codec.alloc_stream(W, H, n, slot_stride=slot)
        offs, sizes = codec.encode_frames(imgs, W, H, n, buf, lead, cap, first_index=7, slot_stride=slot)
        codec.sync()
        o, s = offs.cpu().numpy(), sizes.cpu().numpy()
        assert (o == np.arange(n) * slot).all()
        host = buf.cpu().numpy()
        for f in (0, 1, n // 2, n - 2, n - 1):
            want = oracle.pack_frame(7 + f, oracle.synth_frame(MODES[mode], SEED, 7 + f, W, H), W, H)
            assert host[lead + o[f]: lead + o[f] + s[f]].tobytes() == want.tobytes(), (W, H, mode, f)
        back, res = codec.decode_frames(buf, lead, cap, offs, W, H, n)
        codec.sync()
        assert torch.equal(back, imgs)
        assert all(r == (2, 7 + f, 0, int(s[f])) for f, r in enumerate(codec.parse_results(res)))
        del imgs, buf, back


@pytest.mark.parametrize("W,H,n", [(1, 2000, 2), (3000, 1, 3), (16384, 8, 2), (12000, 17, 2), (8, 8, 1), (9, 9, 65),
                                   (4096, 3072, 1)])
def test_extreme_shapes(codec, oracle, W, H, n):
    """Column / row images, one-tile frames, more frames than workgroups' worth of tiny chunks."""
    import torch
    imgs = codec.synth_frames("mixed", SEED, 7, n, W, H)
    imgs_h = imgs.cpu().numpy()
    for slot in (0, ((codec.L.dbde_hip_max_frame_bytes(W, H) + 255) // 256) * 256):
        frames, (buf, lead, offs, sizes) = gpu_encode(codec, imgs, W, H, n, first_index=7, slot_stride=slot)
        for f in (0, n // 2, n - 1):
            assert frames[f].tobytes() == oracle.pack_frame(7 + f, imgs_h[f], W, H).tobytes(), (W, H, slot, f)
        total = int((offs[-1] + sizes[-1]).item())
        back, res = codec.decode_frames(buf, lead, total, offs, W, H, n)
        codec.sync()
        assert torch.equal(back, imgs), (W, H, slot)


def test_ticket_mode_fallback(codec, oracle, dv):
    """The encoder's fallback chunk assignment (tickets instead of static strides), forced."""
    import os
    import torch
    W, H, n = 1024, 512, 24
    os.environ["DBDE_HIP_EXPERIMENT"] = "1"
    try:
        c2 = dv.Codec(0)
    finally:
        os.environ.pop("DBDE_HIP_EXPERIMENT", None)
    try:
        imgs = c2.synth_frames("mixed", SEED, 0, n, W, H)
        frames, (buf, lead, offs, sizes) = gpu_encode(c2, imgs, W, H, n)
        imgs_h = imgs.cpu().numpy()
        for f in range(n):
            assert frames[f].tobytes() == oracle.pack_frame(f, imgs_h[f], W, H).tobytes()
    finally:
        c2.close()


@pytest.mark.parametrize("flags", [512, 513])
def test_scanner_role_is_taken_over_when_workgroup_0_is_late(codec, oracle, dv, flags):
    """Workgroup 0 of a persistent launch checks the arrivals, settles the mode and scans.  If it is not running (here: it
    turns up 60 us late, $DBDE_HIP_EXPERIMENT bit 9), the encoders give up waiting for the mode after 20 us, the launch
    falls back to tickets and the workgroup that decided so takes the scanner's role; the late workgroup 0 finds the role
    taken and encodes on tickets like the others.  Bit-exact, no time-out."""
    import os
    import torch
    os.environ["DBDE_HIP_EXPERIMENT"] = str(flags)
    try:
        c2 = dv.Codec(0)
    finally:
        os.environ.pop("DBDE_HIP_EXPERIMENT", None)
    try:
        for (W, H, n, concat) in ((1920, 1080, 70, False), (1921, 1081, 64, True)):
            imgs = codec.synth_frames("mixed", SEED, 5, n, W, H)
            slot = 0 if concat else ((dv.max_frame_bytes(W, H) + 255) // 256) * 256
            for rep in range(3):
                frames, (buf, lead, offs, sizes) = gpu_encode(c2, imgs, W, H, n, first_index=5, slot_stride=slot)   # (sync inside: a time-out would raise)
            imgs_h = imgs.cpu().numpy()
            for f in (0, 1, n // 2, n - 1):
                assert frames[f].tobytes() == oracle.pack_frame(5 + f, imgs_h[f], W, H).tobytes(), (W, f)
            cap = (n - 1) * slot + dv.max_frame_bytes(W, H) if slot else int((offs[-1] + sizes[-1]).item())
            back, _ = c2.decode_frames(buf, lead, cap, offs, W, H, n)
            c2.sync()
            assert torch.equal(back, imgs)
    finally:
        c2.close()


@pytest.mark.parametrize("W,H,n,concat", [(1920, 1080, 96, False), (1921, 1081, 80, True), (2048, 2048, 40, True)])
def test_persistent_encoder_on_tickets(codec, oracle, dv, W, H, n, concat):
    """The PERSISTENT encoder's fallback (every chunk id a ticket: what a launch falls back to when not all of its
    workgroups are seen running), forced, on launches of several rounds: bit-exact, and not pathologically slow -- round 4
    found a form that was correct but chained the workgroups one behind the other through the in-order prefix (two tickets
    drawn at once are neighbouring chunk ids), ten times the static mode's time, which no correctness test noticed."""
    import os
    import torch
    os.environ["DBDE_HIP_EXPERIMENT"] = "1"
    try:
        c2 = dv.Codec(0)
    finally:
        os.environ.pop("DBDE_HIP_EXPERIMENT", None)
    try:
        assert dv.encode_plan(W, H, n)["kernel"] == 0          # the persistent encoder (more chunks than resident workgroups)
        imgs = codec.synth_frames("mixed", SEED, 11, n, W, H)
        slot = 0 if concat else ((dv.max_frame_bytes(W, H) + 255) // 256) * 256
        times = {}
        for name, c in (("static", codec), ("tickets", c2)):
            frames, (buf, lead, offs, sizes) = gpu_encode(c, imgs, W, H, n, first_index=11, slot_stride=slot)
            imgs_h = imgs.cpu().numpy()
            for f in (0, 1, n // 2, n - 1):
                assert frames[f].tobytes() == oracle.pack_frame(11 + f, imgs_h[f], W, H).tobytes(), (name, f)
            cap = (n - 1) * slot + dv.max_frame_bytes(W, H) if slot else int((offs[-1] + sizes[-1]).item())
            back, _ = c.decode_frames(buf, lead, cap, offs, W, H, n)
            c.sync()
            assert torch.equal(back, imgs), name
            c.timing(True)
            c.timing_read(reset=True)
            for _ in range(5):
                c.encode_frames(imgs, W, H, n, buf, lead, cap if slot else n * dv.max_frame_bytes(W, H), first_index=11, offsets=offs, nbytes=sizes,
                                slot_stride=slot)
            c.sync()
            times[name] = c.timing_read(reset=True)["encode"][0] / 5
            c.timing(False)
        assert times["tickets"] < 4.0 * times["static"], times
    finally:
        c2.close()


def test_two_persistent_encodes_at_once(oracle, dv):
    """Two contexts on two streams, each launching persistent encodes sized for the whole device, at the same time: the
    second launch's workgroups are not all resident while the first runs, so it cannot prove co-residency -- its workgroups
    give up waiting for the mode after 20 us, fall back to tickets, and if its workgroup 0 is not running yet one of them
    takes the scanner's role.  Forward progress must not depend on who got the device first; both streams bit-exact."""
    import torch
    W, H, n = 2048, 1024, 160                      # 32 chunks per frame: 5120 chunks, ten rounds of 511 workgroups
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    c1, c2 = dv.Codec(0, stream=s1), dv.Codec(0, stream=s2)
    try:
        assert dv.encode_plan(W, H, n)["kernel"] == 0
        imgs = [c.synth_frames("mixed", SEED, 100 * k, n, W, H) for k, c in enumerate((c1, c2))]
        bufs = [c.alloc_stream(W, H, n) for c in (c1, c2)]
        c1.sync(); c2.sync()
        outs = [None, None]
        for it in range(12):
            for k, c in enumerate((c1, c2) if it % 2 == 0 else (c2, c1)):      # who launches first alternates
                kk = k if it % 2 == 0 else 1 - k
                buf, lead, cap = bufs[kk]
                outs[kk] = c.encode_frames(imgs[kk], W, H, n, buf, lead, cap, first_index=100 * kk)
            c1.sync(); c2.sync()                       # dbde_hip_sync: a look-back time-out would surface here
        for kk, c in enumerate((c1, c2)):
            buf, lead, cap = bufs[kk]
            offs, sizes = outs[kk]
            o, s_ = offs.cpu().numpy(), sizes.cpu().numpy()
            assert o[0] == 0 and (o[1:] == (o + s_)[:-1]).all()
            ih = imgs[kk].cpu().numpy()
            for f in (0, 1, n // 2, n - 1):
                want = oracle.pack_frame(100 * kk + f, ih[f], W, H)
                got = buf[lead + int(o[f]):lead + int(o[f] + s_[f])].cpu().numpy()
                assert got.tobytes() == want.tobytes(), (kk, f)
            back, _ = c.decode_frames(buf, lead, int(o[-1] + s_[-1]), offs, W, H, n)
            c.sync()
            assert torch.equal(back, imgs[kk]), kk
    finally:
        c1.close(); c2.close()


def test_argument_and_capacity_errors(codec, dv):
    """C-ABI error behaviour: bad geometry, short buffers and null pointers are refused before
    any launch; an empty batch is a no-op."""
    import torch
    W, H, n = 64, 64, 4
    imgs = codec.synth_frames("mixed", SEED, 0, n, W, H)
    buf, lead, cap = codec.alloc_stream(W, H, n)
    L, h = codec.L, codec.h
    offs = torch.zeros(n, dtype=torch.int64, device=imgs.device)
    # empty batch: OK, nothing written
    buf.fill_(0xEE)
    assert L.dbde_hip_encode_frames(h, imgs.data_ptr(), W, H, 0, 0, None, None, buf.data_ptr() + lead, cap, 0, None, None) == dv.OK
    assert L.dbde_hip_decode_frames(h, buf.data_ptr() + lead, cap, offs.data_ptr(), W, H, 0, imgs.data_ptr(), None) == dv.OK
    codec.sync()
    assert bool((buf == 0xEE).all())
    # geometry
    for (w_, h_) in ((0, 8), (8, 0), (-1, 8)):
        assert L.dbde_hip_encode_frames(h, imgs.data_ptr(), w_, h_, 1, 0, None, None, buf.data_ptr() + lead, cap, 0, None, None) == dv.ERR_ARG
        assert L.dbde_hip_max_frame_bytes(w_, h_) == 0
    assert L.dbde_hip_encode_frames(h, imgs.data_ptr(), W, H, -1, 0, None, None, buf.data_ptr() + lead, cap, 0, None, None) == dv.ERR_ARG
    # null pointers
    assert L.dbde_hip_encode_frames(h, None, W, H, n, 0, None, None, buf.data_ptr() + lead, cap, 0, None, None) == dv.ERR_ARG
    assert L.dbde_hip_encode_frames(h, imgs.data_ptr(), W, H, n, 0, None, None, None, cap, 0, None, None) == dv.ERR_ARG
    assert L.dbde_hip_decode_frames(h, buf.data_ptr(), cap, None, W, H, n, imgs.data_ptr(), None) == dv.ERR_ARG
    # capacity below the worst case (concatenated and slots), slot stride below one frame
    worst = dv.max_frame_bytes(W, H)
    assert L.dbde_hip_encode_frames(h, imgs.data_ptr(), W, H, n, 0, None, None, buf.data_ptr() + lead, n * worst - 1, 0, None, None) == dv.ERR_CAPACITY
    assert L.dbde_hip_encode_frames(h, imgs.data_ptr(), W, H, n, 0, None, None, buf.data_ptr() + lead, cap, worst - 1, None, None) == dv.ERR_ARG
    assert L.dbde_hip_encode_frames(h, imgs.data_ptr(), W, H, n, 0, None, None, buf.data_ptr() + lead, 3 * worst, worst, None, None) == dv.ERR_CAPACITY
    assert b"capacity" in L.dbde_hip_last_error(h)
    codec.sync()
    assert bool((buf == 0xEE).all()), "a refused call must not launch anything"
    # the context is still usable
    o, s = codec.encode_frames(imgs, W, H, n, buf, lead, cap)
    back, _ = codec.decode_frames(buf, lead, cap, o, W, H, n)
    codec.sync()
    assert torch.equal(back, imgs)


def test_crafted_minimum_wraps_per_byte(codec, oracle):
    """min + value > 255 cannot come out of an encoder but is decodable: each byte wraps on its
    own, as the reference's _mm_add_epi8 does (dbde_util.cpp:245-277)."""
    import torch
    W, H = 64, 24
    T = 8 * 3
    img = oracle.synth_frame(1, SEED, 3, W, H)
    frame = oracle.pack_frame(9, img, W, H).copy()
    frame[28 + T: 28 + 2 * T] = 250          # every tile's minimum
    adv, fh, want = oracle.unpack_frame(frame, W, H)
    assert fh[0] == 2
    dev = torch.device("cuda", 0)
    buf = torch.from_numpy(np.concatenate([np.zeros(32, np.uint8), frame, np.zeros(64, np.uint8)])).to(dev)
    offs = torch.zeros(1, dtype=torch.int64, device=dev)
    back, res = codec.decode_frames(buf, 32, len(frame), offs, W, H, 1)
    codec.sync()
    assert np.array_equal(back.cpu().numpy()[0], want)
    assert codec.parse_results(res)[0] == (2, 9, 0, adv)


def test_two_contexts_on_concurrent_streams(dv, oracle):
    """Two contexts on two HIP streams, launches interleaved without synchronisation: the
    persistent encoder may find the device shared (not all of its workgroups resident at once)
    and must still produce the same bytes -- it then takes chunks by ticket instead of by stride."""
    import torch
    W, H, n = 2048, 1024, 24
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    codecs = [dv.Codec(0, stream=s) for s in streams]
    try:
        state = []
        for k, c in enumerate(codecs):
            with torch.cuda.stream(streams[k]):
                imgs = c.synth_frames("mixed", SEED, 1000 * k, n, W, H)
                buf, lead, cap = c.alloc_stream(W, H, n)
                back = torch.empty_like(imgs)
                state.append([imgs, buf, lead, cap, back, None, None])
        torch.cuda.synchronize(dev)
        for rep in range(6):           # interleave: A enc, B enc, A dec, B dec, ...
            for k, c in enumerate(codecs):
                imgs, buf, lead, cap, back, _, _ = state[k]
                with torch.cuda.stream(streams[k]):
                    offs, sizes = c.encode_frames(imgs, W, H, n, buf, lead, cap, first_index=1000 * k)
                state[k][5], state[k][6] = offs, sizes
            for k, c in enumerate(codecs):
                imgs, buf, lead, cap, back, offs, sizes = state[k]
                with torch.cuda.stream(streams[k]):
                    c.decode_frames(buf, lead, cap, offs, W, H, n, images=back)
        for c in codecs:
            c.sync()
        for k, c in enumerate(codecs):
            imgs, buf, lead, cap, back, offs, sizes = state[k]
            assert torch.equal(back, imgs)
            host = buf.cpu().numpy()
            o, s = offs.cpu().numpy(), sizes.cpu().numpy()
            imgs_h = imgs.cpu().numpy()
            for f in (0, n // 2, n - 1):
                want = oracle.pack_frame(1000 * k + f, imgs_h[f], W, H)
                assert host[lead + o[f]: lead + o[f] + s[f]].tobytes() == want.tobytes(), (k, f)
    finally:
        for c in codecs:
            c.close()


def test_seeded_fuzz_against_oracle(codec, oracle):
    """120 seeded random cases: shape, frame count, layout, byte misalignment of the stream,
    and content built to stress depth transitions (random per-tile depth and minimum, saturated
    tiles, single-pixel outliers).  Encode bytes == oracle bytes, decode(encode) == image,
    decode of the ORACLE's bytes == image."""
    import torch
    rng = np.random.default_rng(0xDBDE)
    dev = torch.device("cuda", 0)
    for case in range(120):
        W = int(rng.integers(1, 200)) if case % 3 else int(rng.choice([8, 16, 64, 128, 256, 1024]))
        H = int(rng.integers(1, 120))
        n = int(rng.integers(1, 6))
        w, h = (W + 7) // 8, (H + 7) // 8
        # per-tile depth and minimum, pixel = min + noise < 2^depth, clipped to 255
        d = rng.integers(0, 9, (n, h, w))
        m = rng.integers(0, 256, (n, h, w))
        noise = rng.integers(0, 256, (n, h * 8, w * 8))
        dd = np.repeat(np.repeat(d, 8, 1), 8, 2)
        mm = np.repeat(np.repeat(m, 8, 1), 8, 2)
        img = np.minimum(mm + (noise & ((1 << dd) - 1)), 255).astype(np.uint8)[:, :H, :W]
        if case % 5 == 0:      # outliers: one pixel per frame at 0 and one at 255
            for f in range(n):
                img[f, rng.integers(0, H), rng.integers(0, W)] = 0
                img[f, rng.integers(0, H), rng.integers(0, W)] = 255
        img = np.ascontiguousarray(img)
        imgs = torch.from_numpy(img).to(dev)
        worst = codec.L.dbde_hip_max_frame_bytes(W, H)
        slot = 0 if case % 2 else int(worst + rng.integers(0, 64))
        mis = int(rng.integers(0, 16))
        frames, (buf, lead, offs, sizes) = gpu_encode(codec, imgs, W, H, n, first_index=case, slot_stride=slot,
                                                      misalign=mis)
        want = [oracle.pack_frame(case + f, img[f], W, H) for f in range(n)]
        for f in range(n):
            assert frames[f].tobytes() == want[f].tobytes(), (case, W, H, n, slot, mis, f)
        total = int((offs[-1] + sizes[-1]).item())
        back, res = codec.decode_frames(buf, lead, total, offs, W, H, n)
        # the oracle's stream, at another misalignment, through the index scanner
        mis2 = int(rng.integers(0, 16))
        stream = np.concatenate([np.zeros(mis2, np.uint8)] + want + [np.zeros(64, np.uint8)])
        sbuf = torch.from_numpy(stream).to(dev)
        slen = len(stream) - mis2 - 64
        offs2, found = codec.index_stream(sbuf, mis2, slen, W, H, n + 3)
        assert found == n
        back2, _ = codec.decode_frames(sbuf, mis2, slen, offs2, W, H, n)
        codec.sync()
        assert torch.equal(back, imgs) and torch.equal(back2, imgs), (case, W, H, n, slot, mis, mis2)


@pytest.mark.parametrize("W,H,n", [(8, 8, 1000), (64, 64, 300), (10, 10, 77), (33, 31, 50), (24, 16, 5), (1, 1, 9), (512, 8, 40),
                                   (7, 300, 33), (61, 59, 129),
                                   # round 4, the persistent small-frame encoder (8-byte rows, frames whole 16-byte blocks): 16, 15
                                   # (rows below the image repeated) and 4 tiles a frame, a last group that is part empty
                                   (32, 32, 333), (40, 20, 100), (16, 16, 1000),
                                   # rows of 4 mod 8 bytes: the staged decoder with two dwords per tile row, one for the row's last tile
                                   (20, 20, 300), (60, 60, 77), (36, 44, 100), (12, 9, 500),
                                   # ... frames that are no whole 16-byte blocks: the persistent encoder's image shifts from group to group
                                   (20, 10, 200), (28, 9, 150)])
@pytest.mark.parametrize("mode", ["noise8", "mixed", "smooth", "flat"])
def test_tiny_frames_many_per_wave(codec, codec_staged_decode, codec_three_workgroups, oracle, W, H, n, mode):
    """Frames of at most 64 tiles (the reference's randomized test is 1024 single-tile frames, dbde_util_test.cpp:66-96):
    one tile per lane, several frames per wave in both directions (encode_tiny_kernel for slots, decode_mid_kernel: whole frames per 256-thread workgroup);
    every frame byte for byte against the oracle, both layouts, partial tiles, a frame count that leaves lanes idle."""
    import torch
    imgs = codec.synth_frames(mode, SEED, 50, n, W, H)
    imgs_h = imgs.cpu().numpy()
    slot_bytes = ((codec.L.dbde_hip_max_frame_bytes(W, H) + 255) // 256) * 256
    for slot in (slot_bytes, 0):
        frames, (buf, lead, offs, sizes) = gpu_encode(codec, imgs, W, H, n, first_index=50, slot_stride=slot)
        for f in range(n):
            assert frames[f].tobytes() == oracle.pack_frame(50 + f, imgs_h[f], W, H).tobytes(), (W, H, mode, slot, f)
        if slot == slot_bytes:   # the persistent small-frame encoder on three workgroups: its double-buffered loop many times over
            frames3, _ = gpu_encode(codec_three_workgroups, imgs, W, H, n, first_index=50, slot_stride=slot)
            for f in range(n):
                assert frames3[f].tobytes() == frames[f].tobytes(), (W, H, mode, "three workgroups", f)
        total = int((offs[-1] + sizes[-1]).item())
        # (the default forms, the staged whole-frame decoder where it applies, three persistent workgroups walking their loop)
        for dec in (codec, codec_staged_decode, codec_three_workgroups):
            canvas = torch.full_like(imgs, 0xEE)
            back, res = dec.decode_frames(buf, lead, total, offs, W, H, n, images=canvas)
            dec.sync()
            assert torch.equal(back, imgs), (W, H, mode, slot)
            assert dec.parse_results(res) == [(2, 50 + f, 0, len(frames[f])) for f in range(n)]


@pytest.mark.parametrize("W,H,n", [(72, 72, 200), (96, 96, 150), (128, 128, 64), (160, 120, 90), (71, 73, 100), (176, 144, 41),
                                   (520, 8, 30), (9, 600, 25), (180, 180, 7), (130, 121, 1), (150, 150, 40), (220, 215, 9), (65, 64, 513),
                                   # round 4, the staged whole-frame kernels (8-byte rows): odd tile counts, rows below the image
                                   # repeated (H % 8 != 0), one frame per workgroup, a last workgroup that is part empty
                                   (72, 72, 7), (200, 168, 7), (100, 75, 50), (104, 100, 33), (200, 150, 19), (168, 161, 10), (224, 200, 5), (176, 144, 1),
                                   (520, 65, 9), (8, 5200, 3)])
@pytest.mark.parametrize("mode", ["noise8", "mixed", "smooth", "flat"])
def test_mid_frames_many_per_workgroup(codec, codec_staged_decode, codec_three_workgroups, oracle, W, H, n, mode):
    """Frames of 65 .. 512 tiles (72 .. 180 pixels a side): one tile per lane, as many whole frames per 256 / 512 / 1024
    thread workgroup as fit (encode_mid_kernel for slots; decode_mid_kernel where it is the faster form); every frame
    byte for byte against the oracle, both layouts, partial tiles, frame counts that leave the last workgroup part empty."""
    import torch
    imgs = codec.synth_frames(mode, SEED, 50, n, W, H)
    imgs_h = imgs.cpu().numpy()
    slot_bytes = ((codec.L.dbde_hip_max_frame_bytes(W, H) + 255) // 256) * 256
    for slot in (slot_bytes, slot_bytes + 3, 0):
        frames, (buf, lead, offs, sizes) = gpu_encode(codec, imgs, W, H, n, first_index=50, slot_stride=slot)
        for f in range(n):
            assert frames[f].tobytes() == oracle.pack_frame(50 + f, imgs_h[f], W, H).tobytes(), (W, H, mode, slot, f)
        if slot == slot_bytes:   # the persistent small-frame encoder on three workgroups: its double-buffered loop many times over
            frames3, _ = gpu_encode(codec_three_workgroups, imgs, W, H, n, first_index=50, slot_stride=slot)
            for f in range(n):
                assert frames3[f].tobytes() == frames[f].tobytes(), (W, H, mode, "three workgroups", f)
        total = int((offs[-1] + sizes[-1]).item())
        # the default forms, the staged whole-frame decoder where it applies, and decode_mid_kernel on three persistent
        # workgroups: every workgroup walks its software-pipelined loop several times (metadata of the next group and the
        # offset of the one after it in flight), the last iterations with groups that do not exist
        for dec in (codec, codec_staged_decode, codec_three_workgroups):
            canvas = torch.full_like(imgs, 0xEE)
            back, res = dec.decode_frames(buf, lead, total, offs, W, H, n, images=canvas)
            dec.sync()
            assert torch.equal(back, imgs), (W, H, mode, slot)
            assert dec.parse_results(res) == [(2, 50 + f, 0, len(frames[f])) for f in range(n)]


def _codec_with_experiment(dv, bits):
    import os
    old = os.environ.get("DBDE_HIP_EXPERIMENT")
    os.environ["DBDE_HIP_EXPERIMENT"] = str(bits | int(old or "0", 0))
    c = dv.Codec(0)
    if old is None:
        del os.environ["DBDE_HIP_EXPERIMENT"]
    else:
        os.environ["DBDE_HIP_EXPERIMENT"] = old
    return c


@pytest.fixture(scope="module")
def codec_three_workgroups(dv):
    """A context whose decode_mid_kernel launches have three (persistent) workgroups ($DBDE_HIP_EXPERIMENT bit 10)."""
    c = _codec_with_experiment(dv, 1024)
    yield c
    c.close()


@pytest.fixture(scope="module")
def codec_staged_decode(dv):
    """A context whose decode calls take decode_frames_kernel where the geometry allows ($DBDE_HIP_EXPERIMENT bit 8: the
    staged whole-frame decoder is built and correct but not the default, it measured no faster)."""
    import os
    old = os.environ.get("DBDE_HIP_EXPERIMENT")
    os.environ["DBDE_HIP_EXPERIMENT"] = str(256 | int(old or "0", 0))
    c = dv.Codec(0)
    if old is None:
        del os.environ["DBDE_HIP_EXPERIMENT"]
    else:
        os.environ["DBDE_HIP_EXPERIMENT"] = old
    yield c
    c.close()


@pytest.mark.parametrize("which", ["default", "staged", "three_workgroups"])
@pytest.mark.parametrize("W,H,n", [(72, 72, 50), (160, 120, 23), (96, 96, 61), (200, 150, 9), (60, 60, 40), (100, 76, 30)])
def test_staged_frame_decoder_takes_any_offsets_and_rejects_like_the_reference(codec, codec_staged_decode, codec_three_workgroups, oracle, W, H, n, which):
    codec = {"default": codec, "staged": codec_staged_decode, "three_workgroups": codec_three_workgroups}[which]
    """decode_frames_kernel: frames wherever they lie (concatenated: every alignment mod 16; a stream base that is odd), a
    readable extent that ends with the last frame, and malformed frames among good ones -- nb / nm / n64 wrong, a depth
    byte above 8, a truncated frame, a wild offset: rejected with u64s = 0xFFFFFFFF and consumed = 20 (dbde_util.cpp:
    295-303, 335, 342), their images untouched, their neighbours decoded."""
    import torch
    T = ((W + 7) // 8) * ((H + 7) // 8)
    imgs = codec.synth_frames("mixed", SEED, 3, n, W, H)
    imgs_h = imgs.cpu().numpy()
    for misalign in (0, 1, 6):
        frames, (buf, lead, offs, sizes) = gpu_encode(codec, imgs, W, H, n, first_index=3, slot_stride=0, misalign=misalign)
        for f in (0, n // 2, n - 1):
            assert frames[f].tobytes() == oracle.pack_frame(3 + f, imgs_h[f], W, H).tobytes()
        total = int((offs[-1] + sizes[-1]).item())
        canvas = torch.full_like(imgs, 0xEE)
        back, res = codec.decode_frames(buf, lead, total, offs, W, H, n, images=canvas)      # extent ends with the last frame
        codec.sync()
        assert torch.equal(back, imgs), (W, H, misalign)
        o, s_ = offs.cpu().numpy(), sizes.cpu().numpy()
        bad = buf.clone()
        victims = {}
        def poke(f, pos, delta):
            a = lead + int(o[f]) + pos
            bad[a] = (int(bad[a].item()) + delta) % 256
        if n >= 9:
            poke(1, 20, 1); victims[1] = "nb"
            poke(2, 24 + T, 1); victims[2] = "nm"
            poke(4, 28 + 2 * T, 1); victims[4] = "n64"
            a = lead + int(o[5]) + 24
            bad[a] = 9; victims[5] = "depth 9"
            poke(7, 0, 1); victims[7] = "frame field"          # header field wrong: the image IS decoded (dbde_util.cpp:339-345), u64s = -1
        offs_bad = offs.clone()
        offs_bad[n - 1] = total - 5                                # a frame that starts 5 bytes before the end of the extent
        victims[n - 1] = "truncated"
        canvas = torch.full_like(imgs, 0xEE)
        back, res = codec.decode_frames(bad, lead, total, offs_bad, W, H, n, images=canvas)
        codec.sync()
        rs = codec.parse_results(res)
        for f in range(n):
            why = victims.get(f)
            if why is None:
                assert torch.equal(back[f], imgs[f]) and rs[f] == (2, 3 + f, 0, int(s_[f])), (f, rs[f])
            elif why == "frame field":
                assert rs[f][0] == 0xFFFFFFFF and torch.equal(back[f], imgs[f]), (f, rs[f])
            else:
                assert rs[f][0] == 0xFFFFFFFF and rs[f][3] == 20 and bool((back[f] == 0xEE).all()), (f, why, rs[f])


@pytest.mark.parametrize("W,H,n,mode,concat", [(1921, 1081, 64, "noise8", True), (2048, 1024, 70, "mixed", False),
                                               (2048, 1024, 48, "mixed", True), (2048, 1024, 80, "smooth", False),
                                               (2048, 1024, 97, "mixed", True)])
def test_persistent_encoder_chunk_counts_around_the_ticket_boundary(codec, oracle, W, H, n, mode, concat):
    """The persistent encoder hands out the ids of the last three rounds as tickets even in static mode; the first two
    chunks of a workgroup (rank, rank + G) are always static.  Launches of 3 G ... 6 G chunks sit on every side of that
    boundary (a soak run found 4 G <= chunks < 5 G handing the ids G ... 2 G out twice: look-back time-out)."""
    import torch
    imgs = codec.synth_frames(mode, SEED, 7, n, W, H)
    slot = 0 if concat else ((codec.L.dbde_hip_max_frame_bytes(W, H) + 255) // 256) * 256
    for rep in range(2):
        frames, (buf, lead, offs, sizes) = gpu_encode(codec, imgs, W, H, n, first_index=7, slot_stride=slot)
    imgs_h = imgs.cpu().numpy()
    for f in (0, 1, n // 2, n - 2, n - 1):
        assert frames[f].tobytes() == oracle.pack_frame(7 + f, imgs_h[f], W, H).tobytes(), f
    cap = (n - 1) * slot + codec.L.dbde_hip_max_frame_bytes(W, H) if slot else int((offs[-1] + sizes[-1]).item())
    back, res = codec.decode_frames(buf, lead, cap, offs, W, H, n)
    codec.sync()
    assert torch.equal(back, imgs)
