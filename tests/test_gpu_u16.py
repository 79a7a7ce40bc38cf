"""GPU: DBDE16 kernels (dbde16_hip_encode_frames / dbde16_hip_decode_frames) against the DBDE16 oracle
(oracle/dbde16_oracle.c = the extension's specification; PARITY UNPINNED, see tests/test_oracle_u16.py for what
holds the oracle in place).  Byte-for-byte frames, round trips, both layouts, edge tiles, full 16-bit range,
malformed frames."""
import ctypes as C

import numpy as np
import pytest

from test_oracle_u16 import o16, pack16   # noqa: F401  (fixture + helper)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def codec():
    import dbde_video_cpp_amd as dv
    dv.build()
    c = dv.Codec(0)
    yield c
    c.close()


def make_images(rng, n, W, H, kind):
    if kind == "full":
        img = rng.integers(0, 65536, size=(n, H, W))
    elif kind == "mixed":      # a different bit depth per 8x8 tile, 0..16
        d = rng.integers(0, 17, size=(n, (H + 7) // 8, (W + 7) // 8))
        dd = np.repeat(np.repeat(d, 8, axis=1), 8, axis=2)[:, :H, :W]
        base = rng.integers(0, 65536, size=(n, (H + 7) // 8, (W + 7) // 8))
        bb = np.repeat(np.repeat(base, 8, axis=1), 8, axis=2)[:, :H, :W]
        img = np.minimum(bb >> 1, 65535 - ((1 << dd) - 1)) + (rng.integers(0, 65536, size=(n, H, W)) & ((1 << dd) - 1))
    else:                       # 8-bit content in 16-bit pixels
        img = rng.integers(0, 256, size=(n, H, W)) >> 3
    return np.ascontiguousarray(img.astype(np.uint16))


@pytest.mark.parametrize("W,H,n", [(8, 8, 1), (10, 10, 3), (64, 64, 4), (200, 123, 5), (1, 1, 2), (7, 300, 2),
                                   (1024, 40, 3), (4104, 16, 2), (33, 31, 7)])
@pytest.mark.parametrize("kind", ["full", "mixed", "small"])
def test_encode_matches_oracle_and_round_trips(codec, o16, W, H, n, kind):
    import torch
    rng = np.random.default_rng(W * 7 + H * 3 + n + len(kind))
    imgs_h = make_images(rng, n, W, H, kind)
    imgs = torch.from_numpy(imgs_h.view(np.int16)).cuda()
    maxf = int(codec.L.dbde16_hip_max_frame_bytes(W, H))
    for slot in (0, ((maxf + 255) // 256) * 256):
        cap = (n - 1) * slot + maxf if slot else n * maxf
        buf = torch.full((32 + cap + 64,), 0xEE, dtype=torch.uint8, device="cuda")
        offs, sizes = codec.encode_frames16(imgs, W, H, n, buf, 32, cap, first_index=11, slot_stride=slot)
        codec.sync()
        host, o, s = buf.cpu().numpy(), offs.cpu().numpy(), sizes.cpu().numpy()
        for f in range(n):
            want = pack16(o16, imgs_h[f], 11 + f)
            got = host[32 + o[f]: 32 + o[f] + s[f]]
            assert int(s[f]) == len(want) and got.tobytes() == want.tobytes(), (W, H, kind, slot, f)
        if slot == 0:
            assert o[0] == 0 and (o[1:] == np.cumsum(s)[:-1]).all()
            assert (host[32 + int(o[-1] + s[-1]):-64] == 0xEE).all()       # nothing written past the stream
        back, res = codec.decode_frames16(buf, 32, cap, offs, W, H, n)
        codec.sync()
        assert (back.cpu().numpy().view(np.uint16) == imgs_h).all(), (W, H, kind, slot)
        for f, r in enumerate(codec.parse_results(res)):
            assert r == (2, 11 + f, 0, int(s[f]))


def test_full_size_round_trip_and_spot_check(codec, o16):
    import torch
    W, H, n = 2048, 1536, 6
    rng = np.random.default_rng(5)
    imgs_h = make_images(rng, n, W, H, "mixed")
    imgs = torch.from_numpy(imgs_h.view(np.int16)).cuda()
    cap = n * int(codec.L.dbde16_hip_max_frame_bytes(W, H))
    buf = torch.empty(32 + cap + 64, dtype=torch.uint8, device="cuda")
    offs, sizes = codec.encode_frames16(imgs, W, H, n, buf, 32, cap)
    back, res = codec.decode_frames16(buf, 32, cap, offs, W, H, n)
    codec.sync()
    assert torch.equal(back, imgs)
    o, s = offs.cpu().numpy(), sizes.cpu().numpy()
    want = pack16(o16, imgs_h[3], 3)
    assert buf[32 + int(o[3]): 32 + int(o[3] + s[3])].cpu().numpy().tobytes() == want.tobytes()


@pytest.mark.parametrize("flags", [0, 1, 32, 33])
@pytest.mark.parametrize("kind", ["mixed", "full"])
def test_more_chunks_than_resident_workgroups(o16, kind, flags):
    """Launches whose chunks outnumber the resident workgroups.  Rows of 16-byte aligned pixels go through the 8-bit
    path's persistent encoder (encode_kernel<.., PIX = 2>); DBDE_HIP_EXPERIMENT bit 5 keeps them on enc16_kernel, the
    kernel of every other geometry.  Both claim chunk ids by static strides once all workgroups have arrived, or by
    tickets (forced through bit 0); `full` content (every tile of depth 16) also takes enc16_kernel's path of chunks
    larger than the LDS payload image."""
    import os
    import torch
    import dbde_video_cpp_amd as dv
    force_tickets = flags
    if flags:
        os.environ["DBDE_HIP_EXPERIMENT"] = str(flags)
    try:
        c2 = dv.Codec(0)
    finally:
        os.environ.pop("DBDE_HIP_EXPERIMENT", None)
    try:
        W, H, n = 2048, 1536, 16    # 3072 chunks of 256 tiles
        rng = np.random.default_rng(17 + len(kind))
        imgs_h = make_images(rng, n, W, H, kind)
        imgs = torch.from_numpy(imgs_h.view(np.int16)).cuda()
        maxf = int(c2.L.dbde16_hip_max_frame_bytes(W, H))
        for slot in (0, ((maxf + 255) // 256) * 256):
            cap = (n - 1) * slot + maxf if slot else n * maxf
            buf = torch.empty(32 + cap + 64, dtype=torch.uint8, device="cuda")
            for rep in range(2):
                offs, sizes = c2.encode_frames16(imgs, W, H, n, buf, 32, cap, slot_stride=slot)
                back, res = c2.decode_frames16(buf, 32, cap, offs, W, H, n)
                c2.sync()
                assert torch.equal(back, imgs), (kind, flags, slot, rep)
            o, s = offs.cpu().numpy(), sizes.cpu().numpy()
            for f in (0, 7, 15):
                want = pack16(o16, imgs_h[f], f)
                assert buf[32 + int(o[f]): 32 + int(o[f] + s[f])].cpu().numpy().tobytes() == want.tobytes(), (kind, slot, f)
    finally:
        c2.close()


@pytest.mark.parametrize("W,H,n,off", [(1000, 1003, 40, 32), (1000, 1003, 40, 33), (1024, 768, 64, 40), (8, 8, 600 * 512, 32),
                                       (4096, 3072, 3, 32), (1001, 1003, 40, 32)])
@pytest.mark.parametrize("kind", ["mixed", "full", "small"])
def test_persistent_encoder_geometries(codec, o16, W, H, n, off, kind):
    """The PIX = 2 instance of the persistent encoder: a last tile row that is padded (H % 8), tile counts that leave
    the U16 minima and the payload unaligned (T % 8, odd output offsets), one-tile frames (every chunk nearly
    empty), frames of many chunks; 1001 wide takes the any-geometry instance (round 4).  Every frame byte for byte against the oracle
    at small n, a sample of frames otherwise."""
    import torch
    rng = np.random.default_rng(W + 3 * H + n + off + len(kind))
    if n > 1000:   # many one-tile frames: a few distinct ones, repeated
        base = make_images(rng, 997, W, H, kind)
        imgs_h = np.ascontiguousarray(base[np.arange(n) % 997])
    else:
        imgs_h = make_images(rng, n, W, H, kind)
    imgs = torch.from_numpy(imgs_h.view(np.int16)).cuda()
    maxf = int(codec.L.dbde16_hip_max_frame_bytes(W, H))
    for slot in (0, ((maxf + 255) // 256) * 256, maxf + 3):
        cap = (n - 1) * slot + maxf if slot else n * maxf
        buf = torch.full((off + cap + 64,), 0xEE, dtype=torch.uint8, device="cuda")
        offs, sizes = codec.encode_frames16(imgs, W, H, n, buf, off, cap, first_index=5, slot_stride=slot)
        back, res = codec.decode_frames16(buf, off, cap, offs, W, H, n)
        codec.sync()
        assert torch.equal(back, imgs), (W, H, kind, slot)
        host, o, s = buf.cpu().numpy(), offs.cpu().numpy(), sizes.cpu().numpy()
        sample = range(n) if n <= 64 else list(range(0, n, max(1, n // 61))) + [n - 1]
        for f in sample:
            want = pack16(o16, imgs_h[f], 5 + f)
            got = host[off + int(o[f]): off + int(o[f] + s[f])]
            assert int(s[f]) == len(want) and got.tobytes() == want.tobytes(), (W, H, kind, slot, f)
        if slot == 0:
            assert o[0] == 0 and (o[1:] == np.cumsum(s)[:-1]).all()
            assert (host[off + int(o[-1] + s[-1]):-64] == 0xEE).all() and (host[:off] == 0xEE).all()


@pytest.mark.parametrize("W,H,n,shift", [(1001, 67, 180, 0), (1002, 67, 180, 0), (1003, 61, 180, 1), (1004, 67, 180, 0), (1005, 67, 180, 3),
                                         (1006, 70, 180, 0), (1007, 67, 180, 0), (1000, 67, 180, 1), (1000, 67, 180, 4), (9, 9, 700, 0),
                                         (15, 8, 600, 1)])
@pytest.mark.parametrize("kind", ["mixed", "full"])
def test_persistent_encoder_any_width(codec, o16, W, H, n, shift, kind):
    """Round 4: widths that are no multiple of 8 pixels (and aligned widths at a base that is not 16-byte aligned) through
    the persistent encoder as well (encode_kernel<kInRaw, .., PIX = 2>): every count of valid columns in a row's last tile,
    images that END with their buffer (the batch's last fetch is moved left), a sample of frames byte for byte against the
    oracle incl. the first and the last, round trip."""
    import torch
    rng = np.random.default_rng(W * 5 + H + n + shift + len(kind))
    imgs_h = make_images(rng, n, W, H, kind)
    imgs_h[-1, -1, -8:] = rng.integers(0, 65536, 8)
    flat = torch.empty(shift + n * H * W, dtype=torch.int16, device="cuda")          # the images end where the buffer ends
    imgs = flat[shift:].view(n, H, W)
    imgs.copy_(torch.from_numpy(imgs_h.view(np.int16)))
    maxf = int(codec.L.dbde16_hip_max_frame_bytes(W, H))
    for slot in (0, ((maxf + 255) // 256) * 256):
        cap = (n - 1) * slot + maxf if slot else n * maxf
        buf = torch.full((32 + cap + 64,), 0xEE, dtype=torch.uint8, device="cuda")
        offs, sizes = codec.encode_frames16(imgs, W, H, n, buf, 32, cap, first_index=9, slot_stride=slot)
        back, res = codec.decode_frames16(buf, 32, cap, offs, W, H, n)
        codec.sync()
        assert torch.equal(back, imgs), (W, H, kind, slot)
        host, o, s = buf.cpu().numpy(), offs.cpu().numpy(), sizes.cpu().numpy()
        for f in sorted(set([0, 1, n // 2, n - 2, n - 1] + [int(x) for x in rng.integers(0, n, 6)])):
            want = pack16(o16, imgs_h[f], 9 + f)
            got = host[32 + int(o[f]): 32 + int(o[f] + s[f])]
            assert int(s[f]) == len(want) and got.tobytes() == want.tobytes(), (W, H, kind, slot, f)


def test_malformed_frames_are_rejected(codec, o16):
    import torch
    W, H = 24, 16
    T = 6
    img = (np.arange(W * H).reshape(H, W) * 377 % 65536).astype(np.uint16)
    good = pack16(o16, img, 4)
    frames = []
    for mutate in ("ok", "depth", "nm", "n64"):
        g = good.copy()
        if mutate == "depth":
            g[24] = 17
        elif mutate == "nm":
            g[24 + T:28 + T] = np.frombuffer(np.uint32(T).tobytes(), np.uint8)
        elif mutate == "n64":
            g[28 + 3 * T] ^= 1
        frames.append(g)
    stream = np.concatenate(frames)
    offs_h = np.cumsum([0] + [len(f) for f in frames[:-1]])
    buf = torch.from_numpy(np.concatenate([stream, np.zeros(64, np.uint8)])).cuda()
    offs = torch.from_numpy(offs_h.astype(np.int64)).cuda()
    canvas = torch.full((4, H, W), 0x5A5A, dtype=torch.int16, device="cuda")
    back, res = codec.decode_frames16(buf, 0, len(stream), offs, W, H, 4, images=canvas)
    codec.sync()
    r = codec.parse_results(res)
    assert r[0] == (2, 4, 0, len(good)) and (back[0].cpu().numpy().view(np.uint16) == img).all()
    for k in (1, 2, 3):
        assert r[k][0] == 0xFFFFFFFF and r[k][3] == 20 and (back[k] == 0x5A5A).all()
