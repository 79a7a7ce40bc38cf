"""GPU parity for the file ends of the path: the batched .dbde writer and reader
(dbde_hip_writer_* / dbde_hip_reader_*, include/dbde_hip.h) against the oracle's bytes and,
where oracle/_ref was built, the reference's own file walker (dbde_util.cpp:362-426).

A file is bit-exact when it equals  video_header(28 B) + dbde_pack_frame(...) for every frame.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0xDBDE2016


@pytest.fixture(scope="module")
def dv():
    import dbde_video_cpp_amd as m
    return m


@pytest.fixture(scope="module")
def codec(dv):
    c = dv.Codec(0)
    yield c
    c.close()


def oracle_file(oracle, imgs_h, W, H, hz, indices):
    parts = [oracle.pack_video_header(3, H, W, hz)]
    parts += [oracle.pack_frame(int(indices[f]), imgs_h[f], W, H) for f in range(len(imgs_h))]
    return np.concatenate(parts)


@pytest.mark.parametrize("W,H,n,batch,mode", [(200, 123, 11, 4, "mixed"), (10, 10, 5, 16, "smooth"),
                                              (64, 64, 9, 3, "noise8"), (1921, 1081, 5, 2, "mixed"),
                                              (33, 31, 7, 1, "mixed"), (8, 8, 3, 2, "flat")])
def test_writer_bytes_match_oracle(codec, oracle, tmp_path, W, H, n, batch, mode):
    imgs = codec.synth_frames(mode, SEED, 40, n, W, H)
    path = str(tmp_path / "w.dbde")
    w = codec.open_writer(path, W, H, frame_hz=29.97, batch_frames=batch)
    # two puts: the second continues the numbering
    k = n // 2
    w.put(imgs[:k], k, first_index=40)
    w.put(imgs[k:], n - k, first_index=40 + k)
    frames, nbytes = w.close()
    got = np.fromfile(path, np.uint8)
    want = oracle_file(oracle, imgs.cpu().numpy(), W, H, 29.97, range(40, 40 + n))
    assert frames == n and nbytes == len(got)
    assert got.tobytes() == want.tobytes()


def test_writer_indices_and_elapsed(codec, oracle, tmp_path):
    import torch
    W, H, n = 40, 24, 6
    imgs = codec.synth_frames("mixed", SEED, 0, n, W, H)
    idx_h = np.array([7, 0, 2**40 + 5, 3, 99, 2**63 + 1], np.uint64)
    el_h = np.array([0, 1, 1000000007, 2**53, 123456789, 5], np.uint64)
    idx = torch.from_numpy(idx_h.view(np.int64)).to(imgs.device)
    el = torch.from_numpy(el_h.view(np.int64)).to(imgs.device)
    path = str(tmp_path / "ie.dbde")
    with codec.open_writer(path, W, H, batch_frames=4) as w:
        w.put(imgs, n, indices=idx, elapsed_ns=el)
    got = np.fromfile(path, np.uint8)
    imgs_h = imgs.cpu().numpy()
    parts = [oracle.pack_video_header(3, H, W, 30.0)]
    for f in range(n):
        parts.append(oracle.pack_frame_header(2, int(idx_h[f]), int(el_h[f])))
        parts.append(oracle.pack_image(imgs_h[f], W, H))
    assert got.tobytes() == np.concatenate(parts).tobytes()
    # and back through the reader: headers as dbde_unpack_frame_header returns them
    with codec.open_reader(path, batch_frames=4) as r:
        hdrs, back = [], []
        for im, hd in r:
            hdrs += hd
            back.append(im.clone())
    assert torch.equal(torch.cat(back), imgs)
    for f, h in enumerate(hdrs):
        _, want = oracle.unpack_frame_header(oracle.pack_frame_header(2, int(idx_h[f]), int(el_h[f])))
        assert h == tuple(want)


@pytest.mark.parametrize("W,H,n,batch,ask", [(200, 123, 11, 4, None), (10, 10, 5, 16, None), (64, 64, 9, 3, 2),
                                             (1921, 1081, 5, 2, None), (1, 1, 6, 4, 3)])
def test_reader_reads_oracle_file(codec, oracle, tmp_path, W, H, n, batch, ask):
    import torch
    imgs_h = np.stack([oracle.synth_frame(1, SEED, 300 + f, W, H) for f in range(n)])
    path = str(tmp_path / "o.dbde")
    oracle_file(oracle, imgs_h, W, H, 60.0, range(300, 300 + n)).tofile(path)
    r = codec.open_reader(path, batch_frames=batch)
    assert r.video_header == (3, H, W, 60.0)
    seen = 0
    while True:
        im, hd = r.next(max_frames=ask)
        if not hd:
            break
        assert len(hd) <= (ask or batch)
        for k, h in enumerate(hd):
            assert h == (2, 300 + seen + k, 0)
        assert np.array_equal(im.cpu().numpy().reshape(len(hd), H, W), imgs_h[seen:seen + len(hd)])
        seen += len(hd)
    assert seen == n
    # a finished walk stays finished
    assert r.next()[1] == []
    r.close()


def test_reference_walker_reads_writer_file(codec, reference, tmp_path):
    """The reference's own dbde_walk_a_file decodes what the HIP writer wrote."""
    W, H, n = 333, 77, 13
    imgs = codec.synth_frames("mixed", SEED, 0, n, W, H)
    path = str(tmp_path / "ref.dbde")
    with codec.open_writer(path, W, H, batch_frames=5) as w:
        w.put(imgs, n, first_index=1000)
    for keep in (1, 7, 13):
        cnt, hw, last, img = reference.walk_file(path, 3, W, H, keep=keep)
        assert cnt == n and hw == (H, W) and last == 1000 + n - 1
        assert np.array_equal(img, imgs[keep - 1].cpu().numpy())


def test_reader_stops_like_the_walker(codec, oracle, tmp_path):
    """Truncated tail / frame that does not parse: frames before it are delivered, then the
    walk ends (dbde_walk_a_file returns false, dbde_util.cpp:412-420)."""
    W, H, n = 48, 40, 8
    T = 6 * 5
    imgs_h = np.stack([oracle.synth_frame(1, SEED, f, W, H) for f in range(n)])
    frames = [oracle.pack_frame(f, imgs_h[f], W, H) for f in range(n)]
    head = oracle.pack_video_header(3, H, W, 30.0)

    def walk(data, batch):
        path = str(tmp_path / f"bad{batch}.dbde")
        np.asarray(data, np.uint8).tofile(path)
        out = []
        with codec.open_reader(path, batch_frames=batch) as r:
            for im, hd in r:
                out += [(h, im[k].cpu().numpy().copy()) for k, h in enumerate(hd)]
        return out

    for batch in (1, 3, 16):
        # cut in the middle of frame 5's payload
        cut = np.concatenate([head] + frames[:5] + [frames[5][:len(frames[5]) - 9]])
        got = walk(cut, batch)
        assert [h[1] for h, _ in got] == [0, 1, 2, 3, 4]
        # frame 3: nb field wrong -> does not parse
        bad = [f.copy() for f in frames]
        bad[3][20] ^= 1
        got = walk(np.concatenate([head] + bad), batch)
        assert [h[1] for h, _ in got] == [0, 1, 2]
        assert all(np.array_equal(im, imgs_h[k]) for k, (_, im) in enumerate(got))
        # frame 6: a depth byte of 9 (the documented deviation: rejected)
        bad = [f.copy() for f in frames]
        bad[6][24 + 2] = 9
        got = walk(np.concatenate([head] + bad), batch)
        assert [h[1] for h, _ in got] == [0, 1, 2, 3, 4, 5]
        # frame 2: n64 larger than any frame could carry
        bad = [f.copy() for f in frames]
        bad[2][28 + 2 * T: 32 + 2 * T] = np.frombuffer(np.uint32(8 * T + 1).tobytes(), np.uint8)
        got = walk(np.concatenate([head] + bad), batch)
        assert [h[1] for h, _ in got] == [0, 1]
        # only a video header: zero frames
        assert walk(head, batch) == []


def test_reader_rejects_bad_video_header(codec, dv, oracle, tmp_path):
    for vh in (oracle.pack_video_header(2, 8, 8, 30.0), oracle.pack_video_header(3, 0, 8, 30.0),
               oracle.pack_video_header(3, 0x38000000, 1, 30.0), oracle.pack_video_header(3, 8, 8, 30.0)[:20]):
        path = str(tmp_path / "h.dbde")
        np.asarray(vh, np.uint8).tofile(path)
        with pytest.raises(dv.DbdeError):
            codec.open_reader(path)
    with pytest.raises(dv.DbdeError):
        codec.open_reader(str(tmp_path / "missing.dbde"))


def test_file_round_trip_full_size(codec, tmp_path):
    """BASELINE configs[1] shape through a file: 6 frames of 4096x3072, windows of 4."""
    import torch
    W, H, n = 4096, 3072, 6
    imgs = codec.synth_frames("mixed", SEED, 0, n, W, H)
    path = str(tmp_path / "big.dbde")
    with codec.open_writer(path, W, H, batch_frames=4) as w:
        w.put(imgs, n, first_index=5)
    back = torch.empty_like(imgs)
    got = 0
    with codec.open_reader(path, batch_frames=4) as r:
        while True:
            im, hd = r.next(images=back[got:got + 4] if got + 4 <= n else None)
            if not hd:
                break
            if got + 4 > n:
                back[got:got + len(hd)] = im
            assert [h[1] for h in hd] == list(range(5 + got, 5 + got + len(hd)))
            got += len(hd)
    assert got == n and torch.equal(back, imgs)
