"""GPU, one rank: the native decode-side shard (dbde_hip_scatter_*, csrc/dbde_scatter.cpp) end to end on an MI355X --
communicator from a unique id, the block table worked out on the device from the scanner's offsets and count, its
broadcast and the capacity all-gather, the root's block in place and (DBDE_HIP_SCATTER_LOOPBACK) through ncclSend /
ncclRecv in pieces, the offsets rebased on the device -- and the scattered block decoded bit-exact.  Rank-to-rank
traffic needs more than one GPU and has not run (tests/test_scatter_plan.py plays worlds 1-8 on the CPU)."""
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


@pytest.mark.parametrize("W,H,n,mode", [(1921, 1081, 40, "mixed"), (64, 64, 700, "mixed"), (2048, 2048, 24, "noise8")])
def test_one_rank_scatter_in_place_and_loopback(dv, codec, oracle, W, H, n, mode):
    import torch
    imgs = codec.synth_frames(mode, SEED, 0, n, W, H)
    buf, lead, cap = codec.alloc_stream(W, H, n)
    offs, sizes = codec.encode_frames(imgs, W, H, n, buf, lead, cap, first_index=0)
    codec.sync()
    total = int((offs[-1] + sizes[-1]).item())
    dev = imgs.device
    found = torch.empty(n + 4, dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    sc = dv.Scatter(codec, dv.gather_unique_id(), 1, 0, max_message_bytes=3 << 20)
    try:
        for slot, loopback in ((0, False), (1, True), (0, True)):
            codec.index_stream_async(buf, lead, total, W, H, n + 4, found, count)     # a reader's view: bytes + length only
            seg = torch.full((total + 64,), 0xEE, dtype=torch.uint8, device=dev)
            my_offs = torch.full((n + 4,), -1, dtype=torch.int64, device=dev)
            sc.set_capacity(total, n)
            sc.begin(slot, buf, lead, total, found, count)
            mine, table = sc.post(slot, seg, my_offs, loopback=loopback)
            assert mine == (0, n, 0, total) and table == [mine]
            sc.join(slot)
            src, src_off = (seg, 0) if loopback else (buf, lead + mine[2])
            back, res = codec.decode_frames(src, src_off, mine[3], my_offs, W, H, mine[1])
            codec.sync()
            assert torch.equal(my_offs[:n], offs) and (my_offs[n:] == -1).all()
            assert torch.equal(back, imgs), (W, H, loopback)
            if loopback:
                assert torch.equal(seg[:total], buf[lead:lead + total]) and (seg[total:] == 0xEE).all()
            for f, (u64s, index, el, consumed) in enumerate(codec.parse_results(res)):
                assert (u64s, index, el) == (2, f, 0)
        # one frame of the scattered block against the oracle, byte for byte
        f = n // 2
        o, s = int(offs[f].item()), int(sizes[f].item())
        want = oracle.pack_frame(f, oracle.synth_frame(dv.MODES[mode], SEED, f, W, H), W, H)
        assert seg[o:o + s].cpu().numpy().tobytes() == want.tobytes()
        # a block that does not fit the declared buffers: one verdict, nothing posted, the handle stays usable
        sc.set_capacity(total - 1, n)
        codec.index_stream_async(buf, lead, total, W, H, n + 4, found, count)
        sc.begin(0, buf, lead, total, found, count)
        with pytest.raises(dv.DbdeError, match="does not fit"):
            sc.post(0, seg, my_offs, loopback=True)
        sc.set_capacity(total, n - 1)
        sc.begin(0, buf, lead, total, found, count)
        with pytest.raises(dv.DbdeError, match="does not fit"):
            sc.post(0, seg, my_offs, loopback=True)
        sc.set_capacity(total, n)
        sc.begin(0, buf, lead, total, found, count)
        mine, _ = sc.post(0, seg, my_offs, loopback=True)
        sc.sync(0)
        assert mine == (0, n, 0, total)
    finally:
        sc.close()


def test_empty_and_truncated_streams(dv, codec):
    """No frame found (empty stream) and a truncated tail: the table follows the scanner's count."""
    import torch
    W, H, n = 200, 123, 9
    imgs = codec.synth_frames("mixed", SEED, 0, n, W, H)
    buf, lead, cap = codec.alloc_stream(W, H, n)
    offs, sizes = codec.encode_frames(imgs, W, H, n, buf, lead, cap, first_index=0)
    codec.sync()
    total = int((offs[-1] + sizes[-1]).item())
    dev = imgs.device
    found = torch.empty(n, dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    seg = torch.empty(total, dtype=torch.uint8, device=dev)
    my_offs = torch.empty(n, dtype=torch.int64, device=dev)
    sc = dv.Scatter(codec, dv.gather_unique_id(), 1, 0)
    try:
        sc.set_capacity(total, n)
        for nbytes, want_frames in ((total - 3, n - 1), (10, 0)):
            codec.index_stream_async(buf, lead, nbytes, W, H, n, found, count)
            sc.begin(0, buf, lead, nbytes, found, count)
            mine, _ = sc.post(0, seg, my_offs, loopback=True)
            sc.sync(0)
            assert mine[:2] == (0, want_frames)
            if want_frames:
                assert mine[3] == nbytes      # the block runs to the end of the readable extent
                back, _ = codec.decode_frames(seg, 0, int(offs[want_frames].item()), my_offs, W, H, want_frames)
                codec.sync()
                assert torch.equal(back, imgs[:want_frames])
    finally:
        sc.close()
