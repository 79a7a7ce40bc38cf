"""GPU: the streaming driver (dbde_video_cpp_amd.streaming.RoundTripStream) -- frames produced on a side stream,
two input and two stream slots, every batch encoded, decoded back and compared -- on one rank.  (The multi-rank
gather it overlaps is exercised on gloo in tests/test_distributed_gloo.py and by bench.py's rehearsal mode.)"""
import pytest

pytestmark = pytest.mark.gpu
SEED = 0xDBDE2016


@pytest.fixture(scope="module")
def dv():
    import dbde_video_cpp_amd as dv
    dv.build()
    return dv


def test_round_trip_stream_side_stream_source(dv, oracle):
    import torch
    from dbde_video_cpp_amd.streaming import RoundTripStream
    W, H, n, batch = 200, 123, 13, 3          # uneven last batch, edge tiles
    codec = dv.Codec(0)
    side = torch.cuda.Stream()
    src = dv.Codec(0, stream=side)
    produced = []

    def source(first, k, out):
        produced.append((first, k))
        src.synth_frames("mixed", SEED, first, k, W, H, out=out)

    rts = RoundTripStream(codec, W, H, batch, source=source, source_stream=side, check=True)
    r = rts.run(40, n)
    assert r["frames"] == n and r["batches"] == 5 and rts.mismatches == 0
    assert produced == [(40, 3), (43, 3), (46, 3), (49, 3), (52, 1)]
    # the last batch's stream slot still holds frame 52, byte for byte the oracle's
    buf, lead, cap = rts.out[(r["batches"] - 1) % 2]
    size = int(rts.sizes[(r["batches"] - 1) % 2][0].item())
    want = oracle.pack_frame(52, oracle.synth_frame(1, SEED, 52, W, H), W, H)
    assert size == len(want) and buf[lead:lead + size].cpu().numpy().tobytes() == want.tobytes()
    src.close()
    codec.close()


def test_round_trip_stream_resident_ring(dv):
    """source=None: the caller's frames stay in the two input slots (a ring of resident batches)."""
    import torch
    from dbde_video_cpp_amd.streaming import RoundTripStream
    W, H, batch = 64, 64, 4
    codec = dv.Codec(0)
    rts = RoundTripStream(codec, W, H, batch, check=True)
    for k in range(2):
        codec.synth_frames("smooth", SEED, 10 * k, batch, W, H, out=rts.inp[k])
    codec.sync()
    r = rts.run(0, 5 * batch)
    assert r["batches"] == 5 and rts.mismatches == 0
    codec.close()
