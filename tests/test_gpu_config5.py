"""GPU: BASELINE configs[4] -- the 4096x3072 frame stream walked in batches by the streaming driver, with the gather
of every batch's compressed bytes to the root -- on ONE rank (what a one-GPU box can run): the native RCCL gather of
the C-ABI (dbde_hip_gather_*: communicator from a unique id, ncclAllGather of the byte counts, root window; and, in
loopback, the ncclSend / ncclRecv path moving the root's bytes) and the torch.distributed form under a one-rank `nccl`
process group.  Checked: round-trip identity of every batch (device compare), exact sizes, the root window's bytes ==
the rank's stream, and frames byte for byte against reference-made SHA-256 (tests/golden, frames 0 and 3) and the oracle.
The gathered stream must simply be the frames one after another (reference README.md:12-23; per frame
dbde_util.cpp:190-196).  Rank-to-rank traffic needs a multi-GPU box: tests/test_gather_plan.py plays the plan of
worlds 1..8 on the CPU, tests/test_distributed_gloo.py runs the driver's loop at world size 2."""
import hashlib
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0xDBDE2016
W, H = 4096, 3072
BATCH, N_FRAMES = 64, 200            # 4 batches (64, 64, 64, 8); one input slot = 805 MB > the 256 MiB Infinity Cache


@pytest.fixture(scope="module")
def dv():
    import dbde_video_cpp_amd as dv
    dv.build()
    return dv


@pytest.fixture(scope="module")
def nccl_group():
    """A one-rank RCCL process group in this process (the N > 1 path's init, barrier and size all-gather)."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        yield dist
        return
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def _run(dv, content, gather, loopback=False, native=None):
    import torch
    from dbde_video_cpp_amd.streaming import RoundTripStream
    codec = dv.Codec(0)
    side = torch.cuda.Stream()
    src = dv.Codec(0, stream=side)
    kept = {}

    def source(first, k, out):
        src.synth_frames(content, SEED, first, k, W, H, out=out)

    def tap(k, slot, n):          # on the codec stream, behind batch k's encode: keep what the checks below need
        buf, lead, cap = rts.out[slot]
        o, s = rts.offs[slot][:n].clone(), rts.sizes[slot][:n].clone()
        kept[k] = (o, s, buf[lead:lead + cap].clone() if k in (0, 3) else None)

    g = dv.Gather(codec, dv.gather_unique_id(), 1, 0, max_message_bytes=200 << 20) if native else None
    rts = RoundTripStream(codec, W, H, BATCH, source=source, source_stream=side, gather=gather, check=True,
                          native=g, world=1, rank=0, loopback=loopback, tap=tap)
    r = rts.run(0, N_FRAMES, world=1, rank=0)
    codec.sync()
    return codec, src, g, rts, r, kept


def _check(dv, oracle, golden, content, rts, r, kept):
    import torch
    manifest, _ = golden
    assert r["frames"] == N_FRAMES and r["batches"] == 4 and rts.mismatches == 0
    # exact sizes: frames concatenated inside a batch, byte counts add up to what the gather reported
    total = 0
    for k in range(4):
        o, s, _ = kept[k]
        o, s = o.cpu().numpy(), s.cpu().numpy()
        assert o[0] == 0 and (o[1:] == (o + s)[:-1]).all()
        total += int(o[-1] + s[-1])
    assert r["packed_bytes"] == total == r["gathered_bytes"]
    # frames 0 and 3: SHA-256 of the reference's own output (tests/golden/make_golden.py)
    o, s, seg = kept[0]
    for e in [e for e in manifest["big"] if e["name"] == "cfg2_4096x3072" and e["mode"] == content]:
        f = e["frame"]
        got = seg[int(o[f]):int(o[f] + s[f])].cpu().numpy()
        assert len(got) == e["packed_bytes"] and hashlib.sha256(got.tobytes()).hexdigest() == e["packed_sha"], (content, f)
    # the very last frame of the stream, byte for byte the oracle's
    o, s, seg = kept[3]
    last = N_FRAMES - 1
    want = oracle.pack_frame(last, oracle.synth_frame(dv.MODES[content], SEED, last, W, H), W, H)
    got = seg[int(o[-1]):int(o[-1] + s[-1])].cpu().numpy()
    assert got.tobytes() == want.tobytes()
    # the root window of the last two batches holds exactly the rank's stream
    for k in (2, 3):
        o, s, _ = kept[k]
        n = int((o[-1] + s[-1]).item())
        buf, lead, cap = rts.out[k % 2]
        assert torch.equal(rts.window[k % 2][lead:lead + n], buf[lead:lead + n])
    if kept[3][2] is not None:
        n = int((kept[3][0][-1] + kept[3][1][-1]).item())
        assert torch.equal(rts.window[1][rts.out[1][1]:rts.out[1][1] + n], kept[3][2][:n])


@pytest.mark.parametrize("content,loopback", [("noise8", False), ("mixed", True)])
def test_config5_stream_native_rccl_gather(dv, oracle, golden, content, loopback):
    """dbde_hip_gather_*: in place (rank 0 encodes into its window, nothing is copied) and in loopback (the bytes
    travel through ncclSend / ncclRecv in 200 MB pieces from a separate segment buffer into the window)."""
    assert dv.lib().dbde_hip_gather_rccl_version() >= 22000
    codec, src, g, rts, r, kept = _run(dv, content, "native", loopback=loopback, native=True)
    if loopback:
        assert rts.out[0][0].data_ptr() != rts.window[0].data_ptr()
    else:
        assert rts.out[0][0].data_ptr() == rts.window[0].data_ptr()      # no self-copy: the segment IS the window
    _check(dv, oracle, golden, content, rts, r, kept)
    g.close()
    src.close()
    codec.close()


def test_config5_stream_torch_nccl_gather(dv, oracle, golden, nccl_group):
    """The same stream with the exchange through torch.distributed on a one-rank RCCL group (init, size all-gather,
    root window; rank 0's segment is produced in place)."""
    import torch
    assert nccl_group.get_backend() == "nccl" and nccl_group.get_world_size() == 1
    t = torch.ones(1, device="cuda")
    nccl_group.all_reduce(t)
    codec, src, g, rts, r, kept = _run(dv, "noise8", "nccl")
    _check(dv, oracle, golden, "noise8", rts, r, kept)
    src.close()
    codec.close()
