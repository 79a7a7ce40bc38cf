"""CPU, world_size 2, gloo: the N > 1 path -- frame-block sharding and the variable-length
gather of the compressed stream -- with the oracle standing in for the per-rank encoder
(the GPU encoder has no CPU form; what is tested here is the host logic around it)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
W, H, N = 37, 21, 11      # odd sizes, uneven split over 2 ranks
SEED = 0xDBDE2016


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    for p in (HERE, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dbde_video_cpp_amd as dv
    from dbde_video_cpp_amd import distributed as dd
    from oracle_ffi import Oracle
    ora = Oracle()
    lo, hi = dd.shard_frames(N, rank, world)
    frames = [ora.pack_frame(f, ora.synth_frame(1, SEED, f, W, H), W, H) for f in range(lo, hi)]
    seg = torch.from_numpy(np.concatenate(frames + [np.zeros(16, np.uint8)]))
    nbytes = sum(len(f) for f in frames)
    stream, sizes = dd.gather_stream(seg, nbytes, dst=0)
    # the same gather with each segment split into many small messages
    stream2, sizes2 = dd.gather_stream(seg, nbytes, dst=0, max_message_bytes=97)
    assert sizes2 == sizes
    if rank == 0:
        assert sum(sizes) == stream.numel() and torch.equal(stream, stream2)
        # the gathered stream is exactly what one rank would have produced for all N frames
        want = np.concatenate([ora.pack_frame(f, ora.synth_frame(1, SEED, f, W, H), W, H) for f in range(N)])
        ok = stream.numpy().tobytes() == want.tobytes()
        # and a full file = video header + stream parses frame by frame
        body = stream.numpy()
        at, n = 0, 0
        while at < len(body):
            adv, fh, img = ora.unpack_frame(body[at:], W, H)
            ok = ok and fh == (2, n, 0) and (img == ora.synth_frame(1, SEED, n, W, H)).all()
            at += adv
            n += 1
        q.put(bool(ok and n == N))
    # the pipelined form (what the streaming driver does per batch): the gather of batch k is posted and only
    # waited for after batch k+1 has been posted, into two alternating root windows
    batches = [frames[i:i + 2] for i in range(0, len(frames), 2)]
    windows = [torch.empty(4096, dtype=torch.uint8) for _ in range(2)] if rank == 0 else [None, None]
    pending, got_batches, okp = None, [], True
    for kb, bf in enumerate(batches + [None]):
        if bf is not None:
            segb = torch.from_numpy(np.concatenate(bf + [np.zeros(8, np.uint8)]))
            nb = sum(len(f) for f in bf)
            view, szs, works = dd.gather_stream_begin(segb, nb, dst=0, out=windows[kb % 2], max_message_bytes=113)
            posted = (view, szs, works, segb)
        if pending is not None:
            dd.gather_stream_end(pending[2])
            if rank == 0:
                got_batches.append((pending[0].clone(), pending[1]))
        pending = posted if bf is not None else None
    if rank == 0:
        # batch k of the gathered result = rank 0's batch k followed by rank 1's batch k
        for kb, (view, szs) in enumerate(got_batches):
            mine = np.concatenate(batches[kb]) if kb < len(batches) else np.zeros(0, np.uint8)
            okp = okp and view[:szs[0]].numpy().tobytes() == mine.tobytes() and sum(szs) == view.numel()
        q.put(("pipelined", bool(okp and len(got_batches) == len(batches))))
    # decode side: the root scatters frame blocks of the gathered stream back out; every rank
    # decodes its own block (the oracle standing in for the GPU decoder) and finds its frames
    if rank == 0:
        all_sizes = [len(ora.pack_frame(f, ora.synth_frame(1, SEED, f, W, H), W, H)) for f in range(N)]
        seg2, offs2, (lo2, hi2) = dd.scatter_stream(stream, all_sizes, src=0, max_message_bytes=61)
    else:
        seg2, offs2, (lo2, hi2) = dd.scatter_stream(None, None, src=0, max_message_bytes=61)
    assert (lo2, hi2) == (lo, hi) and len(offs2) == hi - lo
    body2 = seg2.numpy()
    ok2 = True
    for k, f in enumerate(range(lo2, hi2)):
        adv, fh, img = ora.unpack_frame(body2[offs2[k]:], W, H)
        ok2 = ok2 and fh == (2, f, 0) and (img == ora.synth_frame(1, SEED, f, W, H)).all()
    q.put(("scatter", rank, bool(ok2)))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_frames_partition():
    sys.path.insert(0, ROOT)
    from dbde_video_cpp_amd import distributed as dd
    for n in (0, 1, 7, 64, 10000):
        for world in (1, 2, 3, 8):
            blocks = [dd.shard_frames(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
            assert max(b[1] - b[0] for b in blocks) - min(b[1] - b[0] for b in blocks) <= 1
    offs, total = dd.frame_offsets_from_sizes([[10, 20], [5], [], [7, 7]])
    assert offs == [0, 10, 30, 35, 42] and total == 49


def test_gather_and_scatter_stream_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = [q.get(timeout=5) for _ in range(4)]
    assert True in got                                     # rank 0: the gathered stream is the full stream
    assert ("pipelined", True) in got                      # rank 0: overlapped per-batch gathers, two windows
    assert sorted(g for g in got if isinstance(g, tuple) and g[0] == "scatter") == [("scatter", 0, True), ("scatter", 1, True)]
