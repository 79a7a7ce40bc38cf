"""CPU: the host logic of the native decode-side shard (dbde_hip_scatter_blocks / _check / _plan, csrc/dbde_scatter.cpp)
and the symmetric capacity verdict of the gather (dbde_hip_gather_check).  Every rank derives its send / receive list
from the same broadcast table; RCCL hangs if two ends disagree, so the arithmetic is exposed through the C-ABI and
checked here by playing every rank of worlds 1..8 against the root: each of the root's sends must meet, in order and
byte for byte, the peer's receive; the blocks must tile the stream (frames following each other, README.md:12-23) and the
frame index exactly once; and "it does not fit" must be ONE verdict, the same on every rank."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def random_index(rng, n):
    sizes = rng.integers(34, 5000, n)
    offs = np.concatenate([[0], np.cumsum(sizes)])[:-1]
    return [int(x) for x in offs], int(sizes.sum())


def test_blocks_tile_the_stream_and_are_the_gathers_frame_blocks():
    import dbde_video_cpp_amd as dv
    from dbde_video_cpp_amd import distributed as dd
    rng = np.random.default_rng(5)
    for world in range(1, 9):
        for n in (0, 1, world - 1, world, 3 * world + 1, 1000):
            offs, total = random_index(rng, n)
            blocks = dv.scatter_blocks(world, offs, total)
            at_f, at_b = 0, 0
            for r, (f0, nf, b0, nb) in enumerate(blocks):
                lo, hi = dd.shard_frames(n, r, world)           # the encode side's contiguous frame blocks
                assert (f0, nf) == (lo, hi - lo) and f0 == at_f
                assert b0 == (offs[lo] if lo < n else total) and b0 == (at_b if nf else b0)
                assert nb == (offs[hi] if hi < n else total) - b0
                at_f += nf
                at_b = b0 + nb if nf else at_b
            assert at_f == n and sum(b[3] for b in blocks) == (total if n else 0)


def test_every_send_meets_its_receive():
    import dbde_video_cpp_amd as dv
    rng = np.random.default_rng(6)
    for world in range(1, 9):
        for root in sorted({0, world - 1, world // 2}):
            for piece in (0, 1 << 16, 4096, 7):
                offs, total = random_index(rng, int(rng.integers(0, 40)))
                blocks = dv.scatter_blocks(world, offs, total)
                plans = [dv.scatter_plan(world, r, root, blocks, piece) for r in range(world)]
                root_ops = plans[root]
                assert all(k in (dv.SCATTER_SEND_BYTES, dv.SCATTER_SEND_OFFSETS, dv.SCATTER_OWN) for _, k, _, _, _ in root_ops)
                own = [o for o in root_ops if o[1] == dv.SCATTER_OWN]
                assert len(own) == (1 if blocks[root][1] else 0)
                for _, _, so, do, b in own:
                    assert (so, do, b) == (blocks[root][2], 0, blocks[root][3])
                for r in range(world):
                    if r == root:
                        continue
                    sends = [o for o in root_ops if o[0] == r]
                    recvs = plans[r]
                    assert len(sends) == len(recvs) and all(p == root for p, _, _, _, _ in recvs)
                    got_bytes, got_offs = 0, 0
                    for (_, ks, sso, sdo, sb), (_, kr, rso, rdo, rb) in zip(sends, recvs):
                        assert (ks, kr) in ((dv.SCATTER_SEND_BYTES, dv.SCATTER_RECV_BYTES), (dv.SCATTER_SEND_OFFSETS, dv.SCATTER_RECV_OFFSETS))
                        assert (sso, sdo, sb) == (rso, rdo, rb) and 0 < sb <= (piece or 1 << 30)
                        if ks == dv.SCATTER_SEND_BYTES:
                            assert sso == blocks[r][2] + got_bytes and sdo == got_bytes and got_offs == 0   # bytes first, in order
                            got_bytes += sb
                        else:
                            assert sso == 8 * blocks[r][0] + got_offs and sdo == got_offs
                            got_offs += sb
                    assert got_bytes == blocks[r][3] and got_offs == 8 * blocks[r][1]


def test_capacity_verdict_is_the_same_on_every_rank():
    """Scatter: a block that does not fit ITS rank's declared buffers; gather: a total that does not fit the root's window.
    The verdict is a function of the exchanged numbers alone -- no rank argument -- so all post or none does."""
    import dbde_video_cpp_amd as dv
    rng = np.random.default_rng(7)
    for world in range(1, 9):
        offs, total = random_index(rng, 5 * world)
        blocks = dv.scatter_blocks(world, offs, total)
        caps = [(b[3], b[1]) for b in blocks]                      # exactly enough
        assert dv.scatter_check(blocks, caps) == dv.OK
        for r in range(world):
            short = list(caps)
            short[r] = (caps[r][0] - 1, caps[r][1])
            assert dv.scatter_check(blocks, short) == dv.ERR_CAPACITY
            short[r] = (caps[r][0], caps[r][1] - 1)
            assert dv.scatter_check(blocks, short) == dv.ERR_CAPACITY
        sizes = [int(x) for x in rng.integers(0, 10000, world)]
        for root in range(world):
            caps = [0] * world
            caps[root] = sum(sizes)
            assert dv.gather_check(world, root, sizes, caps) == (dv.OK, sum(sizes))
            caps[root] -= 1
            if sum(sizes):
                assert dv.gather_check(world, root, sizes, caps) == (dv.ERR_CAPACITY, sum(sizes))
            # what the OTHER ranks put in their capacity word does not matter: only the root's counts
            caps = [1 << 60] * world
            caps[root] = sum(sizes)
            assert dv.gather_check(world, root, sizes, caps)[0] == dv.OK


def test_plan_rejects_bad_arguments():
    import ctypes as C
    import dbde_video_cpp_amd as dv
    L = dv.lib()
    table = (dv.ScatterBlock * 1)()
    assert L.dbde_hip_scatter_plan(0, 0, 0, table, 0, None, 0) < 0
    assert L.dbde_hip_scatter_plan(2, 2, 0, table, 0, None, 0) < 0
    assert L.dbde_hip_scatter_plan(2, 0, 0, None, 0, None, 0) < 0
    assert L.dbde_hip_scatter_blocks(0, 0, None, 0, table) < 0
    assert L.dbde_hip_scatter_blocks(1, 3, None, 10, table) < 0
