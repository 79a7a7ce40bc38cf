"""CPU: the host logic of the native gather (dbde_hip_gather_plan, csrc/dbde_gather.cpp).  Both ends of every
transfer derive their send / receive lists from the same all-gathered byte counts; RCCL hangs if they disagree, so
the plan is exposed through the C-ABI and checked here by playing every rank of worlds of 1..8 against the root:
each sender's messages must meet, in order and byte for byte, the root's receives from that peer, and the pieces
must tile the gathered stream (ranks' segments in rank order, README.md:12-23) exactly once."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_every_send_meets_its_receive_and_the_window_is_tiled():
    import dbde_video_cpp_amd as dv
    rng = np.random.default_rng(11)
    for world in range(1, 9):
        for root in sorted({0, world - 1, world // 2}):
            for piece in (0, 1 << 20, 4096, 7):
                sizes = [int(x) for x in rng.integers(0, 3 * max(piece, 1000), world)]
                sizes[int(rng.integers(world))] = 0                      # a rank with nothing to send
                if piece:
                    sizes[int(rng.integers(world))] = 5 * piece           # exact multiple of the piece size
                plans = [dv.gather_plan(world, r, root, sizes, piece) for r in range(world)]
                assert all(t == sum(sizes) for _, t in plans)
                root_ops, total = plans[root]
                disp = np.concatenate([[0], np.cumsum(sizes)])
                covered = np.zeros(total, np.uint8)
                for r in range(world):
                    ops = plans[r][0]
                    if r == root:
                        own = [o for o in ops if o[1] == dv.GATHER_OWN]
                        assert len(own) == (1 if sizes[r] else 0)
                        for peer, kind, so, wo, b in own:
                            assert (peer, so, wo, b) == (root, 0, disp[r], sizes[r])
                            covered[wo:wo + b] += 1
                        continue
                    sends = ops
                    assert all(k == dv.GATHER_SEND and p == root for p, k, _, _, _ in sends)
                    recvs = [o for o in root_ops if o[1] == dv.GATHER_RECV and o[0] == r]
                    assert len(sends) == len(recvs)
                    at = 0
                    for (_, _, so, wo, b), (_, _, _, rwo, rb) in zip(sends, recvs):
                        assert b == rb and wo == rwo and so == at and wo == disp[r] + at
                        assert 0 < b <= (piece or 1 << 30)
                        covered[wo:wo + b] += 1
                        at += b
                    assert at == sizes[r]
                assert (covered == 1).all()
                # a non-root rank never sees a receive; the root never sends
                assert all(k != dv.GATHER_SEND for _, k, _, _, _ in root_ops)


def test_plan_rejects_bad_arguments():
    import ctypes as C
    import dbde_video_cpp_amd as dv
    L = dv.lib()
    one = (C.c_uint64 * 1)(5)
    assert L.dbde_hip_gather_plan(0, 0, 0, one, 0, None, 0, None) < 0
    assert L.dbde_hip_gather_plan(1, 1, 0, one, 0, None, 0, None) < 0
    assert L.dbde_hip_gather_plan(1, 0, 2, one, 0, None, 0, None) < 0
    assert L.dbde_hip_gather_plan(1, 0, 0, None, 0, None, 0, None) < 0
    assert L.dbde_hip_gather_plan(1, 0, 0, one, 0, None, 0, None) == 1
