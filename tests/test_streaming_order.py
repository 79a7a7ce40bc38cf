"""CPU: the control flow of the streaming driver (dbde_video_cpp_amd.streaming.drive_rounds, what RoundTripStream.run
executes) with a recording stand-in for the GPU operations.  The native gather's begin / post are COLLECTIVE (an
all-gather of the counts, then grouped sends / receives): every rank must issue them in the same order whatever the
size of its frame block, or RCCL pairs the wrong calls and hangs.  Ranks whose blocks have fewer batches take part in
the remaining rounds with nothing to send."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Recorder:
    def __init__(self, decode=True):
        self.decode = decode
        self.calls = []

    def op_encode(self, k, slot, n):
        self.calls.append(("join", slot))          # the codec's stream waits for the transfer that last read the slot ...
        self.calls.append(("encode", slot, k, n))  # ... before the slot is encoded into again

    def op_begin(self, slot, n):
        self.calls.append(("begin", slot, n))

    def op_decode(self, slot, n):
        self.calls.append(("decode", slot, n))

    def op_produce(self, k, slot):
        self.calls.append(("produce", slot, k))

    def op_post(self, slot, n):
        self.calls.append(("post", slot, n))


def run_rank(n_frames, batch, rounds, native=True):
    import torch  # noqa: F401  (the module imports it)
    from dbde_video_cpp_amd.streaming import drive_rounds
    nb = (n_frames + batch - 1) // batch
    count = lambda k: max(0, min(batch, n_frames - k * batch))
    rec = Recorder()
    drive_rounds(rec, max(rounds, nb), nb, count, True, native)
    return rec.calls


def test_collective_calls_are_in_the_same_order_on_every_rank():
    batch = 4
    for blocks in ([20, 12], [17, 17], [4, 0], [1, 9, 5], [8, 8, 8, 3]):
        rounds = max((b + batch - 1) // batch for b in blocks)
        per_rank = [run_rank(b, batch, rounds) for b in blocks]
        collective = [[(c[0], c[1]) for c in calls if c[0] in ("begin", "post")] for calls in per_rank]
        assert all(c == collective[0] for c in collective), blocks
        assert len(collective[0]) == 2 * rounds                     # one size exchange and one transfer per round
        for calls, b in zip(per_rank, blocks):
            nb = (b + batch - 1) // batch
            assert [c[3] for c in calls if c[0] == "encode"] == [min(batch, b - k * batch) for k in range(nb)]
            # rounds past the block: a begin and a post with nothing to send, no encode
            assert [c[2] for c in calls if c[0] == "begin"] == [max(0, min(batch, b - k * batch)) for k in range(rounds)]


def test_slot_reuse_is_ordered():
    """Per slot: begin -> post -> (join -> encode -> begin) ...: a slot's transfer is posted before its next size exchange,
    and the codec waits for that transfer (join) before it encodes into the slot again."""
    for (n_frames, batch, rounds) in ((40, 4, 10), (9, 4, 5), (4, 4, 3)):
        calls = run_rank(n_frames, batch, rounds)
        for slot in (0, 1):
            seq = [c[0] for c in calls if c[1] == slot and c[0] in ("begin", "post", "join", "encode")]
            state = "idle"                       # idle -> (join, encode)? -> begin -> post -> idle
            for op in seq:
                if op == "join":
                    assert state == "idle", (slot, seq)
                    state = "joined"
                elif op == "encode":
                    assert state == "joined", (slot, seq)
                    state = "encoded"
                elif op == "begin":
                    assert state in ("encoded", "idle"), (slot, seq)      # (idle: an empty round)
                    state = "begun"
                elif op == "post":
                    assert state == "begun", (slot, seq)
                    state = "idle"
            assert state == "idle"
        # the transfer of round k is posted one round behind: after round k + 1's encode has been enqueued
        order = [(c[0], c[1]) for c in calls if c[0] in ("encode", "post")]
        first_post = order.index(("post", 0))
        assert order[:first_post].count(("encode", 0)) + order[:first_post].count(("encode", 1)) >= min(2, (n_frames + batch - 1) // batch)
