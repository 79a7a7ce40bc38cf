"""Streaming driver for long frame sequences (SURVEY.md 7 "HBM capacity at config 5", 8e).

BASELINE configs[4] is a 10,000-frame 4096x3072 stream: 126 GB of pixels and up to 130 GB of DBDE bytes.
It is never all resident: a rank walks ITS contiguous block of frames (distributed.shard_frames) in
batches through two input slots and two stream slots, and three things overlap on separate HIP streams:

    source stream : frames of batch k+2 are produced into the input slot batch k has just left
    codec stream  : encode batch k+1 (-> concatenated .dbde body) and decode it back (the round trip)
    comm stream   : the compressed bytes of batch k travel to the root (RCCL send/recv over xGMI) into one of two
                    root windows -- gather="native": the C-ABI's dbde_hip_gather_* (csrc/dbde_gather.cpp: byte counts
                    exchanged device to device, rank 0 encodes straight into its window, no host stall on the codec
                    stream); gather="nccl": the same exchange through torch.distributed
                    (distributed.gather_stream_begin/_end)

The host never waits on the codec stream: the byte count of batch k (needed to post matching send/recv
pieces) is read through a side stream that waits only on batch k's encode event, while batch k+1 is already
queued.  There is no collective on the encode/decode path itself; frames are independent
(dbde_util.cpp:137-180, 291-328 keep no state between frames).
"""
import time

import torch

from . import distributed as dd


def drive_rounds(ops, rounds, nb, count, gather, native):
    """The control flow of RoundTripStream.run, separated from the GPU so that it can be tested without one
    (tests/test_streaming_order.py): `ops` supplies op_encode / op_begin / op_decode / op_produce / op_post.
    Every rank walks the same `rounds` iterations, so that the collective calls (size exchange of round k, then the
    transfers of round k - 1) are issued in the same order everywhere; a rank whose block has fewer batches (`nb`) takes
    part in the remaining rounds with nothing to send (count(k) == 0: no encode, but the same begin and post)."""
    for k in range(rounds):
        slot, n = k % 2, count(k)
        if n:
            ops.op_encode(k, slot, n)        # (waits, on the codec's stream, for the transfer that last read this slot)
        if native:
            ops.op_begin(slot, n)
        if n and ops.decode:
            ops.op_decode(slot, n)
        if k + 2 < nb:
            ops.op_produce(k + 2, slot)
        if gather and k >= 1:                # one round behind: the host wait for the counts ends early
            ops.op_post((k - 1) % 2, count(k - 1))
    if gather and rounds:
        ops.op_post((rounds - 1) % 2, count(rounds - 1))


class RoundTripStream:
    """Encode+decode `n_frames` frames [first_frame, first_frame + n_frames) in batches of `batch`.

    codec   : dbde_video_cpp_amd.Codec bound to the stream the codec kernels run on
    source  : callable(first_frame, n, out_tensor) that ENQUEUES the production of n frames on
              `source_stream` (e.g. synth_frames of a second Codec created on that stream), or None to re-use
              what is in the input slots (caller filled them: a ring of resident frames)
    gather  : None | "native" | "nccl" | "host": send every batch's compressed bytes to rank 0 ("native": the
              C-ABI gather, `native` = a dbde_video_cpp_amd.Gather; "host": staged through pinned memory, for gloo
              rehearsals)
    """

    def __init__(self, codec, W, H, batch, source=None, source_stream=None, gather=None, decode=True, check=False,
                 native=None, world=1, rank=0, loopback=False, tap=None):
        self.codec, self.W, self.H, self.batch = codec, W, H, batch
        self.source, self.gather, self.decode, self.check = source, gather, decode, check
        dev = codec.device
        self.dev = dev
        self.s_codec = codec.stream
        self.s_src = source_stream if source_stream is not None else torch.cuda.Stream(dev)
        self.s_comm = torch.cuda.Stream(dev)
        self.s_copy = torch.cuda.Stream(dev)
        self.inp = [torch.empty((batch, H, W), dtype=torch.uint8, device=dev) for _ in range(2)]
        self.out = [codec.alloc_stream(W, H, batch) for _ in range(2)]            # (buf, lead, cap)
        self.img = torch.empty((batch, H, W), dtype=torch.uint8, device=dev) if decode else None
        self.offs = [torch.zeros(batch, dtype=torch.int64, device=dev) for _ in range(2)]
        self.sizes = [torch.zeros(batch, dtype=torch.int64, device=dev) for _ in range(2)]
        self.res = torch.empty((batch, 4), dtype=torch.int64, device=dev)
        self.total_pinned = torch.empty(2, dtype=torch.int64).pin_memory()
        self.ev_src = [torch.cuda.Event() for _ in range(2)]
        self.ev_enc = [torch.cuda.Event() for _ in range(2)]
        self.ev_gath = [torch.cuda.Event() for _ in range(2)]
        self.ev_copy = torch.cuda.Event()
        self.window = None
        self.mismatches = 0
        self.native, self.loopback = native, loopback
        self.tap = tap      # tests: callable(k, slot, n), called on the codec stream right after batch k's encode is enqueued
        if gather == "native":
            assert native is not None, 'gather="native" needs a Gather'
        if gather in ("native", "nccl") and rank == 0:
            # the root's two windows: every rank's worst case; rank 0's own segment is ENCODED here (its displacement
            # in the gathered stream is 0), so the root never copies its own bytes
            lead, cap = self.out[0][1], self.out[0][2]
            self.window = [torch.empty(lead + world * cap + 64, dtype=torch.uint8, device=dev) for _ in range(2)]
            if not loopback:
                self.out = [(self.window[k], lead, cap) for k in range(2)]
            if native is not None:
                native.set_window(world * cap)     # travels with every size exchange: an overflow is one verdict on all ranks

    def _produce(self, slot, first, n):
        if self.source is None:
            return
        with torch.cuda.stream(self.s_src):
            self.s_src.wait_event(self.ev_enc[slot])      # the encode that read this slot has finished
            self.source(first, n, self.inp[slot][:n])
            self.ev_src[slot].record(self.s_src)

    def _batch_bytes(self, slot, n):
        """Compressed bytes of the batch in `slot`, read without touching the codec stream."""
        if n == 0:
            return 0
        with torch.cuda.stream(self.s_copy):
            self.s_copy.wait_event(self.ev_enc[slot])
            self.total_pinned[slot:slot + 1].copy_((self.offs[slot][n - 1:n] + self.sizes[slot][n - 1:n]), non_blocking=True)
            self.ev_copy.record(self.s_copy)
        self.ev_copy.synchronize()
        return int(self.total_pinned[slot].item())

    def _post_gather(self, slot, n, world, rank):
        nbytes = self._batch_bytes(slot, n)
        buf, lead, cap = self.out[slot]
        seg = buf[lead:lead + cap]
        if self.gather == "host":      # gloo rehearsal: CPU tensors, blocking
            host = seg[:nbytes].cpu()
            _, sizes = dd.gather_stream(host, nbytes, dst=0)
            self.ev_gath[slot].record(self.s_comm)
            return nbytes, sum(sizes)
        with torch.cuda.stream(self.s_comm):
            self.s_comm.wait_event(self.ev_enc[slot])
            _, sizes, works = dd.gather_stream_begin(seg, nbytes, dst=0, out=self.window[slot][lead:] if rank == 0 else None)
            dd.gather_stream_end(works)
            self.ev_gath[slot].record(self.s_comm)
        return nbytes, sum(sizes)

    def _native_begin(self, slot, n):
        last = slice(n - 1, n)
        self.native.begin(slot, self.offs[slot][last] if n else None, self.sizes[slot][last] if n else None)

    def _native_post(self, slot, world):
        buf, lead, cap = self.out[slot]
        win = self.window[slot] if self.window is not None else None
        sizes = self.native.post(slot, buf, lead, win, lead, world * cap, loopback=self.loopback)
        return sizes[self.native.rank], sum(sizes)

    # ---- the operations of one round (drive_rounds calls them in the order that keeps the collectives in step) ----
    def op_join(self, slot):
        if self.gather == "native":
            self.native.join(slot)                           # the transfer that read this slot is done
        else:
            self.s_codec.wait_event(self.ev_gath[slot])

    def op_encode(self, k, slot, n):
        first_frame = self._geo[0]
        buf, lead, cap = self.out[slot]
        with torch.cuda.stream(self.s_codec):
            if self.source is not None:
                self.s_codec.wait_event(self.ev_src[slot])
            self.op_join(slot)
            self.codec.encode_frames(self.inp[slot], self.W, self.H, n, buf, lead, cap, first_index=first_frame + k * self.batch,
                                     offsets=self.offs[slot], nbytes=self.sizes[slot])
            self.ev_enc[slot].record(self.s_codec)
            if self.tap is not None:
                self.tap(k, slot, n)

    def op_begin(self, slot, n):
        with torch.cuda.stream(self.s_codec):
            self._native_begin(slot, n)      # size exchange enqueued behind the encode; returns at once

    def op_decode(self, slot, n):
        buf, lead, cap = self.out[slot]
        with torch.cuda.stream(self.s_codec):
            self.codec.decode_frames(buf, lead, cap, self.offs[slot], self.W, self.H, n, images=self.img, results=self.res)
            if self.check:
                self.mismatches += int((self.img[:n] != self.inp[slot][:n]).any().item())

    def op_produce(self, k, slot):
        first_frame, _, _, _, count = self._geo
        self._produce(slot, first_frame + k * self.batch, count(k))

    def op_post(self, slot, n):
        _, world, rank, _, _ = self._geo
        a, b = self._native_post(slot, world) if self.gather == "native" else self._post_gather(slot, n, world, rank)
        self._acc[0] += a
        self._acc[1] += b

    def run(self, first_frame, n_frames, world=1, rank=0, rounds=None):
        """Returns a dict: frames, seconds (host wall, everything drained), packed bytes of this rank,
        gathered bytes (root's view of every rank) when gathering.  `rounds`: gather rounds every rank takes
        part in (the batch count of the LARGEST rank block; a rank with fewer batches posts empty ones)."""
        codec, W, H, B = self.codec, self.W, self.H, self.batch
        nb = (n_frames + B - 1) // B
        native = self.gather == "native"
        rounds = max(rounds or nb, nb) if self.gather else nb
        count = lambda k: max(0, min(B, n_frames - k * B))
        for ev in self.ev_enc + self.ev_gath:
            ev.record(self.s_codec)
        for k in range(min(2, nb)):
            self._produce(k, first_frame + k * B, count(k))
        self._acc = [0, 0]
        self._geo = (first_frame, world, rank, nb, count)
        torch.cuda.synchronize(self.dev)      # the first two batches are resident when the clock starts
        t0 = time.perf_counter()
        drive_rounds(self, rounds, nb, count, bool(self.gather), native)
        packed, gathered = self._acc
        if native:
            for slot in range(2):
                self.native.sync(slot)
        torch.cuda.synchronize(self.dev)
        dt = time.perf_counter() - t0
        codec.sync()
        return {"frames": n_frames, "batches": nb, "seconds": dt, "packed_bytes": packed, "gathered_bytes": gathered}
