"""Frame-level data parallelism for the DBDE path.

Frames are independent (no inter-frame state in dbde_pack_image / dbde_unpack_image), so the
path shards by CONTIGUOUS FRAME BLOCKS: rank g of G owns frames [g*N/G, (g+1)*N/G).  Each
rank's compressed output is then one contiguous, in-order segment of the final stream and
there is no collective on the encode/decode data path.  The only exchange step is the
variable-length gather of the compressed byte stream to a root (RCCL has no gatherv):
  1. all_gather of the per-rank byte counts,
  2. exclusive scan -> displacement of each rank's segment,
  3. one grouped isend / irecv per peer straight into the root's buffer at its displacement.
Works on any torch.distributed backend: "nccl" (= RCCL over xGMI) on device tensors,
"gloo" on CPU tensors (tests/test_distributed_gloo.py).
"""
import torch
import torch.distributed as dist


def shard_frames(n_frames, rank, world):
    """Contiguous block of frame indices [lo, hi) owned by `rank`."""
    lo = (n_frames * rank) // world
    hi = (n_frames * (rank + 1)) // world
    return lo, hi


MAX_MESSAGE_BYTES = 1 << 30   # a rank's segment travels in pieces of at most this many bytes


def _pieces(nbytes, piece):
    return [(at, min(piece, nbytes - at)) for at in range(0, nbytes, piece)]


def gather_stream_begin(segment, nbytes, dst=0, group=None, out=None, max_message_bytes=None):
    """Posts the variable-length gather of `segment[:nbytes]` (uint8, 1-D) to rank `dst` and returns
    without waiting for the bytes: (stream_view_or_None, sizes, works).  The size exchange (a few bytes,
    all_gather) is synchronous -- both ends need the sizes to post matching pieces -- the payload is not:
    call gather_stream_end(works) before `segment` is overwritten or `out` is read.  This is what lets the
    gather of batch k run beside the encode of batch k+1 (streaming.RoundTripStream)."""
    piece = int(max_message_bytes or MAX_MESSAGE_BYTES)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = segment.device
    mine = torch.tensor([int(nbytes)], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, mine, group=group)
    sizes = [int(s.item()) for s in all_sizes]
    if rank == dst:
        total = sum(sizes)
        if out is None:
            out = torch.empty(total, dtype=torch.uint8, device=dev)
        assert out.numel() >= total
        at, ops = 0, []
        for r in range(world):
            if r == rank:
                if out.data_ptr() + at != segment.data_ptr():    # (a root that produced its segment in place has nothing to move)
                    out[at:at + sizes[r]].copy_(segment[:sizes[r]], non_blocking=True)
            else:
                for o, n in _pieces(sizes[r], piece):
                    ops.append(dist.P2POp(dist.irecv, out[at + o:at + o + n], r, group))
            at += sizes[r]
        return out[:total], sizes, (dist.batch_isend_irecv(ops) if ops else [])
    ops = [dist.P2POp(dist.isend, segment[o:o + n], dst, group) for o, n in _pieces(sizes[rank], piece)]
    return None, sizes, (dist.batch_isend_irecv(ops) if ops else [])


def gather_stream_end(works):
    """Waits for the transfers posted by gather_stream_begin (on nccl: makes the current stream wait)."""
    for w in works:
        w.wait()


def gather_stream(segment, nbytes, dst=0, group=None, out=None, max_message_bytes=None):
    """Gathers every rank's first `nbytes` bytes of `segment` (uint8, 1-D) to rank `dst`, in rank
    order.  Returns (stream, sizes) on dst -- `stream` holds sum(sizes) bytes -- and (None, sizes)
    elsewhere.  `out` may supply the destination buffer on dst (>= sum(sizes) bytes).  Segments
    of many GB (1024 frames of 4096x3072 are 13 GB) are split into <= max_message_bytes pieces,
    all posted in one group; both ends derive the same piece boundaries from the gathered sizes."""
    stream, sizes, works = gather_stream_begin(segment, nbytes, dst, group, out, max_message_bytes)
    gather_stream_end(works)
    return stream, sizes


def scatter_stream(stream, frame_bytes, src=0, group=None, device=None, max_message_bytes=None):
    """The decode-side counterpart of gather_stream: rank `src` holds a concatenated frame
    sequence (`stream`, uint8 1-D, no video header) and the byte length of every frame
    (`frame_bytes`, list of ints); every rank receives the bytes of ITS contiguous frame block
    (shard_frames) and the frame offsets inside that segment.  Returns (segment, offsets, (lo, hi)) on
    every rank -- `segment` is a uint8 tensor on `device` (default: the stream's device on src, cpu
    elsewhere), `offsets` a list of ints relative to the segment, [lo, hi) the global frame numbers.
    Other ranks pass stream=None, frame_bytes=None."""
    piece = int(max_message_bytes or MAX_MESSAGE_BYTES)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    meta = [list(map(int, frame_bytes))] if rank == src else [None]
    dist.broadcast_object_list(meta, src=src, group=group)      # frame lengths: a few bytes per frame
    sizes = meta[0]
    n = len(sizes)
    starts = [0] * (n + 1)
    for i, b in enumerate(sizes):
        starts[i + 1] = starts[i] + b
    lo, hi = shard_frames(n, rank, world)
    my_bytes = starts[hi] - starts[lo]
    offsets = [starts[f] - starts[lo] for f in range(lo, hi)]
    if rank == src:
        dev = stream.device if device is None else device
        ops = []
        for r in range(world):
            rlo, rhi = shard_frames(n, r, world)
            if r == rank:
                continue
            a, nb = starts[rlo], starts[rhi] - starts[rlo]
            for o, m in _pieces(nb, piece):
                ops.append(dist.P2POp(dist.isend, stream[a + o:a + o + m], r, group))
        seg = stream[starts[lo]:starts[hi]].to(dev)
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return seg, offsets, (lo, hi)
    dev = torch.device("cpu") if device is None else device
    seg = torch.empty(my_bytes, dtype=torch.uint8, device=dev)
    ops = [dist.P2POp(dist.irecv, seg[o:o + m], src, group) for o, m in _pieces(my_bytes, piece)]
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    return seg, offsets, (lo, hi)


def frame_offsets_from_sizes(per_rank_frame_bytes):
    """Root-side frame index of the gathered stream: list (per rank) of per-frame byte counts ->
    flat list of frame offsets relative to the first frame (exclusive scan)."""
    offs, at = [], 0
    for sizes in per_rank_frame_bytes:
        for s in sizes:
            offs.append(at)
            at += int(s)
    return offs, at
