"""dbde_video_cpp_amd -- Python doorway to the MI355X DBDE codec (libdbde_hip.so).

This is plumbing, not the product: every function is a ctypes call into the C-ABI declared
in include/dbde_hip.h.  PyTorch is used only for what it is good at here -- device memory
(`tensor.data_ptr()`), streams and `torch.distributed`.  Method names follow the reference's
dbde_util.h (pack_frame, unpack_frame, pack_8x8, ...), so tests read like the reference's.

There is no CPU fallback.  If the in-tree library is missing, importing raises; if no
gfx950 device is usable, `Codec()` raises.

The directory is called `dbde-video-cpp_amd`; import it as `dbde_video_cpp_amd` (the
repository root carries a one-file loader of that name).
"""
import ctypes as C
import os
import subprocess

import numpy as np

try:  # torch first: its bundled libamdhip64.so.7 must be the one HIP runtime in the process
    import torch
except Exception:  # pragma: no cover - host-only use of the header helpers
    torch = None

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
LIB_PATH = os.path.join(PKG_DIR, "libdbde_hip.so")
SHIM_PATH = os.path.join(PKG_DIR, "libdbde_util_hip.so")

OK, ERR_ARG, ERR_HIP, ERR_CAPACITY, ERR_DEVICE = 0, -1, -2, -3, -4

MODES = {"noise8": 0, "mixed": 1, "flat": 2, "smooth": 3}

# every symbol include/dbde_hip.h declares (tests/test_capi_symbols.py checks the export list)
C_ABI_SYMBOLS = [
    "dbde_hip_create", "dbde_hip_destroy", "dbde_hip_sync", "dbde_hip_last_error", "dbde_hip_device_arch",
    "dbde_hip_max_frame_bytes", "dbde_hip_image_bytes",
    "dbde_hip_encode_frames", "dbde_hip_decode_frames", "dbde_hip_index_stream", "dbde_hip_index_stream_async",
    "dbde_hip_scan_ahead", "dbde_hip_scan_join", "dbde_hip_synth_frames",
    "dbde_hip_pack_8x8", "dbde_hip_pack_8x8_partial", "dbde_hip_pack_image", "dbde_hip_pack_frame",
    "dbde_hip_unpack_8x8", "dbde_hip_unpack_8x8_partial", "dbde_hip_unpack_image", "dbde_hip_unpack_frame",
    "dbde_hip_pack_frame_header", "dbde_hip_pack_video_header",
    "dbde_hip_unpack_frame_header", "dbde_hip_unpack_video_header",
    "dbde_hip_timing_enable", "dbde_hip_timing_read", "dbde_hip_encode_plan", "dbde_hip_decode_plan",
    "dbde_hip_stream_handle", "dbde_hip_device_index",
    "dbde16_hip_max_frame_bytes", "dbde16_hip_encode_frames", "dbde16_hip_decode_frames",
    "dbde_hip_writer_open", "dbde_hip_writer_put", "dbde_hip_writer_error", "dbde_hip_writer_close",
    "dbde_hip_reader_open", "dbde_hip_reader_next", "dbde_hip_reader_close",
    "dbde_hip_gather_unique_id", "dbde_hip_gather_create", "dbde_hip_gather_attach", "dbde_hip_gather_destroy",
    "dbde_hip_gather_error", "dbde_hip_gather_set_max_message", "dbde_hip_gather_begin", "dbde_hip_gather_post",
    "dbde_hip_gather_join", "dbde_hip_gather_sync", "dbde_hip_gather_rccl_version", "dbde_hip_gather_plan",
    "dbde_hip_gather_set_window", "dbde_hip_gather_check",
    "dbde_hip_scatter_create", "dbde_hip_scatter_attach", "dbde_hip_scatter_destroy", "dbde_hip_scatter_error",
    "dbde_hip_scatter_set_max_message", "dbde_hip_scatter_set_capacity", "dbde_hip_scatter_begin", "dbde_hip_scatter_post",
    "dbde_hip_scatter_join", "dbde_hip_scatter_sync", "dbde_hip_scatter_blocks", "dbde_hip_scatter_check",
    "dbde_hip_scatter_plan", "dbde_hip_create_on_own_stream", "dbde_hip_set_host_staging",
]


def build(verbose=False):
    """Compile the HIP library and the dbde_util.h shim in-tree (hipcc --offload-arch=gfx950)."""
    r = subprocess.run(["make", "-C", os.path.join(PKG_DIR, "csrc")], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode:
        raise RuntimeError("building libdbde_hip.so failed")


class FrameHeader(C.Structure):
    _fields_ = [("u64s", C.c_uint32), ("index", C.c_uint64), ("elapsed_ns", C.c_uint64)]


class VideoHeader(C.Structure):
    _fields_ = [("u64s", C.c_uint32), ("height", C.c_uint64), ("width", C.c_uint64), ("frame_hz", C.c_double)]


class FrameResult(C.Structure):
    _fields_ = [("header", FrameHeader), ("consumed", C.c_uint64)]


class ScatterBlock(C.Structure):
    _fields_ = [("first_frame", C.c_uint64), ("n_frames", C.c_uint64), ("byte_start", C.c_uint64), ("byte_count", C.c_uint64)]


class ScatterOp(C.Structure):
    _fields_ = [("peer", C.c_int32), ("kind", C.c_int32), ("source_offset", C.c_uint64), ("dest_offset", C.c_uint64),
                ("bytes", C.c_uint64)]


SCATTER_SEND_BYTES, SCATTER_RECV_BYTES, SCATTER_SEND_OFFSETS, SCATTER_RECV_OFFSETS, SCATTER_OWN = 1, 2, 3, 4, 5
SCATTER_LOOPBACK = 1


class GatherOp(C.Structure):
    _fields_ = [("peer", C.c_int32), ("kind", C.c_int32), ("segment_offset", C.c_uint64),
                ("window_offset", C.c_uint64), ("bytes", C.c_uint64)]


GATHER_SEND, GATHER_RECV, GATHER_OWN = 1, 2, 3
GATHER_LOOPBACK = 1
GATHER_ID_BYTES = 128


u8p = C.POINTER(C.c_uint8)
_lib = None


def lib():
    """The loaded C-ABI library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, i, u64, sz = C.c_void_p, C.c_int, C.c_uint64, C.c_size_t
    L.dbde_hip_create.restype = i
    L.dbde_hip_create.argtypes = [i, vp, C.POINTER(vp)]
    L.dbde_hip_destroy.restype = None
    L.dbde_hip_destroy.argtypes = [vp]
    L.dbde_hip_sync.restype = i
    L.dbde_hip_sync.argtypes = [vp]
    L.dbde_hip_last_error.restype = C.c_char_p
    L.dbde_hip_last_error.argtypes = [vp]
    L.dbde_hip_device_arch.restype = C.c_char_p
    L.dbde_hip_device_arch.argtypes = [vp]
    L.dbde_hip_max_frame_bytes.restype = sz
    L.dbde_hip_max_frame_bytes.argtypes = [i, i]
    L.dbde_hip_image_bytes.restype = sz
    L.dbde_hip_image_bytes.argtypes = [i, i, u64]
    L.dbde_hip_encode_frames.restype = i
    L.dbde_hip_encode_frames.argtypes = [vp, vp, i, i, i, u64, vp, vp, vp, sz, u64, vp, vp]
    L.dbde_hip_decode_frames.restype = i
    L.dbde_hip_decode_frames.argtypes = [vp, vp, sz, vp, i, i, i, vp, vp]
    L.dbde_hip_index_stream.restype = i
    L.dbde_hip_index_stream.argtypes = [vp, vp, sz, i, i, i, vp, C.POINTER(i)]
    L.dbde_hip_index_stream_async.restype = i
    L.dbde_hip_index_stream_async.argtypes = [vp, vp, sz, i, i, i, vp, vp]
    L.dbde_hip_scan_ahead.restype = i
    L.dbde_hip_scan_ahead.argtypes = [vp, vp, sz, i, i, i, vp, vp, vp]
    L.dbde_hip_scan_join.restype = i
    L.dbde_hip_scan_join.argtypes = [vp]
    L.dbde_hip_synth_frames.restype = i
    L.dbde_hip_synth_frames.argtypes = [vp, i, u64, u64, i, i, i, vp]
    L.dbde_hip_pack_8x8.restype = C.c_uint32
    L.dbde_hip_pack_8x8.argtypes = [vp, vp, i, vp]
    L.dbde_hip_pack_8x8_partial.restype = C.c_uint32
    L.dbde_hip_pack_8x8_partial.argtypes = [vp, vp, i, i, i, vp]
    L.dbde_hip_pack_image.restype = sz
    L.dbde_hip_pack_image.argtypes = [vp, vp, i, i, vp]
    L.dbde_hip_pack_frame.restype = sz
    L.dbde_hip_pack_frame.argtypes = [vp, u64, vp, i, i, vp]
    L.dbde_hip_unpack_8x8.restype = None
    L.dbde_hip_unpack_8x8.argtypes = [vp, C.c_uint8, C.c_uint8, vp, sz, vp]
    L.dbde_hip_unpack_8x8_partial.restype = None
    L.dbde_hip_unpack_8x8_partial.argtypes = [vp, C.c_uint8, C.c_uint8, vp, sz, i, i, vp]
    L.dbde_hip_unpack_image.restype = sz
    L.dbde_hip_unpack_image.argtypes = [vp, vp, i, i, vp]
    L.dbde_hip_unpack_frame.restype = FrameHeader
    L.dbde_hip_unpack_frame.argtypes = [vp, C.POINTER(vp), i, i, vp]
    L.dbde_hip_pack_frame_header.restype = sz
    L.dbde_hip_pack_frame_header.argtypes = [C.POINTER(FrameHeader), vp]
    L.dbde_hip_pack_video_header.restype = sz
    L.dbde_hip_pack_video_header.argtypes = [C.POINTER(VideoHeader), vp]
    L.dbde_hip_unpack_frame_header.restype = FrameHeader
    L.dbde_hip_unpack_frame_header.argtypes = [C.POINTER(vp)]
    L.dbde_hip_unpack_video_header.restype = VideoHeader
    L.dbde_hip_unpack_video_header.argtypes = [C.POINTER(vp)]
    L.dbde_hip_timing_enable.restype = i
    L.dbde_hip_timing_enable.argtypes = [vp, i]
    L.dbde_hip_timing_read.restype = i
    L.dbde_hip_timing_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(u64), i]
    L.dbde_hip_stream_handle.restype = vp
    L.dbde_hip_stream_handle.argtypes = [vp]
    L.dbde_hip_device_index.restype = i
    L.dbde_hip_device_index.argtypes = [vp]
    L.dbde16_hip_max_frame_bytes.restype = sz
    L.dbde16_hip_max_frame_bytes.argtypes = [i, i]
    L.dbde16_hip_encode_frames.restype = i
    L.dbde16_hip_encode_frames.argtypes = [vp, vp, i, i, i, u64, vp, sz, u64, vp, vp]
    L.dbde16_hip_decode_frames.restype = i
    L.dbde16_hip_decode_frames.argtypes = [vp, vp, sz, vp, i, i, i, vp, vp]
    L.dbde_hip_writer_open.restype = i
    L.dbde_hip_writer_open.argtypes = [vp, C.c_char_p, i, i, C.c_double, i, C.POINTER(vp)]
    L.dbde_hip_writer_put.restype = i
    L.dbde_hip_writer_put.argtypes = [vp, vp, i, u64, vp, vp]
    L.dbde_hip_writer_error.restype = C.c_char_p
    L.dbde_hip_writer_error.argtypes = [vp]
    L.dbde_hip_writer_close.restype = i
    L.dbde_hip_writer_close.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.dbde_hip_reader_open.restype = i
    L.dbde_hip_reader_open.argtypes = [vp, C.c_char_p, i, C.POINTER(VideoHeader), C.POINTER(vp)]
    L.dbde_hip_reader_next.restype = i
    L.dbde_hip_reader_next.argtypes = [vp, vp, i, C.POINTER(FrameHeader), C.POINTER(i)]
    L.dbde_hip_reader_close.restype = None
    L.dbde_hip_reader_close.argtypes = [vp]
    L.dbde_hip_gather_unique_id.restype = i
    L.dbde_hip_gather_unique_id.argtypes = [vp]
    L.dbde_hip_gather_create.restype = i
    L.dbde_hip_gather_create.argtypes = [vp, vp, i, i, i, C.POINTER(vp)]
    L.dbde_hip_gather_attach.restype = i
    L.dbde_hip_gather_attach.argtypes = [vp, vp, i, i, i, C.POINTER(vp)]
    L.dbde_hip_gather_destroy.restype = None
    L.dbde_hip_gather_destroy.argtypes = [vp]
    L.dbde_hip_gather_error.restype = C.c_char_p
    L.dbde_hip_gather_error.argtypes = [vp]
    L.dbde_hip_gather_set_max_message.restype = i
    L.dbde_hip_gather_set_max_message.argtypes = [vp, u64]
    L.dbde_hip_gather_begin.restype = i
    L.dbde_hip_gather_begin.argtypes = [vp, i, vp, vp]
    L.dbde_hip_gather_post.restype = i
    L.dbde_hip_gather_post.argtypes = [vp, i, vp, vp, sz, C.POINTER(u64), C.c_uint32]
    L.dbde_hip_gather_join.restype = i
    L.dbde_hip_gather_join.argtypes = [vp, i]
    L.dbde_hip_gather_sync.restype = i
    L.dbde_hip_gather_sync.argtypes = [vp, i]
    L.dbde_hip_gather_rccl_version.restype = i
    L.dbde_hip_gather_rccl_version.argtypes = []
    L.dbde_hip_gather_plan.restype = i
    L.dbde_hip_gather_plan.argtypes = [i, i, i, C.POINTER(u64), u64, C.POINTER(GatherOp), i, C.POINTER(u64)]
    L.dbde_hip_gather_set_window.restype = i
    L.dbde_hip_gather_set_window.argtypes = [vp, u64]
    L.dbde_hip_gather_check.restype = i
    L.dbde_hip_gather_check.argtypes = [i, i, C.POINTER(u64), C.POINTER(u64)]
    L.dbde_hip_scatter_create.restype = i
    L.dbde_hip_scatter_create.argtypes = [vp, vp, i, i, i, C.POINTER(vp)]
    L.dbde_hip_scatter_attach.restype = i
    L.dbde_hip_scatter_attach.argtypes = [vp, vp, i, i, i, C.POINTER(vp)]
    L.dbde_hip_scatter_destroy.restype = None
    L.dbde_hip_scatter_destroy.argtypes = [vp]
    L.dbde_hip_scatter_error.restype = C.c_char_p
    L.dbde_hip_scatter_error.argtypes = [vp]
    L.dbde_hip_scatter_set_max_message.restype = i
    L.dbde_hip_scatter_set_max_message.argtypes = [vp, u64]
    L.dbde_hip_scatter_set_capacity.restype = i
    L.dbde_hip_scatter_set_capacity.argtypes = [vp, u64, u64]
    L.dbde_hip_scatter_begin.restype = i
    L.dbde_hip_scatter_begin.argtypes = [vp, i, vp, u64, vp, vp]
    L.dbde_hip_scatter_post.restype = i
    L.dbde_hip_scatter_post.argtypes = [vp, i, vp, vp, C.POINTER(ScatterBlock), C.POINTER(ScatterBlock), C.c_uint32]
    L.dbde_hip_scatter_join.restype = i
    L.dbde_hip_scatter_join.argtypes = [vp, i]
    L.dbde_hip_scatter_sync.restype = i
    L.dbde_hip_scatter_sync.argtypes = [vp, i]
    L.dbde_hip_scatter_blocks.restype = i
    L.dbde_hip_scatter_blocks.argtypes = [i, u64, C.POINTER(u64), u64, C.POINTER(ScatterBlock)]
    L.dbde_hip_scatter_check.restype = i
    L.dbde_hip_scatter_check.argtypes = [i, C.POINTER(ScatterBlock), C.POINTER(u64)]
    L.dbde_hip_scatter_plan.restype = i
    L.dbde_hip_scatter_plan.argtypes = [i, i, i, C.POINTER(ScatterBlock), u64, C.POINTER(ScatterOp), i]
    _lib = L
    return L


def max_frame_bytes(W, H):
    return int(lib().dbde_hip_max_frame_bytes(W, H))


def tiles(W, H):
    return ((W + 7) // 8) * ((H + 7) // 8)


def gather_plan(nranks, rank, root, sizes, max_piece=0):
    """dbde_hip_gather_plan: the ordered transfers of `rank` given every rank's byte count (host arithmetic only).
    -> (list of (peer, kind, segment_offset, window_offset, bytes), total bytes)."""
    arr = (C.c_uint64 * nranks)(*[int(x) for x in sizes])
    total = C.c_uint64(0)
    n = lib().dbde_hip_gather_plan(nranks, rank, root, arr, max_piece, None, 0, C.byref(total))
    if n < 0:
        raise ValueError(f"dbde_hip_gather_plan({nranks}, {rank}, {root}) -> {n}")
    ops = (GatherOp * max(n, 1))()
    lib().dbde_hip_gather_plan(nranks, rank, root, arr, max_piece, ops, n, None)
    return [(o.peer, o.kind, o.segment_offset, o.window_offset, o.bytes) for o in ops[:n]], total.value


class LaunchPlan(C.Structure):
    """dbde_hip_launch_plan (include/dbde_hip.h)."""
    _fields_ = [("kernel", C.c_int32), ("input_mode", C.c_int32), ("image_mode", C.c_int32), ("index_mode", C.c_int32),
                ("threads", C.c_int32), ("aligned_out", C.c_int32), ("chunks_per_frame", C.c_uint32),
                ("chunk_tiles", C.c_uint32), ("n_chunks", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def encode_plan(W, H, n_frames, image_address=0, out_address=0, slot_stride=0, resident_workgroups=513):
    """dbde_hip_encode_plan: which kernel form an encode call with these arguments runs (host arithmetic only)."""
    pl = LaunchPlan()
    L = lib()
    L.dbde_hip_encode_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(LaunchPlan)]
    rc = L.dbde_hip_encode_plan(W, H, n_frames, image_address, out_address, slot_stride, resident_workgroups, C.byref(pl))
    if rc != OK:
        raise ValueError(f"dbde_hip_encode_plan({W}, {H}, {n_frames}) -> {rc}")
    return pl.as_dict()


def decode_plan(W, H, n_frames, image_address=0, n_cu=256):
    """dbde_hip_decode_plan: which kernel form a decode call with these arguments runs (host arithmetic only)."""
    pl = LaunchPlan()
    L = lib()
    L.dbde_hip_decode_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.POINTER(LaunchPlan)]
    rc = L.dbde_hip_decode_plan(W, H, n_frames, image_address, n_cu, C.byref(pl))
    if rc != OK:
        raise ValueError(f"dbde_hip_decode_plan({W}, {H}, {n_frames}) -> {rc}")
    return pl.as_dict()


def gather_check(nranks, root, sizes, caps):
    """dbde_hip_gather_check: (verdict, total) every rank reaches from the exchanged {count, capacity} pairs."""
    pairs = (C.c_uint64 * (2 * nranks))(*[int(x) for r in range(nranks) for x in (sizes[r], caps[r])])
    total = C.c_uint64(0)
    rc = lib().dbde_hip_gather_check(nranks, root, pairs, C.byref(total))
    return rc, int(total.value)


def scatter_blocks(nranks, frame_offsets, stream_bytes):
    """dbde_hip_scatter_blocks: [(first_frame, n_frames, byte_start, byte_count)] per rank from a host-side frame index."""
    n = len(frame_offsets)
    arr = (C.c_uint64 * max(n, 1))(*[int(x) for x in frame_offsets])
    table = (ScatterBlock * nranks)()
    rc = lib().dbde_hip_scatter_blocks(nranks, n, arr, int(stream_bytes), table)
    if rc != OK:
        raise ValueError(f"dbde_hip_scatter_blocks -> {rc}")
    return [(int(b.first_frame), int(b.n_frames), int(b.byte_start), int(b.byte_count)) for b in table]


def _scatter_table(blocks):
    table = (ScatterBlock * len(blocks))()
    for k, b in enumerate(blocks):
        table[k].first_frame, table[k].n_frames, table[k].byte_start, table[k].byte_count = [int(x) for x in b]
    return table


def scatter_check(blocks, caps):
    """dbde_hip_scatter_check: OK or ERR_CAPACITY; caps = [(segment_bytes, max_frames)] per rank."""
    flat = (C.c_uint64 * (2 * len(caps)))(*[int(x) for c in caps for x in c])
    return lib().dbde_hip_scatter_check(len(blocks), _scatter_table(blocks), flat)


def scatter_plan(nranks, rank, root, blocks, max_piece=0):
    """dbde_hip_scatter_plan: the ordered transfers of `rank` -> [(peer, kind, source_offset, dest_offset, bytes)]."""
    table = _scatter_table(blocks)
    n = lib().dbde_hip_scatter_plan(nranks, rank, root, table, max_piece, None, 0)
    if n < 0:
        raise ValueError(f"dbde_hip_scatter_plan({nranks}, {rank}, {root}) -> {n}")
    ops = (ScatterOp * max(n, 1))()
    lib().dbde_hip_scatter_plan(nranks, rank, root, table, max_piece, ops, n)
    return [(o.peer, o.kind, int(o.source_offset), int(o.dest_offset), int(o.bytes)) for o in ops[:n]]


def gather_unique_id():
    """Rendezvous token of the native gather (ncclGetUniqueId), as a 128-byte uint8 numpy array."""
    out = np.zeros(GATHER_ID_BYTES, np.uint8)
    rc = lib().dbde_hip_gather_unique_id(out.ctypes.data)
    if rc != OK:
        raise RuntimeError(f"dbde_hip_gather_unique_id failed ({rc}): librccl could not be opened")
    return out


# ---- header wire format (host only) ------------------------------------------------------

def pack_frame_header(u64s, index, elapsed_ns):
    out = np.zeros(20, np.uint8)
    fh = FrameHeader(u64s, index, elapsed_ns)
    assert lib().dbde_hip_pack_frame_header(C.byref(fh), out.ctypes.data) == 20
    return out


def pack_video_header(u64s, height, width, frame_hz):
    out = np.zeros(28, np.uint8)
    vh = VideoHeader(u64s, height, width, frame_hz)
    assert lib().dbde_hip_pack_video_header(C.byref(vh), out.ctypes.data) == 28
    return out


def unpack_frame_header(packed):
    buf = np.ascontiguousarray(np.asarray(packed, np.uint8)[:20])
    cur = C.c_void_p(buf.ctypes.data)
    fh = lib().dbde_hip_unpack_frame_header(C.byref(cur))
    return cur.value - buf.ctypes.data, (fh.u64s, fh.index, fh.elapsed_ns)


def unpack_video_header(packed):
    buf = np.ascontiguousarray(np.asarray(packed, np.uint8)[:28])
    cur = C.c_void_p(buf.ctypes.data)
    vh = lib().dbde_hip_unpack_video_header(C.byref(cur))
    return cur.value - buf.ctypes.data, (vh.u64s, vh.height, vh.width, vh.frame_hz)


class DbdeError(RuntimeError):
    pass


class Codec:
    """One C-ABI context: a HIP device + stream.  Not thread-safe (one per thread)."""

    def __init__(self, device=0, stream=None):
        self.L = lib()
        if torch is None or not torch.cuda.is_available():
            raise DbdeError("no HIP device visible to PyTorch; the DBDE codec has no CPU path")
        self.device = torch.device("cuda", device)
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        self.stream = stream
        h = C.c_void_p()
        rc = self.L.dbde_hip_create(device, C.c_void_p(stream.cuda_stream), C.byref(h))
        if rc != OK or not h.value:
            raise DbdeError(f"dbde_hip_create failed ({rc}): needs a gfx950 device; no CPU path")
        self.h = h

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.L.dbde_hip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != OK:
            raise DbdeError(f"{what} failed ({rc}): {self.L.dbde_hip_last_error(self.h).decode()}")

    @property
    def arch(self):
        return self.L.dbde_hip_device_arch(self.h).decode()

    def sync(self):
        self._check(self.L.dbde_hip_sync(self.h), "dbde_hip_sync")

    # ---- batch API on device tensors ---------------------------------------------------
    def synth_frames(self, mode, seed, first_frame, n, W, H, out=None):
        if out is None:
            out = torch.empty((n, H, W), dtype=torch.uint8, device=self.device)
        m = MODES[mode] if isinstance(mode, str) else mode
        self._check(self.L.dbde_hip_synth_frames(self.h, m, seed, first_frame, n, W, H, out.data_ptr()),
                    "dbde_hip_synth_frames")
        return out

    def alloc_stream(self, W, H, n, slot_stride=0, lead=32):
        """Device buffer for n worst-case frames.  `lead` bytes are kept in front of the first
        frame (28-byte video header at lead-28) so that frames start 8-byte aligned."""
        cap = (n - 1) * slot_stride + max_frame_bytes(W, H) if slot_stride else n * max_frame_bytes(W, H)
        buf = torch.empty(lead + cap + 64, dtype=torch.uint8, device=self.device)
        return buf, lead, cap

    def encode_frames(self, images, W, H, n, out, out_offset, capacity, first_index=0, indices=None,
                      elapsed_ns=None, slot_stride=0, offsets=None, nbytes=None):
        """images: uint8 device tensor of n*H*W bytes.  Frames are written from
        out.data_ptr()+out_offset.  Returns (offsets, nbytes) int64 device tensors (per frame)."""
        if offsets is None:
            offsets = torch.empty(n, dtype=torch.int64, device=self.device)
        if nbytes is None:
            nbytes = torch.empty(n, dtype=torch.int64, device=self.device)
        rc = self.L.dbde_hip_encode_frames(
            self.h, images.data_ptr(), W, H, n, first_index,
            indices.data_ptr() if indices is not None else None,
            elapsed_ns.data_ptr() if elapsed_ns is not None else None,
            out.data_ptr() + out_offset, capacity, slot_stride, offsets.data_ptr(), nbytes.data_ptr())
        self._check(rc, "dbde_hip_encode_frames")
        return offsets, nbytes

    def decode_frames(self, stream, stream_offset, stream_bytes, offsets, W, H, n, images=None, results=None):
        """Decodes n frames; frame f starts at stream.data_ptr()+stream_offset+offsets[f]."""
        if images is None:
            images = torch.empty((n, H, W), dtype=torch.uint8, device=self.device)
        if results is None:
            results = torch.empty((n, 4), dtype=torch.int64, device=self.device)
        rc = self.L.dbde_hip_decode_frames(self.h, stream.data_ptr() + stream_offset, stream_bytes,
                                           offsets.data_ptr(), W, H, n, images.data_ptr(), results.data_ptr())
        self._check(rc, "dbde_hip_decode_frames")
        return images, results

    def index_stream(self, stream, stream_offset, stream_bytes, W, H, max_frames):
        offsets = torch.empty(max(max_frames, 1), dtype=torch.int64, device=self.device)
        n = C.c_int(0)
        rc = self.L.dbde_hip_index_stream(self.h, stream.data_ptr() + stream_offset, stream_bytes, W, H,
                                          max_frames, offsets.data_ptr(), C.byref(n))
        self._check(rc, "dbde_hip_index_stream")
        return offsets[:n.value], n.value

    def index_stream_async(self, stream, stream_offset, stream_bytes, W, H, max_frames, offsets, count=None):
        """Enqueues the frame-to-frame walk; offsets (int64 device tensor, >= max_frames) and the device
        word `count` (int32 tensor, 1 element) are valid once the stream reaches this point."""
        if count is None:
            count = torch.empty(1, dtype=torch.int32, device=self.device)
        rc = self.L.dbde_hip_index_stream_async(self.h, stream.data_ptr() + stream_offset, stream_bytes, W, H,
                                                max_frames, offsets.data_ptr(), count.data_ptr())
        self._check(rc, "dbde_hip_index_stream_async")
        return offsets, count

    def scan_ahead(self, stream, stream_offset, stream_bytes, W, H, max_frames, cursor, offsets, count):
        """Enqueues the walk of the next batch on the context's second stream (see dbde_hip.h): `cursor`
        (int64 device tensor, 1 element, zeroed before the first call) is advanced past the frames found."""
        rc = self.L.dbde_hip_scan_ahead(self.h, stream.data_ptr() + stream_offset, stream_bytes, W, H, max_frames,
                                        cursor.data_ptr(), offsets.data_ptr(), count.data_ptr())
        self._check(rc, "dbde_hip_scan_ahead")

    def scan_join(self):
        self._check(self.L.dbde_hip_scan_join(self.h), "dbde_hip_scan_join")

    @staticmethod
    def parse_results(results):
        """results tensor (n,4) int64 -> list of (u64s, index, elapsed_ns, consumed)."""
        r = results.cpu().numpy().view(np.uint64)
        return [(int(a) & 0xFFFFFFFF, int(b), int(c), int(d)) for a, b, c, d in r]

    # ---- DBDE16 (higher-bit-depth extension, parity unpinned) ---------------------------------
    def encode_frames16(self, images, W, H, n, out, out_offset, capacity, first_index=0, slot_stride=0):
        """images: int16/uint16-sized device tensor of n*H*W pixels.  Returns (offsets, nbytes) int64 device tensors."""
        offsets = torch.empty(n, dtype=torch.int64, device=self.device)
        nbytes = torch.empty(n, dtype=torch.int64, device=self.device)
        rc = self.L.dbde16_hip_encode_frames(self.h, images.data_ptr(), W, H, n, first_index, out.data_ptr() + out_offset,
                                             capacity, slot_stride, offsets.data_ptr(), nbytes.data_ptr())
        self._check(rc, "dbde16_hip_encode_frames")
        return offsets, nbytes

    def decode_frames16(self, stream, stream_offset, stream_bytes, offsets, W, H, n, images=None):
        if images is None:
            images = torch.empty((n, H, W), dtype=torch.int16, device=self.device)
        results = torch.empty((n, 4), dtype=torch.int64, device=self.device)
        rc = self.L.dbde16_hip_decode_frames(self.h, stream.data_ptr() + stream_offset, stream_bytes, offsets.data_ptr(),
                                             W, H, n, images.data_ptr(), results.data_ptr())
        self._check(rc, "dbde16_hip_decode_frames")
        return images, results

    # ---- host-pointer API: the reference's functions -------------------------------------
    def pack_frame(self, index, image, W, H):
        src = np.ascontiguousarray(image, np.uint8).reshape(-1)
        out = np.full(max_frame_bytes(W, H) + 64, 0xEE, np.uint8)
        n = self.L.dbde_hip_pack_frame(self.h, index, src.ctypes.data, W, H, out.ctypes.data)
        assert (out[n:] == 0xEE).all(), "wrote past the returned size"
        return out[:n].copy()

    def pack_image(self, image, W, H):
        src = np.ascontiguousarray(image, np.uint8).reshape(-1)
        out = np.full(max_frame_bytes(W, H) + 64, 0xEE, np.uint8)
        n = self.L.dbde_hip_pack_image(self.h, src.ctypes.data, W, H, out.ctypes.data)
        assert (out[n:] == 0xEE).all(), "wrote past the returned size"
        return out[:n].copy()

    def unpack_image(self, packed, W, H, fill=0xEE):
        img = np.full(W * H, fill, np.uint8)
        buf = np.concatenate([np.asarray(packed, np.uint8), np.zeros(64, np.uint8)])
        n = self.L.dbde_hip_unpack_image(self.h, buf.ctypes.data, W, H, img.ctypes.data)
        return int(n), img.reshape(H, W)

    def unpack_frame(self, packed, W, H, fill=0xEE):
        img = np.full(W * H, fill, np.uint8)
        buf = np.concatenate([np.asarray(packed, np.uint8), np.zeros(64, np.uint8)])
        cur = C.c_void_p(buf.ctypes.data)
        fh = self.L.dbde_hip_unpack_frame(self.h, C.byref(cur), W, H, img.ctypes.data)
        return cur.value - buf.ctypes.data, (fh.u64s, fh.index, fh.elapsed_ns), img.reshape(H, W)

    def pack_8x8(self, image, off, stride):
        out = np.full(64 + 16, 0xEE, np.uint8)
        code = self.L.dbde_hip_pack_8x8(self.h, image.ctypes.data + off, stride, out.ctypes.data)
        return int(code), out[:8 * (code >> 8)].copy(), out

    def pack_8x8_partial(self, image, off, stride, rm, dm):
        out = np.full(64 + 16, 0xEE, np.uint8)
        code = self.L.dbde_hip_pack_8x8_partial(self.h, image.ctypes.data + off, stride, rm, dm, out.ctypes.data)
        return int(code), out[:8 * (code >> 8)].copy(), out

    def unpack_8x8(self, depth, minval, packed, stride, canvas, off=0):
        buf = np.zeros(64 + 8, np.uint8)
        buf[:len(packed)] = packed
        self.L.dbde_hip_unpack_8x8(self.h, depth, minval, buf.ctypes.data, stride, canvas.ctypes.data + off)
        return canvas

    def unpack_8x8_partial(self, depth, minval, packed, stride, rm, dm, canvas, off=0):
        buf = np.zeros(64 + 8, np.uint8)
        buf[:len(packed)] = packed
        self.L.dbde_hip_unpack_8x8_partial(self.h, depth, minval, buf.ctypes.data, stride, rm, dm,
                                           canvas.ctypes.data + off)
        return canvas

    # ---- .dbde files: batched writer / reader (include/dbde_hip.h, file I/O) ----------------
    def open_writer(self, path, W, H, frame_hz=30.0, batch_frames=16):
        return FileWriter(self, path, W, H, frame_hz, batch_frames)

    def open_reader(self, path, batch_frames=16):
        return FileReader(self, path, batch_frames)

    # ---- timing hook ------------------------------------------------------------------
    def timing(self, on=True):
        self._check(self.L.dbde_hip_timing_enable(self.h, 1 if on else 0), "dbde_hip_timing_enable")

    def timing_read(self, reset=True):
        ms = (C.c_double * 4)()
        n = (C.c_uint64 * 4)()
        self._check(self.L.dbde_hip_timing_read(self.h, ms, n, 1 if reset else 0), "dbde_hip_timing_read")
        return {"encode": (ms[0], n[0]), "decode_index": (ms[1], n[1]), "decode": (ms[2], n[2]), "scan": (ms[3], n[3])}


class Gather:
    """dbde_hip_gather_*: this rank's end of the variable-length gather of the compressed stream (RCCL).
    `unique_id`: 128 uint8 (gather_unique_id() on one rank, handed to the others), or pass `comm`=an ncclComm_t."""

    def __init__(self, codec, unique_id, nranks, rank, root=0, comm=None, max_message_bytes=None):
        self.codec, self.nranks, self.rank, self.root = codec, nranks, rank, root
        self.h = C.c_void_p()
        if comm is not None:
            rc = codec.L.dbde_hip_gather_attach(codec.h, C.c_void_p(comm), nranks, rank, root, C.byref(self.h))
        else:
            uid = np.ascontiguousarray(np.asarray(unique_id, np.uint8))
            assert uid.size == GATHER_ID_BYTES
            rc = codec.L.dbde_hip_gather_create(codec.h, uid.ctypes.data, nranks, rank, root, C.byref(self.h))
        if rc != OK or not self.h.value:
            raise DbdeError(f"dbde_hip_gather_create failed ({rc})")
        if max_message_bytes:
            self._check(codec.L.dbde_hip_gather_set_max_message(self.h, int(max_message_bytes)), "set_max_message")

    def _check(self, rc, what):
        if rc != OK:
            raise DbdeError(f"dbde_hip_gather_{what} failed ({rc}): {self.codec.L.dbde_hip_gather_error(self.h).decode()}")

    def set_window(self, window_bytes):
        """Root: the bytes its window holds; travels with every size exchange so that an overflow is ONE verdict on all ranks."""
        self._check(self.codec.L.dbde_hip_gather_set_window(self.h, int(window_bytes)), "set_window")

    def begin(self, slot, last_offset, last_bytes):
        """last_offset / last_bytes: 1-element int64 device tensors (views of the encoder's outputs) or None."""
        self._check(self.codec.L.dbde_hip_gather_begin(self.h, slot, last_offset.data_ptr() if last_offset is not None else None,
                                                       last_bytes.data_ptr() if last_bytes is not None else None), "begin")

    def post(self, slot, segment, segment_offset, window, window_offset, window_bytes, loopback=False):
        """-> list of every rank's byte count.  segment / window: uint8 device tensors (either may be None where the
        rank does not need it)."""
        sizes = (C.c_uint64 * self.nranks)()
        seg = segment.data_ptr() + segment_offset if segment is not None else None
        win = window.data_ptr() + window_offset if window is not None else None
        self._check(self.codec.L.dbde_hip_gather_post(self.h, slot, seg, win, window_bytes, sizes,
                                                      GATHER_LOOPBACK if loopback else 0), "post")
        return [int(x) for x in sizes]

    def join(self, slot):
        self._check(self.codec.L.dbde_hip_gather_join(self.h, slot), "join")

    def sync(self, slot):
        self._check(self.codec.L.dbde_hip_gather_sync(self.h, slot), "sync")

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.codec.L.dbde_hip_gather_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scatter:
    """dbde_hip_scatter_*: this rank's end of the scatter of a .dbde body to the ranks' frame blocks (RCCL): the decode-side
    mirror of Gather.  `unique_id` as for Gather, or `comm` = an ncclComm_t."""

    def __init__(self, codec, unique_id, nranks, rank, root=0, comm=None, max_message_bytes=None):
        self.codec, self.nranks, self.rank, self.root = codec, nranks, rank, root
        self.h = C.c_void_p()
        if comm is not None:
            rc = codec.L.dbde_hip_scatter_attach(codec.h, C.c_void_p(comm), nranks, rank, root, C.byref(self.h))
        else:
            uid = np.ascontiguousarray(np.asarray(unique_id, np.uint8))
            assert uid.size == GATHER_ID_BYTES
            rc = codec.L.dbde_hip_scatter_create(codec.h, uid.ctypes.data, nranks, rank, root, C.byref(self.h))
        if rc != OK or not self.h.value:
            raise DbdeError(f"dbde_hip_scatter_create failed ({rc})")
        if max_message_bytes:
            self._check(codec.L.dbde_hip_scatter_set_max_message(self.h, int(max_message_bytes)), "set_max_message")

    def _check(self, rc, what):
        if rc != OK:
            raise DbdeError(f"dbde_hip_scatter_{what} failed ({rc}): {self.codec.L.dbde_hip_scatter_error(self.h).decode()}")

    def set_capacity(self, segment_bytes, max_frames):
        self._check(self.codec.L.dbde_hip_scatter_set_capacity(self.h, int(segment_bytes), int(max_frames)), "set_capacity")

    def begin(self, slot, stream=None, stream_offset=0, stream_bytes=0, offsets=None, count=None):
        """Root: the stream (uint8 device tensor), its extent, the scanner's offsets (int64) and count (int32, 1 element)."""
        self._check(self.codec.L.dbde_hip_scatter_begin(
            self.h, slot, stream.data_ptr() + stream_offset if stream is not None else None, int(stream_bytes),
            offsets.data_ptr() if offsets is not None else None, count.data_ptr() if count is not None else None), "begin")

    def post(self, slot, segment, offsets_out, loopback=False):
        """-> (mine, table): (first_frame, n_frames, byte_start, byte_count) of this rank and of every rank."""
        mine, table = ScatterBlock(), (ScatterBlock * self.nranks)()
        self._check(self.codec.L.dbde_hip_scatter_post(self.h, slot, segment.data_ptr() if segment is not None else None,
                                                       offsets_out.data_ptr(), C.byref(mine), table,
                                                       SCATTER_LOOPBACK if loopback else 0), "post")
        t = lambda b: (int(b.first_frame), int(b.n_frames), int(b.byte_start), int(b.byte_count))
        return t(mine), [t(b) for b in table]

    def join(self, slot):
        self._check(self.codec.L.dbde_hip_scatter_join(self.h, slot), "join")

    def sync(self, slot):
        self._check(self.codec.L.dbde_hip_scatter_sync(self.h, slot), "sync")

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.codec.L.dbde_hip_scatter_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FileWriter:
    """dbde_hip_writer_*: appends device-resident frames to a .dbde file, a batch per launch."""

    def __init__(self, codec, path, W, H, frame_hz, batch_frames):
        self.codec, self.W, self.H = codec, W, H
        self.h = C.c_void_p()
        rc = codec.L.dbde_hip_writer_open(codec.h, os.fsencode(path), W, H, frame_hz, batch_frames, C.byref(self.h))
        if rc != OK:
            raise DbdeError(f"dbde_hip_writer_open({path!r}) failed ({rc})")

    def put(self, images, n, first_index=0, indices=None, elapsed_ns=None):
        rc = self.codec.L.dbde_hip_writer_put(self.h, images.data_ptr(), n, first_index,
                                              indices.data_ptr() if indices is not None else None,
                                              elapsed_ns.data_ptr() if elapsed_ns is not None else None)
        if rc != OK:
            raise DbdeError(f"dbde_hip_writer_put failed ({rc}): {self.codec.L.dbde_hip_writer_error(self.h).decode()}")

    def close(self):
        """Returns (frames written, file bytes)."""
        if not self.h.value:
            return None
        fr, by = C.c_uint64(0), C.c_uint64(0)
        rc = self.codec.L.dbde_hip_writer_close(self.h, C.byref(fr), C.byref(by))
        self.h = C.c_void_p()
        if rc != OK:
            raise DbdeError(f"dbde_hip_writer_close failed ({rc})")
        return fr.value, by.value

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class FileReader:
    """dbde_hip_reader_*: walks a .dbde file a batch at a time, images land in HBM."""

    def __init__(self, codec, path, batch_frames):
        self.codec, self.batch = codec, batch_frames
        self.h = C.c_void_p()
        vh = VideoHeader()
        rc = codec.L.dbde_hip_reader_open(codec.h, os.fsencode(path), batch_frames, C.byref(vh), C.byref(self.h))
        if rc != OK:
            raise DbdeError(f"dbde_hip_reader_open({path!r}) failed ({rc})")
        self.video_header = (vh.u64s, vh.height, vh.width, vh.frame_hz)
        self.W, self.H = int(vh.width), int(vh.height)

    def next(self, max_frames=None, images=None):
        """-> (images[:n] device tensor, [(u64s, index, elapsed_ns)] * n); n == 0 ends the walk."""
        m = self.batch if max_frames is None else min(max_frames, self.batch)
        if images is None:
            images = torch.empty((m, self.H, self.W), dtype=torch.uint8, device=self.codec.device)
        hdr = (FrameHeader * max(m, 1))()
        n = C.c_int(0)
        rc = self.codec.L.dbde_hip_reader_next(self.h, images.data_ptr(), m, hdr, C.byref(n))
        if rc != OK:
            raise DbdeError(f"dbde_hip_reader_next failed ({rc}): {self.codec.L.dbde_hip_last_error(self.codec.h).decode()}")
        return images[:n.value], [(hdr[k].u64s, hdr[k].index, hdr[k].elapsed_ns) for k in range(n.value)]

    def __iter__(self):
        while True:
            imgs, hdrs = self.next()
            if not hdrs:
                return
            yield imgs, hdrs

    def close(self):
        if self.h.value:
            self.codec.L.dbde_hip_reader_close(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
