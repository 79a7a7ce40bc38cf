"""ctypes doorway to the parity checker (oracle/) -- test infrastructure only.

`Oracle` wraps oracle/liboracle.so (the CPU restatement); `Reference` wraps
oracle/_ref/libdbde_ref.so (the real reference, built by oracle/Makefile from
/root/reference where that exists; the prebuilt .so travels to the GPU box).
Both expose the same Python-level API so tests can run one against the other.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libdbde_ref.so")

u8p = C.POINTER(C.c_uint8)


def build_oracle():
    """Compile oracle/ (and oracle/_ref when /root/reference is present)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def _p(a):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u8p)


def max_frame_bytes(W, H):
    T = ((W + 7) // 8) * ((H + 7) // 8)
    return 20 + 12 + 66 * T


class _FH(C.Structure):
    _fields_ = [("u64s", C.c_uint32), ("index", C.c_uint64), ("elapsed_ns", C.c_uint64)]


class _VH(C.Structure):
    _fields_ = [("u64s", C.c_uint32), ("height", C.c_uint64), ("width", C.c_uint64),
                ("frame_hz", C.c_double)]


class Oracle:
    name = "oracle"

    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            build_oracle()
        L = self.L = C.CDLL(path)
        L.dbde_oracle_pack_8x8.restype = C.c_uint32
        L.dbde_oracle_pack_8x8.argtypes = [u8p, C.c_int, u8p]
        L.dbde_oracle_pack_8x8_partial.restype = C.c_uint32
        L.dbde_oracle_pack_8x8_partial.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p]
        L.dbde_oracle_unpack_8x8.restype = None
        L.dbde_oracle_unpack_8x8.argtypes = [C.c_uint8, C.c_uint8, u8p, C.c_size_t, u8p]
        L.dbde_oracle_unpack_8x8_partial.restype = None
        L.dbde_oracle_unpack_8x8_partial.argtypes = [C.c_uint8, C.c_uint8, u8p, C.c_size_t,
                                                     C.c_int, C.c_int, u8p]
        for f in ("pack_image", "unpack_image"):
            fn = getattr(L, "dbde_oracle_" + f)
            fn.restype = C.c_size_t
            fn.argtypes = [u8p, C.c_int, C.c_int, u8p]
        L.dbde_oracle_pack_frame.restype = C.c_size_t
        L.dbde_oracle_pack_frame.argtypes = [C.c_uint64, u8p, C.c_int, C.c_int, u8p]
        L.dbde_oracle_pack_frame_header.restype = C.c_size_t
        L.dbde_oracle_pack_frame_header.argtypes = [C.POINTER(_FH), u8p]
        L.dbde_oracle_pack_video_header.restype = C.c_size_t
        L.dbde_oracle_pack_video_header.argtypes = [C.POINTER(_VH), u8p]
        L.dbde_oracle_unpack_frame_header.restype = C.c_size_t
        L.dbde_oracle_unpack_frame_header.argtypes = [u8p, C.POINTER(_FH)]
        L.dbde_oracle_unpack_frame.restype = C.c_size_t
        L.dbde_oracle_unpack_frame.argtypes = [u8p, C.c_int, C.c_int, u8p, C.POINTER(_FH)]
        L.dbde_oracle_unpack_video_header.restype = C.c_size_t
        L.dbde_oracle_unpack_video_header.argtypes = [u8p, C.POINTER(_VH)]
        L.dbde_oracle_synth_frame.restype = None
        L.dbde_oracle_synth_frame.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, u8p]
        L.dbde_oracle_time_roundtrip.restype = C.c_double
        L.dbde_oracle_time_roundtrip.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p, u8p,
                                                 C.POINTER(C.c_double), C.POINTER(C.c_double),
                                                 C.POINTER(C.c_uint64)]

    # -- tile level -------------------------------------------------------------------
    def pack_8x8(self, image, off, stride):
        out = np.full(64 + 16, 0xEE, np.uint8)
        code = self.L.dbde_oracle_pack_8x8(C.cast(image.ctypes.data + off, u8p), stride, _p(out))
        return code, out[:8 * (code >> 8)].copy(), out

    def pack_8x8_partial(self, image, off, stride, rm, dm):
        out = np.full(64 + 16, 0xEE, np.uint8)
        code = self.L.dbde_oracle_pack_8x8_partial(C.cast(image.ctypes.data + off, u8p), stride,
                                                   rm, dm, _p(out))
        return code, out[:8 * (code >> 8)].copy(), out

    def unpack_8x8(self, depth, minval, packed, stride=8, canvas=None, off=0):
        img = np.full(8 * stride + 8, 0xEE, np.uint8) if canvas is None else canvas
        buf = np.zeros(64 + 8, np.uint8)
        buf[:len(packed)] = packed
        self.L.dbde_oracle_unpack_8x8(depth, minval, _p(buf), stride,
                                      C.cast(img.ctypes.data + off, u8p))
        return img

    def unpack_8x8_partial(self, depth, minval, packed, stride, rm, dm, canvas, off=0):
        buf = np.zeros(64 + 8, np.uint8)
        buf[:len(packed)] = packed
        self.L.dbde_oracle_unpack_8x8_partial(depth, minval, _p(buf), stride, rm, dm,
                                              C.cast(canvas.ctypes.data + off, u8p))
        return canvas

    # -- frame level ------------------------------------------------------------------
    def pack_image(self, image, W, H):
        out = np.full(max_frame_bytes(W, H) + 64, 0xEE, np.uint8)
        n = self.L.dbde_oracle_pack_image(_p(np.ascontiguousarray(image).reshape(-1)), W, H, _p(out))
        assert (out[n:] == 0xEE).all(), "encoder wrote past its returned size"
        return out[:n].copy()

    def pack_frame(self, index, image, W, H):
        out = np.full(max_frame_bytes(W, H) + 64, 0xEE, np.uint8)
        n = self.L.dbde_oracle_pack_frame(index, _p(np.ascontiguousarray(image).reshape(-1)), W, H,
                                          _p(out))
        assert (out[n:] == 0xEE).all(), "encoder wrote past its returned size"
        return out[:n].copy()

    def pack_frame_header(self, u64s, index, elapsed_ns):
        out = np.zeros(20, np.uint8)
        fh = _FH(u64s, index, elapsed_ns)
        assert self.L.dbde_oracle_pack_frame_header(C.byref(fh), _p(out)) == 20
        return out

    def pack_video_header(self, u64s, height, width, hz):
        out = np.zeros(28, np.uint8)
        vh = _VH(u64s, height, width, hz)
        assert self.L.dbde_oracle_pack_video_header(C.byref(vh), _p(out)) == 28
        return out

    def unpack_image(self, packed, W, H, fill=0xEE):
        """-> (bytes consumed, image as (H, W) array pre-filled with `fill`)."""
        img = np.full(W * H, fill, np.uint8)
        buf = np.concatenate([np.asarray(packed, np.uint8), np.zeros(64, np.uint8)])
        n = self.L.dbde_oracle_unpack_image(_p(buf), W, H, _p(img))
        return n, img.reshape(H, W)

    def unpack_frame_header(self, packed):
        fh = _FH()
        buf = np.ascontiguousarray(np.asarray(packed, np.uint8)[:20])
        n = self.L.dbde_oracle_unpack_frame_header(_p(buf), C.byref(fh))
        return n, (fh.u64s, fh.index, fh.elapsed_ns)

    def unpack_frame(self, packed, W, H, fill=0xEE):
        """-> (bytes advanced, (u64s, index, elapsed_ns), image)."""
        img = np.full(W * H, fill, np.uint8)
        buf = np.concatenate([np.asarray(packed, np.uint8), np.zeros(64, np.uint8)])
        fh = _FH()
        n = self.L.dbde_oracle_unpack_frame(_p(buf), W, H, _p(img), C.byref(fh))
        return n, (fh.u64s, fh.index, fh.elapsed_ns), img.reshape(H, W)

    def unpack_video_header(self, packed):
        vh = _VH()
        buf = np.ascontiguousarray(np.asarray(packed, np.uint8)[:28])
        n = self.L.dbde_oracle_unpack_video_header(_p(buf), C.byref(vh))
        return n, (vh.u64s, vh.height, vh.width, vh.frame_hz)

    # -- generators / baseline --------------------------------------------------------
    def synth_frame(self, mode, seed, frame, W, H):
        img = np.empty(W * H, np.uint8)
        self.L.dbde_oracle_synth_frame(mode, seed, frame, W, H, _p(img))
        return img.reshape(H, W)

    def time_roundtrip(self, images, n, W, H, reps=1):
        sp = np.empty(max_frame_bytes(W, H) + 64, np.uint8)
        si = np.empty(W * H, np.uint8)
        te, td, bad = C.c_double(), C.c_double(), C.c_uint64()
        t = self.L.dbde_oracle_time_roundtrip(_p(images.reshape(-1)), n, W, H, reps, _p(sp), _p(si),
                                              C.byref(te), C.byref(td), C.byref(bad))
        return t, te.value, td.value, bad.value


class Reference:
    """The real reference (oracle/_ref/libdbde_ref.so) behind the same Python API."""
    name = "reference"

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self, path=REF_SO):
        L = self.L = C.CDLL(path)
        L.ref_pack_8x8.restype = C.c_uint32
        L.ref_pack_8x8.argtypes = [u8p, C.c_int, u8p]
        L.ref_pack_8x8_partial.restype = C.c_uint32
        L.ref_pack_8x8_partial.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p]
        L.ref_unpack_8x8.restype = None
        L.ref_unpack_8x8.argtypes = [C.c_uint8, C.c_uint8, u8p, C.c_size_t, u8p]
        L.ref_unpack_8x8_partial.restype = None
        L.ref_unpack_8x8_partial.argtypes = [C.c_uint8, C.c_uint8, u8p, C.c_size_t, C.c_int, C.c_int, u8p]
        for f in ("ref_pack_image", "ref_unpack_image"):
            fn = getattr(L, f)
            fn.restype = C.c_size_t
            fn.argtypes = [u8p, C.c_int, C.c_int, u8p]
        L.ref_pack_frame.restype = C.c_size_t
        L.ref_pack_frame.argtypes = [C.c_uint64, u8p, C.c_int, C.c_int, u8p]
        L.ref_pack_frame_header.restype = C.c_size_t
        L.ref_pack_frame_header.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, u8p]
        L.ref_pack_video_header.restype = C.c_size_t
        L.ref_pack_video_header.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, C.c_double, u8p]
        L.ref_unpack_frame_header.restype = C.c_size_t
        L.ref_unpack_frame_header.argtypes = [u8p, C.POINTER(C.c_uint64)]
        L.ref_unpack_frame.restype = C.c_size_t
        L.ref_unpack_frame.argtypes = [u8p, C.c_int, C.c_int, u8p, C.POINTER(C.c_uint64)]
        L.ref_unpack_video_header.restype = C.c_size_t
        L.ref_unpack_video_header.argtypes = [u8p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        L.ref_walk_file.restype = C.c_int
        L.ref_walk_file.argtypes = [C.c_char_p, C.c_int, u8p, C.c_int, C.POINTER(C.c_uint64),
                                    C.POINTER(C.c_uint64)]
        L.ref_time_roundtrip.restype = C.c_double
        L.ref_time_roundtrip.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p, u8p,
                                         C.POINTER(C.c_double), C.POINTER(C.c_double),
                                         C.POINTER(C.c_uint64)]

    def pack_8x8(self, image, off, stride):
        out = np.full(64 + 16, 0xEE, np.uint8)
        code = self.L.ref_pack_8x8(C.cast(image.ctypes.data + off, u8p), stride, _p(out))
        return code, out[:8 * (code >> 8)].copy(), out

    def pack_8x8_partial(self, image, off, stride, rm, dm):
        out = np.full(64 + 16, 0xEE, np.uint8)
        code = self.L.ref_pack_8x8_partial(C.cast(image.ctypes.data + off, u8p), stride, rm, dm, _p(out))
        return code, out[:8 * (code >> 8)].copy(), out

    def unpack_8x8(self, depth, minval, packed, stride=8, canvas=None, off=0):
        img = np.full(8 * stride + 8, 0xEE, np.uint8) if canvas is None else canvas
        buf = np.zeros(64 + 8, np.uint8)
        buf[:len(packed)] = packed
        self.L.ref_unpack_8x8(depth, minval, _p(buf), stride, C.cast(img.ctypes.data + off, u8p))
        return img

    def unpack_8x8_partial(self, depth, minval, packed, stride, rm, dm, canvas, off=0):
        buf = np.zeros(64 + 8, np.uint8)
        buf[:len(packed)] = packed
        self.L.ref_unpack_8x8_partial(depth, minval, _p(buf), stride, rm, dm,
                                      C.cast(canvas.ctypes.data + off, u8p))
        return canvas

    def pack_image(self, image, W, H):
        out = np.full(max_frame_bytes(W, H) + 64, 0xEE, np.uint8)
        src = np.ascontiguousarray(image).reshape(-1)
        # the reference reads 8 bytes per tile row even for the last one: pad the source
        src = np.concatenate([src, np.zeros(64, np.uint8)])
        n = self.L.ref_pack_image(_p(src), W, H, _p(out))
        assert (out[n:] == 0xEE).all()
        return out[:n].copy()

    def pack_frame(self, index, image, W, H):
        out = np.full(max_frame_bytes(W, H) + 64, 0xEE, np.uint8)
        src = np.concatenate([np.ascontiguousarray(image).reshape(-1), np.zeros(64, np.uint8)])
        n = self.L.ref_pack_frame(index, _p(src), W, H, _p(out))
        assert (out[n:] == 0xEE).all()
        return out[:n].copy()

    def pack_frame_header(self, u64s, index, elapsed_ns):
        out = np.zeros(20, np.uint8)
        assert self.L.ref_pack_frame_header(u64s, index, elapsed_ns, _p(out)) == 20
        return out

    def pack_video_header(self, u64s, height, width, hz):
        out = np.zeros(28, np.uint8)
        assert self.L.ref_pack_video_header(u64s, height, width, hz, _p(out)) == 28
        return out

    def unpack_image(self, packed, W, H, fill=0xEE):
        img = np.full(W * H + 64, fill, np.uint8)
        buf = np.concatenate([np.asarray(packed, np.uint8), np.zeros(64, np.uint8)])
        n = self.L.ref_unpack_image(_p(buf), W, H, _p(img))
        return n, img[:W * H].reshape(H, W)

    def unpack_frame_header(self, packed):
        o = (C.c_uint64 * 3)()
        buf = np.ascontiguousarray(np.asarray(packed, np.uint8)[:20])
        n = self.L.ref_unpack_frame_header(_p(buf), o)
        return n, (o[0], o[1], o[2])

    def unpack_frame(self, packed, W, H, fill=0xEE):
        img = np.full(W * H + 64, fill, np.uint8)
        buf = np.concatenate([np.asarray(packed, np.uint8), np.zeros(64, np.uint8)])
        o = (C.c_uint64 * 3)()
        n = self.L.ref_unpack_frame(_p(buf), W, H, _p(img), o)
        return n, (o[0], o[1], o[2]), img[:W * H].reshape(H, W)

    def unpack_video_header(self, packed):
        o = (C.c_uint64 * 3)()
        hz = C.c_double()
        buf = np.ascontiguousarray(np.asarray(packed, np.uint8)[:28])
        n = self.L.ref_unpack_video_header(_p(buf), o, C.byref(hz))
        return n, (o[0], o[1], o[2], hz.value)

    def walk_file(self, path, frames_buffered, W, H, keep=0):
        img = np.zeros(W * H, np.uint8)
        hw = (C.c_uint64 * 2)()
        last = C.c_uint64()
        n = self.L.ref_walk_file(path.encode(), frames_buffered, _p(img), keep, hw, C.byref(last))
        return n, (hw[0], hw[1]), last.value, img.reshape(H, W)

    def time_roundtrip(self, images, n, W, H, reps=1):
        sp = np.empty(max_frame_bytes(W, H) + 64, np.uint8)
        si = np.empty(W * H + 64, np.uint8)
        te, td, bad = C.c_double(), C.c_double(), C.c_uint64()
        src = np.concatenate([images.reshape(-1), np.zeros(64, np.uint8)])
        t = self.L.ref_time_roundtrip(_p(src), n, W, H, reps, _p(sp), _p(si),
                                      C.byref(te), C.byref(td), C.byref(bad))
        return t, te.value, td.value, bad.value
