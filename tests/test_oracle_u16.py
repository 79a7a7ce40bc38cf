"""CPU: DBDE16, the higher-bit-depth extension README.md:65 hints at (specification: oracle/dbde16_oracle.c).
PARITY UNPINNED -- the reference defines no 16-bit format.  What holds it in place instead:
  * on images that fit 8 bits it agrees field for field with the PINNED 8-bit oracle (same depth bytes, same
    payload words; minima widened to 16 bits, nm = 2T);
  * depth boundaries 0..16, round trips on full-range 16-bit data incl. edge tiles and wrap-around minima;
  * validation: an 8-bit frame (nm = T), a depth byte > 16 and a wrong n64 are rejected, image untouched."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle_ffi import ORACLE_SO, Oracle

u8p, u16p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint16)


@pytest.fixture(scope="module")
def o16():
    Oracle()   # builds liboracle.so when missing
    L = C.CDLL(ORACLE_SO)
    L.dbde16_oracle_max_frame_bytes.restype = C.c_size_t
    L.dbde16_oracle_max_frame_bytes.argtypes = [C.c_int, C.c_int]
    L.dbde16_oracle_pack_frame.restype = C.c_size_t
    L.dbde16_oracle_pack_frame.argtypes = [C.c_uint64, u16p, C.c_int, C.c_int, u8p]
    L.dbde16_oracle_pack_image.restype = C.c_size_t
    L.dbde16_oracle_pack_image.argtypes = [u16p, C.c_int, C.c_int, u8p]
    L.dbde16_oracle_unpack_image.restype = C.c_size_t
    L.dbde16_oracle_unpack_image.argtypes = [u8p, C.c_int, C.c_int, u16p]
    return L


def pack16(L, img, index=0):
    H, W = img.shape
    img = np.ascontiguousarray(img, np.uint16)
    out = np.full(L.dbde16_oracle_max_frame_bytes(W, H) + 16, 0xEE, np.uint8)
    n = L.dbde16_oracle_pack_frame(index, img.ctypes.data_as(u16p), W, H, out.ctypes.data_as(u8p))
    assert (out[n:] == 0xEE).all()
    return out[:n].copy()


def unpack16(L, frame, W, H, fill=0xEEEE):
    img = np.full((H, W), fill, np.uint16)
    body = np.ascontiguousarray(np.concatenate([frame[20:], np.zeros(16, np.uint8)]))
    n = L.dbde16_oracle_unpack_image(body.ctypes.data_as(u8p), W, H, img.ctypes.data_as(u16p))
    return n, img


@pytest.mark.parametrize("W,H", [(8, 8), (10, 10), (33, 17), (64, 40), (1, 1), (7, 300)])
def test_agrees_with_the_pinned_8bit_oracle_on_8bit_images(o16, oracle, W, H):
    rng = np.random.default_rng(W * 1000 + H)
    T = ((W + 7) // 8) * ((H + 7) // 8)
    for trial in range(4):
        depth = rng.integers(0, 9, size=(H + 7) // 8 * 8 * ((W + 7) // 8 * 8)).reshape((H + 7) // 8 * 8, -1)
        img8 = (rng.integers(0, 256, size=(H, W)) >> (8 - rng.integers(0, 9))).astype(np.uint8)
        f8 = oracle.pack_frame(5, img8, W, H)
        f16 = pack16(o16, img8.astype(np.uint16), 5)
        assert f16[:24].tobytes() == f8[:24].tobytes()                               # header + nb
        assert f16[24:24 + T].tobytes() == f8[24:24 + T].tobytes()                   # depth bytes
        assert int(f16[24 + T:28 + T].view("<u4")[0]) == 2 * T                       # nm = bytes of the minima
        assert (f16[28 + T:28 + 3 * T].view("<u2") == f8[28 + T:28 + 2 * T]).all()   # minima, widened
        assert f16[28 + 3 * T:].tobytes() == f8[28 + 2 * T:].tobytes()               # n64 + payload words
        n, back = unpack16(o16, f16, W, H)
        assert n == len(f16) - 20 and (back == img8).all()


def test_depth_boundaries_and_full_range(o16):
    for d in range(17):
        tile = np.full((8, 8), 1000, np.uint16)
        if d:
            tile[0, 1] = 1000 + (1 << (d - 1))      # smallest range that needs d bits
            tile[7, 7] = 1000 + (1 << d) - 1 if 1000 + (1 << d) - 1 < 65536 else 65535
        f = pack16(o16, tile)
        assert f[24] == d and len(f) == 20 + 12 + 3 + 8 * d
        n, back = unpack16(o16, f, 8, 8)
        assert n == len(f) - 20 and (back == tile).all()
    rng = np.random.default_rng(7)
    for (W, H) in [(17, 9), (200, 123), (1921, 16)]:
        img = rng.integers(0, 65536, size=(H, W)).astype(np.uint16)
        img[: H // 2] >>= 6     # a mix of depths
        f = pack16(o16, img, 9)
        n, back = unpack16(o16, f, W, H)
        assert n == len(f) - 20 and (back == img).all()


def test_validation_and_wraparound(o16, oracle):
    W, H = 24, 16
    T = 6
    img = (np.arange(W * H).reshape(H, W) * 37 % 65536).astype(np.uint16)
    f = pack16(o16, img)
    for mutate in ("depth", "nm", "n64", "nb"):
        g = f.copy()
        if mutate == "depth":
            g[24] = 17
        elif mutate == "nm":
            g[24 + T:28 + T] = np.frombuffer(np.uint32(T).tobytes(), np.uint8)   # what a classic 8-bit frame says
        elif mutate == "n64":
            g[28 + 3 * T] ^= 1
        else:
            g[20] ^= 1
        n, back = unpack16(o16, g, W, H)
        assert n == 0 and (back == 0xEEEE).all(), mutate
    # an 8-bit reader rejects a DBDE16 frame outright (nm != T) instead of mis-decoding it
    adv, fh, img8 = oracle.unpack_frame(f, W, H)
    assert fh[0] == 0xFFFFFFFF and adv == 20
    # crafted minimum + value beyond 65535 wraps modulo 2^16
    g = pack16(o16, np.array([[65535, 65534] + [65535] * 6] * 8, np.uint16))
    g[28 + 1:28 + 3] = [0xFF, 0xFF]      # minimum 65535, depth 1, payload bit set for the 65535 pixels
    n, back = unpack16(o16, g, 8, 8)
    assert n and back[0, 1] == 65535 and back[0, 0] == 0
