"""CPU: the per-lane bit manipulation the kernels run (csrc/dbde_bits.h is host+device code)
compiled with g++ and checked against the oracle on every depth."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include <stdint.h>
#include <string.h>
#include "dbde_bits.h"
using namespace dbde;
// encode + decode one dense 8x8 tile exactly as a kernel lane does; returns depth<<8|min
extern "C" uint32_t lane_tile(const uint8_t *px, uint8_t *payload, uint8_t *back) {
    uint32_t v[16];
    memcpy(v, px, 64);
    uint32_t mn, mx;
    tile_minmax(v, mn, mx);
    const uint32_t d = depth_of_range(mx - mn), m4 = mn * 0x01010101u;
    Funnel fn; fn.reset();
    uint64_t words[8]; int nw = 0;
    for (int r = 0; r < 8; r++) {
        uint64_t row = pack_row(v[2*r] - m4, v[2*r+1] - m4, d), w;
        if (fn.push(row, 8u*d, w)) words[nw++] = w;
    }
    if (nw != (int)d || fn.fill != 0) return 0xFFFFFFFFu;
    memcpy(payload, words, 8*d);
    // decode: row r is the 8d-bit integer at byte r*d
    uint8_t buf[64 + 16] = {0};
    memcpy(buf, payload, 8*d);
    for (int r = 0; r < 8; r++) {
        uint64_t row; memcpy(&row, buf + r*d, 8);
        if (d < 8) row &= (1ull << (8*d)) - 1;
        uint32_t lo, hi; expand_row(row, d, lo, hi);
        lo = add_bytes(lo, m4); hi = add_bytes(hi, m4);
        memcpy(back + 8*r, &lo, 4); memcpy(back + 8*r + 4, &hi, 4);
    }
    return (d << 8) | mn;
}
extern "C" uint32_t add_bytes_c(uint32_t a, uint32_t b) { return add_bytes(a, b); }
'''


def test_lane_code_matches_oracle(tmp_path, oracle):
    import ctypes as C
    src = tmp_path / "bits.cpp"
    src.write_text(SRC)
    so = tmp_path / "bits.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "dbde-video-cpp_amd", "csrc"),
                    "-o", str(so), str(src)], check=True)
    L = C.CDLL(str(so))
    L.lane_tile.restype = C.c_uint32
    L.lane_tile.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5)
    for it in range(3000):
        d = it % 9
        lo = int(rng.integers(0, 257 - (1 << d)))
        px = (lo + rng.integers(0, 1 << d, 64)).astype(np.uint8)
        if it % 7 == 0:
            px[rng.integers(0, 64)] = lo
        payload, back = np.zeros(64, np.uint8), np.zeros(64, np.uint8)
        code = L.lane_tile(px.ctypes.data, payload.ctypes.data, back.ctypes.data)
        wcode, wpayload, _ = oracle.pack_8x8(px, 0, 8)
        assert code == wcode, (it, code, wcode)
        assert payload[:8 * (code >> 8)].tobytes() == wpayload.tobytes()
        assert (back == px).all()
    L.add_bytes_c.restype = C.c_uint32
    L.add_bytes_c.argtypes = [C.c_uint32, C.c_uint32]
    for _ in range(2000):
        a, b = int(rng.integers(0, 2**32)), int(rng.integers(0, 2**32))
        want = sum((((a >> s) & 255) + ((b >> s) & 255)) % 256 << s for s in (0, 8, 16, 24))
        assert L.add_bytes_c(a, b) == want
