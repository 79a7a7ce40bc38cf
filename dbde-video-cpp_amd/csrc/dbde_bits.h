// dbde_bits.h -- per-lane bit manipulation of the DBDE tile codec (no memory access).
//
// Everything here is pure integer arithmetic on registers, marked host+device so that the
// exact code the HIP kernels run can also be exercised on the CPU by
// tests/test_bits_host.py (compiled with g++, compared against the oracle).
//
// Format facts used (reference README.md:52-54, dbde_util.cpp:70-101, 229-244):
//   * a tile is 64 pixels, row-major; pixel i occupies bits [i*depth, (i+1)*depth) of an
//     LSB-first little-endian bitstream of 8*depth bytes;
//   * therefore tile row r (8 pixels) is the self-contained 8*depth-bit integer at bytes
//     [r*depth, (r+1)*depth) of the tile payload.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DBDE_HD __host__ __device__ __forceinline__
#else
#define DBDE_HD inline
#endif

namespace dbde {

// bit_length(hi - lo): 0 for 0, 1 for 1, 2 for 2..3, ... 8 for 128..255 (dbde_util.cpp:48,57,66-68).
DBDE_HD uint32_t depth_of_range(uint32_t range) {
    return range ? 32u - (uint32_t)__builtin_clz(range) : 0u;
}

// Four bytes (each < 2^d, d in 0..8) -> the 4d-bit field b0 | b1<<d | b2<<2d | b3<<3d.
DBDE_HD uint32_t pack4(uint32_t a, uint32_t d) {
    uint32_t even = a & 0x00FF00FFu;
    uint32_t odd = (a >> 8) & 0x00FF00FFu;
    uint32_t f = even | (odd << d);                    // two 2d-bit fields in 16-bit lanes
    return (f & 0xFFFFu) | ((f >> 16) << (2u * d));    // one 4d-bit field
}

// One tile row: 8 min-subtracted bytes (lo = pixels 0..3, hi = pixels 4..7) -> 8d-bit integer.
DBDE_HD uint64_t pack_row(uint32_t lo, uint32_t hi, uint32_t d) {
    return (uint64_t)pack4(lo, d) | ((uint64_t)pack4(hi, d) << (4u * d));
}

// Inverse of pack4 for a 4d-bit field g.
DBDE_HD uint32_t expand4(uint32_t g, uint32_t d) {
    uint32_t m1 = ((1u << d) - 1u) * 0x00010001u;      // d <= 8
    uint32_t m2 = (1u << (2u * d)) - 1u;               // 2d <= 16
    uint32_t f = (g & m2) | (((g >> (2u * d)) & m2) << 16);
    return (f & m1) | (((f >> d) & m1) << 8);
}

// Inverse of pack_row: the low 8d bits of `row` -> 8 bytes.  Bits above 8d are ignored.
DBDE_HD void expand_row(uint64_t row, uint32_t d, uint32_t &lo, uint32_t &hi) {
    uint32_t m4 = (uint32_t)((1ull << (4u * d)) - 1ull);   // 4d <= 32
    lo = expand4((uint32_t)row & m4, d);
    hi = expand4((uint32_t)(row >> (4u * d)) & m4, d);
}

// Byte-wise wrapping add of four bytes (the reference adds the minimum with _mm_add_epi8,
// dbde_util.cpp:245-277, so a crafted min+value > 255 wraps inside its byte).
DBDE_HD uint32_t add_bytes(uint32_t a, uint32_t b) {
    return ((a & 0x7F7F7F7Fu) + (b & 0x7F7F7F7Fu)) ^ ((a ^ b) & 0x80808080u);
}

// Bit funnel used by the encoder to concatenate the eight row integers of a tile into
// whole U64 words.  push() returns true when a full word is ready in `out`.
struct Funnel {
    uint64_t acc;
    uint32_t fill;   // valid bits in acc, always < 64
    DBDE_HD void reset() { acc = 0; fill = 0; }
    DBDE_HD bool push(uint64_t bits, uint32_t nbits, uint64_t &out) {   // nbits in 0..64
        uint64_t merged = acc | (bits << fill);
        uint32_t nf = fill + nbits;
        if (nf >= 64u) {
            out = merged;
            acc = fill ? (bits >> (64u - fill)) : 0ull;
            fill = nf - 64u;
            return true;
        }
        acc = merged;
        fill = nf;
        return false;
    }
};

// Packed-u16 min/max over the 16 dwords (64 bytes) of one tile.  Even bytes are isolated
// with a mask; odd bytes are compared through the high byte of each u16 (the u16 min/max
// is decided by the high byte first, so its high byte IS the min/max of the odd bytes).
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
DBDE_HD uint32_t pk_min_u16(uint32_t a, uint32_t b) {   // v_pk_min_u16
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2_t, a),
                                                                  __builtin_bit_cast(u16x2_t, b)));
}
DBDE_HD uint32_t pk_max_u16(uint32_t a, uint32_t b) {   // v_pk_max_u16
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, a),
                                                                  __builtin_bit_cast(u16x2_t, b)));
}
#else
DBDE_HD uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    uint32_t lo = ((a & 0xFFFFu) < (b & 0xFFFFu)) ? (a & 0xFFFFu) : (b & 0xFFFFu);
    uint32_t hi = ((a >> 16) < (b >> 16)) ? (a >> 16) : (b >> 16);
    return lo | (hi << 16);
}
DBDE_HD uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    uint32_t lo = ((a & 0xFFFFu) > (b & 0xFFFFu)) ? (a & 0xFFFFu) : (b & 0xFFFFu);
    uint32_t hi = ((a >> 16) > (b >> 16)) ? (a >> 16) : (b >> 16);
    return lo | (hi << 16);
}
#endif

// min and max over the 64 bytes held in 16 dwords.
DBDE_HD void tile_minmax(const uint32_t (&v)[16], uint32_t &mn, uint32_t &mx) {
    uint32_t e = v[0] & 0x00FF00FFu;
    uint32_t emin = e, emax = e, omin = v[0], omax = v[0];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 1; i < 16; i++) {
        e = v[i] & 0x00FF00FFu;
        emin = pk_min_u16(emin, e);
        emax = pk_max_u16(emax, e);
        omin = pk_min_u16(omin, v[i]);
        omax = pk_max_u16(omax, v[i]);
    }
    uint32_t a = emin & 0xFFFFu, b = emin >> 16, c = (omin >> 8) & 0xFFu, d = omin >> 24;
    uint32_t m1 = a < b ? a : b, m2 = c < d ? c : d;
    mn = m1 < m2 ? m1 : m2;
    a = emax & 0xFFFFu; b = emax >> 16; c = (omax >> 8) & 0xFFu; d = omax >> 24;
    m1 = a > b ? a : b; m2 = c > d ? c : d;
    mx = m1 > m2 ? m1 : m2;
}

}  // namespace dbde
