/*
 * dbde16_oracle.c -- DBDE16, the higher-bit-depth extension the reference's README points at
 * ("Minimum intensities (note: could expand size to handle higher bit depth images!)", README.md:65).
 * TEST INFRASTRUCTURE / SPECIFICATION ONLY: plain scalar C, nothing under dbde-video-cpp_amd/ links it.
 *
 * PARITY UNPINNED: the reference defines no 16-bit format and has no code for one, so there is nothing to be
 * bit-exact against.  What pins this file instead (tests/test_oracle_u16.py): on images whose pixels all fit
 * 8 bits it produces, field for field, what the PINNED 8-bit oracle produces (same depth bytes, same payload
 * words, minima widened to 16 bits), plus round trips and the depth boundaries.
 *
 * The extension changes exactly what README.md:65 names and nothing else:
 *
 *   frame      := frame_header(20 B, unchanged)  frame_data16
 *   frame_data16 := I32 nb (= T) | U8  depth[T]            depth = bit_length(max - min) in 0..16
 *                 | I32 nm (= 2T) | U16 min[T] (little-endian)
 *                 | I32 n64 (= sum(depth)) | U64 data[n64]
 *
 *   * pixels are U16, row-major, pitch W pixels; tiles, their order and the constant padding are the 8-bit
 *     format's (README.md:52; dbde_util.cpp:105-135);
 *   * a tile's payload is its 64 values (p - min), `depth` bits each, least significant first, in `depth` U64
 *     words (64 * depth bits) -- tile row r is still the 8*depth-bit integer at byte r*depth of the payload;
 *   * the second I32 is, as README.md:63 words it, the NUMBER OF BYTES of the minimum array: T says "classic
 *     DBDE", 2T says DBDE16.  A reader needs no other flag, and an 8-bit reader rejects a DBDE16 frame on
 *     its nm == T check (dbde_util.cpp:298-300) instead of mis-decoding it;
 *   * frame bytes = 20 + 12 + 3T + 8*sum(depth); worst case 20 + 12 + 131T.
 *   * decode adds the minimum modulo 2^16 (the 16-bit analogue of the reference's _mm_add_epi8,
 *     dbde_util.cpp:245-277); a depth byte > 16 is rejected.
 */
#include "dbde_oracle.h"

#include <string.h>

static void put32(uint8_t *p, uint32_t v) { for (int i = 0; i < 4; i++) p[i] = (uint8_t)(v >> (8 * i)); }
static uint32_t get32(const uint8_t *p) { uint32_t v = 0; for (int i = 0; i < 4; i++) v |= (uint32_t)p[i] << (8 * i); return v; }

static int bit_length(unsigned range) {
    int n = 0;
    while (range) { n++; range >>= 1; }
    return n;
}

/* 64 pixels (dense 8x8) -> depth, min, 8*depth payload bytes. */
static int encode_tile16(const uint16_t px[64], uint16_t *minval, uint8_t *target) {
    unsigned lo = 65535, hi = 0;
    for (int i = 0; i < 64; i++) {
        if (px[i] < lo) lo = px[i];
        if (px[i] > hi) hi = px[i];
    }
    *minval = (uint16_t)lo;
    const int depth = bit_length(hi - lo);
    uint64_t acc = 0;   /* bits not yet written, LSB first (at most 7 + 16 live bits) */
    int nacc = 0;
    for (int i = 0; i < 64 && depth; i++) {
        acc |= (uint64_t)(px[i] - lo) << nacc;
        nacc += depth;
        while (nacc >= 8) { *target++ = (uint8_t)acc; acc >>= 8; nacc -= 8; }
    }
    return depth;
}

static void decode_tile16(int depth, uint16_t minval, const uint8_t *packed, uint16_t px[64]) {
    uint64_t acc = 0;
    int nacc = 0;
    const uint32_t mask = depth >= 16 ? 0xFFFFu : ((1u << depth) - 1u);
    for (int i = 0; i < 64; i++) {
        while (nacc < depth) { acc |= (uint64_t)(*packed++) << nacc; nacc += 8; }
        px[i] = (uint16_t)(((uint32_t)acc & mask) + minval);   /* modulo 2^16 */
        acc >>= depth;
        nacc -= depth;
    }
}

size_t dbde16_oracle_max_frame_bytes(int W, int H) {
    const size_t T = (size_t)((W + 7) / 8) * (size_t)((H + 7) / 8);
    return 20 + 12 + 131 * T;
}

/* frame_data16 of one image; returns its byte count (12 + 3T + 8*sum(depth)). */
size_t dbde16_oracle_pack_image(const uint16_t *image, int W, int H, uint8_t *target) {
    const int w = (W + 7) / 8, h = (H + 7) / 8;
    const size_t T = (size_t)w * (size_t)h;
    uint8_t *depth_arr = target + 4, *min_arr = target + 8 + T, *data = target + 12 + 3 * T;
    uint32_t n64 = 0;
    put32(target, (uint32_t)T);
    put32(target + 4 + T, (uint32_t)(2 * T));
    for (int ty = 0; ty < h; ty++) {
        for (int tx = 0; tx < w; tx++) {
            uint16_t px[64], mn;
            for (int r = 0; r < 8; r++) {          /* constant padding = clamp-to-edge addressing */
                const int yy = 8 * ty + r < H ? 8 * ty + r : H - 1;
                for (int c = 0; c < 8; c++) {
                    const int xx = 8 * tx + c < W ? 8 * tx + c : W - 1;
                    px[8 * r + c] = image[(size_t)yy * (size_t)W + (size_t)xx];
                }
            }
            const int d = encode_tile16(px, &mn, data + 8 * (size_t)n64);
            const size_t t = (size_t)ty * (size_t)w + (size_t)tx;
            depth_arr[t] = (uint8_t)d;
            min_arr[2 * t] = (uint8_t)mn;
            min_arr[2 * t + 1] = (uint8_t)(mn >> 8);
            n64 += (uint32_t)d;
        }
    }
    put32(target + 8 + 3 * T, n64);
    return 12 + 3 * T + 8 * (size_t)n64;
}

size_t dbde16_oracle_pack_frame(uint64_t index, const uint16_t *image, int W, int H, uint8_t *target) {
    dbde_oracle_frame_header fh = {2, index, 0};
    const size_t n = dbde_oracle_pack_frame_header(&fh, target);   /* the frame header is the 8-bit format's */
    return n + dbde16_oracle_pack_image(image, W, H, target + n);
}

/* Returns bytes consumed, 0 when the frame data does not validate (image untouched). */
size_t dbde16_oracle_unpack_image(const uint8_t *packed, int W, int H, uint16_t *image) {
    const int w = (W + 7) / 8, h = (H + 7) / 8;
    const size_t T = (size_t)w * (size_t)h;
    if (get32(packed) != (uint32_t)T) return 0;
    if (get32(packed + 4 + T) != (uint32_t)(2 * T)) return 0;   /* T here would be a classic 8-bit frame */
    const uint8_t *depth_arr = packed + 4, *min_arr = packed + 8 + T, *data = packed + 12 + 3 * T;
    uint64_t sum = 0;
    for (size_t t = 0; t < T; t++) {
        if (depth_arr[t] > 16) return 0;
        sum += depth_arr[t];
    }
    if (get32(packed + 8 + 3 * T) != (uint32_t)sum) return 0;
    size_t at = 0;
    for (int ty = 0; ty < h; ty++) {
        for (int tx = 0; tx < w; tx++) {
            const size_t t = (size_t)ty * (size_t)w + (size_t)tx;
            uint16_t px[64];
            decode_tile16(depth_arr[t], (uint16_t)(min_arr[2 * t] | (min_arr[2 * t + 1] << 8)), data + 8 * at, px);
            at += depth_arr[t];
            for (int r = 0; r < 8 && 8 * ty + r < H; r++)     /* only the valid region is written */
                for (int c = 0; c < 8 && 8 * tx + c < W; c++)
                    image[(size_t)(8 * ty + r) * (size_t)W + (size_t)(8 * tx + c)] = px[8 * r + c];
        }
    }
    return 12 + 3 * T + 8 * (size_t)sum;
}
