/*
 * dbde_oracle.c -- scalar CPU restatement of the DBDE codec.  TEST INFRASTRUCTURE ONLY
 * (see dbde_oracle.h for who may use it and how its parity is pinned).
 *
 * Written from the format description; every function names the reference lines whose
 * observable behaviour it restates.  No SIMD, no type punning: bytes are assembled
 * explicitly little-endian (reference README.md:27).
 */
#include "dbde_oracle.h"

#include <string.h>
#include <time.h>

/* ---- little-endian helpers ------------------------------------------------------------ */

static void put_le32(uint8_t *p, uint32_t v) {
    for (int i = 0; i < 4; i++) p[i] = (uint8_t)(v >> (8 * i));
}
static void put_le64(uint8_t *p, uint64_t v) {
    for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i));
}
static uint32_t get_le32(const uint8_t *p) {
    uint32_t v = 0;
    for (int i = 0; i < 4; i++) v |= (uint32_t)p[i] << (8 * i);
    return v;
}
static uint64_t get_le64(const uint8_t *p) {
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v |= (uint64_t)p[i] << (8 * i);
    return v;
}
static uint64_t f64_bits(double d) {
    uint64_t u;
    memcpy(&u, &d, 8);
    return u;
}
static double bits_f64(uint64_t u) {
    double d;
    memcpy(&d, &u, 8);
    return d;
}

/* Number of bits needed for the range hi-lo: 0 for 0, 1 for 1, 2 for 2..3, ... 8 for 128..255.
 * Restates the decision tree of dbde_util.cpp:48,57,66-68. */
static int bit_length_u8(unsigned range) {
    int n = 0;
    while (range) {
        n++;
        range >>= 1;
    }
    return n;
}

/* ---- tile encode ---------------------------------------------------------------------- */

/* Encode 64 pixels already laid out as a dense 8x8 (row-major).  Emits 8*depth bytes:
 * pixel i (row-major within the tile) occupies bits [i*depth, (i+1)*depth) of a
 * little-endian bitstream (README.md:54,114; dbde_util.cpp:70-101).  depth 8 is the 64
 * min-subtracted bytes verbatim (dbde_util.cpp:57-63); depth 0 emits nothing (:48). */
static uint32_t encode_dense_tile(const uint8_t px[64], uint8_t *target) {
    unsigned lo = 255, hi = 0;
    for (int i = 0; i < 64; i++) {
        if (px[i] < lo) lo = px[i];
        if (px[i] > hi) hi = px[i];
    }
    int depth = bit_length_u8(hi - lo);
    if (depth == 0) return lo;

    uint64_t acc = 0; /* bits not yet written, LSB first */
    int nacc = 0;     /* how many of them */
    for (int i = 0; i < 64; i++) {
        acc |= (uint64_t)(px[i] - lo) << nacc;
        nacc += depth;
        while (nacc >= 8) {
            *target++ = (uint8_t)acc;
            acc >>= 8;
            nacc -= 8;
        }
    }
    /* 64*depth bits is a whole number of bytes, so nothing is left over. */
    return ((uint32_t)depth << 8) | lo;
}

/* dbde_util.cpp:22-103: one full tile read at `stride`; returns (depth<<8)|min. */
uint32_t dbde_oracle_pack_8x8(const uint8_t *image, int stride, uint8_t *target) {
    uint8_t px[64];
    for (int r = 0; r < 8; r++)
        for (int c = 0; c < 8; c++) px[8 * r + c] = image[(ptrdiff_t)r * stride + c];
    return encode_dense_tile(px, target);
}

/* dbde_util.cpp:105-135: constant padding.  Each valid row is extended to the right with its
 * last valid pixel (:116-128), then missing rows repeat the last padded row (:129-132).
 * That is clamp-to-edge addressing. */
uint32_t dbde_oracle_pack_8x8_partial(const uint8_t *image, int stride, int rightmargin,
                                      int downmargin, uint8_t *target) {
    uint8_t px[64];
    int rm = rightmargin > 8 ? 8 : rightmargin;
    int dm = downmargin > 8 ? 8 : downmargin;
    for (int r = 0; r < 8; r++) {
        int rr = r < dm ? r : dm - 1;
        for (int c = 0; c < 8; c++) {
            int cc = c < rm ? c : rm - 1;
            px[8 * r + c] = image[(ptrdiff_t)rr * stride + cc];
        }
    }
    return encode_dense_tile(px, target);
}

/* ---- tile decode ---------------------------------------------------------------------- */

static void decode_dense_tile(int depth, uint8_t minval, const uint8_t *packed, uint8_t px[64]) {
    if (depth == 0) { /* dbde_util.cpp:218-226 */
        memset(px, minval, 64);
        return;
    }
    /* dbde_util.cpp:229-244 (depth 1..7) and :245-277 (add min, byte-wise wrapping add:
     * _mm_add_epi8).  Reads exactly 8*depth bytes. */
    uint64_t acc = 0;
    int nacc = 0;
    unsigned mask = (1u << depth) - 1u;
    for (int i = 0; i < 64; i++) {
        while (nacc < depth) {
            acc |= (uint64_t)(*packed++) << nacc;
            nacc += 8;
        }
        px[i] = (uint8_t)((acc & mask) + minval);
        acc >>= depth;
        nacc -= depth;
    }
}

/* dbde_util.cpp:216-279.  depth must be 0..8. */
void dbde_oracle_unpack_8x8(uint8_t depth, uint8_t minval, const uint8_t *packed, size_t stride,
                            uint8_t *image) {
    uint8_t px[64];
    decode_dense_tile(depth, minval, packed, px);
    for (int r = 0; r < 8; r++) memcpy(image + r * stride, px + 8 * r, 8);
}

/* dbde_util.cpp:281-289: only the valid rightmargin x downmargin region is written. */
void dbde_oracle_unpack_8x8_partial(uint8_t depth, uint8_t minval, const uint8_t *packed,
                                    size_t stride, int rightmargin, int downmargin,
                                    uint8_t *image) {
    uint8_t px[64];
    decode_dense_tile(depth, minval, packed, px);
    for (int r = 0; r < downmargin && r < 8; r++)
        for (int c = 0; c < rightmargin && c < 8; c++) image[r * stride + c] = px[8 * r + c];
}

/* ---- frame encode --------------------------------------------------------------------- */

/* dbde_util.cpp:137-180.  Layout at target:
 *   I32 T | U8 depth[T] | I32 T | U8 min[T] | I32 n64 | U64 data[n64],  T = ceil(W/8)*ceil(H/8),
 * tiles visited row-major (:150-178), payload cursor advances 8*depth per tile (:155). */
size_t dbde_oracle_pack_image(const uint8_t *image, int W, int H, uint8_t *target) {
    int w = (W + 7) / 8, h = (H + 7) / 8;
    int T = w * h;
    uint8_t *depth_arr = target + 4;
    uint8_t *min_arr = target + 8 + T;
    uint8_t *n64_at = target + 8 + 2 * (size_t)T;
    uint8_t *out = target + 12 + 2 * (size_t)T;
    put_le32(target, (uint32_t)T);
    put_le32(target + 4 + T, (uint32_t)T);

    uint32_t n64 = 0;
    int t = 0;
    for (int ty = 0; ty < h; ty++) {
        int dm = H - 8 * ty;
        if (dm > 8) dm = 8;
        for (int tx = 0; tx < w; tx++, t++) {
            int rm = W - 8 * tx;
            if (rm > 8) rm = 8;
            const uint8_t *src = image + (size_t)8 * ty * W + 8 * tx;
            uint32_t code = (rm == 8 && dm == 8)
                                ? dbde_oracle_pack_8x8(src, W, out)
                                : dbde_oracle_pack_8x8_partial(src, W, rm, dm, out);
            uint32_t d = code >> 8;
            depth_arr[t] = (uint8_t)d;
            min_arr[t] = (uint8_t)(code & 0xFF);
            out += 8 * d;
            n64 += d;
        }
    }
    put_le32(n64_at, n64);
    return 12 + 2 * (size_t)T + 8 * (size_t)n64;
}

/* dbde_util.cpp:182-188.  NOTE the third field is written as an IEEE-754 double holding
 * (double)elapsed_ns, not a U64 (SURVEY.md trap T1). */
size_t dbde_oracle_pack_frame_header(const dbde_oracle_frame_header *fh, uint8_t *target) {
    put_le32(target, fh->u64s);
    put_le64(target + 4, fh->index);
    put_le64(target + 12, f64_bits((double)fh->elapsed_ns));
    return 20;
}

/* dbde_util.cpp:190-196: header {2, index, 0} then the image. */
size_t dbde_oracle_pack_frame(uint64_t index, const uint8_t *image, int W, int H, uint8_t *target) {
    dbde_oracle_frame_header fh = {2, index, 0};
    size_t n = dbde_oracle_pack_frame_header(&fh, target);
    return n + dbde_oracle_pack_image(image, W, H, target + n);
}

/* dbde_util.cpp:198-209 (default build: frame_hz as F64; height before width). */
size_t dbde_oracle_pack_video_header(const dbde_oracle_video_header *vh, uint8_t *target) {
    put_le32(target, vh->u64s);
    put_le64(target + 4, vh->height);
    put_le64(target + 12, vh->width);
    put_le64(target + 20, f64_bits(vh->frame_hz));
    return 28;
}

/* ---- frame decode --------------------------------------------------------------------- */

/* dbde_util.cpp:291-328.  Returns 0 (image untouched) unless nb == T, nm == T and
 * n64 == sum(depth) (:295-303); otherwise bytes consumed. */
size_t dbde_oracle_unpack_image(const uint8_t *packed, int W, int H, uint8_t *image) {
    int w = (W + 7) / 8, h = (H + 7) / 8;
    int T = w * h;
    if ((int32_t)get_le32(packed) != T) return 0;
    const uint8_t *depth_arr = packed + 4;
    if ((int32_t)get_le32(packed + 4 + T) != T) return 0;
    const uint8_t *min_arr = packed + 8 + T;
    int32_t n64 = (int32_t)get_le32(packed + 8 + 2 * (size_t)T);
    for (int i = 0; i < T; i++) {
        if (depth_arr[i] > 8) return 0; /* the one intentional deviation, see header */
        n64 -= depth_arr[i];
    }
    if (n64 != 0) return 0;

    const uint8_t *in = packed + 12 + 2 * (size_t)T;
    int t = 0;
    for (int ty = 0; ty < h; ty++) {
        int dm = H - 8 * ty;
        if (dm > 8) dm = 8;
        for (int tx = 0; tx < w; tx++, t++) {
            int rm = W - 8 * tx;
            if (rm > 8) rm = 8;
            uint8_t *dst = image + (size_t)8 * ty * W + 8 * tx;
            if (rm == 8 && dm == 8)
                dbde_oracle_unpack_8x8(depth_arr[t], min_arr[t], in, (size_t)W, dst);
            else
                dbde_oracle_unpack_8x8_partial(depth_arr[t], min_arr[t], in, (size_t)W, rm, dm, dst);
            in += 8 * (size_t)depth_arr[t];
        }
    }
    return (size_t)(in - packed);
}

/* dbde_util.cpp:330-337: 20 bytes; u64s becomes 0xFFFFFFFF unless the field is 2;
 * elapsed_ns = (uint64_t)(double on the wire). */
size_t dbde_oracle_unpack_frame_header(const uint8_t *packed, dbde_oracle_frame_header *out) {
    out->u64s = get_le32(packed);
    out->index = get_le64(packed + 4);
    out->elapsed_ns = (uint64_t)bits_f64(get_le64(packed + 12));
    if (out->u64s != 2) out->u64s = 0xFFFFFFFFu;
    return 20;
}

/* dbde_util.cpp:339-345: the image is decoded whatever the header said; a failed image
 * sets u64s = 0xFFFFFFFF and the cursor stays just past the header (SURVEY.md trap T9). */
size_t dbde_oracle_unpack_frame(const uint8_t *packed, int W, int H, uint8_t *image,
                                dbde_oracle_frame_header *out) {
    size_t adv = dbde_oracle_unpack_frame_header(packed, out);
    size_t n = dbde_oracle_unpack_image(packed + adv, W, H, image);
    if (n == 0)
        out->u64s = 0xFFFFFFFFu;
    else
        adv += n;
    return adv;
}

/* dbde_util.cpp:347-359. */
size_t dbde_oracle_unpack_video_header(const uint8_t *packed, dbde_oracle_video_header *out) {
    out->u64s = get_le32(packed);
    out->height = get_le64(packed + 4);
    out->width = get_le64(packed + 12);
    out->frame_hz = bits_f64(get_le64(packed + 20));
    if (out->u64s != 3) out->u64s = 0xFFFFFFFFu;
    return 28;
}

size_t dbde_oracle_max_frame_bytes(int W, int H) {
    size_t T = (size_t)((W + 7) / 8) * (size_t)((H + 7) / 8);
    return 20 + 12 + 66 * T;
}

/* ---- bounded CPU baseline (bench.py cpu_baseline leg, kind "port") -------------------- */

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double dbde_oracle_time_roundtrip(const uint8_t *images, int n, int W, int H, int reps,
                                  uint8_t *scratch_packed, uint8_t *scratch_image,
                                  double *enc_seconds, double *dec_seconds, uint64_t *mismatch) {
    size_t P = (size_t)W * (size_t)H;
    double te = 0, td = 0;
    uint64_t bad = 0;
    for (int r = 0; r < reps; r++) {
        for (int f = 0; f < n; f++) {
            const uint8_t *img = images + P * (size_t)f;
            double t0 = now_s();
            dbde_oracle_pack_frame((uint64_t)f, img, W, H, scratch_packed);
            double t1 = now_s();
            dbde_oracle_frame_header fh;
            dbde_oracle_unpack_frame(scratch_packed, W, H, scratch_image, &fh);
            double t2 = now_s();
            te += t1 - t0;
            td += t2 - t1;
            if (r == reps - 1)
                for (size_t i = 0; i < P; i++) bad += img[i] != scratch_image[i];
        }
    }
    if (enc_seconds) *enc_seconds = te;
    if (dec_seconds) *dec_seconds = td;
    if (mismatch) *mismatch = bad;
    return te + td;
}
