/*
 * ref_api.cpp -- extern "C" doorway to the REAL reference, for tests and the CPU baseline.
 * TEST INFRASTRUCTURE ONLY.
 *
 * Compiled by oracle/Makefile together with the reference's own dbde_util.cpp, taken where
 * it lies under $(REF) (= /root/reference), with the reference makefile's flags
 * (makefile:6,9: -O3 -std=c++14 -march=corei7), into oracle/_ref/libdbde_ref.so.  No
 * reference source is copied into this repository; this file only declares thin C wrappers
 * around the C++-mangled API of the reference's dbde_util.h so Python (ctypes) can call it.
 */
#include <smmintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "dbde_util.h" /* found through -I$(REF) */

extern "C" {

uint32_t ref_pack_8x8(uint8_t *image, int stride, uint8_t *target) {
    return dbde_pack_8x8(image, stride, target);
}
uint32_t ref_pack_8x8_partial(uint8_t *image, int stride, int rm, int dm, uint8_t *target) {
    return dbde_pack_8x8_partial(image, stride, rm, dm, target);
}
void ref_unpack_8x8(uint8_t depth, uint8_t minval, uint8_t *packed, size_t stride, uint8_t *image) {
    dbde_unpack_8x8(depth, minval, packed, stride, image);
}
void ref_unpack_8x8_partial(uint8_t depth, uint8_t minval, uint8_t *packed, size_t stride, int rm,
                            int dm, uint8_t *image) {
    dbde_unpack_8x8_partial(depth, minval, packed, stride, rm, dm, image);
}
size_t ref_pack_image(uint8_t *image, int W, int H, uint8_t *target) {
    return dbde_pack_image(image, W, H, target);
}
size_t ref_pack_frame_header(uint32_t u64s, uint64_t index, uint64_t elapsed_ns, uint8_t *target) {
    frame_header fh;
    fh.u64s = u64s;
    fh.index = index;
    fh.elapsed_ns = elapsed_ns;
    return dbde_pack_frame_header(fh, target);
}
size_t ref_pack_frame(uint64_t index, uint8_t *image, int W, int H, uint8_t *target) {
    return dbde_pack_frame(index, image, W, H, target);
}
size_t ref_pack_video_header(uint32_t u64s, uint64_t height, uint64_t width, double hz,
                             uint8_t *target) {
    video_header vh;
    vh.u64s = u64s;
    vh.height = height;
    vh.width = width;
    vh.frame_hz = hz;
    return dbde_pack_video_header(vh, target);
}
size_t ref_unpack_image(uint8_t *packed, int W, int H, uint8_t *image) {
    return dbde_unpack_image(packed, W, H, image);
}
/* Unpackers: return bytes advanced; fields come back through out[3] = {u64s, a, b}. */
size_t ref_unpack_frame_header(uint8_t *packed, uint64_t *out) {
    uint8_t *p = packed;
    frame_header fh = dbde_unpack_frame_header(&p);
    out[0] = fh.u64s;
    out[1] = fh.index;
    out[2] = fh.elapsed_ns;
    return (size_t)(p - packed);
}
size_t ref_unpack_frame(uint8_t *packed, int W, int H, uint8_t *image, uint64_t *out) {
    uint8_t *p = packed;
    frame_header fh = dbde_unpack_frame(&p, W, H, image);
    out[0] = fh.u64s;
    out[1] = fh.index;
    out[2] = fh.elapsed_ns;
    return (size_t)(p - packed);
}
size_t ref_unpack_video_header(uint8_t *packed, uint64_t *out_u, double *out_hz) {
    uint8_t *p = packed;
    video_header vh = dbde_unpack_video_header(&p);
    out_u[0] = vh.u64s;
    out_u[1] = vh.height;
    out_u[2] = vh.width;
    *out_hz = vh.frame_hz;
    return (size_t)(p - packed);
}

/* File walker (dbde_util.cpp:362-426): count frames, optionally keep frame `keep` (1-based). */
int ref_walk_file(const char *name, int frames_buffered, uint8_t *image, int keep, uint64_t *hw_out,
                  uint64_t *last_index) {
    video_header vh;
    frame_header fh;
    dbde_file_walker w = dbde_start_file_walk(name, frames_buffered, &vh);
    if (!w.fptr) return -1;
    hw_out[0] = vh.height;
    hw_out[1] = vh.width;
    uint8_t *tmp = (uint8_t *)malloc((size_t)vh.height * vh.width + 64);
    int n = 0;
    while (dbde_walk_a_file(&w, &fh, tmp)) {
        n++;
        *last_index = fh.index;
        if (n == keep && image) memcpy(image, tmp, (size_t)vh.height * vh.width);
    }
    dbde_end_file_walk(&w);
    free(tmp);
    free(w.buffer);
    return n;
}

/* CPU baseline (bench.py cpu_baseline leg, kind "reference"): same contract as
 * dbde_oracle_time_roundtrip. */
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
double ref_time_roundtrip(uint8_t *images, int n, int W, int H, int reps, uint8_t *scratch_packed,
                          uint8_t *scratch_image, double *enc_seconds, double *dec_seconds,
                          uint64_t *mismatch) {
    size_t P = (size_t)W * (size_t)H;
    double te = 0, td = 0;
    uint64_t bad = 0;
    for (int r = 0; r < reps; r++) {
        for (int f = 0; f < n; f++) {
            uint8_t *img = images + P * (size_t)f;
            double t0 = now_s();
            dbde_pack_frame((uint64_t)f, img, W, H, scratch_packed);
            double t1 = now_s();
            uint8_t *p = scratch_packed;
            dbde_unpack_frame(&p, W, H, scratch_image);
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

} /* extern "C" */
