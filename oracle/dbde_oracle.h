/*
 * dbde_oracle.h -- CPU restatement of the DBDE frame codec.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity checker for the HIP path.  It restates, in plain scalar C written
 * from the format description (reference README.md:8-67) and the observable behaviour
 * of the reference implementation (reference dbde_util.cpp), what each hot-path function
 * of dbde_util.h computes.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  Nothing under dbde-video-cpp_amd/ links or calls it.
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks it byte-for-byte against
 *   (1) the reference's own known-answer vector (dbde_util_test.cpp:135-178),
 *   (2) fixtures in tests/golden/ produced by the real reference (oracle/_ref, built by
 *       oracle/Makefile from /root/reference/dbde_util.cpp) with tests/golden/make_golden.py,
 *   (3) when oracle/_ref/libdbde_ref.so is present, live randomized differential runs.
 *
 * One intentional deviation from the reference, reachable only on malformed input:
 * a depth byte > 8 makes dbde_oracle_unpack_image return 0 (the reference does not
 * validate it, dbde_util.cpp:229-244, and reads/writes out of bounds).
 */
#ifndef DBDE_ORACLE_H
#define DBDE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* In-memory headers: same field order/types as reference dbde_util.h:8-19. */
typedef struct {
    uint32_t u64s;
    uint64_t height;
    uint64_t width;
    double   frame_hz;
} dbde_oracle_video_header;

typedef struct {
    uint32_t u64s;
    uint64_t index;
    uint64_t elapsed_ns;
} dbde_oracle_frame_header;

/* Tile level (reference dbde_util.h:21-22,30-31). */
uint32_t dbde_oracle_pack_8x8(const uint8_t *image, int stride, uint8_t *target);
uint32_t dbde_oracle_pack_8x8_partial(const uint8_t *image, int stride, int rightmargin,
                                      int downmargin, uint8_t *target);
void dbde_oracle_unpack_8x8(uint8_t depth, uint8_t minval, const uint8_t *packed, size_t stride,
                            uint8_t *image);
void dbde_oracle_unpack_8x8_partial(uint8_t depth, uint8_t minval, const uint8_t *packed,
                                    size_t stride, int rightmargin, int downmargin,
                                    uint8_t *image);

/* Frame level (reference dbde_util.h:24-28,33-37).  Headers are passed by pointer here
 * (plain C ABI for ctypes); the wire bytes are what is compared. */
size_t dbde_oracle_pack_image(const uint8_t *image, int W, int H, uint8_t *target);
size_t dbde_oracle_pack_frame_header(const dbde_oracle_frame_header *fh, uint8_t *target);
size_t dbde_oracle_pack_frame(uint64_t index, const uint8_t *image, int W, int H, uint8_t *target);
size_t dbde_oracle_pack_video_header(const dbde_oracle_video_header *vh, uint8_t *target);

size_t dbde_oracle_unpack_image(const uint8_t *packed, int W, int H, uint8_t *image);
/* The three unpackers return the number of bytes the reference would have advanced
 * *packed by, and fill *out. */
size_t dbde_oracle_unpack_frame_header(const uint8_t *packed, dbde_oracle_frame_header *out);
size_t dbde_oracle_unpack_frame(const uint8_t *packed, int W, int H, uint8_t *image,
                                dbde_oracle_frame_header *out);
size_t dbde_oracle_unpack_video_header(const uint8_t *packed, dbde_oracle_video_header *out);

/* Worst-case bytes of one packed frame: 20 + 12 + 66*T. */
size_t dbde_oracle_max_frame_bytes(int W, int H);

/* Synthetic frame generators shared (as a specification) with the HIP generator
 * dbde_hip_synth_frames: mode 0 = noise8, 1 = mixed, 2 = flat, 3 = smooth.  See synth.c. */
void dbde_oracle_synth_frame(int mode, uint64_t seed, uint64_t frame, int W, int H, uint8_t *image);

/* Bounded CPU-baseline helper for bench.py: encode+decode `n` frames (each W*H bytes,
 * contiguous) `reps` times on the calling thread; returns elapsed seconds and writes the
 * number of mismatching pixels of the last round trip to *mismatch. */
double dbde_oracle_time_roundtrip(const uint8_t *images, int n, int W, int H, int reps,
                                  uint8_t *scratch_packed, uint8_t *scratch_image,
                                  double *enc_seconds, double *dec_seconds, uint64_t *mismatch);

/* ---- DBDE16: the higher-bit-depth extension README.md:65 points at (oracle/dbde16_oracle.c holds the
 * specification).  PARITY UNPINNED: the reference defines no such format; tied to the pinned 8-bit oracle on
 * images that fit 8 bits (tests/test_oracle_u16.py). */
size_t dbde16_oracle_max_frame_bytes(int W, int H);
size_t dbde16_oracle_pack_image(const uint16_t *image, int W, int H, uint8_t *target);
size_t dbde16_oracle_pack_frame(uint64_t index, const uint16_t *image, int W, int H, uint8_t *target);
size_t dbde16_oracle_unpack_image(const uint8_t *packed, int W, int H, uint16_t *image);

#ifdef __cplusplus
}
#endif
#endif
