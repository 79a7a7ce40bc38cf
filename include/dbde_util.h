/*
 * dbde_util.h -- drop-in replacement for the reference's public header (dbde_util.h:1-54),
 * backed by the MI355X HIP codec in libdbde_hip.so.
 *
 * Same structs, same C++ function names and signatures, hence the same Itanium-mangled
 * symbols as the reference's dbde_util.o: code written against the
 * reference -- including its own dbde_util_test.cpp -- compiles and links unchanged against
 * libdbde_util_hip.so (csrc/dbde_util_shim.cpp).  Unlike the reference header this one
 * includes what it needs (the reference requires the includer to pull in <stdint.h> and
 * <stdio.h> first; doing so again is harmless).
 *
 * Every codec function below forwards to the C-ABI entry point of the same name in
 * include/dbde_hip.h (dbde_pack_frame -> dbde_hip_pack_frame, ...), using one lazily
 * created process-wide context on HIP device $DBDE_HIP_DEVICE (default 0).  If no gfx950
 * device is usable the first call prints a diagnostic and aborts: there is no CPU path.
 */
#ifndef DBDE_UTIL
#define DBDE_UTIL

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

/* ---- headers: field-for-field the reference structs (dbde_util.h:8-19) ------------------ */

struct video_header {
    uint32_t u64s;     /* number of 8-byte fields that follow on the wire: always 3 */
    uint64_t height;
    uint64_t width;
    double frame_hz;
};

struct frame_header {
    uint32_t u64s;     /* always 2; readers set 0xFFFFFFFF on a bad header or bad frame data */
    uint64_t index;
    uint64_t elapsed_ns;   /* stored on the wire as an IEEE-754 double (dbde_util.cpp:186) */
};

/* ---- encode (reference dbde_util.h:21-28) ----------------------------------------------- */

/* One full 8x8 tile read at `stride`; returns (depth<<8)|min and writes 8*depth bytes. */
uint32_t dbde_pack_8x8(uint8_t *image, int stride, uint8_t *target);
/* Edge tile, constant-padded from its valid rightmargin x downmargin pixels. */
uint32_t dbde_pack_8x8_partial(uint8_t *image, int stride, int rightmargin, int downmargin,
                               uint8_t *target);
/* Frame data: I32 T | depth[T] | I32 T | min[T] | I32 n64 | U64 data[n64]; returns its size. */
size_t dbde_pack_image(uint8_t *image, int W, int H, uint8_t *target);
/* 20-byte frame header. */
size_t dbde_pack_frame_header(frame_header fh, uint8_t *target);
/* Frame header {2, index, 0} followed by the frame data; returns total bytes. */
size_t dbde_pack_frame(uint64_t index, uint8_t *image, int W, int H, uint8_t *target);
/* 28-byte video header (height before width). */
size_t dbde_pack_video_header(video_header vh, uint8_t *target);

/* ---- decode (reference dbde_util.h:30-37) ----------------------------------------------- */

void dbde_unpack_8x8(uint8_t depth, uint8_t minval, uint8_t *packed, size_t stride, uint8_t *image);
void dbde_unpack_8x8_partial(uint8_t depth, uint8_t minval, uint8_t *packed, size_t stride,
                             int rightmargin, int downmargin, uint8_t *image);
/* Returns bytes consumed, or 0 (image untouched) when the frame data does not validate. */
size_t dbde_unpack_image(uint8_t *packed, int W, int H, uint8_t *image);
/* The three below advance *packed past what they consumed. */
frame_header dbde_unpack_frame_header(uint8_t **packed);
frame_header dbde_unpack_frame(uint8_t **packed, int W, int H, uint8_t *image);
video_header dbde_unpack_video_header(uint8_t **packed);

/* ---- file walker (reference dbde_util.h:39-52) ------------------------------------------ */

struct dbde_file_walker {
    FILE *fptr;       /* file being read; NULL once closed or on error */
    int32_t frames;   /* kept for layout compatibility (the reference never updates it) */
    size_t i;         /* first unread byte in buffer */
    size_t n;         /* end of valid bytes in buffer */
    size_t N;         /* capacity of buffer */
    int32_t width;
    int32_t height;
    uint8_t *buffer;
};

dbde_file_walker dbde_start_file_walk(const char *name, int frames_buffered, video_header *vh);
bool dbde_walk_a_file(dbde_file_walker *walker, frame_header *fh, uint8_t *image);
void dbde_end_file_walk(dbde_file_walker *walker);
/* Not declared by the reference's header but exported by its object file (dbde_util.cpp:394): refills the walker's
 * byte window; false on a read error. */
bool dbde_advance_file_buffer(dbde_file_walker &w);

#endif
