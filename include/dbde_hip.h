/*
 * dbde_hip.h -- C-ABI of the MI355X (gfx950) DBDE frame codec: libdbde_hip.so.
 *
 * This is the drop-in boundary for the reference's hot path, dbde_util.h:21-37
 * (dbde_pack_* / dbde_unpack_*).  Plain C: pointers, sizes and fixed-width integers only.
 * Two layers live behind it:
 *
 *   1. the batch API on DEVICE-RESIDENT buffers (dbde_hip_encode_frames /
 *      dbde_hip_decode_frames): N frames per launch, what bench.py measures;
 *   2. host-pointer entry points with the argument meaning of the reference functions they
 *      replace (each cites its reference line); include/dbde_util.h re-exports them under
 *      the reference's own C++ names, so code written against the reference links unchanged.
 *
 * All compute runs in hand-written HIP kernels (csrc/dbde_kernels.hip).  There is no CPU
 * fallback: every entry point fails with DBDE_HIP_ERR_HIP when no gfx950 device is usable.
 *
 * Wire format (all little-endian; reference README.md:12-67, dbde_util.cpp:137-209):
 *   stream := video_header(28 B) { frame_header(20 B) frame_data }*
 *   frame_data := I32 T | U8 depth[T] | I32 T | U8 min[T] | I32 n64 | U64 data[n64]
 *   T = ceil(W/8)*ceil(H/8) 8x8 tiles, row-major; n64 = sum(depth)
 */
#ifndef DBDE_HIP_H
#define DBDE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dbde_hip_ctx dbde_hip_ctx;

enum {
    DBDE_HIP_OK = 0,
    DBDE_HIP_ERR_ARG = -1,       /* bad argument (null pointer, W/H out of range, ...) */
    DBDE_HIP_ERR_HIP = -2,       /* HIP runtime error or no usable gfx950 device */
    DBDE_HIP_ERR_CAPACITY = -3,  /* output buffer smaller than the worst case */
    DBDE_HIP_ERR_DEVICE = -4     /* a kernel reported failure (look-back time-out) */
};

/* In-memory headers, field-for-field the reference's structs (dbde_util.h:8-19). */
typedef struct {
    uint32_t u64s;
    uint64_t height;
    uint64_t width;
    double frame_hz;
} dbde_hip_video_header;

typedef struct {
    uint32_t u64s;       /* 2, or 0xFFFFFFFF when the frame failed to parse (dbde_util.cpp:335,342) */
    uint64_t index;
    uint64_t elapsed_ns;
} dbde_hip_frame_header;

/* Per-frame result of a batch decode (device memory, one per frame). */
typedef struct {
    dbde_hip_frame_header header;  /* as dbde_unpack_frame would return it */
    uint64_t consumed;             /* bytes the reference would advance *packed by (20 on failure) */
} dbde_hip_frame_result;

/* ---- context -------------------------------------------------------------------------- */

/* Creates a context on HIP device `device`.  `stream` is a hipStream_t (or NULL for the
 * default stream) on which every kernel and copy of this context is enqueued; the caller
 * keeps ownership of it.  Workspace (look-back state, decode index, staging buffers for the
 * host-pointer entry points) is owned by the context and grown on demand. */
int dbde_hip_create(int device, void *stream, dbde_hip_ctx **out);
/* The same with a stream of the context's OWN (non-blocking, destroyed with it): what a caller without the HIP headers
 * needs to run several contexts side by side -- the drop-in shim keeps a pool of these, one per calling thread at a
 * time, so that the reference's re-entrant API (dbde_util.h:21-37: no global state) scales with the caller's threads. */
int dbde_hip_create_on_own_stream(int device, dbde_hip_ctx **out);
/* How the host-pointer entry points (dbde_hip_pack_frame ...) move the caller's bytes: 0 (default) straight from / to
 * the caller's pageable memory (the runtime pins it per call: fastest for ONE caller, but that pinning serialises
 * concurrent callers), 1 through pinned buffers of the context (a memcpy by the calling thread each way, DMA that never
 * pins: scales with the number of calling threads).  The drop-in shim switches per call by how many calls are in flight. */
int dbde_hip_set_host_staging(dbde_hip_ctx *ctx, int pinned);
void dbde_hip_destroy(dbde_hip_ctx *ctx);
/* Blocks until everything enqueued by this context has finished; returns
 * DBDE_HIP_ERR_DEVICE if a kernel raised its failure flag since the last call. */
int dbde_hip_sync(dbde_hip_ctx *ctx);
const char *dbde_hip_last_error(const dbde_hip_ctx *ctx);
/* Name of the device the context runs on (e.g. "gfx950:sramecc+:xnack-"). */
const char *dbde_hip_device_arch(const dbde_hip_ctx *ctx);
/* The hipStream_t and HIP device index the context was created with. */
void *dbde_hip_stream_handle(const dbde_hip_ctx *ctx);
int dbde_hip_device_index(const dbde_hip_ctx *ctx);

/* ---- sizes ---------------------------------------------------------------------------- */

/* Worst-case bytes of one packed frame incl. its 20-byte header: 20 + 12 + 66*T
 * (the bound the reference's test allocates, dbde_util_test.cpp:78). */
size_t dbde_hip_max_frame_bytes(int W, int H);
/* Exact bytes of frame_data for a frame with n64 payload words: 12 + 2T + 8*n64. */
size_t dbde_hip_image_bytes(int W, int H, uint64_t n64);

/* ---- batch API, device pointers (the measured path) ----------------------------------- */

/* Encodes n_frames images (contiguous, W*H bytes each, row-major U8, pitch W) into DBDE
 * frames (frame header + frame data each), replacing n_frames calls of dbde_pack_frame
 * (dbde_util.cpp:190-196).
 *   d_indices    : optional frame numbers (device, n_frames); NULL -> first_index + f
 *   d_elapsed_ns : optional elapsed_ns per frame (device); NULL -> 0 as dbde_pack_frame writes
 *   d_out        : output bytes (device), capacity out_capacity
 *   slot_stride  : 0 -> frames are CONCATENATED from d_out (a ready-to-write .dbde body);
 *                  else frame f starts at d_out + f*slot_stride (>= dbde_hip_max_frame_bytes)
 *   d_frame_offsets / d_frame_bytes : optional outputs (device, n_frames each): byte offset of
 *                  each frame from d_out and its exact length.
 * Asynchronous on the context's stream.  out_capacity must cover the worst case
 * (n_frames * dbde_hip_max_frame_bytes, or (n_frames-1)*slot_stride + max). */
int dbde_hip_encode_frames(dbde_hip_ctx *ctx, const uint8_t *d_images, int W, int H, int n_frames,
                           uint64_t first_index, const uint64_t *d_indices,
                           const uint64_t *d_elapsed_ns, uint8_t *d_out, size_t out_capacity,
                           uint64_t slot_stride, uint64_t *d_frame_offsets,
                           uint64_t *d_frame_bytes);

/* Decodes n_frames frames, replacing n_frames calls of dbde_unpack_frame
 * (dbde_util.cpp:339-345).  Frame f starts at d_stream + d_frame_offsets[f] (any byte
 * alignment).  stream_bytes is the readable extent of d_stream.  A frame whose frame data
 * fails validation (nb != T, nm != T, n64 != sum(depth): dbde_util.cpp:295-303; or a depth
 * byte > 8, the one documented deviation) leaves its image untouched and reports
 * header.u64s = 0xFFFFFFFF, consumed = 20.  d_results may be NULL. */
int dbde_hip_decode_frames(dbde_hip_ctx *ctx, const uint8_t *d_stream, size_t stream_bytes,
                           const uint64_t *d_frame_offsets, int W, int H, int n_frames,
                           uint8_t *d_images, dbde_hip_frame_result *d_results);

/* Builds the frame index of a concatenated frame sequence starting at d_stream (no video
 * header): hops 20 + 12 + 2T + 8*n64 from frame to frame (README.md:12-23) until max_frames
 * or the end of stream_bytes.  Writes offsets (device, max_frames) and returns the number of
 * whole frames found in *n_found (host).  Synchronous. */
int dbde_hip_index_stream(dbde_hip_ctx *ctx, const uint8_t *d_stream, size_t stream_bytes, int W,
                          int H, int max_frames, uint64_t *d_frame_offsets, int *n_found);
/* The same walk, enqueued on the context's stream without waiting for it: the frame count lands in the
 * DEVICE word *d_n_found.  A dbde_hip_decode_frames call enqueued behind it may use d_frame_offsets
 * directly (a reader that knows how many frames it expects, or that bounds the decode by max_frames and
 * inspects the per-frame results: entries past the count are set to 0xFFFFFFFFFFFFFFFF, an offset every
 * extent check rejects -- those frames report header.u64s = 0xFFFFFFFF, consumed = 20, index = elapsed_ns = 0
 * and leave their image untouched; the same holds for any offset outside [0, stream_bytes), however large). */
int dbde_hip_index_stream_async(dbde_hip_ctx *ctx, const uint8_t *d_stream, size_t stream_bytes, int W,
                                int H, int max_frames, uint64_t *d_frame_offsets, uint32_t *d_n_found);

/* Reading an un-indexed stream a batch at a time without the walk on the critical path: dbde_hip_scan_ahead
 * enqueues the walk of the next (up to) max_frames frames on a SECOND stream owned by the context -- behind
 * everything enqueued on the context's stream so far -- starting at the byte offset in the device word *d_cursor
 * and leaving the offset of the first unvisited byte there (zero it before the first call); frame offsets are
 * relative to d_stream.  dbde_hip_scan_join makes the context's stream wait for the walks enqueued so far.
 * A reader enqueues: scan_ahead(batch 0); then per batch b: scan_join, scan_ahead(batch b+1), decode_frames(batch b)
 * -- the dependent pointer chase of batch b+1 (about a microsecond per frame) then runs beside the decode of b. */
int dbde_hip_scan_ahead(dbde_hip_ctx *ctx, const uint8_t *d_stream, size_t stream_bytes, int W, int H,
                        int max_frames, uint64_t *d_cursor, uint64_t *d_frame_offsets,
                        uint32_t *d_n_found);
int dbde_hip_scan_join(dbde_hip_ctx *ctx);

/* Counter-based synthetic frames (same bytes as oracle/synth.c): mode 0 noise8, 1 mixed,
 * 2 flat, 3 smooth; modes 4..12 (profiling only, no oracle twin): every tile of depth mode - 4.
 * Used by bench.py and the parity tests to build inputs in HBM. */
int dbde_hip_synth_frames(dbde_hip_ctx *ctx, int mode, uint64_t seed, uint64_t first_frame,
                          int n_frames, int W, int H, uint8_t *d_images);

/* ---- host-pointer entry points: the reference signatures, on the GPU -------------------- */
/* Each copies its operands to the device, runs the same kernels as the batch API, copies
 * the result back and synchronises.  Return values and written bytes are the reference's. */

/* dbde_pack_8x8 (dbde_util.h:21, dbde_util.cpp:22-103): returns (depth<<8)|min, writes
 * exactly 8*depth bytes at target. */
uint32_t dbde_hip_pack_8x8(dbde_hip_ctx *ctx, const uint8_t *image, int stride, uint8_t *target);
/* dbde_pack_8x8_partial (dbde_util.h:22, dbde_util.cpp:105-135). */
uint32_t dbde_hip_pack_8x8_partial(dbde_hip_ctx *ctx, const uint8_t *image, int stride,
                                   int rightmargin, int downmargin, uint8_t *target);
/* dbde_pack_image (dbde_util.h:24, dbde_util.cpp:137-180): returns 12 + 2T + 8*n64. */
size_t dbde_hip_pack_image(dbde_hip_ctx *ctx, const uint8_t *image, int W, int H, uint8_t *target);
/* dbde_pack_frame (dbde_util.h:26, dbde_util.cpp:190-196). */
size_t dbde_hip_pack_frame(dbde_hip_ctx *ctx, uint64_t index, const uint8_t *image, int W, int H,
                           uint8_t *target);
/* dbde_unpack_8x8 (dbde_util.h:30, dbde_util.cpp:216-279); depth > 8 is ignored (no write). */
void dbde_hip_unpack_8x8(dbde_hip_ctx *ctx, uint8_t depth, uint8_t minval, const uint8_t *packed,
                         size_t stride, uint8_t *image);
/* dbde_unpack_8x8_partial (dbde_util.h:31, dbde_util.cpp:281-289). */
void dbde_hip_unpack_8x8_partial(dbde_hip_ctx *ctx, uint8_t depth, uint8_t minval,
                                 const uint8_t *packed, size_t stride, int rightmargin,
                                 int downmargin, uint8_t *image);
/* dbde_unpack_image (dbde_util.h:33, dbde_util.cpp:291-328): bytes consumed, 0 on failure. */
size_t dbde_hip_unpack_image(dbde_hip_ctx *ctx, const uint8_t *packed, int W, int H,
                             uint8_t *image);
/* dbde_unpack_frame (dbde_util.h:35, dbde_util.cpp:339-345): *packed is advanced exactly as
 * the reference advances it (by 20 only when the frame data is rejected). */
dbde_hip_frame_header dbde_hip_unpack_frame(dbde_hip_ctx *ctx, uint8_t **packed, int W, int H,
                                            uint8_t *image);

/* ---- header wire format (host only; a few bytes, no kernel) ----------------------------- */
/* dbde_pack_frame_header (dbde_util.cpp:182-188): 20 bytes; elapsed_ns travels as an F64. */
size_t dbde_hip_pack_frame_header(const dbde_hip_frame_header *fh, uint8_t *target);
/* dbde_pack_video_header (dbde_util.cpp:198-209): 28 bytes, height before width. */
size_t dbde_hip_pack_video_header(const dbde_hip_video_header *vh, uint8_t *target);
/* dbde_unpack_frame_header (dbde_util.cpp:330-337): advances *packed by 20. */
dbde_hip_frame_header dbde_hip_unpack_frame_header(uint8_t **packed);
/* dbde_unpack_video_header (dbde_util.cpp:347-359): advances *packed by 28. */
dbde_hip_video_header dbde_hip_unpack_video_header(uint8_t **packed);

/* ---- file I/O: batched .dbde writer and reader -------------------------------------------- */
/* The other end of the path: the reference reads a file frame by frame
 * (dbde_start_file_walk / dbde_walk_a_file / dbde_end_file_walk, dbde_util.cpp:362-426; that
 * API itself is served by libdbde_util_hip.so) and has no writer, its test hand-rolls one
 * (dbde_util_test.cpp:204-211).  These move whole batches: images are DEVICE buffers, the
 * compressed bytes cross PCIe through two pinned windows so that file I/O of one batch
 * overlaps the kernels of the next.  Files are byte-identical to a video header followed by
 * dbde_pack_frame output for every frame. */
typedef struct dbde_hip_writer dbde_hip_writer;
typedef struct dbde_hip_reader dbde_hip_reader;

/* Creates `path` and writes the 28-byte video header {3, H, W, frame_hz}
 * (dbde_pack_video_header).  batch_frames = frames encoded per launch (window size). */
int dbde_hip_writer_open(dbde_hip_ctx *ctx, const char *path, int W, int H, double frame_hz,
                         int batch_frames, dbde_hip_writer **out);
/* Appends n_frames device-resident images (arguments as dbde_hip_encode_frames).  On return
 * d_images may be reused; the bytes reach the file by the next put or by close. */
int dbde_hip_writer_put(dbde_hip_writer *w, const uint8_t *d_images, int n_frames,
                        uint64_t first_index, const uint64_t *d_indices,
                        const uint64_t *d_elapsed_ns);
const char *dbde_hip_writer_error(const dbde_hip_writer *w);
/* Flushes, closes the file and frees the writer; totals are optional outputs. */
int dbde_hip_writer_close(dbde_hip_writer *w, uint64_t *frames_written, uint64_t *bytes_written);

/* Opens `path`, parses and checks the video header with the walker's limits
 * (dbde_util.cpp:371-381: u64s == 3, 0 < H, W, H*W <= 0x37FFFFFF) and starts reading. */
int dbde_hip_reader_open(dbde_hip_ctx *ctx, const char *path, int batch_frames,
                         dbde_hip_video_header *vh, dbde_hip_reader **out);
/* Decodes the next up-to-max_frames frames (capped at batch_frames) into d_images (device,
 * W*H bytes each) and their headers into `headers` (host, optional).  *n_out = frames
 * delivered; 0 = end of file or, as dbde_walk_a_file returns false (dbde_util.cpp:412-420),
 * the first frame that is truncated or does not parse -- the walk ends there. */
int dbde_hip_reader_next(dbde_hip_reader *r, uint8_t *d_images, int max_frames,
                         dbde_hip_frame_header *headers, int *n_out);
void dbde_hip_reader_close(dbde_hip_reader *r);

/* ---- DBDE16: higher-bit-depth frames (extension; PARITY UNPINNED) ------------------------------------------- */
/* The reference's README notes that the minimum array "could expand size to handle higher bit depth images"
 * (README.md:65) and defines nothing further.  DBDE16 is that expansion and nothing else -- U16 pixels (pitch W
 * pixels), depth bytes 0..16, U16 little-endian minima, the second I32 = 2T (it is the BYTE count of the minimum
 * array, README.md:63), payload and tiling rules unchanged; an 8-bit reader rejects such a frame on nm != T.
 * Full specification: oracle/dbde16_oracle.c.  There is no reference behaviour to be bit-exact against: the kernels
 * are checked against that oracle, which agrees with the pinned 8-bit oracle on images that fit 8 bits.
 * Batch API on device buffers only, arguments as dbde_hip_encode_frames / dbde_hip_decode_frames (frame headers
 * carry index first_index + f, elapsed 0); worst case per frame 20 + 12 + 131*T bytes. */
size_t dbde16_hip_max_frame_bytes(int W, int H);
int dbde16_hip_encode_frames(dbde_hip_ctx *ctx, const uint16_t *d_images, int W, int H, int n_frames,
                             uint64_t first_index, uint8_t *d_out, size_t out_capacity,
                             uint64_t slot_stride, uint64_t *d_frame_offsets, uint64_t *d_frame_bytes);
int dbde16_hip_decode_frames(dbde_hip_ctx *ctx, const uint8_t *d_stream, size_t stream_bytes,
                             const uint64_t *d_frame_offsets, int W, int H, int n_frames,
                             uint16_t *d_images, dbde_hip_frame_result *d_results);

/* ---- multi-GPU: variable-length gather of the compressed stream to a root (RCCL over xGMI) ------------------- */
/* Frames are independent, so the path shards by contiguous frame blocks (rank g of G owns frames
 * [g*N/G, (g+1)*N/G)): every rank encodes its block with dbde_hip_encode_frames (slot_stride 0) and its output is
 * one in-order segment of the final stream (README.md:12-23: frames simply follow each other).  The only exchange
 * step is this gather (the reference is single-threaded and has no counterpart).  RCCL has no gatherv: byte counts
 * are all-gathered (ncclAllGather, 8 bytes per rank), every rank derives the same displacements, and the bytes
 * travel as grouped ncclSend / ncclRecv into the root's window at their displacement.  One process per GPU; the
 * library opens librccl itself (dlopen; a process that never calls these needs no RCCL).
 *
 * Per batch, with two slots so that the gather of batch k overlaps the encode of batch k+1:
 *   dbde_hip_gather_join(g, slot)            codec stream waits until the slot's previous transfer has finished
 *   dbde_hip_encode_frames(... d_frame_offsets, d_frame_bytes)        (root 0: straight into its window)
 *   dbde_hip_gather_begin(g, slot, &d_frame_offsets[n-1], &d_frame_bytes[n-1])     returns at once
 *   ... enqueue more work (decode, the next batch's encode) ...
 *   dbde_hip_gather_post(g, slot, d_segment, d_window, window_bytes, sizes, 0)     host waits for the counts only
 * Collective: every rank of the communicator makes the same begin / post calls in the same order. */
typedef struct dbde_hip_gather dbde_hip_gather;
#define DBDE_HIP_GATHER_ID_BYTES 128
/* Rendezvous token (ncclGetUniqueId): one rank creates it, the caller hands it to every other rank by any means. */
int dbde_hip_gather_unique_id(uint8_t id[DBDE_HIP_GATHER_ID_BYTES]);
/* Rank `rank` of `nranks` on the context's device; `root` receives.  Collective (ncclCommInitRank). */
int dbde_hip_gather_create(dbde_hip_ctx *ctx, const uint8_t id[DBDE_HIP_GATHER_ID_BYTES], int nranks, int rank,
                           int root, dbde_hip_gather **out);
/* The same on a communicator the caller already owns (an ncclComm_t, passed as void*); it is not destroyed. */
int dbde_hip_gather_attach(dbde_hip_ctx *ctx, void *nccl_comm, int nranks, int rank, int root,
                           dbde_hip_gather **out);
void dbde_hip_gather_destroy(dbde_hip_gather *g);
const char *dbde_hip_gather_error(const dbde_hip_gather *g);
/* Messages are cut into pieces of at most this many bytes (default 1 GiB), all posted in one group. */
int dbde_hip_gather_set_max_message(dbde_hip_gather *g, uint64_t bytes);
/* Root: the bytes its window holds (default: no limit declared).  The value travels with every size exchange, so that
 * "the batch does not fit" is a verdict EVERY rank reaches from the same numbers (dbde_hip_gather_post then returns
 * DBDE_HIP_ERR_CAPACITY on all ranks and nothing has been posted anywhere; a root that found out alone used to leave
 * its peers' sends unmatched).  Call it once, before the first dbde_hip_gather_begin. */
int dbde_hip_gather_set_window(dbde_hip_gather *g, uint64_t window_bytes);
/* The verdict itself, pure arithmetic: pairs[2 r] = rank r's count, pairs[2 r + 1] = the window capacity rank r declared
 * (only the root's counts).  DBDE_HIP_OK or DBDE_HIP_ERR_CAPACITY; *total_out = sum of the counts. */
int dbde_hip_gather_check(int nranks, int root, const uint64_t *pairs, uint64_t *total_out);
/* Enqueues the size exchange of `slot` (0 or 1) behind everything on the context's stream: this rank's byte count
 * is the sum of the two DEVICE words (either may be NULL = 0; the encoder's offset and length of the batch's last
 * frame).  Does not wait for anything. */
int dbde_hip_gather_begin(dbde_hip_gather *g, int slot, const uint64_t *d_last_offset,
                          const uint64_t *d_last_bytes);
/* Blocks the HOST (not the codec's stream) until the slot's counts have arrived, then posts the transfers on the
 * gather's own stream: a non-root rank sends d_segment[0, its count) to the root; the root receives rank r's bytes
 * at d_window + sum(counts of ranks < r).  The root's own bytes are not moved when d_segment already is
 * d_window + its displacement (root 0 encoding straight into its window), else copied once device-to-device.
 * window_bytes (root) must hold the sum of all counts -- size it as nranks x the per-rank capacity and declare it
 * with dbde_hip_gather_set_window: DBDE_HIP_ERR_CAPACITY is then returned by EVERY rank, with nothing posted.  (A root
 * whose window_bytes is below what it declared gets DBDE_HIP_ERR_ARG: a caller's error, not the data's.)  sizes_out: optional host
 * array of nranks counts.  flags: DBDE_HIP_GATHER_LOOPBACK (tests and one-GPU rehearsals) sends the root's own
 * segment to itself through ncclSend / ncclRecv instead. */
#define DBDE_HIP_GATHER_LOOPBACK 1u
int dbde_hip_gather_post(dbde_hip_gather *g, int slot, const uint8_t *d_segment, uint8_t *d_window,
                         size_t window_bytes, uint64_t *sizes_out, uint32_t flags);
/* Makes the context's stream wait for the slot's posted transfers (before its segment / window is overwritten). */
int dbde_hip_gather_join(dbde_hip_gather *g, int slot);
/* Blocks the host until they have finished (before the root reads the window from the host side). */
int dbde_hip_gather_sync(dbde_hip_gather *g, int slot);
/* NCCL_VERSION_CODE of the RCCL in use; 0 when librccl cannot be opened. */
int dbde_hip_gather_rccl_version(void);
/* The transfer plan, exposed because it is the host logic both ends must agree on (pure arithmetic, no GPU):
 * for `rank`, the ordered list of operations given every rank's count.  Returns the number of operations
 * (filling at most max_ops of them) or a negative error; *total_out = sum of counts. */
enum { DBDE_HIP_GATHER_SEND = 1, DBDE_HIP_GATHER_RECV = 2, DBDE_HIP_GATHER_OWN = 3 };
typedef struct {
    int32_t peer;              /* rank at the other end (OWN: the root itself) */
    int32_t kind;              /* DBDE_HIP_GATHER_SEND / _RECV / _OWN */
    uint64_t segment_offset;   /* SEND: byte offset in this rank's segment */
    uint64_t window_offset;    /* byte offset in the root's window (displacement + piece offset) */
    uint64_t bytes;
} dbde_hip_gather_op;
int dbde_hip_gather_plan(int nranks, int rank, int root, const uint64_t *sizes, uint64_t max_piece,
                         dbde_hip_gather_op *ops, int max_ops, uint64_t *total_out);

/* ---- multi-GPU: scatter of a .dbde body to the ranks' frame blocks (the decode-side mirror of the gather) ------ */
/* A root holds a stream of frames following each other (README.md:12-23) and the frame starts the device scanner found
 * (dbde_hip_index_stream_async; the serial reader this replaces is dbde_util.cpp:408-421).  Rank r of G gets the bytes of
 * frames [r n / G, (r + 1) n / G) -- the gather's blocks -- and the offsets of those frames relative to its segment, and
 * decodes them with dbde_hip_decode_frames.  The block table is worked out on the device from the scanner's outputs and
 * broadcast (ncclBroadcast), every rank's buffer capacities are all-gathered beside it, the bytes and the offsets travel
 * as grouped ncclSend / ncclRecv; the root's own block is not moved.  Two slots, as for the gather.  Per batch:
 *   root:   dbde_hip_index_stream_async(ctx, d_stream, bytes, W, H, max, d_offsets, d_count)
 *   all:    dbde_hip_scatter_begin(s, slot, d_stream, bytes, d_offsets, d_count)      (non-root ranks pass NULL / 0)
 *   all:    dbde_hip_scatter_post(s, slot, d_segment, d_my_offsets, &mine, NULL, 0)   host waits for the table only
 *   all:    dbde_hip_scatter_join(s, slot)
 *   all:    dbde_hip_decode_frames(ctx, root ? d_stream + mine.byte_start : d_segment, mine.byte_count, d_my_offsets, W, H,
 *                                  (int)mine.n_frames, d_images, d_results)
 * Collective: every rank makes the same begin / post calls in the same order.  UNMEASURED on hardware beyond one rank. */
typedef struct dbde_hip_scatter dbde_hip_scatter;
typedef struct {
    uint64_t first_frame, n_frames;   /* the rank's frame block: global frame numbers [first_frame, first_frame + n_frames) */
    uint64_t byte_start, byte_count;  /* its bytes in the root's stream */
} dbde_hip_scatter_block;
int dbde_hip_scatter_create(dbde_hip_ctx *ctx, const uint8_t id[DBDE_HIP_GATHER_ID_BYTES], int nranks, int rank,
                            int root, dbde_hip_scatter **out);      /* collective (ncclCommInitRank); id: dbde_hip_gather_unique_id */
int dbde_hip_scatter_attach(dbde_hip_ctx *ctx, void *nccl_comm, int nranks, int rank, int root, dbde_hip_scatter **out);
void dbde_hip_scatter_destroy(dbde_hip_scatter *s);
const char *dbde_hip_scatter_error(const dbde_hip_scatter *s);
int dbde_hip_scatter_set_max_message(dbde_hip_scatter *s, uint64_t bytes);
/* What this rank's receive buffers hold: segment bytes and frame offsets.  Travels with every table exchange, so that "a
 * block does not fit" is DBDE_HIP_ERR_CAPACITY on EVERY rank with nothing posted (the root needs none: it decodes in place). */
int dbde_hip_scatter_set_capacity(dbde_hip_scatter *s, uint64_t segment_bytes, uint64_t max_frames);
/* Enqueues the table exchange of `slot` behind everything on the context's stream (the scanner).  Root: the stream, its
 * readable extent, the scanner's offsets and its count word (all device).  Other ranks: NULL, 0, NULL, NULL. */
int dbde_hip_scatter_begin(dbde_hip_scatter *s, int slot, const uint8_t *d_stream, uint64_t stream_bytes,
                           const uint64_t *d_frame_offsets, const uint32_t *d_n_frames);
/* Blocks the HOST until the table has arrived, then posts the transfers on the scatter's own stream.  d_segment: where a
 * non-root rank's bytes land (root: unused unless DBDE_HIP_SCATTER_LOOPBACK, which sends the root's own block to itself
 * through ncclSend / ncclRecv -- tests and one-GPU rehearsals); d_offsets_out: the block's frame offsets relative to its
 * first byte (every rank).  mine_out / table_out (nranks entries): optional host copies of the table. */
#define DBDE_HIP_SCATTER_LOOPBACK 1u
int dbde_hip_scatter_post(dbde_hip_scatter *s, int slot, uint8_t *d_segment, uint64_t *d_offsets_out,
                          dbde_hip_scatter_block *mine_out, dbde_hip_scatter_block *table_out, uint32_t flags);
int dbde_hip_scatter_join(dbde_hip_scatter *s, int slot);   /* the context's stream waits for the slot's transfers */
int dbde_hip_scatter_sync(dbde_hip_scatter *s, int slot);   /* the host does */
/* The host logic both ends must agree on, pure arithmetic (no GPU): the block table from a host-side frame index, the
 * capacity verdict (caps[2 r] = rank r's segment bytes, caps[2 r + 1] = its frame capacity), and the ordered transfers
 * of `rank` (returns their number, filling at most max_ops). */
enum { DBDE_HIP_SCATTER_SEND_BYTES = 1, DBDE_HIP_SCATTER_RECV_BYTES = 2, DBDE_HIP_SCATTER_SEND_OFFSETS = 3,
       DBDE_HIP_SCATTER_RECV_OFFSETS = 4, DBDE_HIP_SCATTER_OWN = 5 };
typedef struct {
    int32_t peer, kind;
    uint64_t source_offset;    /* bytes: offset in the root's stream; offsets: byte offset in the root's offset array */
    uint64_t dest_offset;      /* byte offset in the receiver's segment / offsets buffer */
    uint64_t bytes;
} dbde_hip_scatter_op;
int dbde_hip_scatter_blocks(int nranks, uint64_t n_frames, const uint64_t *frame_offsets, uint64_t stream_bytes,
                            dbde_hip_scatter_block *table);
int dbde_hip_scatter_check(int nranks, const dbde_hip_scatter_block *table, const uint64_t *caps);
int dbde_hip_scatter_plan(int nranks, int rank, int root, const dbde_hip_scatter_block *table, uint64_t max_piece,
                          dbde_hip_scatter_op *ops, int max_ops);

/* ---- launch plans: which kernels a batch call runs (pure host arithmetic: no context, no device) ------------
 * The batch calls choose among several kernel forms by shape, batch size and buffer alignment;
 * these two functions ARE that choice (dbde_hip_encode_frames / _decode_frames call the same code), exposed so that
 * an integrator can see -- and a CPU-only test can pin -- what a given geometry runs.  No reference counterpart. */
typedef struct dbde_hip_launch_plan {
    int32_t kernel;            /* 0 = chunk kernels (encode: persistent encoder), 1 = encode: one workgroup per chunk
                                  (small launches), 2 = whole frames per wave (encode, T <= 64 tiles), 3 = whole frames per workgroup
                                  (decode: 1 .. 256 tiles, and up to 768 where chunks would store tile by tile; persistent), 4 = whole frames per
                                  workgroup, staged through LDS,
                                  5 = encode: 4 .. 64 (and 77 .. 85) tiles in 16-byte aligned slots, rows of 4-byte multiples: persistent
                                  workgroups, pixels double-buffered */
    int32_t input_mode;        /* encode: 0 = 16-byte aligned rows, 1 = any geometry (W >= 16), 2 = byte by byte (W < 16), 3 = any geometry with dword-aligned fetches (63 tile pairs per wave), 4 = the same with one wave per segment of a tile row */
    int32_t image_mode;        /* decode, kernel 0: 0 = one aligned 16-byte store per lane and image row, 1 = chunks of whole
                                  tile rows staged in LDS (16-byte rows: per chunk, only all-depth-8 chunks stage),
                                  2 = tile by tile */
    int32_t index_mode;        /* decode, kernel 0: 0 = index kernel + table, 1 = self-indexing workgroups, 2 = fused index + decode */
    int32_t threads;           /* workgroup size */
    int32_t aligned_out;       /* encode: 1 = 8-byte aligned frames and fields (wide stores) */
    uint32_t chunks_per_frame; /* kernels 0 / 1 */
    uint32_t chunk_tiles;      /* kernels 0 / 1: tile slots of a chunk (decode: whole tile rows, or 512) */
    uint64_t n_chunks;         /* kernels 0 / 1: chunks (= workgroups of the decoder) in the launch */
} dbde_hip_launch_plan;
/* resident_workgroups: what the device holds of the persistent encoder (2 per CU + 1 on MI355X: 513). */
int dbde_hip_encode_plan(int width, int height, int n_frames, uint64_t image_address, uint64_t out_address,
                         uint64_t slot_stride, int resident_workgroups, dbde_hip_launch_plan *plan);
int dbde_hip_decode_plan(int width, int height, int n_frames, uint64_t image_address, int n_cu,
                         dbde_hip_launch_plan *plan);

/* ---- kernel timing hook for bench.py ---------------------------------------------------- */
/* When enabled, every encode / decode call brackets its kernels with HIP events on the
 * context's stream; dbde_hip_timing_read returns accumulated milliseconds and launch counts
 * ([0]=encode kernel, [1]=decode index kernel, [2]=decode kernel, [3]=stream scanner) after synchronising. */
int dbde_hip_timing_enable(dbde_hip_ctx *ctx, int on);
int dbde_hip_timing_read(dbde_hip_ctx *ctx, double ms[4], uint64_t launches[4], int reset);

#ifdef __cplusplus
}
#endif
#endif
