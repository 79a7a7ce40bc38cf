// dbde_kernels.h -- launch interface between the C-ABI (dbde_capi.cpp) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dbde {

// A chunk is the unit of one 256-thread workgroup: 512 consecutive tiles in stream order
// (row-major over the frame's tiles), two tiles per lane.
// Capacity of one decode workgroup: 256 threads, two tiles per lane.
constexpr uint32_t kChunkTiles = 512;
// ... and of the smaller decode workgroup (192 threads) that whole-tile-row chunks take when they fill it better
constexpr uint32_t kChunkTilesSmall = 384;

// Chunk geometry of the DECODER.  A chunk is the unit of one workgroup and always lines up with the image:
//   * row-aligned form, w <= 512 tiles across: a chunk is a run of WHOLE tile rows (rows * w <= 512 tiles) --
//     its 8 * rows image rows are then ONE contiguous byte range of the frame (the staged-image decode path);
//   * row-aligned form, wider frames: a chunk is one piece (<= 512 tiles) of a tile row;
//   * plain form: 512 consecutive tiles, wherever they fall.
// Chunks partition the frame's tiles in stream order (row-major), so a chunk's payload is one contiguous
// byte range of the stream as well.  (The north star's "one wavefront per tile-row strip".)
struct DecGeom {
    uint32_t w, h, T;
    uint32_t ct;        // whole-row form: tiles per chunk (rows * w); piece form: 512
    uint32_t pieces;    // chunks per tile row; 1 = whole-row form
    uint32_t cpf;       // chunks per frame
};
// row_aligned = false: plain runs of 512 consecutive tiles (a workgroup's time is nearly independent of how
// many of its 512 tile slots are used, so full chunks win wherever alignment with the image buys nothing).
// `cap`: tile slots of the workgroup that takes whole-tile-row chunks (kChunkTiles, or kChunkTilesSmall).
__host__ __device__ inline DecGeom dec_geometry(uint32_t w, uint32_t h, bool row_aligned, uint32_t plain_ct = kChunkTiles,
                                                uint32_t cap = kChunkTiles) {
    DecGeom g;
    g.w = w; g.h = h; g.T = w * h;
    if (!row_aligned) {
        g.ct = plain_ct; g.pieces = 1u; g.cpf = (g.T + plain_ct - 1u) / plain_ct;
    } else if (w <= cap) {
        uint32_t rows = cap / w;   // the staged image, 8 * rows * W <= 64 * rows * w bytes, fits the workgroup's LDS
        if (rows < 1u) rows = 1u;
        g.ct = rows * w; g.pieces = 1u; g.cpf = (h + rows - 1u) / rows;
    } else {
        g.ct = kChunkTiles; g.pieces = (w + kChunkTiles - 1u) / kChunkTiles; g.cpf = h * g.pieces;
    }
    return g;
}
// First tile (stream order) of chunk c; c == cpf gives T.
__host__ __device__ inline uint32_t dec_chunk_begin(const DecGeom &g, uint32_t c) {
    if (g.pieces == 1u) { const uint64_t b = (uint64_t)c * g.ct; return b < g.T ? (uint32_t)b : g.T; }
    if (c >= g.cpf) return g.T;
    const uint32_t ty = c / g.pieces, pc = c - ty * g.pieces;
    return ty * g.w + pc * kChunkTiles;
}
// Chunk that holds tile `pos` (pos < T).
__host__ __device__ inline uint32_t dec_chunk_of(const DecGeom &g, uint32_t pos) {
    if (g.pieces == 1u) return pos / g.ct;
    const uint32_t ty = pos / g.w, tx = pos - ty * g.w;
    return ty * g.pieces + tx / kChunkTiles;
}

// The encoder walks larger chunks (fewer ticket draws): one 512-thread workgroup per 1024 tiles.
#ifndef DBDE_ENC_CHUNK_TILES
#define DBDE_ENC_CHUNK_TILES 1024   // A/B switch (512: four-wave workgroups, half the step, twice the records)
#endif
constexpr uint32_t kEncChunkTiles = DBDE_ENC_CHUNK_TILES;

// Decoupled look-back record, one per chunk, 64 bits, written and read with relaxed
// agent-scope atomics (the record IS the flag, so no fence is needed):
//   [63:62] status  0 = not yet, 1 = AGG (chunk total only), 2 = INC (inclusive prefixes), 3 = POISON
//   AGG : [31:0]  payload words of this chunk
//   INC : [61:32] payload words of this frame up to and including this chunk (< 2^30)
//         [31:0]  payload words of the whole launch up to and including this chunk (mod 2^32)
constexpr unsigned long long kStAgg = 1ull << 62, kStInc = 2ull << 62, kStPoison = 3ull << 62;

// floor(2^32 / d) for the kernels' div_magic (d = 1: 2^32 - 1, which the one correction step absorbs).
inline uint32_t div_magic_of(uint32_t d) { return d <= 1u ? 0xFFFFFFFFu : (uint32_t)((1ull << 32) / d); }

#ifndef DBDE_CTRL_SLOT_WORDS
#define DBDE_CTRL_SLOT_WORDS 64                 // u32 between two counters of the persistent encoder's control words: a 256-byte slot each (A/B: 4 KB slots measured the same)
#endif
constexpr uint32_t kEncCtrlWords = 16 * DBDE_CTRL_SLOT_WORDS + 1024;   // u32 per set of control words: 16 tail-ticket counters, a slot each, + the rest
constexpr uint32_t kEncMaxGrid = 4096;          // workgroups the mode flags have room for (dbde_capi.cpp: the look-back block's header)

struct EncParams {
    const uint8_t *images;         // n_frames * W*H
    uint8_t *out;
    uint64_t *frame_offsets;       // optional [n_frames]
    uint64_t *frame_bytes;         // optional [n_frames]
    const uint64_t *indices;       // optional [n_frames]
    const uint64_t *elapsed_ns;    // optional [n_frames]
    uint64_t first_index;
    unsigned long long *state;     // [n_chunks] look-back records: no AGG / INC record when the launch starts (each is cleared by its reader)
    uint32_t *ctrl;                // this launch's control words (kEncCtrlWords u32: group arrival / tail-ticket counters, mode, ticket counter; dbde_kernels.hip), zero when the launch starts
    uint32_t *ctrl_next;           // the NEXT launch's set (the host alternates between two): cleared by the scanner
    uint32_t *mode_flags;          // [grid] one word per workgroup: launch_epoch << 2 | claim mode, written by whoever settles the mode
    uint32_t *arrive_flags;        // [grid] one word per workgroup: launch_epoch << 2 | 1, written by the workgroup when it starts
    uint32_t launch_epoch;         // 1 .. 2^30 - 1: tag of THIS persistent launch's mode flags (never cleared between launches)
    uint32_t *sticky;              // context-wide failure word, OR-ed on look-back time-out
    uint64_t slot_stride;          // 0 = frames concatenated
    uint64_t frame_pixels;         // W*H: BYTES of one frame's image (2 W H for launch_encode16_fast)
    int W, H;
    uint32_t w, h, T;              // tiles across, down, total
    uint32_t chunks_per_frame, n_chunks;
    // Any-geometry input path: lanes_per_row = ceil(w / 2) != 0 -> lanes are dealt to tile PAIRS that never leave a
    // tile row (the last lane of a row holds one tile when w is odd); a chunk is 512 consecutive pairs in stream
    // order, wherever they fall.  lanes_per_row == 0: plain runs of 1024 tiles (w even: pairs never straddle).
    uint32_t lanes_per_row;
    uint32_t pairs_per_wave;       // 64, or 63: dword-aligned fetches, a wave's 64th lane only feeds the 63rd (kInRaw4, dbde_kernels.hip)
    uint32_t magic_w, magic_cpf, magic_lpr;   // div_magic_of(w), (chunks_per_frame), (lanes_per_row): divisions by launch constants
    // kInRow (dbde_kernels.hip): a tile row is cut into seg_per_row segments of seg_q or seg_q + 1 pairs (the first seg_rem
    // of them), one wave each; 0 = lanes dealt to pairs linearly
    uint32_t seg_per_row, seg_q, seg_rem, magic_seg;
    uint32_t last_frame;           // n_frames - 1: no pixel load reaches past the end of that frame
    uint32_t flags;                // bit 0: force ticket mode (A/B measurements); bit 6 (tests): small launches, odd chunks publish nothing; bit 9 (tests): workgroup 0 of a persistent launch arrives 60 us late
    uint32_t grid_blocks;          // resident workgroups of the persistent encoder
    unsigned long long *diag;      // [16] cycle counters, written only by -DDBDE_DIAG builds (profiles/variants.sh)
    uint32_t small_epoch;          // launch_encode_small: tag of THIS launch's records (1 .. 2^30 - 1; records of other launches do not match)
};

struct DecParams {
    const uint8_t *stream;
    const uint64_t *frame_offsets;  // [n_frames] byte offset of each frame header
    uint64_t stream_bytes;          // readable extent of stream: nothing beyond it is touched
    uint8_t *images;
    const uint32_t *chunk_off;      // [n_frames][chunks_per_frame + 1] payload word offset of each chunk inside its frame (+ total)
    const uint32_t *frame_ok;       // [n_frames] 1 = frame data validated
    void *results;                  // optional dbde_hip_frame_result[n_frames] (written by the self-indexing form)
    uint64_t frame_pixels;
    int W, H;
    uint32_t w, h, T;
    uint32_t chunks_per_frame, n_chunks;
    uint32_t magic_W;               // div_magic_of(W): byte offset in a chunk's range -> (image row, column), staged copy-out
    unsigned long long *fuse_rec;   // index_mode 2: [n_chunks] records {epoch 32 | depth > 8 seen 1 | payload words 31}
    uint32_t fuse_epoch;            // ... of THIS launch (never 0; records of other launches do not match)
    uint32_t fuse_flags;            // bit 0 (tests): odd chunks publish nothing -- the waiting waves' fallback does the work
    DecGeom geom;
    unsigned long long *diag;       // [16] written only by -DDBDE_DIAG builds (wave 0's timeline of the fused launch)
};

struct IdxParams {
    const uint8_t *stream;
    const uint64_t *frame_offsets;
    uint64_t stream_bytes;
    uint32_t *chunk_off;            // out [n_frames][chunks_per_frame + 1]
    uint32_t *frame_ok;             // out [n_frames]
    void *results;                  // optional dbde_hip_frame_result[n_frames]
    uint32_t T, chunks_per_frame;
    uint32_t min_bytes;             // bytes per tile minimum: 1 = DBDE, 2 = DBDE16 (depth <= 8 * min_bytes, nm = T * min_bytes)
    DecGeom geom;                   // which tiles a chunk holds
    uint32_t split;                 // workgroups per frame (1 = decode_index_kernel, >1 = the split form)
    uint32_t *frame_ctr;            // [n_frames] arrivals per frame, zero between launches (split form)
    uint32_t *frame_flag;           // [n_frames] bit0 = a depth byte > 8 was seen (split form)
};

struct FrameResultDev {             // layout of dbde_hip_frame_result
    uint32_t u64s;
    uint32_t pad_;
    uint64_t index;
    uint64_t elapsed_ns;
    uint64_t consumed;
};

hipError_t launch_encode(const EncParams &p, bool fast_in, bool aligned_out, hipStream_t s);
// DBDE16 (U16 pixels, W % 8 == 0, 16-byte aligned base) through the persistent encoder: 512 tiles per chunk, frame_pixels in bytes
hipError_t launch_encode16_fast(const EncParams &p, bool fast_in, bool aligned_out, hipStream_t s);
// Launches with at most as many chunks as the device holds workgroups: one workgroup per chunk (chunk id = workgroup id),
// no scanner; records are tagged with EncParams::small_epoch and never cleared (the workspace only has to have been
// zeroed once since it was allocated).
hipError_t launch_encode_small(const EncParams &p, bool fast_in, bool aligned_out, hipStream_t s);
int encode_blocks_per_cu();
// Frames of 65 .. 512 tiles, one slot per frame: one tile per lane, as many whole frames per 256 / 512 / 1024-thread
// workgroup as fit (mid_threads_for), no workspace.
hipError_t launch_encode_mid(const EncParams &p, uint32_t n_frames, hipStream_t s);
uint32_t mid_threads_for(uint32_t T, uint32_t max_threads = 1024u);
uint32_t mid_encode_threads_for(uint32_t T);
uint32_t mid_decode_threads_for(uint32_t T);   // the decoder's choice (persistent workgroups: smaller ones, more of them per CU)
// (persistent workgroups, n_cu * 2048 / threads of them; n_cu = 0: three, for tests of the pipelined loop on small batches)
hipError_t launch_decode_mid(const struct DecParams &p, uint32_t n_frames, uint32_t n_cu, hipStream_t s);
// Frames of 65 .. 1024 tiles with 8-byte aligned rows, whole frames per workgroup, pixels and stream bytes staged through
// LDS as aligned 16-byte blocks (encode: one slot per frame).  frames_threads_for: 256 or 512 threads (512 / 1024 tile slots).
hipError_t launch_encode_frames(const EncParams &p, uint32_t n_frames, hipStream_t s);
hipError_t launch_decode_frames(const struct DecParams &p, uint32_t n_frames, hipStream_t s);
uint32_t frames_threads_for(uint32_t T);
// Frames of 1 .. 256 tiles with 8-byte aligned rows, one slot per frame: one tile per lane, persistent 256-thread workgroups,
// the next group's pixels in flight (LDS-DMA into the other of two buffers) while a group is encoded (n_cu = 0: three workgroups, tests)
hipError_t launch_encode_group(const EncParams &p, uint32_t n_frames, uint32_t n_cu, hipStream_t s);
// Frames of at most 64 tiles, one slot per frame: one tile per lane, 64 / T frames per wave, no workspace.
hipError_t launch_encode_tiny(const EncParams &p, uint32_t n_frames, hipStream_t s);
hipError_t launch_decode_index(const IdxParams &p, int n_frames, hipStream_t s);
// img_mode: 0 direct (cache-line friendly geometry), 1 staged linear ranges (W % 8 == 0), 2 tile by tile (any)
// index_mode: 0 = chunk_off / frame_ok come from launch_decode_index; 1 = no index kernel, every workgroup reads the
// frame's whole depth array (few small frames); 2 = no index kernel either, the workgroups exchange their chunks' depth
// sums through p.fuse_rec (few LARGE frames: launches that fit the device's workgroup slots)
hipError_t launch_decode(const DecParams &p, int img_mode, int index_mode, hipStream_t s);
hipError_t launch_synth(int mode, uint64_t seed, uint64_t first_frame, int n_frames, int W, int H,
                        uint8_t *d_images, hipStream_t s);
// Serial frame-to-frame hop over a concatenated stream (one wave); see dbde_hip_index_stream.
hipError_t launch_scan_stream(const uint8_t *stream, uint64_t stream_bytes, uint32_t T, int max_frames,
                              uint64_t *d_offsets, uint32_t *d_count, uint64_t *d_cursor, hipStream_t s);
// Speculative parallel walk (see scan_spec_kernel): the stream in n_seg <= kMaxScanSegments segments, temporary
// position lists.
constexpr uint32_t kMaxScanSegments = 64;
struct ScanParams {
    const uint8_t *stream;
    uint64_t stream_bytes;
    uint32_t T;
    uint32_t gran;                 // frame starts are multiples of this many bytes from the stream's first byte
    uint64_t seg_bytes;            // segment j covers [j * seg_bytes, (j + 1) * seg_bytes)
    uint32_t seg_cap;              // entries per temporary list
    uint64_t *seg_pos;             // [n_seg][seg_cap]
    uint64_t *seg_start, *seg_end; // [n_seg] where a segment's walk began / the first frame start past the segment
    uint32_t *seg_count, *seg_ended;
    uint32_t wg_per_seg;           // workgroups that share a segment's signature search
    unsigned long long *seg_found_inv;   // [n_seg] complement of the best match so far; zero between calls
    uint32_t *seg_arrive;          // [n_seg] zero between calls
};
hipError_t launch_scan_spec(const ScanParams &p, uint32_t n_seg, int max_frames, uint64_t *d_offsets, uint32_t *d_count,
                            hipStream_t s);
// Maximum chunks_per_frame the decode index kernel can hold in LDS.
constexpr uint32_t kMaxChunksPerFrame = 32768;

}  // namespace dbde
