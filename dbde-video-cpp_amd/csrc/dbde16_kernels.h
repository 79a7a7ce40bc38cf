// dbde16_kernels.h -- launch interface of the DBDE16 kernels (dbde16_kernels.hip); see oracle/dbde16_oracle.c for
// the format.  One tile per lane, 256 tiles per workgroup (32 KB of worst-case payload in LDS: four workgroups per CU);
// the 8-bit decoder's index kernels are shared (DecGeom::ct = 256).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dbde16 {

constexpr uint32_t kChunkTiles16 = 256;

struct Params16 {                   // encode
    const uint16_t *images;         // n_frames * W*H pixels, pitch W
    uint8_t *out;
    uint64_t *frame_offsets;        // optional [n_frames]
    uint64_t *frame_bytes;          // optional [n_frames]
    uint64_t first_index;
    uint64_t slot_stride;           // 0 = frames concatenated
    uint64_t frame_pixels;
    int W, H;
    uint32_t w, h, T, chunks_per_frame;
    uint32_t n_frames;
    // workspace, zeroed before every launch
    unsigned long long *state;      // [n_frames * cpf] bit 63 = published, rest = payload words of the chunk
    unsigned long long *gsum;       // [n_frames * ceil(cpf / 64)] words of a frame's group of 64 chunks
    unsigned long long *fsize;      // [n_frames] words of a frame (concatenated layout)
    unsigned long long *fgsum;      // [ceil(n_frames / 64)] words of a group of 64 frames (concatenated layout)
    uint32_t *ticket;               // [0] arrival / ticket counter, [1] how chunk ids are claimed (0 undecided, 1 static, 2 tickets)
    uint32_t force_tickets;         // tests: skip the static assignment
    uint32_t *sticky;               // context-wide failure word, OR-ed on a look-back time-out
    unsigned long long *diag;       // -DDBDE_DIAG builds: phase time sums ([0..7], 10 ns ticks), else unused
};

struct DecParams16 {
    const uint8_t *stream;
    uint64_t stream_bytes;          // readable extent of stream
    const uint64_t *frame_offsets;
    uint16_t *images;
    const uint32_t *chunk_off;      // [n_frames][cpf + 1] from the index kernels
    const uint32_t *frame_ok;
    uint64_t frame_pixels;
    int W, H;
    uint32_t w, h, T, chunks_per_frame;
    unsigned long long *diag;       // -DDBDE_DIAG builds only
};

int encode16_blocks_per_cu();   // resident workgroups per CU of the persistent encoder (occupancy query)
hipError_t launch_encode16(const Params16 &p, int n_frames, uint32_t resident_blocks, hipStream_t s);
hipError_t launch_decode16(const DecParams16 &p, int n_frames, hipStream_t s);

}  // namespace dbde16
