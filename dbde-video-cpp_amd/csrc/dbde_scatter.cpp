// dbde_scatter.cpp -- the decode-side exchange step of the multi-GPU path (SURVEY.md 8e "decode side, if sharded",
// include/dbde_hip.h "multi-GPU: scatter"): a root that holds a .dbde body -- frames simply following each other,
// reference README.md:12-23 -- and the frame starts the device scanner found in it (dbde_hip_index_stream_async: sizes
// are only in-band, the serial reader it replaces is dbde_util.cpp:408-421) hands every rank the bytes of ITS contiguous
// frame block and the offsets of its frames inside them.  The mirror of dbde_gather.cpp:
//   * blocks are dbde_hip_gather's frame blocks (rank r of G owns frames [r n / G, (r + 1) n / G)), so a stream that was
//     gathered from G ranks scatters back to the same ranks;
//   * what each rank gets -- {first frame, frames, first byte, bytes} -- is worked out ON THE DEVICE from the scanner's
//     offsets and count (nothing visits the host on the root's critical path) and broadcast (ncclBroadcast, 32 bytes per
//     rank) together with an all-gather of every rank's buffer capacities, so that "it does not fit" is one verdict
//     reached by every rank from the same numbers (all post their transfers or none does);
//   * the bytes travel as grouped ncclSend / ncclRecv, the frame offsets of a block (8 bytes per frame) behind them; the
//     receiver rebases them to its segment on the device.  The root's own block is not moved: it decodes from the stream
//     where it lies.
// RCCL is opened at run time (dbde_rccl.h).  Rank-to-rank traffic has not run on hardware (one-GPU boxes only); the
// plan both ends derive is pure arithmetic and is tested for worlds 1-8 without a GPU (tests/test_scatter_plan.py).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dbde_hip.h"
#include "dbde_rccl.h"

namespace {

using dbde_rccl::Rccl;
using dbde_rccl::rccl;

constexpr int kSlots = 2;

// Root, one thread: the block table from the scanner's outputs.  Frames past the count (or an empty stream) give empty
// blocks; a block's bytes end where the next block's first frame starts, the last one's at stream_bytes.
__global__ void scatter_table_kernel(const uint64_t *offsets, const uint32_t *n_found, uint64_t stream_bytes, int nranks,
                                     uint64_t *table) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint64_t n = n_found ? *n_found : 0u;
    for (int r = 0; r < nranks; r++) {
        const uint64_t lo = n * (uint64_t)r / (uint64_t)nranks, hi = n * (uint64_t)(r + 1) / (uint64_t)nranks;
        const uint64_t b0 = lo < n ? offsets[lo] : stream_bytes, b1 = hi < n ? offsets[hi] : stream_bytes;
        table[4 * r + 0] = lo;
        table[4 * r + 1] = hi - lo;
        table[4 * r + 2] = b0;
        table[4 * r + 3] = b1 >= b0 ? b1 - b0 : 0ull;   // (offsets are the scanner's: ascending)
    }
}

__global__ void scatter_caps_kernel(uint64_t seg_cap, uint64_t max_frames, uint64_t *out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = seg_cap; out[1] = max_frames; }
}

// offsets of a block's frames, relative to the block's first byte
__global__ void scatter_rebase_kernel(const uint64_t *in, uint64_t *out, uint64_t n, uint64_t byte_start) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = in[i] - byte_start;
}

}  // namespace

struct dbde_hip_scatter {
    dbde_hip_ctx *ctx = nullptr;
    int device = 0, nranks = 1, rank = 0, root = 0;
    ncclComm_t comm = nullptr;
    bool own_comm = false;
    hipStream_t ctx_stream = nullptr, comm_stream = nullptr;
    uint64_t max_piece = 1ull << 30;
    uint64_t seg_cap = 0, max_frames = 0;     // this rank's receive buffers (dbde_hip_scatter_set_capacity)
    struct Slot {
        uint64_t *d_table = nullptr;          // [4 nranks] the block table (valid on every rank after the broadcast)
        uint64_t *d_mine = nullptr;           // [2] this rank's capacities
        uint64_t *d_caps = nullptr;           // [2 nranks] every rank's
        uint64_t *h_words = nullptr;          // pinned: table, then caps
        hipEvent_t ev_ready = nullptr, ev_table = nullptr, ev_done = nullptr;
        const uint8_t *root_stream = nullptr;
        const uint64_t *root_offsets = nullptr;
        bool begun = false;
    } slot[kSlots];
    std::string err;
};

namespace {

int sfail(dbde_hip_scatter *s, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (s) s->err = buf;
    return code;
}

#define S_HIP(s, expr)                                                                                  \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return sfail(s, DBDE_HIP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define S_NCCL(s, expr)                                                                                 \
    do {                                                                                                \
        ncclResult_t r_ = (expr);                                                                       \
        if (r_ != ncclSuccess) return sfail(s, DBDE_HIP_ERR_HIP, "%s failed: %s", #expr, rccl()->GetErrorString(r_)); \
    } while (0)

int setup(dbde_hip_scatter *s) {
    S_HIP(s, hipSetDevice(s->device));
    S_HIP(s, hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking));
    const size_t n = (size_t)s->nranks;
    for (auto &sl : s->slot) {
        void *p = nullptr;
        S_HIP(s, hipMalloc(&p, 8 * (4 * n + 2 + 2 * n)));
        sl.d_table = reinterpret_cast<uint64_t *>(p);
        sl.d_mine = sl.d_table + 4 * n;
        sl.d_caps = sl.d_mine + 2;
        S_HIP(s, hipHostMalloc(&p, 8 * (4 * n + 2 * n), hipHostMallocDefault));
        sl.h_words = reinterpret_cast<uint64_t *>(p);
        S_HIP(s, hipEventCreateWithFlags(&sl.ev_ready, hipEventDisableTiming));
        S_HIP(s, hipEventCreateWithFlags(&sl.ev_table, hipEventDisableTiming));
        S_HIP(s, hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming));
        S_HIP(s, hipEventRecord(sl.ev_done, s->comm_stream));
    }
    return DBDE_HIP_OK;
}

dbde_hip_scatter *make(dbde_hip_ctx *ctx, int nranks, int rank, int root) {
    dbde_hip_scatter *s = new dbde_hip_scatter;
    s->ctx = ctx;
    s->device = dbde_hip_device_index(ctx);
    s->ctx_stream = reinterpret_cast<hipStream_t>(dbde_hip_stream_handle(ctx));
    s->nranks = nranks; s->rank = rank; s->root = root;
    return s;
}

}  // namespace

extern "C" {

// ---- pure arithmetic: the block table and the transfers each rank derives from it ---------------------------------

int dbde_hip_scatter_blocks(int nranks, uint64_t n_frames, const uint64_t *frame_offsets, uint64_t stream_bytes,
                            dbde_hip_scatter_block *table) {
    if (nranks < 1 || !table || (n_frames && !frame_offsets)) return DBDE_HIP_ERR_ARG;
    for (int r = 0; r < nranks; r++) {
        const uint64_t lo = n_frames * (uint64_t)r / (uint64_t)nranks, hi = n_frames * (uint64_t)(r + 1) / (uint64_t)nranks;
        const uint64_t b0 = lo < n_frames ? frame_offsets[lo] : stream_bytes, b1 = hi < n_frames ? frame_offsets[hi] : stream_bytes;
        table[r].first_frame = lo;
        table[r].n_frames = hi - lo;
        table[r].byte_start = b0;
        table[r].byte_count = b1 >= b0 ? b1 - b0 : 0;
    }
    return DBDE_HIP_OK;
}

int dbde_hip_scatter_check(int nranks, const dbde_hip_scatter_block *table, const uint64_t *caps) {
    if (nranks < 1 || !table || !caps) return DBDE_HIP_ERR_ARG;
    for (int r = 0; r < nranks; r++)
        if (table[r].byte_count > caps[2 * r] || table[r].n_frames > caps[2 * r + 1]) return DBDE_HIP_ERR_CAPACITY;
    return DBDE_HIP_OK;
}

int dbde_hip_scatter_plan(int nranks, int rank, int root, const dbde_hip_scatter_block *table, uint64_t max_piece,
                          dbde_hip_scatter_op *ops, int max_ops) {
    if (nranks < 1 || rank < 0 || rank >= nranks || root < 0 || root >= nranks || !table) return DBDE_HIP_ERR_ARG;
    if (max_piece == 0) max_piece = 1ull << 30;
    int n = 0;
    auto emit = [&](int peer, int kind, uint64_t src, uint64_t dst, uint64_t bytes) {
        if (ops && n < max_ops) { ops[n].peer = peer; ops[n].kind = kind; ops[n].source_offset = src; ops[n].dest_offset = dst; ops[n].bytes = bytes; }
        n++;
    };
    for (int r = 0; r < nranks; r++) {
        if (rank != root && rank != r) continue;
        const dbde_hip_scatter_block &b = table[r];
        if (r == root) {   // the root's own block: decoded where it lies
            if (rank == root && b.n_frames) emit(root, DBDE_HIP_SCATTER_OWN, b.byte_start, 0, b.byte_count);
            continue;
        }
        // the bytes in pieces, then the block's frame offsets (8 bytes per frame, as they stand in the root's array)
        for (uint64_t at = 0; at < b.byte_count; at += max_piece) {
            const uint64_t m = b.byte_count - at < max_piece ? b.byte_count - at : max_piece;
            if (rank == root) emit(r, DBDE_HIP_SCATTER_SEND_BYTES, b.byte_start + at, at, m);
            else emit(root, DBDE_HIP_SCATTER_RECV_BYTES, b.byte_start + at, at, m);
        }
        for (uint64_t at = 0; at < 8 * b.n_frames; at += max_piece) {
            const uint64_t m = 8 * b.n_frames - at < max_piece ? 8 * b.n_frames - at : max_piece;
            if (rank == root) emit(r, DBDE_HIP_SCATTER_SEND_OFFSETS, 8 * b.first_frame + at, at, m);
            else emit(root, DBDE_HIP_SCATTER_RECV_OFFSETS, 8 * b.first_frame + at, at, m);
        }
    }
    return n;
}

// ---- the handle -------------------------------------------------------------------------------------------------

int dbde_hip_scatter_create(dbde_hip_ctx *ctx, const uint8_t id[DBDE_HIP_GATHER_ID_BYTES], int nranks, int rank, int root,
                            dbde_hip_scatter **out) {
    if (!ctx || !id || !out || nranks < 1 || rank < 0 || rank >= nranks || root < 0 || root >= nranks) return DBDE_HIP_ERR_ARG;
    *out = nullptr;
    Rccl *R = rccl();
    if (!R) return DBDE_HIP_ERR_HIP;
    dbde_hip_scatter *s = make(ctx, nranks, rank, root);
    if (hipSetDevice(s->device) != hipSuccess) { delete s; return DBDE_HIP_ERR_HIP; }
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    if (R->CommInitRank(&s->comm, nranks, u, rank) != ncclSuccess) { delete s; return DBDE_HIP_ERR_HIP; }
    s->own_comm = true;
    const int rc = setup(s);
    if (rc) { dbde_hip_scatter_destroy(s); return rc; }
    *out = s;
    return DBDE_HIP_OK;
}

int dbde_hip_scatter_attach(dbde_hip_ctx *ctx, void *nccl_comm, int nranks, int rank, int root, dbde_hip_scatter **out) {
    if (!ctx || !nccl_comm || !out || nranks < 1 || rank < 0 || rank >= nranks || root < 0 || root >= nranks) return DBDE_HIP_ERR_ARG;
    *out = nullptr;
    if (!rccl()) return DBDE_HIP_ERR_HIP;
    dbde_hip_scatter *s = make(ctx, nranks, rank, root);
    s->comm = reinterpret_cast<ncclComm_t>(nccl_comm);
    const int rc = setup(s);
    if (rc) { dbde_hip_scatter_destroy(s); return rc; }
    *out = s;
    return DBDE_HIP_OK;
}

void dbde_hip_scatter_destroy(dbde_hip_scatter *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->comm_stream) (void)hipStreamSynchronize(s->comm_stream);
    if (s->own_comm && s->comm && rccl()) (void)rccl()->CommDestroy(s->comm);
    for (auto &sl : s->slot) {
        if (sl.d_table) (void)hipFree(sl.d_table);
        if (sl.h_words) (void)hipHostFree(sl.h_words);
        if (sl.ev_ready) (void)hipEventDestroy(sl.ev_ready);
        if (sl.ev_table) (void)hipEventDestroy(sl.ev_table);
        if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
    }
    if (s->comm_stream) (void)hipStreamDestroy(s->comm_stream);
    delete s;
}

const char *dbde_hip_scatter_error(const dbde_hip_scatter *s) { return s ? s->err.c_str() : "null scatter"; }

int dbde_hip_scatter_set_max_message(dbde_hip_scatter *s, uint64_t bytes) {
    if (!s || bytes == 0) return DBDE_HIP_ERR_ARG;
    s->max_piece = bytes;
    return DBDE_HIP_OK;
}

int dbde_hip_scatter_set_capacity(dbde_hip_scatter *s, uint64_t segment_bytes, uint64_t max_frames) {
    if (!s) return DBDE_HIP_ERR_ARG;
    s->seg_cap = segment_bytes;
    s->max_frames = max_frames;
    return DBDE_HIP_OK;
}

int dbde_hip_scatter_begin(dbde_hip_scatter *s, int slot, const uint8_t *d_stream, uint64_t stream_bytes,
                           const uint64_t *d_frame_offsets, const uint32_t *d_n_frames) {
    if (!s || slot < 0 || slot >= kSlots) return DBDE_HIP_ERR_ARG;
    auto &sl = s->slot[slot];
    const bool is_root = s->rank == s->root;
    if (is_root && (!d_stream || !d_frame_offsets || !d_n_frames)) return sfail(s, DBDE_HIP_ERR_ARG, "scatter_begin: the root needs the stream, its frame offsets and their count");
    S_HIP(s, hipSetDevice(s->device));
    // behind everything the codec's stream holds so far (the scanner that produces the offsets and the count)
    S_HIP(s, hipEventRecord(sl.ev_ready, s->ctx_stream));
    S_HIP(s, hipStreamWaitEvent(s->comm_stream, sl.ev_ready, 0));
    // the root's own block is decoded in place: no capacity of its own to meet
    hipLaunchKernelGGL(scatter_caps_kernel, dim3(1), dim3(64), 0, s->comm_stream, is_root ? ~0ull : s->seg_cap,
                       is_root ? ~0ull : s->max_frames, sl.d_mine);
    S_HIP(s, hipGetLastError());
    if (is_root) {
        hipLaunchKernelGGL(scatter_table_kernel, dim3(1), dim3(64), 0, s->comm_stream, d_frame_offsets, d_n_frames, stream_bytes,
                           s->nranks, sl.d_table);
        S_HIP(s, hipGetLastError());
    }
    Rccl *R = rccl();
    S_NCCL(s, R->AllGather(sl.d_mine, sl.d_caps, 2, ncclUint64, s->comm, s->comm_stream));
    S_NCCL(s, R->Broadcast(sl.d_table, sl.d_table, 4 * (size_t)s->nranks, ncclUint64, s->root, s->comm, s->comm_stream));
    const size_t n = (size_t)s->nranks;
    S_HIP(s, hipMemcpyAsync(sl.h_words, sl.d_table, 8 * 4 * n, hipMemcpyDeviceToHost, s->comm_stream));
    S_HIP(s, hipMemcpyAsync(sl.h_words + 4 * n, sl.d_caps, 8 * 2 * n, hipMemcpyDeviceToHost, s->comm_stream));
    S_HIP(s, hipEventRecord(sl.ev_table, s->comm_stream));
    sl.root_stream = d_stream;
    sl.root_offsets = d_frame_offsets;
    sl.begun = true;
    return DBDE_HIP_OK;
}

int dbde_hip_scatter_post(dbde_hip_scatter *s, int slot, uint8_t *d_segment, uint64_t *d_offsets_out,
                          dbde_hip_scatter_block *mine_out, dbde_hip_scatter_block *table_out, uint32_t flags) {
    if (!s || slot < 0 || slot >= kSlots) return DBDE_HIP_ERR_ARG;
    auto &sl = s->slot[slot];
    if (!sl.begun) return sfail(s, DBDE_HIP_ERR_ARG, "scatter_post: slot %d has no table exchange pending", slot);
    if (!d_offsets_out) return sfail(s, DBDE_HIP_ERR_ARG, "scatter_post: null offsets buffer");
    S_HIP(s, hipSetDevice(s->device));
    S_HIP(s, hipEventSynchronize(sl.ev_table));   // the HOST waits for the table; the codec's stream is not involved
    sl.begun = false;
    const size_t n = (size_t)s->nranks;
    std::vector<dbde_hip_scatter_block> table(n);
    for (size_t r = 0; r < n; r++) {
        table[r].first_frame = sl.h_words[4 * r]; table[r].n_frames = sl.h_words[4 * r + 1];
        table[r].byte_start = sl.h_words[4 * r + 2]; table[r].byte_count = sl.h_words[4 * r + 3];
    }
    if (table_out) memcpy(table_out, table.data(), n * sizeof(dbde_hip_scatter_block));
    if (mine_out) *mine_out = table[(size_t)s->rank];
    const bool is_root = s->rank == s->root;
    const bool loopback = (flags & DBDE_HIP_SCATTER_LOOPBACK) != 0 && is_root;
    // one verdict for everybody, from numbers everybody holds: all post or none
    std::vector<uint64_t> caps(sl.h_words + 4 * n, sl.h_words + 6 * n);
    if (loopback) { caps[2 * (size_t)s->root] = s->seg_cap; caps[2 * (size_t)s->root + 1] = s->max_frames; }
    if (dbde_hip_scatter_check(s->nranks, table.data(), caps.data()) != DBDE_HIP_OK)
        return sfail(s, DBDE_HIP_ERR_CAPACITY, "scatter_post: a rank's block does not fit the buffers it declared (dbde_hip_scatter_set_capacity): nothing posted on any rank");
    const dbde_hip_scatter_block &me = table[(size_t)s->rank];
    if (!is_root && me.byte_count && !d_segment) return sfail(s, DBDE_HIP_ERR_ARG, "scatter_post: null segment buffer");
    if (loopback && me.byte_count && !d_segment) return sfail(s, DBDE_HIP_ERR_ARG, "scatter_post: loopback needs a segment buffer");
    const int n_ops = dbde_hip_scatter_plan(s->nranks, s->rank, s->root, table.data(), s->max_piece, nullptr, 0);
    if (n_ops < 0) return sfail(s, DBDE_HIP_ERR_ARG, "scatter_post: bad plan");
    std::vector<dbde_hip_scatter_op> ops((size_t)n_ops);
    (void)dbde_hip_scatter_plan(s->nranks, s->rank, s->root, table.data(), s->max_piece, ops.data(), n_ops);
    Rccl *R = rccl();
    // the received offsets land in the upper half of the caller's array (2 x max_frames words are NOT required: they land
    // at d_offsets_out itself and are rebased in place)
    ncclResult_t bad = ncclSuccess;
    auto keep = [&](ncclResult_t r) { if (bad == ncclSuccess && r != ncclSuccess) bad = r; };
    bool grouped = false;
    auto open_group = [&] { if (!grouped) { keep(R->GroupStart()); grouped = true; } };
    const uint8_t *stream = sl.root_stream;
    const uint8_t *offs_bytes = reinterpret_cast<const uint8_t *>(sl.root_offsets);
    for (const auto &op : ops) {
        switch (op.kind) {
            case DBDE_HIP_SCATTER_SEND_BYTES:
                open_group();
                keep(R->Send(stream + op.source_offset, (size_t)op.bytes, ncclUint8, op.peer, s->comm, s->comm_stream));
                break;
            case DBDE_HIP_SCATTER_SEND_OFFSETS:
                open_group();
                keep(R->Send(offs_bytes + op.source_offset, (size_t)op.bytes, ncclUint8, op.peer, s->comm, s->comm_stream));
                break;
            case DBDE_HIP_SCATTER_RECV_BYTES:
                open_group();
                keep(R->Recv(d_segment + op.dest_offset, (size_t)op.bytes, ncclUint8, op.peer, s->comm, s->comm_stream));
                break;
            case DBDE_HIP_SCATTER_RECV_OFFSETS:
                open_group();
                keep(R->Recv(reinterpret_cast<uint8_t *>(d_offsets_out) + op.dest_offset, (size_t)op.bytes, ncclUint8, op.peer, s->comm, s->comm_stream));
                break;
            default:   // the root's own block
                if (loopback) {   // tests and one-GPU rehearsals: through ncclSend / ncclRecv to itself, in pieces
                    open_group();
                    for (uint64_t at = 0; at < op.bytes; at += s->max_piece) {
                        const uint64_t m = op.bytes - at < s->max_piece ? op.bytes - at : s->max_piece;
                        keep(R->Send(stream + op.source_offset + at, (size_t)m, ncclUint8, s->rank, s->comm, s->comm_stream));
                        keep(R->Recv(d_segment + at, (size_t)m, ncclUint8, s->rank, s->comm, s->comm_stream));
                    }
                    const uint64_t ob = 8 * me.n_frames;
                    for (uint64_t at = 0; at < ob; at += s->max_piece) {
                        const uint64_t m = ob - at < s->max_piece ? ob - at : s->max_piece;
                        keep(R->Send(offs_bytes + 8 * me.first_frame + at, (size_t)m, ncclUint8, s->rank, s->comm, s->comm_stream));
                        keep(R->Recv(reinterpret_cast<uint8_t *>(d_offsets_out) + at, (size_t)m, ncclUint8, s->rank, s->comm, s->comm_stream));
                    }
                }
                break;
        }
    }
    if (grouped) keep(R->GroupEnd());
    if (bad != ncclSuccess) return sfail(s, DBDE_HIP_ERR_HIP, "scatter_post: RCCL: %s", R->GetErrorString(bad));
    // the block's offsets, relative to its first byte
    if (me.n_frames) {
        const uint64_t *src = (is_root && !loopback) ? sl.root_offsets + me.first_frame : d_offsets_out;
        const unsigned blocks = (unsigned)((me.n_frames + 255) / 256 > 1024 ? 1024 : (me.n_frames + 255) / 256);
        hipLaunchKernelGGL(scatter_rebase_kernel, dim3(blocks), dim3(256), 0, s->comm_stream, src, d_offsets_out, me.n_frames, me.byte_start);
        S_HIP(s, hipGetLastError());
    }
    S_HIP(s, hipEventRecord(sl.ev_done, s->comm_stream));
    return DBDE_HIP_OK;
}

int dbde_hip_scatter_join(dbde_hip_scatter *s, int slot) {
    if (!s || slot < 0 || slot >= kSlots) return DBDE_HIP_ERR_ARG;
    S_HIP(s, hipSetDevice(s->device));
    S_HIP(s, hipStreamWaitEvent(s->ctx_stream, s->slot[slot].ev_done, 0));
    return DBDE_HIP_OK;
}

int dbde_hip_scatter_sync(dbde_hip_scatter *s, int slot) {
    if (!s || slot < 0 || slot >= kSlots) return DBDE_HIP_ERR_ARG;
    S_HIP(s, hipSetDevice(s->device));
    S_HIP(s, hipEventSynchronize(s->slot[slot].ev_done));
    return DBDE_HIP_OK;
}

}  // extern "C"
