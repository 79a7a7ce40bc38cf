// dbde_gather.cpp -- the one exchange step of the multi-GPU path (include/dbde_hip.h "multi-GPU"):
// the variable-length gather of every rank's compressed byte stream to a root, over RCCL / xGMI.
//
// The reference has no counterpart (it is a single-threaded library); what has to be preserved is the stream layout
// (README.md:12-23: frames simply follow each other) -- ranks own CONTIGUOUS frame blocks, so the gathered stream is
// the ranks' segments in rank order and nothing is interleaved.  RCCL has no gatherv: the byte counts are
// all-gathered (8 bytes per rank, device to device), both ends derive the same displacements and message pieces
// from them (gather_plan, pure host arithmetic, tested without a GPU), and the bytes travel as grouped
// ncclSend / ncclRecv straight into the root's window at their displacement.
//
// Nothing here stalls the codec's stream or copies a byte twice:
//   * the byte count is a DEVICE word (the encoder's per-frame offset + size of the batch's last frame); its exchange
//     is enqueued on the gather's own stream behind the encode (dbde_hip_gather_begin returns at once);
//   * the host reads the counts only when it posts the transfers (dbde_hip_gather_post) -- one batch behind in a
//     pipelined caller, so that wait ends long before the device runs dry;
//   * the root's own segment is not moved when it already lies at its displacement (root 0 encodes straight into its
//     window: displacement 0).
// librccl is opened at run time (dlopen "librccl.so.1": in a PyTorch process that is the copy torch has loaded, so the
// process holds ONE RCCL), which keeps single-GPU users of libdbde_hip.so free of the dependency.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dbde_hip.h"
#include "dbde_rccl.h"

namespace {

using dbde_rccl::Rccl;
using dbde_rccl::rccl;

// the batch's byte count: offset of its last frame + that frame's length (either word may be absent); beside it the
// capacity of the root's window as this rank knows it (the root's own word is the one that counts)
__global__ void gather_count_kernel(const uint64_t *a, const uint64_t *b, uint64_t cap, uint64_t *out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (a ? *a : 0ull) + (b ? *b : 0ull); out[1] = cap; }
}

constexpr int kSlots = 2;

}  // namespace

struct dbde_hip_gather {
    dbde_hip_ctx *ctx = nullptr;
    int device = 0, nranks = 1, rank = 0, root = 0;
    ncclComm_t comm = nullptr;
    bool own_comm = false;
    hipStream_t ctx_stream = nullptr, comm_stream = nullptr;
    uint64_t max_piece = 1ull << 30;
    uint64_t window_cap = ~0ull;      // root: bytes its window holds (dbde_hip_gather_set_window); travels with every size exchange
    struct Slot {
        uint64_t *d_mine = nullptr, *d_sizes = nullptr;   // device: this rank's {count, window capacity}, every rank's pair
        uint64_t *h_sizes = nullptr;                      // pinned host copy of d_sizes
        hipEvent_t ev_ready = nullptr, ev_sizes = nullptr, ev_done = nullptr;
        bool begun = false;
    } slot[kSlots];
    std::string err;
};

namespace {

int gfail(dbde_hip_gather *g, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g) g->err = buf;
    return code;
}

#define G_HIP(g, expr)                                                                                  \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return gfail(g, DBDE_HIP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define G_NCCL(g, expr)                                                                                 \
    do {                                                                                                \
        ncclResult_t r_ = (expr);                                                                       \
        if (r_ != ncclSuccess) return gfail(g, DBDE_HIP_ERR_HIP, "%s failed: %s", #expr, rccl()->GetErrorString(r_)); \
    } while (0)

int setup(dbde_hip_gather *g) {
    G_HIP(g, hipSetDevice(g->device));
    G_HIP(g, hipStreamCreateWithFlags(&g->comm_stream, hipStreamNonBlocking));
    for (auto &s : g->slot) {
        void *p = nullptr;
        G_HIP(g, hipMalloc(&p, 16 * (size_t)(g->nranks + 1)));
        s.d_mine = reinterpret_cast<uint64_t *>(p);
        s.d_sizes = s.d_mine + 2;
        G_HIP(g, hipHostMalloc(&p, 16 * (size_t)g->nranks, hipHostMallocDefault));
        s.h_sizes = reinterpret_cast<uint64_t *>(p);
        G_HIP(g, hipEventCreateWithFlags(&s.ev_ready, hipEventDisableTiming));
        G_HIP(g, hipEventCreateWithFlags(&s.ev_sizes, hipEventDisableTiming));
        G_HIP(g, hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
        G_HIP(g, hipEventRecord(s.ev_done, g->comm_stream));   // "nothing in flight": join before the first post is a no-op
    }
    return DBDE_HIP_OK;
}

}  // namespace

extern "C" {

// ---- the plan: pure arithmetic, identical on every rank ---------------------------------------------------------
int dbde_hip_gather_plan(int nranks, int rank, int root, const uint64_t *sizes, uint64_t max_piece,
                         dbde_hip_gather_op *ops, int max_ops, uint64_t *total_out) {
    if (nranks < 1 || rank < 0 || rank >= nranks || root < 0 || root >= nranks || !sizes) return DBDE_HIP_ERR_ARG;
    if (max_piece == 0) max_piece = 1ull << 30;
    uint64_t total = 0;
    int n = 0;
    auto emit = [&](int peer, int kind, uint64_t seg_off, uint64_t win_off, uint64_t bytes) {
        if (ops && n < max_ops) {
            ops[n].peer = peer; ops[n].kind = kind;
            ops[n].segment_offset = seg_off; ops[n].window_offset = win_off; ops[n].bytes = bytes;
        }
        n++;
    };
    for (int r = 0; r < nranks; r++) {
        const uint64_t disp = total;
        total += sizes[r];
        if (r == root) {
            if (rank == root && sizes[r]) emit(root, DBDE_HIP_GATHER_OWN, 0, disp, sizes[r]);   // the root's own bytes: in place, or one copy
            continue;
        }
        if (rank != root && rank != r) continue;
        for (uint64_t at = 0; at < sizes[r]; at += max_piece) {
            const uint64_t b = sizes[r] - at < max_piece ? sizes[r] - at : max_piece;
            if (rank == root) emit(r, DBDE_HIP_GATHER_RECV, 0, disp + at, b);
            else emit(root, DBDE_HIP_GATHER_SEND, at, disp + at, b);
        }
    }
    if (total_out) *total_out = total;
    return n;
}

// The verdict every rank reaches from the exchanged {count, capacity} pairs -- the same on all of them, so that either
// everybody posts its transfers or nobody does (a root that found its window too small while its peers had already
// posted their sends left them waiting for ever).
int dbde_hip_gather_check(int nranks, int root, const uint64_t *pairs, uint64_t *total_out) {
    if (nranks < 1 || root < 0 || root >= nranks || !pairs) return DBDE_HIP_ERR_ARG;
    uint64_t total = 0;
    for (int r = 0; r < nranks; r++) {
        if (pairs[2 * r] > ~0ull - total) return DBDE_HIP_ERR_CAPACITY;
        total += pairs[2 * r];
    }
    if (total_out) *total_out = total;
    return total > pairs[2 * root + 1] ? DBDE_HIP_ERR_CAPACITY : DBDE_HIP_OK;
}

int dbde_hip_gather_unique_id(uint8_t id[DBDE_HIP_GATHER_ID_BYTES]) {
    Rccl *R = rccl();
    if (!R || !id) return DBDE_HIP_ERR_HIP;
    static_assert(DBDE_HIP_GATHER_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "rendezvous token size");
    ncclUniqueId u;
    if (R->GetUniqueId(&u) != ncclSuccess) return DBDE_HIP_ERR_HIP;
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return DBDE_HIP_OK;
}

int dbde_hip_gather_create(dbde_hip_ctx *ctx, const uint8_t id[DBDE_HIP_GATHER_ID_BYTES], int nranks, int rank, int root,
                           dbde_hip_gather **out) {
    if (!ctx || !id || !out || nranks < 1 || rank < 0 || rank >= nranks || root < 0 || root >= nranks) return DBDE_HIP_ERR_ARG;
    *out = nullptr;
    Rccl *R = rccl();
    if (!R) return DBDE_HIP_ERR_HIP;
    dbde_hip_gather *g = new dbde_hip_gather;
    g->ctx = ctx;
    g->device = dbde_hip_device_index(ctx);
    g->ctx_stream = reinterpret_cast<hipStream_t>(dbde_hip_stream_handle(ctx));
    g->nranks = nranks; g->rank = rank; g->root = root;
    if (hipSetDevice(g->device) != hipSuccess) { delete g; return DBDE_HIP_ERR_HIP; }
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    if (R->CommInitRank(&g->comm, nranks, u, rank) != ncclSuccess) { delete g; return DBDE_HIP_ERR_HIP; }
    g->own_comm = true;
    const int rc = setup(g);
    if (rc) { dbde_hip_gather_destroy(g); return rc; }
    *out = g;
    return DBDE_HIP_OK;
}

int dbde_hip_gather_attach(dbde_hip_ctx *ctx, void *nccl_comm, int nranks, int rank, int root, dbde_hip_gather **out) {
    if (!ctx || !nccl_comm || !out || nranks < 1 || rank < 0 || rank >= nranks || root < 0 || root >= nranks) return DBDE_HIP_ERR_ARG;
    *out = nullptr;
    if (!rccl()) return DBDE_HIP_ERR_HIP;
    dbde_hip_gather *g = new dbde_hip_gather;
    g->ctx = ctx;
    g->device = dbde_hip_device_index(ctx);
    g->ctx_stream = reinterpret_cast<hipStream_t>(dbde_hip_stream_handle(ctx));
    g->nranks = nranks; g->rank = rank; g->root = root;
    g->comm = reinterpret_cast<ncclComm_t>(nccl_comm);
    const int rc = setup(g);
    if (rc) { dbde_hip_gather_destroy(g); return rc; }
    *out = g;
    return DBDE_HIP_OK;
}

void dbde_hip_gather_destroy(dbde_hip_gather *g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->comm_stream) (void)hipStreamSynchronize(g->comm_stream);
    if (g->own_comm && g->comm && rccl()) (void)rccl()->CommDestroy(g->comm);
    for (auto &s : g->slot) {
        if (s.d_mine) (void)hipFree(s.d_mine);
        if (s.h_sizes) (void)hipHostFree(s.h_sizes);
        if (s.ev_ready) (void)hipEventDestroy(s.ev_ready);
        if (s.ev_sizes) (void)hipEventDestroy(s.ev_sizes);
        if (s.ev_done) (void)hipEventDestroy(s.ev_done);
    }
    if (g->comm_stream) (void)hipStreamDestroy(g->comm_stream);
    delete g;
}

const char *dbde_hip_gather_error(const dbde_hip_gather *g) { return g ? g->err.c_str() : "null gather"; }

int dbde_hip_gather_set_max_message(dbde_hip_gather *g, uint64_t bytes) {
    if (!g || bytes == 0) return DBDE_HIP_ERR_ARG;
    g->max_piece = bytes;
    return DBDE_HIP_OK;
}

int dbde_hip_gather_set_window(dbde_hip_gather *g, uint64_t window_bytes) {
    if (!g) return DBDE_HIP_ERR_ARG;
    g->window_cap = window_bytes;
    return DBDE_HIP_OK;
}

int dbde_hip_gather_begin(dbde_hip_gather *g, int slot, const uint64_t *d_last_offset, const uint64_t *d_last_bytes) {
    if (!g || slot < 0 || slot >= kSlots) return DBDE_HIP_ERR_ARG;
    auto &s = g->slot[slot];
    G_HIP(g, hipSetDevice(g->device));
    // behind everything the codec's stream holds so far (the encode that produces the count and the bytes)
    G_HIP(g, hipEventRecord(s.ev_ready, g->ctx_stream));
    G_HIP(g, hipStreamWaitEvent(g->comm_stream, s.ev_ready, 0));
    hipLaunchKernelGGL(gather_count_kernel, dim3(1), dim3(64), 0, g->comm_stream, d_last_offset, d_last_bytes,
                       g->rank == g->root ? g->window_cap : 0ull, s.d_mine);
    G_HIP(g, hipGetLastError());
    G_NCCL(g, rccl()->AllGather(s.d_mine, s.d_sizes, 2, ncclUint64, g->comm, g->comm_stream));
    G_HIP(g, hipMemcpyAsync(s.h_sizes, s.d_sizes, 16 * (size_t)g->nranks, hipMemcpyDeviceToHost, g->comm_stream));
    G_HIP(g, hipEventRecord(s.ev_sizes, g->comm_stream));
    s.begun = true;
    return DBDE_HIP_OK;
}

int dbde_hip_gather_post(dbde_hip_gather *g, int slot, const uint8_t *d_segment, uint8_t *d_window, size_t window_bytes,
                         uint64_t *sizes_out, uint32_t flags) {
    if (!g || slot < 0 || slot >= kSlots) return DBDE_HIP_ERR_ARG;
    auto &s = g->slot[slot];
    if (!s.begun) return gfail(g, DBDE_HIP_ERR_ARG, "gather_post: slot %d has no size exchange pending", slot);
    if (g->rank == g->root ? !d_window : !d_segment) return gfail(g, DBDE_HIP_ERR_ARG, "gather_post: null buffer");
    G_HIP(g, hipSetDevice(g->device));
    G_HIP(g, hipEventSynchronize(s.ev_sizes));   // the HOST waits for the counts; the codec's stream is not involved
    s.begun = false;
    std::vector<uint64_t> sizes((size_t)g->nranks);
    for (int r = 0; r < g->nranks; r++) sizes[(size_t)r] = s.h_sizes[2 * r];
    if (sizes_out) memcpy(sizes_out, sizes.data(), 8 * (size_t)g->nranks);
    // The data-dependent verdict is the same on every rank (the root's capacity travelled with the counts): all post or none
    uint64_t total = 0;
    if (dbde_hip_gather_check(g->nranks, g->root, s.h_sizes, &total) != DBDE_HIP_OK)
        return gfail(g, DBDE_HIP_ERR_CAPACITY, "gather_post: %llu bytes do not fit the root window (%llu): nothing posted on any rank",
                     (unsigned long long)total, (unsigned long long)s.h_sizes[2 * g->root + 1]);
    if (g->rank == g->root && total > window_bytes)   // (a caller's error, not the data's: the window is smaller than what was declared)
        return gfail(g, DBDE_HIP_ERR_ARG, "gather_post: window_bytes %zu below the declared capacity (dbde_hip_gather_set_window) and the %llu bytes on their way",
                     window_bytes, (unsigned long long)total);
    const int n_ops = dbde_hip_gather_plan(g->nranks, g->rank, g->root, sizes.data(), g->max_piece, nullptr, 0, &total);
    if (n_ops < 0) return gfail(g, DBDE_HIP_ERR_ARG, "gather_post: bad plan");
    std::vector<dbde_hip_gather_op> ops((size_t)n_ops);
    (void)dbde_hip_gather_plan(g->nranks, g->rank, g->root, sizes.data(), g->max_piece, ops.data(), n_ops, nullptr);
    Rccl *R = rccl();
    const bool loopback = (flags & DBDE_HIP_GATHER_LOOPBACK) != 0 && g->rank == g->root;
    if (loopback)
        for (const auto &op : ops)
            if (op.kind == DBDE_HIP_GATHER_OWN && (!d_segment || d_segment == d_window + op.window_offset))
                return gfail(g, DBDE_HIP_ERR_ARG, "gather_post: loopback needs a segment outside the window");
    // one group: every send of this rank (or every receive of the root) is posted together; an error inside the group
    // still closes it
    ncclResult_t bad = ncclSuccess;
    auto keep = [&](ncclResult_t r) { if (bad == ncclSuccess && r != ncclSuccess) bad = r; };
    bool grouped = false;
    for (const auto &op : ops) {
        if (op.kind == DBDE_HIP_GATHER_OWN && !loopback) continue;
        if (!grouped) { keep(R->GroupStart()); grouped = true; }
        if (op.kind == DBDE_HIP_GATHER_SEND) {
            keep(R->Send(d_segment + op.segment_offset, (size_t)op.bytes, ncclUint8, op.peer, g->comm, g->comm_stream));
        } else if (op.kind == DBDE_HIP_GATHER_RECV) {
            keep(R->Recv(d_window + op.window_offset, (size_t)op.bytes, ncclUint8, op.peer, g->comm, g->comm_stream));
        } else {   // loopback (tests, one-GPU rehearsals): the root's own bytes take the send/recv path too, in pieces
            for (uint64_t at = 0; at < op.bytes; at += g->max_piece) {
                const uint64_t b = op.bytes - at < g->max_piece ? op.bytes - at : g->max_piece;
                keep(R->Send(d_segment + at, (size_t)b, ncclUint8, g->rank, g->comm, g->comm_stream));
                keep(R->Recv(d_window + op.window_offset + at, (size_t)b, ncclUint8, g->rank, g->comm, g->comm_stream));
            }
        }
    }
    if (grouped) keep(R->GroupEnd());
    if (bad != ncclSuccess) return gfail(g, DBDE_HIP_ERR_HIP, "gather_post: RCCL: %s", R->GetErrorString(bad));
    if (!loopback) {
        for (const auto &op : ops) {   // the root's own segment: nothing to do when it was encoded where it belongs
            if (op.kind != DBDE_HIP_GATHER_OWN || !d_segment || d_segment == d_window + op.window_offset) continue;
            G_HIP(g, hipMemcpyAsync(d_window + op.window_offset, d_segment, (size_t)op.bytes, hipMemcpyDeviceToDevice, g->comm_stream));
        }
    }
    G_HIP(g, hipEventRecord(s.ev_done, g->comm_stream));
    return DBDE_HIP_OK;
}

int dbde_hip_gather_join(dbde_hip_gather *g, int slot) {
    if (!g || slot < 0 || slot >= kSlots) return DBDE_HIP_ERR_ARG;
    G_HIP(g, hipSetDevice(g->device));
    G_HIP(g, hipStreamWaitEvent(g->ctx_stream, g->slot[slot].ev_done, 0));
    return DBDE_HIP_OK;
}

int dbde_hip_gather_sync(dbde_hip_gather *g, int slot) {
    if (!g || slot < 0 || slot >= kSlots) return DBDE_HIP_ERR_ARG;
    G_HIP(g, hipSetDevice(g->device));
    G_HIP(g, hipEventSynchronize(g->slot[slot].ev_done));
    return DBDE_HIP_OK;
}

int dbde_hip_gather_rccl_version(void) {
    Rccl *R = rccl();
    int v = 0;
    if (!R || R->GetVersion(&v) != ncclSuccess) return 0;
    return v;
}

}  // extern "C"
