// dbde_capi.cpp -- implementation of include/dbde_hip.h (the C-ABI of libdbde_hip.so).
//
// Host side only: argument checking, workspace, launches, and the byte marshalling the
// host-pointer entry points need.  All tile arithmetic is in dbde_kernels.hip; nothing here
// (or anywhere in this library) computes DBDE on the CPU.
#include "../../include/dbde_hip.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dbde16_kernels.h"
#include "dbde_kernels.h"

using namespace dbde;

namespace {

struct TimedSpan {
    hipEvent_t a, b;
    int kind;
};

struct Geometry {
    uint32_t w, h, T;
    uint64_t pixels;
};

}  // namespace

struct dbde_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;         // dbde_hip_create_on_own_stream: destroyed with the context
    std::string err;
    std::string arch;

    // look-back workspace: [two sets of control words: 2 x 8 x u32][state: n_chunks x u64].  Zeroed when it is allocated
    // and after a failed launch; otherwise every launch leaves it as it found it (persistent encoder: each record is
    // cleared by its reader, the scanner clears the control words of the launch after it; small launches tag theirs)
    uint8_t *lb = nullptr;
    size_t lb_bytes = 0;
    uint32_t enc_parity = 0;         // which set of control words the next persistent launch uses
    // scan-ahead: a second stream so that the frame-to-frame walk of the NEXT batch runs beside the decode of this one
    hipStream_t scan_stream = nullptr;
    hipEvent_t scan_ev_main = nullptr, scan_ev_done = nullptr;
    // speculative stream walk: temporary position lists of the segments
    uint8_t *scan_ws = nullptr;
    size_t scan_ws_bytes = 0;
    // DBDE16 encode workspace (per-tile depth / minimum, chunk and frame totals, frame bases, arrival counter)
    uint8_t *w16 = nullptr;
    size_t w16_bytes = 0;
    bool lb_fresh = true;            // the block holds arbitrary bits (fresh allocation): a small launch zeroes it before its first use
    uint32_t enc_epoch = 0;          // tag of the last small encode launch's records (launch_encode_small)
    // decode workspace
    uint32_t *chunk_off = nullptr;
    size_t chunk_off_n = 0;
    uint32_t *frame_ok = nullptr;
    size_t frame_ok_n = 0;
    uint32_t *idx_ctr = nullptr;     // [2 * n]: arrival counters, then flags, of the split index kernel (kept zero)
    size_t idx_ctr_n = 0;
    unsigned long long *fuse_rec = nullptr;   // records of the fused index + decode launch (epoch-tagged, never cleared)
    size_t fuse_rec_n = 0;
    uint32_t fuse_epoch = 0;
    // sticky failure word (device) + scratch
    uint32_t *sticky = nullptr;
    uint64_t *scratch64 = nullptr;   // small device scratch: [0..3]
    // staging for the host-pointer entry points
    uint8_t *st_img = nullptr;
    size_t st_img_bytes = 0;
    uint8_t *st_pack = nullptr;
    size_t st_pack_bytes = 0;
    // ... and their pinned host twins (dbde_hip_set_host_staging): the caller's pageable bytes are copied through these
    // by the calling thread, so that the DMA never has to pin (and the runtime's pinning of pageable memory, which
    // serialises concurrent callers, is out of the way)
    uint64_t *h_words = nullptr;     // pinned, device-visible: [0] the encoder's byte count, [1..4] the decoder's frame result -- the kernels write them
                                     // straight into host memory, so a call has no small copies between its two big ones
    uint8_t *d_hdr = nullptr;        // a frame header {2, 0, 0} in device memory (dbde_hip_unpack_image puts it in front of the caller's frame data)
    int host_staging = 0;
    uint8_t *h_img = nullptr;
    size_t h_img_bytes = 0;
    uint8_t *h_pack = nullptr;
    size_t h_pack_bytes = 0;
    // timing
    uint32_t exp_flags = 0;          // $DBDE_HIP_EXPERIMENT (tuning experiments only)
    uint32_t enc_grid = 0;           // resident workgroups for the persistent encoder
    uint32_t enc16_grid = 0;         // the same for the DBDE16 encoder (queried at its first call)
    int n_cu = 0;
    uint64_t *diag = nullptr;        // [16] phase cycle sums of diagnostic launches
    bool timing = false;
    std::vector<TimedSpan> spans;
    double acc_ms[4] = {0, 0, 0, 0};     // encode, decode index, decode, stream scan
    uint64_t acc_n[4] = {0, 0, 0, 0};
};

namespace {

int fail(dbde_hip_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(ctx, DBDE_HIP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));  \
    } while (0)

bool geometry(int W, int H, Geometry &g) {
    if (W <= 0 || H <= 0) return false;
    uint64_t w = ((uint64_t)W + 7) / 8, h = ((uint64_t)H + 7) / 8;
    uint64_t T = w * h;
    if (T >= (1ull << 27)) return false;   // in-frame payload words must stay below 2^30
    g.w = (uint32_t)w;
    g.h = (uint32_t)h;
    g.T = (uint32_t)T;
    g.pixels = (uint64_t)W * (uint64_t)H;
    return true;
}

template <typename Ptr>
int grow(dbde_hip_ctx *ctx, Ptr &p, size_t &have, size_t want_elems, size_t elem_bytes, bool uncached = false) {
    if (want_elems <= have) return DBDE_HIP_OK;
    // everything queued may still be using the old block
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (p) HIP_TRY(ctx, hipFree(p));
    p = nullptr;
    have = 0;
    size_t n = want_elems + want_elems / 4 + 64;
    void *q = nullptr;
    // chunk records are written by one CU and polled by others through sc1 atomics: keeping them out
    // of the (per-XCD, mutually incoherent) L2s measured 1.5-2.5 % faster; plain memory if refused
    if (uncached && hipExtMallocWithFlags(&q, n * elem_bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        q = nullptr;
    }
    if (!q) HIP_TRY(ctx, hipMalloc(&q, n * elem_bytes));
    p = reinterpret_cast<Ptr>(q);
    have = n;
    return DBDE_HIP_OK;
}

void span_begin(dbde_hip_ctx *ctx, int kind) {
    if (!ctx->timing) return;
    TimedSpan s;
    s.kind = kind;
    if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return;
    (void)hipEventRecord(s.a, ctx->stream);
    ctx->spans.push_back(s);
}
void span_end(dbde_hip_ctx *ctx) {
    if (!ctx->timing || ctx->spans.empty()) return;
    (void)hipEventRecord(ctx->spans.back().b, ctx->stream);
}

void put32(uint8_t *p, uint32_t v) { for (int i = 0; i < 4; i++) p[i] = (uint8_t)(v >> (8 * i)); }
void put64(uint8_t *p, uint64_t v) { for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i)); }
uint32_t get32(const uint8_t *p) { uint32_t v = 0; for (int i = 0; i < 4; i++) v |= (uint32_t)p[i] << (8 * i); return v; }
uint64_t get64(const uint8_t *p) { uint64_t v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)p[i] << (8 * i); return v; }

int grow_pinned(dbde_hip_ctx *ctx, uint8_t *&p, size_t &have, size_t want) {
    if (want <= have) return DBDE_HIP_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (p) HIP_TRY(ctx, hipHostFree(p));
    p = nullptr;
    have = 0;
    void *q = nullptr;
    HIP_TRY(ctx, hipHostMalloc(&q, want + want / 4 + 64, hipHostMallocDefault));
    p = reinterpret_cast<uint8_t *>(q);
    have = want + want / 4 + 64;
    return DBDE_HIP_OK;
}

int ensure_staging(dbde_hip_ctx *ctx, size_t img_bytes, size_t pack_bytes) {
    if (!ctx->h_words) {
        void *q = nullptr;
        HIP_TRY(ctx, hipHostMalloc(&q, 64, hipHostMallocDefault));
        ctx->h_words = reinterpret_cast<uint64_t *>(q);
        memset(ctx->h_words, 0, 64);
        HIP_TRY(ctx, hipMalloc(&q, 32));
        ctx->d_hdr = reinterpret_cast<uint8_t *>(q);
        uint8_t hdr[20];
        const dbde_hip_frame_header fh = {2, 0, 0};
        dbde_hip_pack_frame_header(&fh, hdr);
        HIP_TRY(ctx, hipMemcpy(ctx->d_hdr, hdr, 20, hipMemcpyHostToDevice));   // (once per context)
    }
    int rc = grow(ctx, ctx->st_img, ctx->st_img_bytes, img_bytes + 64, 1);
    if (rc) return rc;
    rc = grow(ctx, ctx->st_pack, ctx->st_pack_bytes, pack_bytes + 64, 1);
    if (rc || !ctx->host_staging) return rc;
    rc = grow_pinned(ctx, ctx->h_img, ctx->h_img_bytes, img_bytes + 64);
    if (rc) return rc;
    return grow_pinned(ctx, ctx->h_pack, ctx->h_pack_bytes, pack_bytes + 64);
}

// Host -> device / device -> host of the host-pointer entry points: straight from / to the caller's (pageable) memory, or
// through the context's pinned twin (host_staging).  `via`: the pinned buffer that shadows the device buffer.
bool h2d(dbde_hip_ctx *ctx, void *d_dst, const void *src, size_t n, uint8_t *via) {
    if (ctx->host_staging && via) { memcpy(via, src, n); src = via; }
    return hipMemcpyAsync(d_dst, src, n, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
}

}  // namespace

extern "C" {

// ---- context ---------------------------------------------------------------------------------

int dbde_hip_create(int device, void *stream, dbde_hip_ctx **out) {
    if (!out) return DBDE_HIP_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return DBDE_HIP_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return DBDE_HIP_ERR_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return DBDE_HIP_ERR_HIP;
    // kernels are built for gfx950 only; refuse anything else instead of failing at launch
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return DBDE_HIP_ERR_HIP;
    dbde_hip_ctx *ctx = new dbde_hip_ctx;
    ctx->device = device;
    ctx->stream = reinterpret_cast<hipStream_t>(stream);
    ctx->arch = prop.gcnArchName;
    {
        int per_cu = encode_blocks_per_cu();
        if (const char *g = getenv("DBDE_HIP_ENC_BLOCKS_PER_CU")) per_cu = atoi(g) > 0 ? atoi(g) : per_cu;
        ctx->enc_grid = (uint32_t)(per_cu * prop.multiProcessorCount);
        if (ctx->enc_grid > kEncMaxGrid) ctx->enc_grid = kEncMaxGrid;   // one mode flag per workgroup (attach_lookback)
        ctx->n_cu = prop.multiProcessorCount;
    }
    if (const char *e = getenv("DBDE_HIP_EXPERIMENT")) ctx->exp_flags = (uint32_t)strtoul(e, nullptr, 0);
    void *p = nullptr;
#ifdef DBDE_DIAG
    const size_t small_block = 64 + 128 + 8 * 16 * 1024;   // + per-workgroup timeline of the persistent encoder ([1024][16] u64)
#else
    const size_t small_block = 64 + 128;
#endif
    if (hipMalloc(&p, small_block) != hipSuccess) { delete ctx; return DBDE_HIP_ERR_HIP; }
    ctx->sticky = reinterpret_cast<uint32_t *>(p);
    ctx->scratch64 = reinterpret_cast<uint64_t *>(reinterpret_cast<uint8_t *>(p) + 16);
    ctx->diag = reinterpret_cast<uint64_t *>(reinterpret_cast<uint8_t *>(p) + 64);
    if (hipMemsetAsync(p, 0, small_block, ctx->stream) != hipSuccess) { (void)hipFree(p); delete ctx; return DBDE_HIP_ERR_HIP; }
    *out = ctx;
    return DBDE_HIP_OK;
}

int dbde_hip_create_on_own_stream(int device, dbde_hip_ctx **out) {
    if (!out) return DBDE_HIP_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return DBDE_HIP_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return DBDE_HIP_ERR_HIP;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return DBDE_HIP_ERR_HIP;
    const int rc = dbde_hip_create(device, s, out);
    if (rc != DBDE_HIP_OK) { (void)hipStreamDestroy(s); return rc; }
    (*out)->own_stream = true;
    return DBDE_HIP_OK;
}

void dbde_hip_destroy(dbde_hip_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &s : ctx->spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    if (ctx->scan_stream) { (void)hipStreamSynchronize(ctx->scan_stream); (void)hipStreamDestroy(ctx->scan_stream); }
    if (ctx->scan_ev_main) (void)hipEventDestroy(ctx->scan_ev_main);
    if (ctx->scan_ev_done) (void)hipEventDestroy(ctx->scan_ev_done);
    if (ctx->scan_ws) (void)hipFree(ctx->scan_ws);
    if (ctx->w16) (void)hipFree(ctx->w16);
    if (ctx->lb) (void)hipFree(ctx->lb);
    if (ctx->chunk_off) (void)hipFree(ctx->chunk_off);
    if (ctx->frame_ok) (void)hipFree(ctx->frame_ok);
    if (ctx->idx_ctr) (void)hipFree(ctx->idx_ctr);
    if (ctx->fuse_rec) (void)hipFree(ctx->fuse_rec);
    if (ctx->sticky) (void)hipFree(ctx->sticky);
    if (ctx->st_img) (void)hipFree(ctx->st_img);
    if (ctx->st_pack) (void)hipFree(ctx->st_pack);
    if (ctx->h_words) (void)hipHostFree(ctx->h_words);
    if (ctx->d_hdr) (void)hipFree(ctx->d_hdr);
    if (ctx->h_img) (void)hipHostFree(ctx->h_img);
    if (ctx->h_pack) (void)hipHostFree(ctx->h_pack);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int dbde_hip_sync(dbde_hip_ctx *ctx) {
    if (!ctx) return DBDE_HIP_ERR_ARG;
    uint32_t flag = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&flag, ctx->sticky, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (flag) {
        ctx->lb_fresh = true;   // the failed launch left its records behind: the next encode clears the workspace first
        HIP_TRY(ctx, hipMemsetAsync(ctx->sticky, 0, 4, ctx->stream));
        return fail(ctx, DBDE_HIP_ERR_DEVICE, "encode kernel: chunk look-back timed out");
    }
    return DBDE_HIP_OK;
}

int dbde_hip_set_host_staging(dbde_hip_ctx *ctx, int pinned) {
    if (!ctx) return DBDE_HIP_ERR_ARG;
    ctx->host_staging = pinned ? 1 : 0;
    return DBDE_HIP_OK;
}

const char *dbde_hip_last_error(const dbde_hip_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }
const char *dbde_hip_device_arch(const dbde_hip_ctx *ctx) { return ctx ? ctx->arch.c_str() : ""; }
void *dbde_hip_stream_handle(const dbde_hip_ctx *ctx) { return ctx ? reinterpret_cast<void *>(ctx->stream) : nullptr; }
int dbde_hip_device_index(const dbde_hip_ctx *ctx) { return ctx ? ctx->device : -1; }

// ---- sizes -----------------------------------------------------------------------------------

size_t dbde_hip_max_frame_bytes(int W, int H) {
    Geometry g;
    if (!geometry(W, H, g)) return 0;
    return 20 + 12 + 66 * (size_t)g.T;
}

size_t dbde_hip_image_bytes(int W, int H, uint64_t n64) {
    Geometry g;
    if (!geometry(W, H, g)) return 0;
    return 12 + 2 * (size_t)g.T + 8 * (size_t)n64;
}

// What every launch of the persistent / small / tiny encoders is told about the batch (`pix` bytes per pixel: 1 = DBDE,
// 2 = DBDE16 through the same kernel; ctrl / state are attached by the caller once the workspace is known).
static EncParams enc_params(dbde_hip_ctx *ctx, const Geometry &g, int W, int H, int n_frames, uint32_t pix, uint32_t chunks_per_frame,
                            uint32_t lanes_per_row, const uint8_t *d_images, uint8_t *d_out, uint64_t slot_stride,
                            uint64_t first_index, uint64_t *d_frame_offsets, uint64_t *d_frame_bytes) {
    EncParams p;
    p.images = d_images;
    p.out = d_out;
    p.frame_offsets = d_frame_offsets;
    p.frame_bytes = d_frame_bytes;
    p.indices = nullptr;
    p.elapsed_ns = nullptr;
    p.first_index = first_index;
    p.state = nullptr;
    p.ctrl = nullptr;
    p.ctrl_next = nullptr;
    p.mode_flags = nullptr;
    p.arrive_flags = nullptr;
    p.launch_epoch = 0;
    p.sticky = ctx->sticky;
    p.slot_stride = slot_stride;
    p.frame_pixels = (uint64_t)pix * g.pixels;   // bytes of one frame's image
    p.W = W; p.H = H; p.w = g.w; p.h = g.h; p.T = g.T;
    p.chunks_per_frame = chunks_per_frame;
    p.n_chunks = (uint32_t)n_frames * chunks_per_frame;
    p.lanes_per_row = lanes_per_row;
    p.pairs_per_wave = 64u;
    p.seg_per_row = p.seg_q = p.seg_rem = p.magic_seg = 0u;
    p.magic_w = div_magic_of(g.w);
    p.magic_cpf = div_magic_of(chunks_per_frame);
    p.magic_lpr = div_magic_of(lanes_per_row);
    p.last_frame = (uint32_t)n_frames - 1u;
    p.flags = ctx->exp_flags;
    p.grid_blocks = ctx->enc_grid;
    p.diag = reinterpret_cast<unsigned long long *>(ctx->diag);
    p.small_epoch = 0;
    return p;
}

// The encoders' shared workspace: [control words, set 0: 8 x u32][set 1][records: n_chunks x u64].  NO memset in front
// of a launch (round 3 cleared it before every persistent launch: a fill kernel and its boundary, 8-10 us of each):
//   * a persistent launch finds zeros where it needs them and leaves zeros behind -- every AGG / INC record is cleared
//     by the workgroup that reads it (wait_inc), the control words it uses were cleared by the launch before it (the
//     scanner clears the OTHER set; launches alternate), and records of small launches (below) have bits 63:62 clear,
//     which no AGG / INC record has;
//   * a small launch tags its records with its epoch and clears nothing.
// The block is zeroed when it is (re)allocated, when the small launches' epoch wraps, and after a launch that failed
// (dbde_hip_sync saw the sticky word: until then the context's launches return at once, encode_kernel).
static int attach_lookback(dbde_hip_ctx *ctx, EncParams &p, uint32_t n_chunks, bool small) {
    // two sets of control words | the mode flags | the arrival flags | records
    constexpr size_t kFlagsAt = 2 * 4 * (size_t)kEncCtrlWords, kHeader = (kFlagsAt + 2 * 4 * (size_t)kEncMaxGrid + 4095) & ~(size_t)4095;
    const size_t lb_need = (kHeader + 8 * (size_t)n_chunks + 15) & ~(size_t)15;
    {   // grown in place: on failure ctx->lb is null and ctx->lb_bytes 0, never a freed pointer
        const size_t had = ctx->lb_bytes;
        int rc = grow(ctx, ctx->lb, ctx->lb_bytes, lb_need, 1, !(ctx->exp_flags & 128u));   // (experiment bit 7: plain cached memory)
        if (rc) return rc;
        if (ctx->lb_bytes != had) ctx->lb_fresh = true;
    }
    if (ctx->lb_fresh || ctx->enc_epoch >= (1u << 30) - 1u) {   // (small launches' record tags and persistent launches' mode flags share the epoch counter)
        HIP_TRY(ctx, hipMemsetAsync(ctx->lb, 0, ctx->lb_bytes, ctx->stream));
        ctx->lb_fresh = false;
        ctx->enc_epoch = 0;
        ctx->enc_parity = 0;
    }
    const uint32_t epoch = ++ctx->enc_epoch;
    p.small_epoch = small ? epoch : 0u;
    p.launch_epoch = epoch;
    p.ctrl = reinterpret_cast<uint32_t *>(ctx->lb) + kEncCtrlWords * ctx->enc_parity;
    p.ctrl_next = reinterpret_cast<uint32_t *>(ctx->lb) + kEncCtrlWords * (ctx->enc_parity ^ 1u);
    p.mode_flags = reinterpret_cast<uint32_t *>(ctx->lb + kFlagsAt);
    p.arrive_flags = p.mode_flags + kEncMaxGrid;
    p.state = reinterpret_cast<unsigned long long *>(ctx->lb + kHeader);
    if (!small) ctx->enc_parity ^= 1u;
    return DBDE_HIP_OK;
}

// ---- batch encode ----------------------------------------------------------------------------
#ifndef DBDE_MID_ENCODE_TILES
// Frames of 65 .. this many tiles, one slot each, encode with whole frames per workgroup (encode_mid_kernel, 0.46 of peak
// on mixed content whatever T, 0.37 on incompressible frames); the chunk-per-frame kernels overtake it as T grows --
// measured: T = 81 / 144 / 196 / 256 / 324 / 396 general path 0.12 / 0.19 / 0.25 / 0.32 / 0.36 / 0.40 mixed and
// 0.15 / 0.27 / 0.35 / 0.44 / 0.50 / 0.53 incompressible.
#define DBDE_MID_ENCODE_TILES 256
#endif

// Which kernel an encode call runs and on what chunk geometry: a pure function of the shape, the batch size, the
// buffers' alignment, the layout and the number of workgroups the device holds (dbde_hip_encode_plan).
struct EncPlan {
    bool fast_in, aligned_out;
    uint32_t enc_cpf, lanes_per_row, pairs_per_wave;
    uint32_t seg_per_row, seg_q, seg_rem;   // kInRow: segments of a tile row (0 = not taken)
    uint64_t n_chunks64;
    int kernel;            // 0 = persistent (encode_kernel), 1 = encode_small_kernel, 2 = encode_tiny_kernel, 3 = encode_mid_kernel, 4 = encode_frames_kernel
};
#ifndef DBDE_ROW_FILL
#define DBDE_ROW_FILL 90
#endif
#ifndef DBDE_FRAMES_ENCODE_TILES
// Frames of 65 .. this many tiles with 8-byte aligned rows encode / decode with whole frames per workgroup and staged,
// coalesced traffic (encode_frames_kernel / decode_frames_kernel); above it the chunk kernels' 512 / 1024 tile slots are
// filled well enough by one frame.
#define DBDE_FRAMES_ENCODE_TILES 640   // measured (mixed, encode): 81 tiles 0.39 -> 0.51, 144 0.44 -> 0.56, 256 0.49 -> 0.65, 300 0.32 -> 0.51, 396 0.39 -> 0.57, 625 0.47 -> 0.48
#endif
#ifndef DBDE_FRAMES_DECODE_TILES
// The decode side is built and tested ($DBDE_HIP_EXPERIMENT bit 8 switches it on) but NOT taken by default: against the
// forms it would replace it measured equal at 81 tiles (0.39; incompressible 0.42 -> 0.49) and slower from 144 tiles on
// (0.43 -> 0.37; 300 tiles: 0.55 with its index kernel counted -> 0.44) -- six dependent phases per workgroup, each a
// memory or LDS round trip, where decode_mid_kernel and the chunk decoder have three.
#define DBDE_FRAMES_DECODE_TILES 640
#endif
// rows 8-byte aligned, frames and base whole 16-byte blocks: what the staged whole-frame kernels take
#ifndef DBDE_GROUP_ROWS4_ALL
#define DBDE_GROUP_ROWS4_ALL 1   // rows of 4 mod 8 bytes: every frame up to 256 tiles (the alternative there is encode_mid_kernel: 84x84 0.33 -> 0.44 / 0.30 -> 0.50, 124x124 0.37 -> 0.53 / 0.32 -> 0.59, 100x75 0.325 -> 0.316 / 0.29 -> 0.37)
#endif
#ifndef DBDE_GROUP_ENCODE_TILES
#define DBDE_GROUP_ENCODE_TILES 85    // largest frame (tiles) of the persistent small-frame encoder where frames above 64 tiles fill 90 % of its lanes (77 .. 85 tiles: three frames)
#endif
static bool frames_geometry(const Geometry &g, int W, uintptr_t images) {
    return W % 8 == 0 && g.pixels % 16 == 0 && (images & 15u) == 0 && g.T > 64u;
}
static EncPlan plan_encode(const Geometry &g, int W, int n_frames, uintptr_t images, uintptr_t out, uint64_t slot_stride,
                           uint32_t enc_grid) {
    EncPlan pl;
    pl.fast_in = (W % 16 == 0) && ((images & 15u) == 0);
#ifdef DBDE_FORCE_GENERIC   // A/B builds only: the any-geometry kernels on aligned images
    pl.fast_in = false;
#endif
    // chunk geometry of the encoder (EncParams): plain runs of 1024 tiles, or -- any-geometry path, W >= 16 --
    // 512 tile PAIRS that never leave a tile row
    pl.enc_cpf = (g.T + kEncChunkTiles - 1) / kEncChunkTiles;
    pl.lanes_per_row = 0;
    pl.pairs_per_wave = 64;
    pl.seg_per_row = pl.seg_q = pl.seg_rem = 0;
    if (!pl.fast_in && W >= 16) {
        pl.lanes_per_row = (g.w + 1u) / 2u;
        // Dword-aligned fetches (kInRaw4: a 16-byte load at an odd address runs at 0.87 of the rate of one at any even
        // address): a wave owns 63 pairs, its 64th lane feeds the 63rd.  Taken when the last pair of a tile row holds at
        // most 13 pixel columns -- the up to 3 bytes its moved fetch is short of are then padding (dbde_kernels.hip).
        // Rows at even addresses (W and the base even) read at the full rate as they are and keep 64 pairs.
#ifndef DBDE_NO_RAW4
        const uint32_t last_cols = (uint32_t)W - 16u * (pl.lanes_per_row - 1u);
        // (32-bit offsets inside a frame: geometry() admits frames of up to 8 GiB)
        if (last_cols <= 13u && ((W | (int)(images & 1u)) & 1) && g.pixels < (1ull << 31)) pl.pairs_per_wave = 63;
#ifdef DBDE_ROW_ALWAYS   // A/B builds: the segment form on rows at even addresses too
        if (last_cols <= 13u && g.pixels < (1ull << 31)) pl.pairs_per_wave = 63;
#endif
#endif
        const uint32_t ppc = pl.pairs_per_wave * (kEncChunkTiles / 128u);
        pl.enc_cpf = (uint32_t)(((uint64_t)g.h * pl.lanes_per_row + ppc - 1u) / ppc);
#ifndef DBDE_NO_ROW
        // kInRow: one wave per segment of a tile row (addresses and shifts in scalar registers) when that keeps at least
        // DBDE_ROW_FILL percent of the lanes busy -- 1921 wide: 121 pairs = 61 + 60 of 128 lanes
        if (pl.pairs_per_wave == 63u) {
            const uint32_t nseg = (pl.lanes_per_row + 62u) / 63u;
            if (100u * pl.lanes_per_row >= (unsigned)DBDE_ROW_FILL * 64u * nseg && pl.lanes_per_row / nseg >= 2u) {
                pl.seg_per_row = nseg;
                pl.seg_q = pl.lanes_per_row / nseg;
                pl.seg_rem = pl.lanes_per_row % nseg;
                const uint32_t spc = kEncChunkTiles / 128u;   // segments (waves) per chunk
                pl.enc_cpf = (uint32_t)(((uint64_t)g.h * nseg + spc - 1u) / spc);
            }
        }
#endif
    }
    pl.n_chunks64 = (uint64_t)n_frames * pl.enc_cpf;
    pl.aligned_out = ((out & 7u) == 0) && (g.T % 4 == 0) && (slot_stride % 8 == 0);
    pl.kernel = 0;
    const bool rows4 = W % 4 == 0 && (images & 3u) == 0 && (out & 15u) == 0 && slot_stride % 16 == 0;
    if (false) {}
#ifndef DBDE_NO_GROUP
    // Small frames in slots, 8-byte aligned rows, frames and buffers whole 16-byte blocks: persistent workgroups, the next
    // group's pixels in flight while a group is encoded (encode_group_kernel: one tile per lane, 256 / T frames per
    // workgroup).  Measured against what it replaces (mixed / incompressible): 64x64 0.40 -> 0.53 / 0.33 -> 0.56, 32x32 0.35
    // -> 0.49 / 0.32 -> 0.47, 40x24 0.31 -> 0.46 / 0.30 -> 0.41; 72x72 (81 tiles, three frames fill 95 % of the lanes) 0.49 ->
    // 0.50 / 0.49 -> 0.56.  Where whole frames leave lanes empty the two-tiles-per-lane kernel below keeps mixed content
    // (96x96, one frame of 144 tiles per workgroup: 0.51 -> 0.42; 128x128 at full fill 0.62 -> 0.56, incompressible 0.54 ->
    // 0.58), and single-tile frames keep the per-wave kernel (8x8: 0.17 -> 0.09, 256 frame images per workgroup).
    // (rows and image bases of 4-byte multiples suffice since the round's last hours: DESIGN 4.1)
    else if (slot_stride != 0 && rows4 && g.T >= 4u &&
             (g.T <= 64u || (g.T <= (unsigned)DBDE_GROUP_ENCODE_TILES && (256u / g.T) * g.T * 10u >= 256u * 9u) ||
              (DBDE_GROUP_ROWS4_ALL && W % 8 != 0 && g.T <= 256u))) pl.kernel = 5;
#endif
    else if (g.T <= 64u && slot_stride != 0) pl.kernel = 2;     // tiny frames in slots: several frames per wave, nothing shared
#ifndef DBDE_NO_FRAMES
    else if (slot_stride != 0 && g.T <= (unsigned)DBDE_FRAMES_ENCODE_TILES && frames_geometry(g, W, images) && (out & 15u) == 0 &&
             slot_stride % 16 == 0) pl.kernel = 4;              // 65 .. 700 tiles, aligned rows: whole frames per workgroup, staged
#endif
#ifndef DBDE_NO_MID
    else if (g.T <= (unsigned)DBDE_MID_ENCODE_TILES && slot_stride != 0) pl.kernel = 3;   // 65 .. 256 tiles in slots: whole frames per workgroup
#endif
    else if (pl.n_chunks64 < enc_grid) pl.kernel = 1;           // fewer chunks than resident workgroups: one workgroup per chunk
    return pl;
}

int dbde_hip_encode_frames(dbde_hip_ctx *ctx, const uint8_t *d_images, int W, int H, int n_frames,
                           uint64_t first_index, const uint64_t *d_indices, const uint64_t *d_elapsed_ns,
                           uint8_t *d_out, size_t out_capacity, uint64_t slot_stride,
                           uint64_t *d_frame_offsets, uint64_t *d_frame_bytes) {
    if (!ctx) return DBDE_HIP_ERR_ARG;
    Geometry g;
    if (!d_images || !d_out || n_frames < 0 || !geometry(W, H, g))
        return fail(ctx, DBDE_HIP_ERR_ARG, "encode_frames: bad argument (W=%d H=%d n=%d)", W, H, n_frames);
    if (n_frames == 0) return DBDE_HIP_OK;
    const uint64_t maxf = 32ull + 66ull * g.T;
    const EncPlan pl = plan_encode(g, W, n_frames, reinterpret_cast<uintptr_t>(d_images), reinterpret_cast<uintptr_t>(d_out), slot_stride,
                                   ctx->enc_grid);
    const bool fast_in = pl.fast_in, aligned_out = pl.aligned_out;
    const uint32_t enc_cpf = pl.enc_cpf, lanes_per_row = pl.lanes_per_row;
    const uint64_t n_chunks64 = pl.n_chunks64;
    if (n_chunks64 >= (1ull << 31)) return fail(ctx, DBDE_HIP_ERR_ARG, "encode_frames: too many chunks in one call");
    if (slot_stride) {
        if (slot_stride < maxf) return fail(ctx, DBDE_HIP_ERR_ARG, "encode_frames: slot_stride below the worst case");
        if ((uint64_t)(n_frames - 1) * slot_stride + maxf > out_capacity)
            return fail(ctx, DBDE_HIP_ERR_CAPACITY, "encode_frames: out_capacity below the worst case");
    } else {
        if ((uint64_t)n_frames * maxf > out_capacity)
            return fail(ctx, DBDE_HIP_ERR_CAPACITY, "encode_frames: out_capacity below the worst case");
        // launch-wide running word count is carried in 32 bits
        if ((uint64_t)n_frames * 8ull * g.T >= (1ull << 32))
            return fail(ctx, DBDE_HIP_ERR_ARG, "encode_frames: more than 2^32 payload words possible; split the batch");
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    const uint32_t n_chunks = (uint32_t)n_chunks64;
    EncParams p = enc_params(ctx, g, W, H, n_frames, 1u, enc_cpf, lanes_per_row, d_images, d_out, slot_stride, first_index,
                             d_frame_offsets, d_frame_bytes);
    p.indices = d_indices;
    p.elapsed_ns = d_elapsed_ns;
    p.pairs_per_wave = pl.pairs_per_wave;
    p.seg_per_row = pl.seg_per_row;
    p.seg_q = pl.seg_q;
    p.seg_rem = pl.seg_rem;
    p.magic_seg = pl.seg_per_row ? div_magic_of(pl.seg_per_row) : 0u;

    if (pl.kernel == 5) {   // 1 .. 256 tiles in slots, aligned rows: persistent workgroups, pixels double-buffered (encode_group_kernel)
        span_begin(ctx, 0);
        HIP_TRY(ctx, launch_encode_group(p, (uint32_t)n_frames, (ctx->exp_flags & 1024u) ? 0u : (uint32_t)ctx->n_cu, ctx->stream));   // (experiment bit 10: three workgroups)
        span_end(ctx);
        return DBDE_HIP_OK;
    }

    if (pl.kernel == 2) {   // tiny frames in slots: several frames per wave, nothing shared (encode_tiny_kernel)
        span_begin(ctx, 0);
        HIP_TRY(ctx, launch_encode_tiny(p, (uint32_t)n_frames, ctx->stream));
        span_end(ctx);
        return DBDE_HIP_OK;
    }

    if (pl.kernel == 4) {   // frames of 65 .. 700 tiles in slots, 8-byte aligned rows: whole frames per workgroup, staged (encode_frames_kernel)
        span_begin(ctx, 0);
        HIP_TRY(ctx, launch_encode_frames(p, (uint32_t)n_frames, ctx->stream));
        span_end(ctx);
        return DBDE_HIP_OK;
    }

    if (pl.kernel == 3) {   // frames of 65 .. 256 tiles in slots: whole frames per workgroup (encode_mid_kernel)
        span_begin(ctx, 0);
        HIP_TRY(ctx, launch_encode_mid(p, (uint32_t)n_frames, ctx->stream));
        span_end(ctx);
        return DBDE_HIP_OK;
    }

    // small launches (one frame per call above all): one workgroup per chunk, epoch-tagged records
    const bool small = pl.kernel == 1;
    span_begin(ctx, 0);
    {
        int rc = attach_lookback(ctx, p, n_chunks, small);
        if (rc) return rc;
    }
    if (small) HIP_TRY(ctx, launch_encode_small(p, fast_in, aligned_out, ctx->stream));
    else HIP_TRY(ctx, launch_encode(p, fast_in, aligned_out, ctx->stream));
    span_end(ctx);
    return DBDE_HIP_OK;
}

// ---- batch decode ----------------------------------------------------------------------------
#ifndef DBDE_MID_DECODE_TILES
// Frames of up to this many tiles decode with whole frames per workgroup (decode_mid_kernel: one tile per lane, from one-tile
// frames up; rounds 2-3 had a kernel of its own for T <= 64, whole frames per wave, which the persistent form beat at every
// size -- 64x64 0.49 -> 0.63, 32x32 0.33 -> 0.56, 8x8 0.19 -> 0.30): no index kernel, no chunk.  Measured wall time per step, mid against chunks + index kernel, round 3:
// T = 81 twice as fast, 144 +24 %, 196 equal (mixed) / -14 % (incompressible), 256 -7 %; with the round-4 form of the kernel
// (persistent 256-thread workgroups, software-pipelined, wide payload loads, pixels staged): 169 tiles 0.91 -> 0.60 ms,
// 225 0.48 -> 0.40 (incompressible 0.53 -> 0.475), 240 0.40 -> 0.315, 256 0.416 -> 0.344 (incompressible 0.414 -> 0.429).
#define DBDE_MID_DECODE_TILES 256
#endif
#ifndef DBDE_MID_DECODE_TILES_UNSTAGED
#define DBDE_MID_DECODE_TILES_UNSTAGED 768
#endif
#ifndef DBDE_STAGED_FILL
// Percent of a workgroup's 512 tile slots that whole tile rows must fill for the staged decode path.  A workgroup's time
// hardly depends on how many of its slots are used, so empty slots are lost throughput -- but tile-by-tile stores
// (partial cache lines) cost more.  Measured on mixed content (round 3): 88 % fill (720, 1200, 600 wide) staged +4..9 %,
// 84 % (3440) +3.5 %, 82 % (1680) equal, 79 % (1080 wide: portrait HD) equal and +26 % on incompressible frames,
// 78 % (1600) -6 %, 70 % (1440) -4 %.
#define DBDE_STAGED_FILL 79
#endif

// Which kernels a decode call runs: a pure function of the geometry, the batch size, the image base's alignment and the
// device's CU count, so that the choice can be inspected and tested without a GPU (dbde_hip_decode_plan).
struct DecPlan {
    int img_mode;          // kImgDirect / kImgLinear / kImgTiles (dbde_kernels.hip)
    uint32_t cap;          // tile slots of the workgroup that takes whole-tile-row chunks
    DecGeom dg;
    uint64_t n_chunks64;
    bool self_index, fused;
    int kernel;            // 0 = chunk kernels, 3 = decode_mid_kernel, 4 = decode_frames_kernel
};
static DecPlan plan_decode(const Geometry &g, int W, int n_frames, uintptr_t ib, int n_cu, uint32_t exp_flags) {
    DecPlan pl;
    // How the pixels reach the image (decode_kernel<IMG>): direct register -> image stores are only FAST when a
    // wave's 1 KB covers whole cache lines (W, the frame size and the base multiples of 128).  Other widths get
    // chunks of whole tile rows where those fill enough of a 512-tile workgroup (DBDE_STAGED_FILL percent): the
    // workgroup stages its pixels in LDS and writes whole cache lines of the chunk's byte range.  Everything else
    // stores tile by tile from plain chunks.
    int img_mode = 2;
    uint32_t cap = kChunkTiles;   // tile slots of the workgroup that takes whole-tile-row chunks
    if (W % 128 == 0 && g.pixels % 128 == 0 && (ib & 127u) == 0) img_mode = 0;
    else if (g.w <= kChunkTiles && W >= 16) {
        // whole tile rows in 512 slots (256 threads) or in 384 (192 threads), whichever they fill better
        // (1366 wide: 2 x 171 tiles = 67 % of 512, 89 % of 384)
        const uint32_t used512 = (kChunkTiles / g.w) * g.w;
        const uint32_t used384 = g.w <= kChunkTilesSmall ? (kChunkTilesSmall / g.w) * g.w : 0u;
        // (16-byte aligned rows are left out: there plain chunks with direct 16-byte stores, below, beat the better-filled
        // small workgroup on mixed content -- 1440 / 2704 wide: 0.70 / 0.70 against 0.68 / 0.67)
#ifndef DBDE_NO_SMALL_WG
        if (W % 16 != 0 && (uint64_t)used384 * kChunkTiles > (uint64_t)used512 * kChunkTilesSmall) cap = kChunkTilesSmall;
#endif
        const uint32_t used = cap == kChunkTiles ? used512 : used384, rows = cap / g.w;
        if (used * 100u >= cap * (unsigned)DBDE_STAGED_FILL) {
            // image rows that are not 8-byte aligned are staged tile-aligned at pitch 8 w + 16: that image must fit the
            // workgroup's LDS (narrow frames have many rows per chunk and do not); chunks of at most 384 tiles run on
            // the smaller workgroup whatever `cap` says (launch_decode)
            const bool a8 = W % 8 == 0 && (ib & 7u) == 0;
            const uint64_t lds = used <= kChunkTilesSmall ? 25600ull : 34048ull;
            if (a8 || 8ull * rows * (8ull * g.w + 16ull) <= lds) img_mode = 1;
        }
    }
    // 16-byte aligned rows that neither cover whole cache lines per wave nor fill staged chunks: still ONE 16-byte store
    // per lane and image row (the direct form) instead of two 8-byte ones (1440 / 1600 wide: 0.62 -> 0.69 / 0.73)
    if (img_mode == 2 && W % 16 == 0 && (ib & 15u) == 0) img_mode = 0;
#ifdef DBDE_FORCE_GENERIC
    img_mode = 2;
#endif
#ifdef DBDE_FORCE_LINEAR
    img_mode = 1;
#endif
    pl.img_mode = img_mode;
    pl.cap = cap;
    pl.dg = dec_geometry(g.w, g.h, img_mode == 1, kChunkTiles, cap);
    pl.n_chunks64 = (uint64_t)n_frames * pl.dg.cpf;
    // Small frames (the tile-level entry points, thumbnails): the decode workgroups index the frame themselves --
    // each reads the T depth bytes -- and the index kernel with its launch boundary is gone.  Only while T is a
    // couple of loads per thread: for a 4096x3072 frame (T = 196,608, 384 workgroups re-reading it) the same
    // idea took 38 us against 11 us for index + decode, measured.
    // (Large batches of one-chunk frames gain nothing from it although each frame's depth bytes would be read only once:
    // 262,144 frames of 128x128 took 1.75 ms self-indexed against 1.50 ms with the index kernel, measured.)
    pl.self_index = g.T <= 8192u && pl.n_chunks64 * (uint64_t)g.T <= (8ull << 20);
    // Few LARGE frames (one 4096x3072 frame per call: 384 chunks): the decode workgroups build the index among
    // themselves (decode_kernel<IMG, kIdxFused>) -- no index kernel, no launch boundary.  Taken when the launch fits the
    // device's workgroup slots (four per CU); correctness does not depend on that, only the latency does.
    pl.fused = !pl.self_index && pl.n_chunks64 <= 4ull * (uint64_t)n_cu && !(exp_flags & 8u);
    // tiny frames (the tile-level entry points, thumbnails) and those just above: whole frames per wave / per workgroup
    pl.kernel = g.T <= (unsigned)DBDE_MID_DECODE_TILES ? 3 : 0;
    // ... and larger frames whose chunks would store tile by tile (rows that are not 8-byte aligned and whole-tile-row chunks
    // that do not fit the staged image): one frame per 512- or 1024-thread workgroup of the same kernel beats two chunks per
    // frame + the index kernel -- 180x180 (529 tiles) 0.25 -> 0.33 mixed / 0.28 -> 0.37 incompressible, 130x121 0.37 -> 0.39 / 0.47,
    // 220x215 (756) 0.36 -> 0.39 / 0.39 -> 0.46; not at 1024 tiles (250x250: 0.51 -> 0.44), not where the chunks stage
    // (300x200: 0.59 -> 0.42) -- profiles/r04b_gain.sh
    if (img_mode == 2 && g.T <= (unsigned)DBDE_MID_DECODE_TILES_UNSTAGED) pl.kernel = 3;
#ifndef DBDE_NO_FRAMES
    if ((exp_flags & 256u) && g.T > 64u && g.T <= (unsigned)DBDE_FRAMES_DECODE_TILES && frames_geometry(g, W, ib)) pl.kernel = 4;
#endif
    return pl;
}

int dbde_hip_decode_frames(dbde_hip_ctx *ctx, const uint8_t *d_stream, size_t stream_bytes,
                           const uint64_t *d_frame_offsets, int W, int H, int n_frames, uint8_t *d_images,
                           dbde_hip_frame_result *d_results) {
    if (!ctx) return DBDE_HIP_ERR_ARG;
    Geometry g;
    if (!d_stream || !d_frame_offsets || !d_images || n_frames < 0 || !geometry(W, H, g))
        return fail(ctx, DBDE_HIP_ERR_ARG, "decode_frames: bad argument (W=%d H=%d n=%d)", W, H, n_frames);
    if (n_frames == 0) return DBDE_HIP_OK;
    const DecPlan pl = plan_decode(g, W, n_frames, reinterpret_cast<uintptr_t>(d_images), ctx->n_cu, ctx->exp_flags);
    const int img_mode = pl.img_mode;
    const DecGeom dg = pl.dg;
    const uint32_t dcpf = dg.cpf;   // whole tile rows (or pieces of a wide one) per chunk, one decode workgroup each
    if (dcpf > kMaxChunksPerFrame) return fail(ctx, DBDE_HIP_ERR_ARG, "decode_frames: frame too large");
    const uint64_t n_chunks64 = pl.n_chunks64;
    if (n_chunks64 >= (1ull << 31)) return fail(ctx, DBDE_HIP_ERR_ARG, "decode_frames: too many chunks in one call");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const bool self_index = pl.self_index;
    if (pl.kernel != 0) {   // tiny frames (the tile-level entry points, thumbnails) and those just above: whole frames per wave / per workgroup, nothing else needed
        DecParams tp;
        memset(&tp, 0, sizeof tp);
        tp.stream = d_stream; tp.frame_offsets = d_frame_offsets; tp.stream_bytes = stream_bytes;
        tp.images = d_images; tp.results = d_results; tp.frame_pixels = g.pixels;
        tp.W = W; tp.H = H; tp.w = g.w; tp.h = g.h; tp.T = g.T;
        span_begin(ctx, 2);
        if (pl.kernel == 4) HIP_TRY(ctx, launch_decode_frames(tp, (uint32_t)n_frames, ctx->stream));
        else HIP_TRY(ctx, launch_decode_mid(tp, (uint32_t)n_frames, (ctx->exp_flags & 1024u) ? 0u : (uint32_t)ctx->n_cu, ctx->stream));   // (experiment bit 10: three workgroups)
        span_end(ctx);
        return DBDE_HIP_OK;
    }
    const bool fused = pl.fused;   // few large frames: the index is built inside the decode launch (plan_decode)
    if (fused) {
        const size_t had = ctx->fuse_rec_n;
        int rc = grow(ctx, ctx->fuse_rec, ctx->fuse_rec_n, (size_t)n_chunks64, sizeof(unsigned long long), true);
        if (rc) return rc;
        if (ctx->fuse_rec_n != had || ctx->fuse_epoch == 0xFFFFFFFFu) {   // a fresh block (or the epoch wrapping): epoch 0 everywhere
            HIP_TRY(ctx, hipMemsetAsync(ctx->fuse_rec, 0, ctx->fuse_rec_n * sizeof(unsigned long long), ctx->stream));
            ctx->fuse_epoch = 0;
        }
        ctx->fuse_epoch++;
    }
    if (!self_index && !fused) {
        int rc = grow(ctx, ctx->chunk_off, ctx->chunk_off_n, (size_t)n_chunks64 + (size_t)n_frames, sizeof(uint32_t));
        if (rc) return rc;
        rc = grow(ctx, ctx->frame_ok, ctx->frame_ok_n, (size_t)n_frames, sizeof(uint32_t));
        if (rc) return rc;

        IdxParams ip;
        ip.stream = d_stream;
        ip.frame_offsets = d_frame_offsets;
        ip.stream_bytes = stream_bytes;
        ip.chunk_off = ctx->chunk_off;
        ip.frame_ok = ctx->frame_ok;
        ip.results = d_results;
        ip.T = g.T;
        ip.chunks_per_frame = dcpf;
        ip.min_bytes = 1;
        ip.geom = dg;
        // Few frames: cut each frame into pieces so that the index pass fills the device too
        // (>= 4 chunks per piece, about 1024 workgroups in all); from 256 frames on, one workgroup per frame.
        ip.split = 1;
        ip.frame_ctr = nullptr;
        ip.frame_flag = nullptr;
        if (n_frames < 256 && dcpf >= 8u) {
            uint32_t sp = 1024u / (uint32_t)n_frames;
            const uint32_t most = (dcpf + 3u) / 4u;
            if (sp > most) sp = most;
            if (sp > 1u) {
                const size_t before = ctx->idx_ctr_n;
                rc = grow(ctx, ctx->idx_ctr, ctx->idx_ctr_n, 2 * (size_t)n_frames, sizeof(uint32_t));
                if (rc) return rc;
                if (ctx->idx_ctr_n != before)   // fresh block: the kernel keeps it zero from here on
                    HIP_TRY(ctx, hipMemsetAsync(ctx->idx_ctr, 0, ctx->idx_ctr_n * sizeof(uint32_t), ctx->stream));
                ip.split = sp;
                ip.frame_ctr = ctx->idx_ctr;
                ip.frame_flag = ctx->idx_ctr + n_frames;
            }
        }
        span_begin(ctx, 1);
        HIP_TRY(ctx, launch_decode_index(ip, n_frames, ctx->stream));
        span_end(ctx);
    }

    DecParams p;
    p.stream = d_stream;
    p.frame_offsets = d_frame_offsets;
    p.stream_bytes = stream_bytes;
    p.images = d_images;
    p.chunk_off = ctx->chunk_off;
    p.frame_ok = ctx->frame_ok;
    p.results = d_results;
    p.frame_pixels = g.pixels;
    p.W = W;
    p.H = H;
    p.w = g.w;
    p.h = g.h;
    p.T = g.T;
    p.chunks_per_frame = dcpf;
    p.n_chunks = (uint32_t)n_chunks64;
    p.magic_W = div_magic_of((uint32_t)W);
    p.fuse_rec = ctx->fuse_rec;
    p.fuse_epoch = ctx->fuse_epoch;
    p.fuse_flags = (ctx->exp_flags & 16u) ? 1u : 0u;
    p.diag = reinterpret_cast<unsigned long long *>(ctx->diag);
    p.geom = dg;
    span_begin(ctx, 2);
    HIP_TRY(ctx, launch_decode(p, img_mode, self_index ? 1 : (fused ? 2 : 0), ctx->stream));
    span_end(ctx);
    return DBDE_HIP_OK;
}

int dbde_hip_index_stream_async(dbde_hip_ctx *ctx, const uint8_t *d_stream, size_t stream_bytes, int W, int H,
                                int max_frames, uint64_t *d_frame_offsets, uint32_t *d_n_found) {
    if (!ctx) return DBDE_HIP_ERR_ARG;
    Geometry g;
    if (!d_stream || !d_frame_offsets || !d_n_found || max_frames < 0 || !geometry(W, H, g))
        return fail(ctx, DBDE_HIP_ERR_ARG, "index_stream: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // Long streams are walked speculatively in up to 16 segments at once (scan_spec_kernel: exact by construction,
    // the plain hop-by-hop walk is what it falls back to); short ones hop by hop.
    const uint64_t maxlen = 32ull + 66ull * g.T, meta = 32ull + 2ull * g.T;
    uint64_t n_seg = stream_bytes / (2 * maxlen);
    if (n_seg > 16) n_seg = 16;   // the signature searches (one maximal frame each, at worst) are the cost: few, wide segments (measured: 8-16)
    span_begin(ctx, 3);
    if (n_seg >= 2 && max_frames > 0) {
        ScanParams sp;
        sp.stream = d_stream;
        sp.stream_bytes = stream_bytes;
        sp.T = g.T;
        sp.gran = g.T % 4 == 0 ? 8u : (g.T % 2 == 0 ? 4u : 2u);   // frame lengths 32 + 2T + 8 n64 are multiples of this
        sp.seg_bytes = (stream_bytes + n_seg - 1) / n_seg;
        sp.seg_cap = (uint32_t)(sp.seg_bytes / meta + 3);
        if (sp.seg_cap > (uint32_t)max_frames + 3u) sp.seg_cap = (uint32_t)max_frames + 3u;   // no list needs more than the caller takes (tiny frames: T = 1, meta = 34)
        // workspace: [found 8 x 64][arrive 4 x 64] (kept zero by the kernel) | lists | start, end, count, ended
        const size_t fixed = 64 * 8 + 64 * 4, lists = (size_t)n_seg * sp.seg_cap * 8, need = fixed + lists + n_seg * 32 + 64;
        const size_t had = ctx->scan_ws_bytes;
        int rc = grow(ctx, ctx->scan_ws, ctx->scan_ws_bytes, need, 1);
        if (rc) return rc;
        if (ctx->scan_ws_bytes != had) HIP_TRY(ctx, hipMemsetAsync(ctx->scan_ws, 0, fixed, ctx->stream));
        sp.seg_found_inv = reinterpret_cast<unsigned long long *>(ctx->scan_ws);
        sp.seg_arrive = reinterpret_cast<uint32_t *>(ctx->scan_ws + 64 * 8);
        sp.seg_pos = reinterpret_cast<uint64_t *>(ctx->scan_ws + fixed);
        sp.seg_start = reinterpret_cast<uint64_t *>(ctx->scan_ws + fixed + lists);
        sp.seg_end = sp.seg_start + n_seg;
        sp.seg_count = reinterpret_cast<uint32_t *>(sp.seg_end + n_seg);
        sp.seg_ended = sp.seg_count + n_seg;
        sp.wg_per_seg = 4;                          // workgroups sharing a segment's signature search (measured: 1-4)
        HIP_TRY(ctx, launch_scan_spec(sp, (uint32_t)n_seg, max_frames, d_frame_offsets, d_n_found, ctx->stream));
    } else {
        HIP_TRY(ctx, launch_scan_stream(d_stream, stream_bytes, g.T, max_frames, d_frame_offsets, d_n_found, nullptr, ctx->stream));
    }
    span_end(ctx);
    return DBDE_HIP_OK;
}

int dbde_hip_scan_ahead(dbde_hip_ctx *ctx, const uint8_t *d_stream, size_t stream_bytes, int W, int H, int max_frames,
                        uint64_t *d_cursor, uint64_t *d_frame_offsets, uint32_t *d_n_found) {
    if (!ctx) return DBDE_HIP_ERR_ARG;
    Geometry g;
    if (!d_stream || !d_frame_offsets || !d_n_found || !d_cursor || max_frames < 0 || !geometry(W, H, g))
        return fail(ctx, DBDE_HIP_ERR_ARG, "scan_ahead: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->scan_stream) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->scan_stream, hipStreamNonBlocking, hi));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->scan_ev_main, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->scan_ev_done, hipEventDisableTiming));
    }
    // the walk may read what the main stream has produced so far (and the cursor a previous walk left)
    HIP_TRY(ctx, hipEventRecord(ctx->scan_ev_main, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->scan_stream, ctx->scan_ev_main, 0));
    // the walk is launched whether or not its timing bracket could be created (as span_begin / span_end)
    TimedSpan sp;
    sp.kind = 3;
    bool timed = false;
    if (ctx->timing && hipEventCreate(&sp.a) == hipSuccess) {
        if (hipEventCreate(&sp.b) == hipSuccess) timed = true;
        else (void)hipEventDestroy(sp.a);
    }
    if (timed) (void)hipEventRecord(sp.a, ctx->scan_stream);
    const hipError_t e_walk = launch_scan_stream(d_stream, stream_bytes, g.T, max_frames, d_frame_offsets, d_n_found, d_cursor, ctx->scan_stream);
    if (timed) {
        (void)hipEventRecord(sp.b, ctx->scan_stream);
        ctx->spans.push_back(sp);
    }
    HIP_TRY(ctx, e_walk);
    HIP_TRY(ctx, hipEventRecord(ctx->scan_ev_done, ctx->scan_stream));
    return DBDE_HIP_OK;
}

int dbde_hip_scan_join(dbde_hip_ctx *ctx) {
    if (!ctx) return DBDE_HIP_ERR_ARG;
    if (ctx->scan_stream) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->scan_ev_done, 0));
    return DBDE_HIP_OK;
}

int dbde_hip_index_stream(dbde_hip_ctx *ctx, const uint8_t *d_stream, size_t stream_bytes, int W, int H,
                          int max_frames, uint64_t *d_frame_offsets, int *n_found) {
    if (!ctx || !n_found) return DBDE_HIP_ERR_ARG;
    uint32_t *d_count = reinterpret_cast<uint32_t *>(ctx->scratch64);
    int rc = dbde_hip_index_stream_async(ctx, d_stream, stream_bytes, W, H, max_frames, d_frame_offsets, d_count);
    if (rc) return rc;
    uint32_t cnt = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&cnt, d_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *n_found = (int)cnt;
    return DBDE_HIP_OK;
}

int dbde_hip_synth_frames(dbde_hip_ctx *ctx, int mode, uint64_t seed, uint64_t first_frame, int n_frames,
                          int W, int H, uint8_t *d_images) {
    if (!ctx) return DBDE_HIP_ERR_ARG;
    if (!d_images || W <= 0 || H <= 0 || n_frames < 0 || mode < 0 || mode > 12)
        return fail(ctx, DBDE_HIP_ERR_ARG, "synth_frames: bad argument");
    if (n_frames == 0) return DBDE_HIP_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_synth(mode, seed, first_frame, n_frames, W, H, d_images, ctx->stream));
    return DBDE_HIP_OK;
}

// ---- launch plans (pure functions: no context, no device) ------------------------------------------------------

int dbde_hip_encode_plan(int W, int H, int n_frames, uint64_t image_address, uint64_t out_address, uint64_t slot_stride,
                         int resident_workgroups, dbde_hip_launch_plan *plan) {
    Geometry g;
    if (!plan || n_frames < 1 || resident_workgroups < 1 || !geometry(W, H, g)) return DBDE_HIP_ERR_ARG;
    const EncPlan pl = plan_encode(g, W, n_frames, (uintptr_t)image_address, (uintptr_t)out_address, slot_stride, (uint32_t)resident_workgroups);
    memset(plan, 0, sizeof *plan);
    plan->kernel = pl.kernel;
    plan->input_mode = pl.fast_in ? 0 : (pl.lanes_per_row ? (pl.seg_per_row ? 4 : pl.pairs_per_wave == 63u ? 3 : 1) : 2);
    plan->aligned_out = pl.aligned_out ? 1 : 0;
    plan->threads = (pl.kernel == 2 || pl.kernel == 5) ? 256 : (pl.kernel == 3 ? (int32_t)mid_encode_threads_for(g.T) : (pl.kernel == 4 ? (int32_t)frames_threads_for(g.T) : (int32_t)(kEncChunkTiles / 2u)));
    plan->chunks_per_frame = pl.kernel >= 2 ? 0u : pl.enc_cpf;
    plan->chunk_tiles = pl.kernel >= 2 ? 0u : (pl.seg_per_row ? 2u * (pl.seg_q + (pl.seg_rem ? 1u : 0u)) * (kEncChunkTiles / 128u) : pl.lanes_per_row ? 2u * pl.pairs_per_wave * (kEncChunkTiles / 128u) : kEncChunkTiles);
    plan->n_chunks = pl.kernel >= 2 ? 0ull : pl.n_chunks64;
    return DBDE_HIP_OK;
}

int dbde_hip_decode_plan(int W, int H, int n_frames, uint64_t image_address, int n_cu, dbde_hip_launch_plan *plan) {
    Geometry g;
    if (!plan || n_frames < 1 || n_cu < 1 || !geometry(W, H, g)) return DBDE_HIP_ERR_ARG;
    const DecPlan pl = plan_decode(g, W, n_frames, (uintptr_t)image_address, n_cu, 0u);
    memset(plan, 0, sizeof *plan);
    plan->kernel = pl.kernel;
    if (pl.kernel == 0) {
        plan->image_mode = pl.img_mode;
        plan->index_mode = pl.self_index ? 1 : (pl.fused ? 2 : 0);
        plan->threads = pl.img_mode == 1 && pl.dg.pieces == 1u && pl.dg.ct <= kChunkTilesSmall ? (int32_t)(kChunkTilesSmall / 2u) : (int32_t)(kChunkTiles / 2u);
        plan->chunks_per_frame = pl.dg.cpf;
        plan->chunk_tiles = pl.dg.ct;
        plan->n_chunks = pl.n_chunks64;
    } else {
        plan->threads = pl.kernel == 4 ? (int32_t)frames_threads_for(g.T) : (int32_t)mid_decode_threads_for(g.T);
    }
    return DBDE_HIP_OK;
}

// ---- DBDE16: the higher-bit-depth extension (include/dbde_hip.h, oracle/dbde16_oracle.c) ----------------------

size_t dbde16_hip_max_frame_bytes(int W, int H) {
    Geometry g;
    if (!geometry(W, H, g)) return 0;
    return 20 + 12 + 131 * (size_t)g.T;
}

int dbde16_hip_encode_frames(dbde_hip_ctx *ctx, const uint16_t *d_images, int W, int H, int n_frames, uint64_t first_index,
                             uint8_t *d_out, size_t out_capacity, uint64_t slot_stride, uint64_t *d_frame_offsets,
                             uint64_t *d_frame_bytes) {
    if (!ctx) return DBDE_HIP_ERR_ARG;
    Geometry g;
    if (!d_images || !d_out || n_frames < 0 || !geometry(W, H, g))
        return fail(ctx, DBDE_HIP_ERR_ARG, "encode16: bad argument (W=%d H=%d n=%d)", W, H, n_frames);
    if (n_frames == 0) return DBDE_HIP_OK;
    const uint64_t maxf = 32ull + 131ull * g.T;
    if (slot_stride ? (slot_stride < maxf || (uint64_t)(n_frames - 1) * slot_stride + maxf > out_capacity)
                    : (uint64_t)n_frames * maxf > out_capacity)
        return fail(ctx, DBDE_HIP_ERR_CAPACITY, "encode16: out_capacity (or slot_stride) below the worst case");
    const uint32_t cpf = (g.T + dbde16::kChunkTiles16 - 1) / dbde16::kChunkTiles16;
    if ((uint64_t)n_frames * cpf >= (1ull << 31) || (uint64_t)g.T * 16ull >= (1ull << 32))
        return fail(ctx, DBDE_HIP_ERR_ARG, "encode16: launch too large");
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    // Large launches of 16-byte aligned rows go through the 8-bit path's persistent encoder (encode_kernel<.., PIX = 2>:
    // central scanner, register prefetch, wave-private payload images, one barrier per 512-tile chunk).  Its prefixes
    // are 32-bit word counts (30 bits inside a frame); everything else -- odd widths, launches the device is not
    // filled by, batches past those limits -- stays with enc16_kernel below.
    {
        const uint32_t cpf2 = (g.T + kEncChunkTiles / 2u - 1u) / (kEncChunkTiles / 2u);
        const uint64_t n_chunks64 = (uint64_t)n_frames * cpf2;
        const bool fits = 16ull * g.T < (1ull << 30) && (slot_stride != 0 || (uint64_t)n_frames * 16ull * g.T < (1ull << 32));
        const bool fast_in = W % 8 == 0 && (reinterpret_cast<uintptr_t>(d_images) & 15u) == 0;
        // (any other geometry from 8 pixels across on: the same kernel with its fetches where they lie -- U16 rows always
        // start at even addresses, which read at the full rate; a frame's bytes stay below 2^32 for its 32-bit offsets)
        const bool raw_in = !fast_in && W >= 8 && 2ull * g.pixels < (1ull << 32) && (reinterpret_cast<uintptr_t>(d_images) & 1u) == 0;
        if ((fast_in || raw_in) && fits && n_chunks64 >= ctx->enc_grid &&
            n_chunks64 < (1ull << 31) && !(ctx->exp_flags & 32u)) {
            EncParams q = enc_params(ctx, g, W, H, n_frames, 2u, cpf2, 0u, reinterpret_cast<const uint8_t *>(d_images), d_out,
                                     slot_stride, first_index, d_frame_offsets, d_frame_bytes);
            const bool aligned_out = (reinterpret_cast<uintptr_t>(d_out) & 7u) == 0 && g.T % 8 == 0 && slot_stride % 8 == 0;
            span_begin(ctx, 0);
            int rc = attach_lookback(ctx, q, q.n_chunks, false);
            if (rc) return rc;
            HIP_TRY(ctx, launch_encode16_fast(q, fast_in, aligned_out, ctx->stream));
            span_end(ctx);
            return DBDE_HIP_OK;
        }
    }
    // workspace, zeroed before the launch: [ticket 16 B][state 8 n cpf][gsum 8 n gpf][fsize 8 n][fgsum 8 ceil(n / 64)]
    const size_t n = (size_t)n_frames, gpf = (cpf + 63) / 64;
    const size_t need = 16 + 8 * (n * cpf + n * gpf + n + (n + 63) / 64);
    int rc = grow(ctx, ctx->w16, ctx->w16_bytes, need, 1, true);
    if (rc) return rc;
    dbde16::Params16 p;
    p.images = d_images;
    p.out = d_out;
    p.frame_offsets = d_frame_offsets;
    p.frame_bytes = d_frame_bytes;
    p.first_index = first_index;
    p.slot_stride = slot_stride;
    p.frame_pixels = g.pixels;
    p.W = W; p.H = H; p.w = g.w; p.h = g.h; p.T = g.T;
    p.chunks_per_frame = cpf;
    p.n_frames = (uint32_t)n_frames;
    p.ticket = reinterpret_cast<uint32_t *>(ctx->w16);
    p.state = reinterpret_cast<unsigned long long *>(ctx->w16 + 16);
    p.gsum = p.state + n * cpf;
    p.fsize = p.gsum + n * gpf;
    p.fgsum = p.fsize + n;
    p.sticky = ctx->sticky;
    p.diag = reinterpret_cast<unsigned long long *>(ctx->diag);
    if (!ctx->enc16_grid) ctx->enc16_grid = (uint32_t)(dbde16::encode16_blocks_per_cu() * ctx->n_cu);
    span_begin(ctx, 0);
    HIP_TRY(ctx, hipMemsetAsync(ctx->w16, 0, need, ctx->stream));
    p.force_tickets = (ctx->exp_flags & 1u) ? 1u : 0u;
    HIP_TRY(ctx, dbde16::launch_encode16(p, n_frames, ctx->enc16_grid, ctx->stream));
    span_end(ctx);
    return DBDE_HIP_OK;
}

int dbde16_hip_decode_frames(dbde_hip_ctx *ctx, const uint8_t *d_stream, size_t stream_bytes, const uint64_t *d_frame_offsets,
                             int W, int H, int n_frames, uint16_t *d_images, dbde_hip_frame_result *d_results) {
    if (!ctx) return DBDE_HIP_ERR_ARG;
    Geometry g;
    if (!d_stream || !d_frame_offsets || !d_images || n_frames < 0 || !geometry(W, H, g))
        return fail(ctx, DBDE_HIP_ERR_ARG, "decode16: bad argument (W=%d H=%d n=%d)", W, H, n_frames);
    if (n_frames == 0) return DBDE_HIP_OK;
    const DecGeom dg = dec_geometry(g.w, g.h, false, dbde16::kChunkTiles16);   // plain runs of 256 tiles
    if (dg.cpf > kMaxChunksPerFrame) return fail(ctx, DBDE_HIP_ERR_ARG, "decode16: frame too large");
    const uint64_t n_chunks64 = (uint64_t)n_frames * dg.cpf;
    if (n_chunks64 >= (1ull << 31)) return fail(ctx, DBDE_HIP_ERR_ARG, "decode16: too many chunks in one call");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = grow(ctx, ctx->chunk_off, ctx->chunk_off_n, (size_t)n_chunks64 + (size_t)n_frames, sizeof(uint32_t));
    if (rc) return rc;
    rc = grow(ctx, ctx->frame_ok, ctx->frame_ok_n, (size_t)n_frames, sizeof(uint32_t));
    if (rc) return rc;
    IdxParams ip;
    ip.stream = d_stream;
    ip.frame_offsets = d_frame_offsets;
    ip.stream_bytes = stream_bytes;
    ip.chunk_off = ctx->chunk_off;
    ip.frame_ok = ctx->frame_ok;
    ip.results = d_results;
    ip.T = g.T;
    ip.chunks_per_frame = dg.cpf;
    ip.min_bytes = 2;        // U16 minima, depth <= 16, nm = 2T
    ip.geom = dg;
    ip.split = 1;            // one index workgroup per frame (the split form is a latency tool of the 8-bit path)
    ip.frame_ctr = nullptr;
    ip.frame_flag = nullptr;
    span_begin(ctx, 1);
    HIP_TRY(ctx, launch_decode_index(ip, n_frames, ctx->stream));
    span_end(ctx);
    dbde16::DecParams16 p;
    p.stream = d_stream;
    p.stream_bytes = stream_bytes;
    p.frame_offsets = d_frame_offsets;
    p.images = d_images;
    p.chunk_off = ctx->chunk_off;
    p.frame_ok = ctx->frame_ok;
    p.frame_pixels = g.pixels;
    p.W = W; p.H = H; p.w = g.w; p.h = g.h; p.T = g.T;
    p.chunks_per_frame = dg.cpf;
    p.diag = reinterpret_cast<unsigned long long *>(ctx->diag);
    span_begin(ctx, 2);
    HIP_TRY(ctx, dbde16::launch_decode16(p, n_frames, ctx->stream));
    span_end(ctx);
    return DBDE_HIP_OK;
}

// ---- host-pointer entry points ---------------------------------------------------------------

// Device -> host on the context's OWN stream (hipMemcpy would go through the null stream, where the calls of every thread
// of a multi-threaded caller queue up behind each other).
static bool d2h(dbde_hip_ctx *ctx, void *dst, const void *src, size_t n, uint8_t *via = nullptr) {
    void *land = ctx->host_staging && via ? via : dst;
    if (hipMemcpyAsync(land, src, n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) return false;
    if (land != dst) memcpy(dst, land, n);
    return true;
}

// Encodes one host image through the GPU; returns the frame's byte count and leaves the
// packed frame (header + data) in ctx->st_pack.  0 on failure.
static size_t encode_one_host(dbde_hip_ctx *ctx, uint64_t index, const uint8_t *image, int W, int H) {
    Geometry g;
    if (!ctx || !image || !geometry(W, H, g)) return 0;
    if (hipSetDevice(ctx->device) != hipSuccess) return 0;   // (the calling thread may never have touched this device)
    const size_t maxf = 32 + 66 * (size_t)g.T;
    if (ensure_staging(ctx, (size_t)g.pixels, maxf)) return 0;
    if (!h2d(ctx, ctx->st_img, image, (size_t)g.pixels, ctx->h_img)) return 0;
    // the byte count lands in pinned host memory, written by the kernel itself: nothing small travels between the image
    // going in and the frame coming out (a copy of 8 bytes costs the link what a hundred kilobytes cost it)
    volatile uint64_t *h_bytes = ctx->h_words;
    *h_bytes = 0;
    if (dbde_hip_encode_frames(ctx, ctx->st_img, W, H, 1, index, nullptr, nullptr, ctx->st_pack, maxf, 0, nullptr,
                               const_cast<uint64_t *>(h_bytes)) != DBDE_HIP_OK)
        return 0;
    // One frame is a launch of the small, tiny or mid encoders, which cannot raise the sticky failure word (their waits
    // end in a fallback); only a frame of more chunks than the device holds workgroups runs the persistent encoder
    Geometry gg = g;
    const EncPlan pl = plan_encode(gg, W, 1, reinterpret_cast<uintptr_t>(ctx->st_img), reinterpret_cast<uintptr_t>(ctx->st_pack), 0, ctx->enc_grid);
    if (pl.kernel == 0) { if (dbde_hip_sync(ctx) != DBDE_HIP_OK) return 0; }
    else if (hipStreamSynchronize(ctx->stream) != hipSuccess) return 0;
    const uint64_t nbytes = *h_bytes;
    return nbytes <= maxf ? (size_t)nbytes : 0;
}

size_t dbde_hip_pack_frame(dbde_hip_ctx *ctx, uint64_t index, const uint8_t *image, int W, int H, uint8_t *target) {
    size_t n = encode_one_host(ctx, index, image, W, H);
    if (!n || !target) return 0;
    if (!d2h(ctx, target, ctx->st_pack, n, ctx->h_pack)) return 0;
    return n;
}

size_t dbde_hip_pack_image(dbde_hip_ctx *ctx, const uint8_t *image, int W, int H, uint8_t *target) {
    size_t n = encode_one_host(ctx, 0, image, W, H);
    if (n <= 20 || !target) return 0;
    if (!d2h(ctx, target, ctx->st_pack + 20, n - 20, ctx->h_pack)) return 0;
    return n - 20;
}

uint32_t dbde_hip_pack_8x8_partial(dbde_hip_ctx *ctx, const uint8_t *image, int stride, int rightmargin,
                                   int downmargin, uint8_t *target) {
    // A rm x dm image is exactly one constant-padded tile (dbde_util.cpp:105-135).
    if (!ctx || !image || rightmargin < 1 || downmargin < 1) return 0;
    const int rm = rightmargin > 8 ? 8 : rightmargin, dm = downmargin > 8 ? 8 : downmargin;
    uint8_t dense[64];
    for (int r = 0; r < dm; r++) memcpy(dense + r * rm, image + (ptrdiff_t)r * stride, (size_t)rm);   // gather only
    size_t n = encode_one_host(ctx, 0, dense, rm, dm);
    if (n < 34) return 0;
    uint8_t head[34 + 64];
    if (!d2h(ctx, head, ctx->st_pack, n)) return 0;
    // T = 1: header 20 | nb 4 | depth 1 | nm 4 | min 1 | n64 4 | payload
    const uint32_t depth = head[24], mn = head[29];
    if (target && depth) memcpy(target, head + 34, 8u * depth);
    return (depth << 8) | mn;
}

uint32_t dbde_hip_pack_8x8(dbde_hip_ctx *ctx, const uint8_t *image, int stride, uint8_t *target) {
    return dbde_hip_pack_8x8_partial(ctx, image, stride, 8, 8, target);
}

// Runs index + decode for one frame_data already resident at ctx->st_pack + 20 (a dummy
// frame header precedes it).  Returns bytes of frame data consumed (0 = rejected) and leaves
// the image in ctx->st_img.
static size_t decode_one_staged(dbde_hip_ctx *ctx, size_t staged_bytes, int W, int H) {
    Geometry g;
    if (!geometry(W, H, g)) return 0;
    uint64_t *d_off = ctx->scratch64 + 2;   // a zero that lives in device memory (cleared when the context was made, never written)
    // the frame's result record is written by the kernel straight into pinned host memory
    volatile dbde_hip_frame_result *h_res = reinterpret_cast<volatile dbde_hip_frame_result *>(ctx->h_words + 1);
    h_res->consumed = 0;
    if (dbde_hip_decode_frames(ctx, ctx->st_pack, staged_bytes, d_off, W, H, 1, ctx->st_img,
                               const_cast<dbde_hip_frame_result *>(h_res)) != DBDE_HIP_OK) return 0;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return 0;   // (decode kernels have no sticky failure)
    const uint64_t consumed = h_res->consumed;
    return consumed > 20 ? (size_t)(consumed - 20) : 0;
}

size_t dbde_hip_unpack_image(dbde_hip_ctx *ctx, const uint8_t *packed, int W, int H, uint8_t *image) {
    Geometry g;
    if (!ctx || !packed || !image || !geometry(W, H, g)) return 0;
    // Only to learn how many bytes to move: the first I32 and the word count
    // (validation proper happens on the device, dbde_util.cpp:295-303 order preserved).
    if ((int32_t)get32(packed) != (int32_t)g.T) return 0;
    if ((int32_t)get32(packed + 4 + g.T) != (int32_t)g.T) return 0;
    const int32_t n64 = (int32_t)get32(packed + 8 + 2 * (size_t)g.T);
    if (n64 < 0 || (uint64_t)n64 > 8ull * g.T) return 0;   // cannot equal sum(depth) with depth <= 8
    const size_t body = 12 + 2 * (size_t)g.T + 8 * (size_t)n64;
    if (hipSetDevice(ctx->device) != hipSuccess) return 0;
    if (ensure_staging(ctx, (size_t)g.pixels, 20 + body + 128)) return 0;
    // a frame header in front of the caller's frame data: device to device (a 20-byte copy from the stack was a trip over the link)
    if (hipMemcpyAsync(ctx->st_pack, ctx->d_hdr, 20, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) return 0;
    if (!h2d(ctx, ctx->st_pack + 20, packed, body, ctx->h_pack)) return 0;
    const size_t used = decode_one_staged(ctx, 20 + body, W, H);
    if (!used) return 0;
    if (!d2h(ctx, image, ctx->st_img, (size_t)g.pixels, ctx->h_img)) return 0;
    return used;
}

dbde_hip_frame_header dbde_hip_unpack_frame(dbde_hip_ctx *ctx, uint8_t **packed, int W, int H, uint8_t *image) {
    dbde_hip_frame_header fh = dbde_hip_unpack_frame_header(packed);   // advances by 20 (dbde_util.cpp:340)
    const size_t n = dbde_hip_unpack_image(ctx, *packed, W, H, image);
    if (n == 0) fh.u64s = 0xFFFFFFFFu;   // dbde_util.cpp:342
    else *packed += n;
    return fh;
}

void dbde_hip_unpack_8x8_partial(dbde_hip_ctx *ctx, uint8_t depth, uint8_t minval, const uint8_t *packed,
                                 size_t stride, int rightmargin, int downmargin, uint8_t *image) {
    if (!ctx || !image || depth > 8 || rightmargin < 1 || downmargin < 1) return;
    const int rm = rightmargin > 8 ? 8 : rightmargin, dm = downmargin > 8 ? 8 : downmargin;
    // frame data of a one-tile rm x dm frame
    uint8_t body[14 + 64];
    put32(body, 1);
    body[4] = depth;
    put32(body + 5, 1);
    body[9] = minval;
    put32(body + 10, depth);
    if (depth) memcpy(body + 14, packed, 8u * depth);
    uint8_t dense[64];
    if (dbde_hip_unpack_image(ctx, body, rm, dm, dense) == 0) return;
    for (int r = 0; r < dm; r++) memcpy(image + r * stride, dense + r * rm, (size_t)rm);   // scatter only
}

void dbde_hip_unpack_8x8(dbde_hip_ctx *ctx, uint8_t depth, uint8_t minval, const uint8_t *packed, size_t stride,
                         uint8_t *image) {
    dbde_hip_unpack_8x8_partial(ctx, depth, minval, packed, stride, 8, 8, image);
}

// ---- header wire format ----------------------------------------------------------------------

size_t dbde_hip_pack_frame_header(const dbde_hip_frame_header *fh, uint8_t *target) {
    put32(target, fh->u64s);
    put64(target + 4, fh->index);
    const double el = (double)fh->elapsed_ns;   // the reference stores a double here (dbde_util.cpp:186)
    uint64_t bits;
    memcpy(&bits, &el, 8);
    put64(target + 12, bits);
    return 20;
}

size_t dbde_hip_pack_video_header(const dbde_hip_video_header *vh, uint8_t *target) {
    put32(target, vh->u64s);
    put64(target + 4, vh->height);
    put64(target + 12, vh->width);
    uint64_t bits;
    memcpy(&bits, &vh->frame_hz, 8);
    put64(target + 20, bits);
    return 28;
}

dbde_hip_frame_header dbde_hip_unpack_frame_header(uint8_t **packed) {
    dbde_hip_frame_header fh;
    const uint8_t *p = *packed;
    fh.u64s = get32(p);
    fh.index = get64(p + 4);
    const uint64_t bits = get64(p + 12);
    double el;
    memcpy(&el, &bits, 8);
    fh.elapsed_ns = (uint64_t)el;
    if (fh.u64s != 2) fh.u64s = 0xFFFFFFFFu;
    *packed += 20;
    return fh;
}

dbde_hip_video_header dbde_hip_unpack_video_header(uint8_t **packed) {
    dbde_hip_video_header vh;
    const uint8_t *p = *packed;
    vh.u64s = get32(p);
    vh.height = get64(p + 4);
    vh.width = get64(p + 12);
    const uint64_t bits = get64(p + 20);
    memcpy(&vh.frame_hz, &bits, 8);
    if (vh.u64s != 3) vh.u64s = 0xFFFFFFFFu;
    *packed += 28;
    return vh;
}

// ---- timing ----------------------------------------------------------------------------------

// Diagnostic builds (-DDBDE_DIAG) accumulate in-kernel cycle counters; this reads (and clears) them.
// Not part of include/dbde_hip.h: tuning tool (profiles/abbench.cpp) only.
int dbde_hip_diag_read(dbde_hip_ctx *ctx, uint64_t out[16]) {
    if (!ctx) return DBDE_HIP_ERR_ARG;
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->diag, 128, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->diag, 0, 128, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DBDE_HIP_OK;
}
#ifdef DBDE_DIAG
// [1024][16] u64: wave 0's wall clock (10 ns) at the points of a persistent-encoder workgroup's life (encode_kernel).
int dbde_hip_diag_trace_read(dbde_hip_ctx *ctx, uint64_t *out, size_t n_u64) {
    if (!ctx || n_u64 > 16 * 1024) return DBDE_HIP_ERR_ARG;
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->diag + 16, 8 * n_u64, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->diag + 16, 0, 8 * 16 * 1024, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DBDE_HIP_OK;
}
#endif

int dbde_hip_timing_enable(dbde_hip_ctx *ctx, int on) {
    if (!ctx) return DBDE_HIP_ERR_ARG;
    ctx->timing = on != 0;
    return DBDE_HIP_OK;
}

int dbde_hip_timing_read(dbde_hip_ctx *ctx, double ms[4], uint64_t launches[4], int reset) {
    if (!ctx) return DBDE_HIP_ERR_ARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->scan_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->scan_stream));
    for (auto &s : ctx->spans) {
        float t = 0;
        if (hipEventElapsedTime(&t, s.a, s.b) == hipSuccess) {
            ctx->acc_ms[s.kind] += t;
            ctx->acc_n[s.kind] += 1;
        }
        (void)hipEventDestroy(s.a);
        (void)hipEventDestroy(s.b);
    }
    ctx->spans.clear();
    for (int k = 0; k < 4; k++) {
        if (ms) ms[k] = ctx->acc_ms[k];
        if (launches) launches[k] = ctx->acc_n[k];
        if (reset) { ctx->acc_ms[k] = 0; ctx->acc_n[k] = 0; }
    }
    return DBDE_HIP_OK;
}

}  // extern "C"
