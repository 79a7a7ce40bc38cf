// dbde_stream_io.cpp -- batched .dbde file writer and reader on top of the batch codec
// (include/dbde_hip.h, "file I/O" section).
//
// The reference reads a file one frame at a time (dbde_start_file_walk / dbde_walk_a_file,
// dbde_util.cpp:362-426) and has no writer: its test hand-rolls one (dbde_util_test.cpp:
// 204-211).  Here both ends move whole batches: images stay in HBM, the compressed bytes
// cross PCIe through two pinned windows so that file I/O of one batch overlaps the kernels
// and copies of the next.  Only container framing happens on the host (the 28-byte video
// header and the in-band frame-length hop 20 + 12 + 2T + 8*n64, README.md:12-23); every tile
// is packed, validated and unpacked by the HIP kernels.
#include "../../include/dbde_hip.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }   // little-endian host (x86-64)

struct Slot {
    uint8_t *dev = nullptr;      // device window
    uint8_t *pin = nullptr;      // pinned host window
    size_t bytes = 0;            // valid bytes
    hipEvent_t moved = nullptr;  // copy between dev and pin finished
    bool pending = false;        // writer: D2H issued, not yet written to the file
};

void free_slot(Slot &s) {
    if (s.dev) (void)hipFree(s.dev);
    if (s.pin) (void)hipHostFree(s.pin);
    if (s.moved) (void)hipEventDestroy(s.moved);
    s = Slot();
}

bool make_slot(Slot &s, size_t cap) {
    void *d = nullptr, *h = nullptr;
    if (hipMalloc(&d, cap) != hipSuccess) return false;
    s.dev = static_cast<uint8_t *>(d);
    if (hipHostMalloc(&h, cap, hipHostMallocDefault) != hipSuccess) return false;
    s.pin = static_cast<uint8_t *>(h);
    return hipEventCreateWithFlags(&s.moved, hipEventDisableTiming) == hipSuccess;
}

}  // namespace

struct dbde_hip_writer {
    dbde_hip_ctx *ctx = nullptr;
    FILE *f = nullptr;
    int W = 0, H = 0, batch = 0;
    size_t maxf = 0, cap = 0;
    Slot slot[2];
    int cur = 0;
    hipStream_t copy = nullptr;
    uint64_t *d_tail = nullptr;   // [2*batch]: offsets, sizes of the batch just encoded
    uint64_t *h_tail = nullptr;   // pinned [2]
    uint64_t frames = 0, bytes = 0;
    bool failed = false;
    std::string err;
};

struct dbde_hip_reader {
    dbde_hip_ctx *ctx = nullptr;
    FILE *f = nullptr;
    int W = 0, H = 0, batch = 0;
    uint32_t T = 0;
    size_t maxf = 0, cap = 0;
    Slot slot[2];
    int cur = 0;
    hipStream_t copy = nullptr;
    uint64_t *d_off = nullptr;             // [batch]
    uint64_t *h_off = nullptr;             // pinned [batch]
    dbde_hip_frame_result *d_res = nullptr;
    dbde_hip_frame_result *h_res = nullptr;   // pinned [batch]
    uint64_t frames = 0;
    bool eof = false, dead = false;
};

namespace {

// ---- writer ------------------------------------------------------------------------------

// Waits for a slot's D2H and appends it to the file.
bool writer_flush(dbde_hip_writer *w, Slot &s) {
    if (!s.pending) return true;
    s.pending = false;
    if (hipEventSynchronize(s.moved) != hipSuccess) { w->err = "writer: D2H copy failed"; return false; }
    if (s.bytes && fwrite(s.pin, 1, s.bytes, w->f) != s.bytes) { w->err = "writer: short write"; return false; }
    w->bytes += s.bytes;
    return true;
}

void writer_free(dbde_hip_writer *w) {
    if (!w) return;
    free_slot(w->slot[0]);
    free_slot(w->slot[1]);
    if (w->d_tail) (void)hipFree(w->d_tail);
    if (w->h_tail) (void)hipHostFree(w->h_tail);
    if (w->copy) (void)hipStreamDestroy(w->copy);
    if (w->f) fclose(w->f);
    delete w;
}

int writer_put_batch(dbde_hip_writer *w, const uint8_t *d_images, int n, uint64_t first_index,
                     const uint64_t *d_indices, const uint64_t *d_elapsed_ns) {
    hipStream_t st = static_cast<hipStream_t>(dbde_hip_stream_handle(w->ctx));
    Slot &s = w->slot[w->cur];
    Slot &other = w->slot[w->cur ^ 1];
    // this window was last used two batches ago: its bytes must be in the file before reuse
    if (!writer_flush(w, s)) return DBDE_HIP_ERR_HIP;
    int rc = dbde_hip_encode_frames(w->ctx, d_images, w->W, w->H, n, first_index, d_indices, d_elapsed_ns, s.dev,
                                    w->cap, 0, w->d_tail, w->d_tail + w->batch);
    if (rc != DBDE_HIP_OK) { w->err = dbde_hip_last_error(w->ctx); return rc; }
    // total = offset + size of the last frame
    if (hipMemcpyAsync(&w->h_tail[0], w->d_tail + (n - 1), 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&w->h_tail[1], w->d_tail + w->batch + (n - 1), 8, hipMemcpyDeviceToHost, st) != hipSuccess) {
        w->err = "writer: hipMemcpyAsync failed";
        return DBDE_HIP_ERR_HIP;
    }
    // while the encoder runs, the previous batch goes to the file
    if (!writer_flush(w, other)) return DBDE_HIP_ERR_HIP;
    rc = dbde_hip_sync(w->ctx);
    if (rc != DBDE_HIP_OK) { w->err = dbde_hip_last_error(w->ctx); return rc; }
    s.bytes = (size_t)(w->h_tail[0] + w->h_tail[1]);
    if (s.bytes > w->cap) { w->err = "writer: encoder reported more bytes than the window holds"; return DBDE_HIP_ERR_DEVICE; }
    if (hipMemcpyAsync(s.pin, s.dev, s.bytes, hipMemcpyDeviceToHost, w->copy) != hipSuccess ||
        hipEventRecord(s.moved, w->copy) != hipSuccess) {
        w->err = "writer: D2H of the packed batch failed";
        return DBDE_HIP_ERR_HIP;
    }
    s.pending = true;
    w->frames += (uint64_t)n;
    w->cur ^= 1;
    return DBDE_HIP_OK;
}

// ---- reader ------------------------------------------------------------------------------

void reader_free(dbde_hip_reader *r) {
    if (!r) return;
    free_slot(r->slot[0]);
    free_slot(r->slot[1]);
    if (r->d_off) (void)hipFree(r->d_off);
    if (r->h_off) (void)hipHostFree(r->h_off);
    if (r->d_res) (void)hipFree(r->d_res);
    if (r->h_res) (void)hipHostFree(r->h_res);
    if (r->copy) (void)hipStreamDestroy(r->copy);
    if (r->f) fclose(r->f);
    delete r;
}

// Tops the window up from the file (after `have` carried-over bytes) and starts its H2D.
bool reader_fill(dbde_hip_reader *r, Slot &s, size_t have) {
    size_t got = 0;
    if (!r->eof) {
        got = fread(s.pin + have, 1, r->cap - have, r->f);
        if (ferror(r->f)) return false;
        if (got < r->cap - have) r->eof = true;
    }
    s.bytes = have + got;
    if (s.bytes && hipMemcpyAsync(s.dev, s.pin, s.bytes, hipMemcpyHostToDevice, r->copy) != hipSuccess) return false;
    return hipEventRecord(s.moved, r->copy) == hipSuccess;
}

}  // namespace

extern "C" {

int dbde_hip_writer_open(dbde_hip_ctx *ctx, const char *path, int W, int H, double frame_hz, int batch_frames,
                         dbde_hip_writer **out) {
    if (!out) return DBDE_HIP_ERR_ARG;
    *out = nullptr;
    const size_t maxf = dbde_hip_max_frame_bytes(W, H);
    if (!ctx || !path || maxf == 0 || batch_frames < 1) return DBDE_HIP_ERR_ARG;
    // one encode call carries its running payload-word count in 32 bits (dbde_hip_encode_frames)
    const uint64_t T = (maxf - 32) / 66;
    while (batch_frames > 1 && (uint64_t)batch_frames * 8ull * T >= (1ull << 32)) batch_frames /= 2;
    if (hipSetDevice(dbde_hip_device_index(ctx)) != hipSuccess) return DBDE_HIP_ERR_HIP;
    dbde_hip_writer *w = new dbde_hip_writer;
    w->ctx = ctx;
    w->W = W;
    w->H = H;
    w->batch = batch_frames;
    w->maxf = maxf;
    w->cap = maxf * (size_t)batch_frames;
    void *p = nullptr;
    bool ok = make_slot(w->slot[0], w->cap) && make_slot(w->slot[1], w->cap) &&
              hipStreamCreateWithFlags(&w->copy, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc(&p, 2 * sizeof(uint64_t) * (size_t)batch_frames) == hipSuccess;
    w->d_tail = static_cast<uint64_t *>(p);
    p = nullptr;
    ok = ok && hipHostMalloc(&p, 2 * sizeof(uint64_t), hipHostMallocDefault) == hipSuccess;
    w->h_tail = static_cast<uint64_t *>(p);
    if (!ok) { writer_free(w); return DBDE_HIP_ERR_HIP; }
    w->f = fopen(path, "wb");
    if (!w->f) { writer_free(w); return DBDE_HIP_ERR_ARG; }
    dbde_hip_video_header vh;
    vh.u64s = 3;
    vh.height = (uint64_t)H;
    vh.width = (uint64_t)W;
    vh.frame_hz = frame_hz;
    uint8_t h28[28];
    dbde_hip_pack_video_header(&vh, h28);
    if (fwrite(h28, 1, 28, w->f) != 28) { writer_free(w); return DBDE_HIP_ERR_ARG; }
    w->bytes = 28;
    *out = w;
    return DBDE_HIP_OK;
}

int dbde_hip_writer_put(dbde_hip_writer *w, const uint8_t *d_images, int n_frames, uint64_t first_index,
                        const uint64_t *d_indices, const uint64_t *d_elapsed_ns) {
    if (!w || !d_images || n_frames < 0) return DBDE_HIP_ERR_ARG;
    if (w->failed) return DBDE_HIP_ERR_HIP;
    if (hipSetDevice(dbde_hip_device_index(w->ctx)) != hipSuccess) return DBDE_HIP_ERR_HIP;
    const size_t pixels = (size_t)w->W * (size_t)w->H;
    for (int done = 0; done < n_frames;) {
        const int n = n_frames - done < w->batch ? n_frames - done : w->batch;
        int rc = writer_put_batch(w, d_images + pixels * (size_t)done, n, first_index + (uint64_t)done,
                                  d_indices ? d_indices + done : nullptr, d_elapsed_ns ? d_elapsed_ns + done : nullptr);
        if (rc != DBDE_HIP_OK) { w->failed = true; return rc; }
        done += n;
    }
    return DBDE_HIP_OK;
}

const char *dbde_hip_writer_error(const dbde_hip_writer *w) { return w ? w->err.c_str() : "null writer"; }

int dbde_hip_writer_close(dbde_hip_writer *w, uint64_t *frames_written, uint64_t *bytes_written) {
    if (!w) return DBDE_HIP_ERR_ARG;
    int rc = DBDE_HIP_OK;
    (void)hipSetDevice(dbde_hip_device_index(w->ctx));
    // oldest window first: w->cur is the one that would be reused next
    if (!w->failed && !(writer_flush(w, w->slot[w->cur]) && writer_flush(w, w->slot[w->cur ^ 1]))) rc = DBDE_HIP_ERR_HIP;
    if (w->failed) rc = DBDE_HIP_ERR_HIP;
    if (w->copy) (void)hipStreamSynchronize(w->copy);
    if (w->f) {
        if (fclose(w->f) != 0 && rc == DBDE_HIP_OK) rc = DBDE_HIP_ERR_HIP;
        w->f = nullptr;
    }
    if (frames_written) *frames_written = w->frames;
    if (bytes_written) *bytes_written = w->bytes;
    writer_free(w);
    return rc;
}

int dbde_hip_reader_open(dbde_hip_ctx *ctx, const char *path, int batch_frames, dbde_hip_video_header *vh,
                         dbde_hip_reader **out) {
    if (!out) return DBDE_HIP_ERR_ARG;
    *out = nullptr;
    if (!ctx || !path || !vh || batch_frames < 1) return DBDE_HIP_ERR_ARG;
    FILE *f = fopen(path, "rb");
    if (!f) return DBDE_HIP_ERR_ARG;
    uint8_t h28[28];
    uint8_t *cur = h28;
    if (fread(h28, 1, 28, f) != 28) { fclose(f); return DBDE_HIP_ERR_ARG; }
    *vh = dbde_hip_unpack_video_header(&cur);
    // acceptance limits of the reference's walker (dbde_util.cpp:371-381)
    if (vh->u64s != 3 || vh->height == 0 || vh->width == 0 || vh->height > 0x37FFFFFF || vh->width > 0x37FFFFFF ||
        vh->height * vh->width > 0x37FFFFFF) {
        fclose(f);
        return DBDE_HIP_ERR_ARG;
    }
    const int W = (int)vh->width, H = (int)vh->height;
    const size_t maxf = dbde_hip_max_frame_bytes(W, H);
    if (maxf == 0 || hipSetDevice(dbde_hip_device_index(ctx)) != hipSuccess) { fclose(f); return DBDE_HIP_ERR_HIP; }
    dbde_hip_reader *r = new dbde_hip_reader;
    r->ctx = ctx;
    r->f = f;
    r->W = W;
    r->H = H;
    r->batch = batch_frames;
    r->T = (uint32_t)((maxf - 32) / 66);
    r->maxf = maxf;
    r->cap = maxf * (size_t)batch_frames;
    void *p = nullptr;
    bool ok = make_slot(r->slot[0], r->cap) && make_slot(r->slot[1], r->cap) &&
              hipStreamCreateWithFlags(&r->copy, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipMalloc(&p, sizeof(uint64_t) * (size_t)batch_frames) == hipSuccess;
    r->d_off = static_cast<uint64_t *>(p);
    p = nullptr;
    ok = ok && hipHostMalloc(&p, sizeof(uint64_t) * (size_t)batch_frames, hipHostMallocDefault) == hipSuccess;
    r->h_off = static_cast<uint64_t *>(p);
    p = nullptr;
    ok = ok && hipMalloc(&p, sizeof(dbde_hip_frame_result) * (size_t)batch_frames) == hipSuccess;
    r->d_res = static_cast<dbde_hip_frame_result *>(p);
    p = nullptr;
    ok = ok && hipHostMalloc(&p, sizeof(dbde_hip_frame_result) * (size_t)batch_frames, hipHostMallocDefault) == hipSuccess;
    r->h_res = static_cast<dbde_hip_frame_result *>(p);
    ok = ok && reader_fill(r, r->slot[0], 0);
    if (!ok) { reader_free(r); return DBDE_HIP_ERR_HIP; }
    *out = r;
    return DBDE_HIP_OK;
}

int dbde_hip_reader_next(dbde_hip_reader *r, uint8_t *d_images, int max_frames, dbde_hip_frame_header *headers,
                         int *n_out) {
    if (!r || !d_images || !n_out || max_frames < 0) return DBDE_HIP_ERR_ARG;
    *n_out = 0;
    if (r->dead || max_frames == 0) return DBDE_HIP_OK;
    if (max_frames > r->batch) max_frames = r->batch;
    if (hipSetDevice(dbde_hip_device_index(r->ctx)) != hipSuccess) return DBDE_HIP_ERR_HIP;
    hipStream_t st = static_cast<hipStream_t>(dbde_hip_stream_handle(r->ctx));
    Slot &s = r->slot[r->cur];
    Slot &nxt = r->slot[r->cur ^ 1];

    // frame boundaries of this window: the in-band hop, on the pinned copy (README.md:12-23)
    const size_t meta = 32 + 2 * (size_t)r->T;
    size_t off = 0;
    int n = 0;
    while (n < max_frames && off + meta <= s.bytes) {
        const uint32_t n64 = rd32(s.pin + off + 28 + 2 * (size_t)r->T);
        const size_t len = meta + 8 * (size_t)n64;
        if (n64 > 8u * r->T || off + len > s.bytes) break;   // not a whole frame (or not a frame at all)
        r->h_off[n++] = off;
        off += len;
    }
    if (n == 0) {   // end of file, a truncated tail, or bytes that cannot be a frame: the walk ends
        r->dead = true;
        return DBDE_HIP_OK;
    }
    // decode the batch once its window has landed in HBM
    if (hipStreamWaitEvent(st, s.moved, 0) != hipSuccess ||
        hipMemcpyAsync(r->d_off, r->h_off, sizeof(uint64_t) * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess)
        return DBDE_HIP_ERR_HIP;
    int rc = dbde_hip_decode_frames(r->ctx, s.dev, s.bytes, r->d_off, r->W, r->H, n, d_images, r->d_res);
    if (rc != DBDE_HIP_OK) return rc;
    if (hipMemcpyAsync(r->h_res, r->d_res, sizeof(dbde_hip_frame_result) * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess)
        return DBDE_HIP_ERR_HIP;
    // while the GPU decodes: carry the unread tail over and read the next window from the file
    const size_t left = s.bytes - off;
    // (the other pinned window is free: its H2D was consumed by the previous call's decode, which was synchronised)
    memcpy(nxt.pin, s.pin + off, left);
    if (!reader_fill(r, nxt, left)) { r->dead = true; return DBDE_HIP_ERR_HIP; }
    rc = dbde_hip_sync(r->ctx);
    if (rc != DBDE_HIP_OK) { r->dead = true; return rc; }
    // the walk stops at the first frame that does not parse (dbde_util.cpp:415-420)
    int good = 0;
    while (good < n && r->h_res[good].header.u64s == 2u) good++;
    if (good < n) r->dead = true;
    if (headers)
        for (int i = 0; i < good; i++) headers[i] = r->h_res[i].header;
    r->frames += (uint64_t)good;
    r->cur ^= 1;
    *n_out = good;
    return DBDE_HIP_OK;
}

void dbde_hip_reader_close(dbde_hip_reader *r) {
    if (!r) return;
    (void)hipSetDevice(dbde_hip_device_index(r->ctx));
    if (r->copy) (void)hipStreamSynchronize(r->copy);
    reader_free(r);
}

}  // extern "C"
