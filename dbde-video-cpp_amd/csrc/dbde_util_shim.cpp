// dbde_util_shim.cpp -- the reference's C++ API (include/dbde_util.h) over the C-ABI.
//
// Exports the same mangled symbols as the reference's dbde_util.o so that programs written
// against the reference link unchanged (SURVEY.md 8b lists them).  Each function is one
// forwarding call into libdbde_hip.so; the only state is the lazily created context.
#include "../../include/dbde_util.h"

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/dbde_hip.h"

namespace {

std::mutex g_lock;
dbde_hip_ctx *g_ctx = nullptr;

// One context per process, on $DBDE_HIP_DEVICE.  No device -> no codec: fail loudly.
dbde_hip_ctx *ctx() {
    if (g_ctx) return g_ctx;
    const char *dev = getenv("DBDE_HIP_DEVICE");
    int rc = dbde_hip_create(dev ? atoi(dev) : 0, nullptr, &g_ctx);
    if (rc != DBDE_HIP_OK || !g_ctx) {
        fprintf(stderr, "dbde_util (HIP): no usable gfx950 device (dbde_hip_create -> %d); "
                        "this build has no CPU path\n", rc);
        abort();
    }
    return g_ctx;
}

}  // namespace

uint32_t dbde_pack_8x8(uint8_t *image, int stride, uint8_t *target) {
    std::lock_guard<std::mutex> g(g_lock);
    return dbde_hip_pack_8x8(ctx(), image, stride, target);
}

uint32_t dbde_pack_8x8_partial(uint8_t *image, int stride, int rightmargin, int downmargin, uint8_t *target) {
    std::lock_guard<std::mutex> g(g_lock);
    return dbde_hip_pack_8x8_partial(ctx(), image, stride, rightmargin, downmargin, target);
}

size_t dbde_pack_image(uint8_t *image, int W, int H, uint8_t *target) {
    std::lock_guard<std::mutex> g(g_lock);
    return dbde_hip_pack_image(ctx(), image, W, H, target);
}

size_t dbde_pack_frame_header(frame_header fh, uint8_t *target) {
    dbde_hip_frame_header h = {fh.u64s, fh.index, fh.elapsed_ns};
    return dbde_hip_pack_frame_header(&h, target);
}

size_t dbde_pack_frame(uint64_t index, uint8_t *image, int W, int H, uint8_t *target) {
    std::lock_guard<std::mutex> g(g_lock);
    return dbde_hip_pack_frame(ctx(), index, image, W, H, target);
}

size_t dbde_pack_video_header(video_header vh, uint8_t *target) {
    dbde_hip_video_header h = {vh.u64s, vh.height, vh.width, vh.frame_hz};
    return dbde_hip_pack_video_header(&h, target);
}

void dbde_unpack_8x8(uint8_t depth, uint8_t minval, uint8_t *packed, size_t stride, uint8_t *image) {
    std::lock_guard<std::mutex> g(g_lock);
    dbde_hip_unpack_8x8(ctx(), depth, minval, packed, stride, image);
}

void dbde_unpack_8x8_partial(uint8_t depth, uint8_t minval, uint8_t *packed, size_t stride, int rightmargin,
                             int downmargin, uint8_t *image) {
    std::lock_guard<std::mutex> g(g_lock);
    dbde_hip_unpack_8x8_partial(ctx(), depth, minval, packed, stride, rightmargin, downmargin, image);
}

size_t dbde_unpack_image(uint8_t *packed, int W, int H, uint8_t *image) {
    std::lock_guard<std::mutex> g(g_lock);
    return dbde_hip_unpack_image(ctx(), packed, W, H, image);
}

frame_header dbde_unpack_frame_header(uint8_t **packed) {
    dbde_hip_frame_header h = dbde_hip_unpack_frame_header(packed);
    frame_header fh;
    fh.u64s = h.u64s;
    fh.index = h.index;
    fh.elapsed_ns = h.elapsed_ns;
    return fh;
}

frame_header dbde_unpack_frame(uint8_t **packed, int W, int H, uint8_t *image) {
    std::lock_guard<std::mutex> g(g_lock);
    dbde_hip_frame_header h = dbde_hip_unpack_frame(ctx(), packed, W, H, image);
    frame_header fh;
    fh.u64s = h.u64s;
    fh.index = h.index;
    fh.elapsed_ns = h.elapsed_ns;
    return fh;
}

video_header dbde_unpack_video_header(uint8_t **packed) {
    dbde_hip_video_header h = dbde_hip_unpack_video_header(packed);
    video_header vh;
    vh.u64s = h.u64s;
    vh.height = h.height;
    vh.width = h.width;
    vh.frame_hz = h.frame_hz;
    return vh;
}
