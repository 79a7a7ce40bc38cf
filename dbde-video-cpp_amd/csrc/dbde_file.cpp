// dbde_file.cpp -- .dbde file reader behind the reference's walker API
// (dbde_util.h:39-52, dbde_util.cpp:362-426) plus the writer the reference lacks.
//
// The walker keeps a host byte window of the file and hands complete frames to the HIP
// decoder (dbde_unpack_frame -> dbde_hip_unpack_frame).  Error conventions follow the
// reference: a walker whose fptr is NULL is dead; dbde_walk_a_file returns false at the end
// of the file or on the first frame that does not parse.
#include <cstdlib>
#include <cstring>

#include "../../include/dbde_hip.h"
#include "../../include/dbde_util.h"

namespace {

size_t worst_frame(int32_t width, int32_t height) { return dbde_hip_max_frame_bytes(width, height); }

void kill(dbde_file_walker *w) {
    if (w->fptr) fclose(w->fptr);
    w->fptr = nullptr;
}

// Slide unread bytes to the front and top the window up from the file.
bool refill(dbde_file_walker *w) {
    if (w->i > 0) {
        if (w->i < w->n) memmove(w->buffer, w->buffer + w->i, w->n - w->i);
        w->n -= w->i;
        w->i = 0;
    }
    if (!feof(w->fptr) && w->n < w->N) {
        w->n += fread(w->buffer + w->n, 1, w->N - w->n, w->fptr);
        if (ferror(w->fptr)) return false;
    }
    return true;
}

}  // namespace

// Exported by the reference's object file although dbde_util.h does not declare it (dbde_util.cpp:394-406): makes sure
// a worst-case frame's worth of bytes lies past the read position, sliding the unread bytes to the front and topping the
// window up from the file.  False only on a read error.  Kept so that the two libraries export the same symbol set.
bool dbde_advance_file_buffer(dbde_file_walker &w) {
    if (!w.fptr || !w.buffer) return false;
    return refill(&w);
}

dbde_file_walker dbde_start_file_walk(const char *name, int frames_buffered, video_header *vh) {
    dbde_file_walker w;
    memset(&w, 0, sizeof w);
    w.width = 1;
    w.height = 1;
    if (frames_buffered < 1) frames_buffered = 2;   // dbde_util.cpp:363
    w.fptr = fopen(name, "rb");
    if (!w.fptr) return w;
    uint8_t head[28];
    uint8_t *cur = head;
    if (fread(head, 1, 28, w.fptr) != 28) { kill(&w); return w; }
    *vh = dbde_unpack_video_header(&cur);
    // same acceptance limits as the reference (dbde_util.cpp:371-381)
    if (vh->u64s != 3 || vh->height == 0 || vh->width == 0 || vh->height > 0x37FFFFFF ||
        vh->width > 0x37FFFFFF || vh->height * vh->width > 0x37FFFFFF) {
        kill(&w);
        return w;
    }
    w.width = (int32_t)vh->width;
    w.height = (int32_t)vh->height;
    const size_t one = worst_frame(w.width, w.height);
    if (one == 0 || one * (size_t)frames_buffered >= 0x7FFFFFFFull) { kill(&w); return w; }
    w.N = one * (size_t)frames_buffered;
    w.buffer = (uint8_t *)malloc(w.N);
    if (!w.buffer) { kill(&w); return w; }
    w.n = fread(w.buffer, 1, w.N, w.fptr);
    if (ferror(w.fptr)) kill(&w);
    return w;
}

bool dbde_walk_a_file(dbde_file_walker *walker, frame_header *fh, uint8_t *image) {
    if (!walker || !walker->fptr) return false;
    const size_t one = worst_frame(walker->width, walker->height);
    if (walker->n - walker->i < one) {
        if (!refill(walker)) { kill(walker); return false; }
        if (walker->n - walker->i < 20) return false;   // end of file (dbde_util.cpp:412)
    }
    // The frame must lie wholly inside the window: its length is only known in-band.
    const size_t avail = walker->n - walker->i;
    const size_t T = (size_t)((walker->width + 7) / 8) * (size_t)((walker->height + 7) / 8);
    if (avail < 32 + 2 * T) { kill(walker); return false; }
    uint8_t *cur = walker->buffer + walker->i;
    uint32_t n64 = 0;
    memcpy(&n64, cur + 28 + 2 * T, 4);
    if (n64 > 8 * T || avail < 32 + 2 * T + 8 * (size_t)n64) { kill(walker); return false; }
    *fh = dbde_unpack_frame(&cur, walker->width, walker->height, image);
    if (fh->u64s != 2) { kill(walker); return false; }
    walker->i = (size_t)(cur - walker->buffer);
    walker->frames += 1;
    return true;
}

void dbde_end_file_walk(dbde_file_walker *walker) {
    if (!walker) return;
    kill(walker);
    free(walker->buffer);
    walker->buffer = nullptr;
}
