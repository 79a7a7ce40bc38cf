// gather1.cpp -- the native gather of the C-ABI (include/dbde_hip.h, dbde_hip_gather_*) from a C++ caller with no
// Python and no torch in the process: one rank (all a one-GPU box allows: RCCL refuses two ranks on one device).
//   1. in place : rank 0 encodes straight into its window; begin / post / sync; the count RCCL all-gathers must be
//                 the encoder's offset + size of the last frame, and nothing is copied;
//   2. loopback : the same frames encoded into a separate segment buffer, moved into a second window by
//                 ncclSend / ncclRecv in 4 KB pieces; both windows must hold the same bytes.
// Build: hipcc --offload-arch=gfx950 gather1.cpp -I include -L <pkg> -ldbde_hip.  Prints "ok".
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "dbde_hip.h"

#define CK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP error at %s\n", #x); return 2; } } while (0)
#define OK(x) do { int rc_ = (x); if (rc_ != DBDE_HIP_OK) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, g ? dbde_hip_gather_error(g) : ""); return 3; } } while (0)

int main() {
    const int W = 200, H = 123, n = 6;
    dbde_hip_ctx *ctx = nullptr;
    dbde_hip_gather *g = nullptr;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    if (dbde_hip_create(0, s, &ctx) != DBDE_HIP_OK) { fprintf(stderr, "no gfx950 device\n"); return 1; }
    if (dbde_hip_gather_rccl_version() < 22000) { fprintf(stderr, "librccl not usable\n"); return 1; }
    uint8_t id[DBDE_HIP_GATHER_ID_BYTES];
    OK(dbde_hip_gather_unique_id(id));
    OK(dbde_hip_gather_create(ctx, id, 1, 0, 0, &g));
    OK(dbde_hip_gather_set_max_message(g, 4096));
    const size_t cap = (size_t)n * dbde_hip_max_frame_bytes(W, H);
    uint8_t *img, *win_a, *win_b, *seg;
    uint64_t *offs, *bytes;
    CK(hipMalloc(&img, (size_t)n * W * H));
    CK(hipMalloc(&win_a, cap + 64)); CK(hipMalloc(&win_b, cap + 64)); CK(hipMalloc(&seg, cap + 64));
    CK(hipMalloc(&offs, 8 * n)); CK(hipMalloc(&bytes, 8 * n));
    CK(hipMemsetAsync(win_a, 0xEE, cap + 64, s)); CK(hipMemsetAsync(win_b, 0xEE, cap + 64, s));
    if (dbde_hip_synth_frames(ctx, 1, 0xDBDE2016ull, 5, n, W, H, img)) return 1;
    uint64_t sizes[1] = {0};
    // 1. in place
    OK(dbde_hip_gather_join(g, 0));
    if (dbde_hip_encode_frames(ctx, img, W, H, n, 5, nullptr, nullptr, win_a, cap, 0, offs, bytes)) return 1;
    OK(dbde_hip_gather_begin(g, 0, offs + (n - 1), bytes + (n - 1)));
    OK(dbde_hip_gather_post(g, 0, win_a, win_a, cap, sizes, 0));
    OK(dbde_hip_gather_sync(g, 0));
    uint64_t ho[n], hb[n];
    CK(hipMemcpy(ho, offs, 8 * n, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb, bytes, 8 * n, hipMemcpyDeviceToHost));
    const uint64_t total = ho[n - 1] + hb[n - 1];
    if (sizes[0] != total || total == 0) { fprintf(stderr, "count %llu != %llu\n", (unsigned long long)sizes[0], (unsigned long long)total); return 4; }
    // 2. loopback through ncclSend / ncclRecv, slot 1
    OK(dbde_hip_gather_join(g, 1));
    if (dbde_hip_encode_frames(ctx, img, W, H, n, 5, nullptr, nullptr, seg, cap, 0, offs, bytes)) return 1;
    OK(dbde_hip_gather_begin(g, 1, offs + (n - 1), bytes + (n - 1)));
    OK(dbde_hip_gather_post(g, 1, seg, win_b, cap, sizes, DBDE_HIP_GATHER_LOOPBACK));
    OK(dbde_hip_gather_sync(g, 1));
    if (sizes[0] != total) return 4;
    std::vector<uint8_t> a(cap + 64), b(cap + 64);
    CK(hipMemcpy(a.data(), win_a, cap + 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), win_b, cap + 64, hipMemcpyDeviceToHost));
    if (memcmp(a.data(), b.data(), total) != 0) { fprintf(stderr, "loopback window differs\n"); return 5; }
    for (size_t i = total; i < cap + 64; i++)
        if (a[i] != 0xEE || b[i] != 0xEE) { fprintf(stderr, "byte %zu behind the stream was touched\n", i); return 6; }
    // an unposted slot refuses a post; a window that is too small is refused before anything is posted
    if (dbde_hip_gather_post(g, 0, win_a, win_a, cap, sizes, 0) != DBDE_HIP_ERR_ARG) return 7;
    // (the DECLARED capacity travels with the counts: "does not fit" is then one verdict on every rank, nothing posted;
    // a window_bytes below what was declared is the caller's error)
    OK(dbde_hip_gather_set_window(g, 16));
    OK(dbde_hip_gather_begin(g, 0, offs + (n - 1), bytes + (n - 1)));
    if (dbde_hip_gather_post(g, 0, win_a, win_a, 16, sizes, 0) != DBDE_HIP_ERR_CAPACITY) return 8;
    OK(dbde_hip_gather_set_window(g, cap));
    OK(dbde_hip_gather_begin(g, 0, offs + (n - 1), bytes + (n - 1)));
    if (dbde_hip_gather_post(g, 0, win_a, win_a, 16, sizes, 0) != DBDE_HIP_ERR_ARG) return 9;
    OK(dbde_hip_gather_begin(g, 0, offs + (n - 1), bytes + (n - 1)));
    OK(dbde_hip_gather_post(g, 0, win_a, win_a, cap, sizes, 0));
    OK(dbde_hip_gather_sync(g, 0));
    if (sizes[0] != total) return 10;
    dbde_hip_gather_destroy(g);
    dbde_hip_destroy(ctx);
    printf("ok\n");
    return 0;
}
