/* A plain C99 client of include/dbde_hip.h: what a C or FFI caller of the reference's
 * dbde_pack_frame / dbde_unpack_frame (dbde_util.h:26,35) does after switching libraries.
 * Built with gcc (no HIP headers needed), linked against libdbde_hip.so only.
 *
 *   roundtrip <image hex> <packed hex>
 *       packs the 10x10 image (100 bytes, given as hex: the README example from tests/golden) as
 *       frame 7, checks it against the 112 golden bytes (SURVEY 8c golden 2), unpacks, checks the
 *       pixels; then a ramp frame of every size 1x40 .. 40x1 round-trips.  Exit code 0 = all good.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dbde_hip.h"

static int unhex(const char *hex, uint8_t *out, size_t n) {
    if (strlen(hex) != 2 * n) return 0;
    for (size_t i = 0; i < n; i++) {
        unsigned v = 0;
        if (sscanf(hex + 2 * i, "%2x", &v) != 1) return 0;
        out[i] = (uint8_t)v;
    }
    return 1;
}

int main(int argc, char **argv) {
    dbde_hip_ctx *ctx = NULL;
    int rc = dbde_hip_create(0, NULL, &ctx);
    if (rc != DBDE_HIP_OK) {
        fprintf(stderr, "dbde_hip_create failed (%d): needs a gfx950 device\n", rc);
        return 2;
    }
    int bad = 0;
    /* 1. README image against the golden bytes */
    {
        uint8_t readme_image[100], golden[112], packed[512], back[100];
        if (argc < 3 || !unhex(argv[1], readme_image, 100) || !unhex(argv[2], golden, 112)) {
            fprintf(stderr, "usage: roundtrip <100-byte image as hex> <112-byte packed frame as hex>\n");
            return 2;
        }
        memset(packed, 0xEE, sizeof packed);
        size_t n = dbde_hip_pack_frame(ctx, 7, readme_image, 10, 10, packed);
        if (n != 112) { fprintf(stderr, "README frame: %zu bytes, expected 112\n", n); bad++; }
        if (packed[112] != 0xEE) { fprintf(stderr, "wrote past the returned size\n"); bad++; }
        if (memcmp(packed, golden, 112) != 0) { fprintf(stderr, "README frame: bytes differ from the golden vector\n"); bad++; }
        uint8_t *cur = packed;
        memset(back, 0, sizeof back);
        dbde_hip_frame_header fh = dbde_hip_unpack_frame(ctx, &cur, 10, 10, back);
        if (fh.u64s != 2 || fh.index != 7 || fh.elapsed_ns != 0 || (size_t)(cur - packed) != n) {
            fprintf(stderr, "README frame: header/cursor wrong\n");
            bad++;
        }
        if (memcmp(back, readme_image, 100) != 0) { fprintf(stderr, "README frame: pixels differ\n"); bad++; }
    }
    /* 2. every small size */
    for (int s = 1; s <= 40 && !bad; s++) {
        int W = s, H = 41 - s;
        size_t P = (size_t)W * (size_t)H, cap = dbde_hip_max_frame_bytes(W, H);
        uint8_t *img = malloc(P), *back = malloc(P), *packed = malloc(cap + 16);
        for (size_t i = 0; i < P; i++) img[i] = (uint8_t)(100 + (i * 7 + (i / (size_t)W) * 13) % (size_t)(1 + s * 3));
        size_t n = dbde_hip_pack_frame(ctx, (uint64_t)s, img, W, H, packed);
        uint8_t *cur = packed;
        dbde_hip_frame_header fh = dbde_hip_unpack_frame(ctx, &cur, W, H, back);
        if (n == 0 || n > cap || fh.u64s != 2 || fh.index != (uint64_t)s || (size_t)(cur - packed) != n ||
            memcmp(img, back, P) != 0) {
            fprintf(stderr, "%dx%d: round trip failed\n", W, H);
            bad++;
        }
        free(img); free(back); free(packed);
    }
    if (dbde_hip_sync(ctx) != DBDE_HIP_OK) { fprintf(stderr, "sync: %s\n", dbde_hip_last_error(ctx)); bad++; }
    dbde_hip_destroy(ctx);
    printf(bad ? "FAILED\n" : "ok\n");
    return bad ? 1 : 0;
}
