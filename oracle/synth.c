/*
 * synth.c -- CPU statement of the synthetic frame generators.
 * TEST INFRASTRUCTURE ONLY.  The HIP generator dbde_hip_synth_frames must produce the same
 * bytes (tests/test_gpu_parity.py::test_synth_matches_oracle); bench.py and the parity
 * tests rely on that to build identical inputs on the CPU and on the GPU.
 *
 * Counter-based: every pixel is a pure function of (mode, seed, frame, y, x), so frames can
 * be produced in any order and on any device.
 *
 *   mix64(z)      : splitmix64 finaliser.
 *   rowkey        : mix64(seed ^ frame<<42 ^ y<<21 ^ (x>>3))      -> 8 random bytes for 8 pixels
 *   tilekey       : mix64(~seed ^ frame<<42 ^ ty<<21 ^ tx)        -> per-tile depth and minimum
 *
 *   mode 0 noise8 : pix = random byte                              (every tile depth 8)
 *   mode 1 mixed  : d = tilekey % 9, m = (tilekey>>32) % (257 - 2^d),
 *                   pix = m + (random byte & (2^d - 1)); tile-local pixel (0,0) forced to m and
 *                   (0,1) to m + 2^d - 1 so a full tile realises exactly depth d
 *                   (uniform depth histogram 0..8, packed/raw ~ 0.53)
 *   mode 2 flat   : pix = seed & 0xFF                              (every tile depth 0)
 *   mode 3 smooth : horizontal+vertical ramp plus 3 bits of noise  (low depths, camera-like)
 */
#include "dbde_oracle.h"

static uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void dbde_oracle_synth_frame(int mode, uint64_t seed, uint64_t frame, int W, int H, uint8_t *image) {
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            uint64_t rk = mix64(seed ^ (frame << 42) ^ ((uint64_t)y << 21) ^ (uint64_t)(x >> 3));
            unsigned rnd = (unsigned)(rk >> (8 * (x & 7))) & 0xFFu;
            unsigned pix;
            if (mode == 0) {
                pix = rnd;
            } else if (mode == 1) {
                uint64_t tk = mix64(~seed ^ (frame << 42) ^ ((uint64_t)(y >> 3) << 21) ^ (uint64_t)(x >> 3));
                unsigned d = (unsigned)(tk % 9u);
                unsigned m = (unsigned)((tk >> 32) % (257u - (1u << d)));
                unsigned top = (1u << d) - 1u;
                pix = m + (rnd & top);
                if ((y & 7) == 0 && (x & 7) == 0) pix = m;
                if ((y & 7) == 0 && (x & 7) == 1) pix = m + top;
            } else if (mode == 2) {
                pix = (unsigned)(seed & 0xFFu);
            } else {
                pix = (((unsigned)x >> 4) + ((unsigned)y >> 5) + (unsigned)(frame & 15u) + (rnd & 7u)) & 0xFFu;
            }
            image[(size_t)y * W + x] = (uint8_t)pix;
        }
    }
}
