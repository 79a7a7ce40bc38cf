// threads.cpp -- a C++ caller of the reference's API (dbde_util.h) from N threads at once, each on its own buffers: the
// reference has no global state and is re-entrant (dbde_util.h:21-37), and so must its drop-in be.  Links against
// libdbde_util_hip.so only (no HIP headers).  Every thread packs ITS frame and unpacks it again, `reps` times; the packed
// bytes are compared with what the REAL reference produced for the same frame (expected file, made by the caller of this
// program from oracle/_ref), the unpacked image with the input.
//   threads W H n_threads reps frames.bin expected.bin
//   frames.bin  : n_threads raw frames of W*H bytes;  expected.bin: per frame, u64 length + that many packed bytes
// Prints one JSON line: round trips per second over all threads (wall clock around the timed loop), mismatches.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "dbde_util.h"

int main(int argc, char **argv) {
    if (argc < 7) { fprintf(stderr, "usage: threads W H n_threads reps frames.bin expected.bin\n"); return 1; }
    const int W = atoi(argv[1]), H = atoi(argv[2]), nt = atoi(argv[3]), reps = atoi(argv[4]);
    const size_t px = (size_t)W * H, T = (size_t)((W + 7) / 8) * ((H + 7) / 8), maxf = 32 + 66 * T;
    std::vector<std::vector<uint8_t>> img(nt), want(nt), out(nt), back(nt);
    FILE *f = fopen(argv[5], "rb"), *e = fopen(argv[6], "rb");
    if (!f || !e) { fprintf(stderr, "cannot open the input files\n"); return 1; }
    for (int t = 0; t < nt; t++) {
        img[t].resize(px);
        if (fread(img[t].data(), 1, px, f) != px) { fprintf(stderr, "frames.bin too short\n"); return 1; }
        uint64_t n = 0;
        if (fread(&n, 8, 1, e) != 1 || n > maxf) { fprintf(stderr, "expected.bin: bad length\n"); return 1; }
        want[t].resize(n);
        if (fread(want[t].data(), 1, n, e) != n) { fprintf(stderr, "expected.bin too short\n"); return 1; }
        out[t].assign(maxf + 64, 0xEE);
        back[t].assign(px, 0xEE);
    }
    fclose(f); fclose(e);
    std::atomic<int> bad{0}, go{0};
    // THREADS_CHECK_LAST_ONLY=1 (throughput runs): bytes are compared on a thread's last repetition only -- the two
    // memcmp of a 4096x3072 round trip cost as much as the round trip itself; sizes and headers are always checked
    const bool last_only = getenv("THREADS_CHECK_LAST_ONLY") && atoi(getenv("THREADS_CHECK_LAST_ONLY"));
    auto work = [&](int t, int n_rep) {
        for (int r = 0; r < n_rep; r++) {
            const bool cmp = !last_only || r == n_rep - 1;
            const size_t n = dbde_pack_frame(1000 + t, img[t].data(), W, H, out[t].data());
            if (n != want[t].size() || (cmp && memcmp(out[t].data(), want[t].data(), n) != 0)) bad++;
            uint8_t *cur = out[t].data();
            const frame_header fh = dbde_unpack_frame(&cur, W, H, back[t].data());
            if (fh.u64s != 2 || fh.index != (uint64_t)(1000 + t) || (size_t)(cur - out[t].data()) != n ||
                (cmp && memcmp(back[t].data(), img[t].data(), px) != 0)) bad++;
        }
    };
    {   // warm-up: every thread once (contexts, staging buffers), not timed
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back(work, t, 1);
        for (auto &x : th) x.join();
    }
    for (int t = 0; t < nt; t++)      // nothing behind the frame was touched
        for (size_t i = want[t].size(); i < out[t].size(); i++) if (out[t][i] != 0xEE) { bad++; break; }
    const auto t0 = std::chrono::steady_clock::now();
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back(work, t, reps);
        for (auto &x : th) x.join();
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    (void)go;
    printf("{\"W\": %d, \"H\": %d, \"threads\": %d, \"reps\": %d, \"round_trips_per_s\": %.1f, \"seconds\": %.3f, "
           "\"mismatches\": %d, \"packed_bytes_thread0\": %zu}\n", W, H, nt, reps, nt * reps / dt, dt, bad.load(), want[0].size());
    return bad.load() ? 3 : 0;
}
