// abbench.cpp -- A/B timing of libdbde_hip.so builds through the C-ABI, without Python (no torch import:
// a variant costs seconds, so one GPU call can rank many).  Build: hipcc --offload-arch=gfx950 -O2
// profiles/abbench.cpp -o profiles/abbench -ldl.  Usage:
//   abbench <libdbde_hip.so> <W> <H> <frames> <noise8|mixed|flat|smooth|depthN> <slots|concat> <steps> [tag]
// Prints one JSON line: kernel times from the library's own HIP-event hook, round trip verified on the device.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
    } while (0)

__global__ void count_diff(const uint4 *a, const uint4 *b, size_t n16, unsigned long long *out) {
    unsigned long long d = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        uint4 x = a[i], y = b[i];
        d += (x.x != y.x) + (x.y != y.y) + (x.z != y.z) + (x.w != y.w);
    }
    if (d) atomicAdd(out, d);
}

typedef struct ctx ctx;
int main(int argc, char **argv) {
    if (argc < 8) { fprintf(stderr, "usage: abbench lib W H frames content layout steps [tag]\n"); return 1; }
    const char *libpath = argv[1];
    const int W = atoi(argv[2]), H = atoi(argv[3]), B = atoi(argv[4]), steps = atoi(argv[7]);
    const char *content = argv[5];
    const bool slots = strcmp(argv[6], "slots") == 0;
    const char *tag = argc > 8 ? argv[8] : libpath;
    int mode = !strcmp(content, "noise8") ? 0 : !strcmp(content, "mixed") ? 1 : !strcmp(content, "flat") ? 2 : 3;
    if (!strncmp(content, "depth", 5)) mode = 4 + atoi(content + 5);   // every tile of that depth
    void *h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    using fn_dbde_hip_create = int (*)(int, void *, ctx **);
    using fn_dbde_hip_destroy = void (*)(ctx *);
    using fn_dbde_hip_sync = int (*)(ctx *);
    using fn_dbde_hip_last_error = const char *(*)(ctx *);
    using fn_dbde_hip_max_frame_bytes = size_t (*)(int, int);
    using fn_dbde_hip_synth_frames = int (*)(ctx *, int, uint64_t, uint64_t, int, int, int, uint8_t *);
    using fn_dbde_hip_encode_frames = int (*)(ctx *, const uint8_t *, int, int, int, uint64_t, const uint64_t *,
                                              const uint64_t *, uint8_t *, size_t, uint64_t, uint64_t *, uint64_t *);
    using fn_dbde_hip_decode_frames = int (*)(ctx *, const uint8_t *, size_t, const uint64_t *, int, int, int, uint8_t *, void *);
    using fn_dbde_hip_timing_enable = int (*)(ctx *, int);
    using fn_dbde_hip_timing_read = int (*)(ctx *, double *, uint64_t *, int);
#define SYM(name) fn_##name name = (fn_##name)dlsym(h, #name); if (!name) { fprintf(stderr, "missing %s\n", #name); return 1; }
    SYM(dbde_hip_create) SYM(dbde_hip_destroy) SYM(dbde_hip_sync) SYM(dbde_hip_last_error) SYM(dbde_hip_max_frame_bytes)
    SYM(dbde_hip_synth_frames) SYM(dbde_hip_encode_frames) SYM(dbde_hip_decode_frames) SYM(dbde_hip_timing_enable)
    SYM(dbde_hip_timing_read)
    hipStream_t s;
    CK(hipStreamCreate(&s));
    ctx *c = nullptr;
    if (dbde_hip_create(0, s, &c) != 0) { fprintf(stderr, "create failed\n"); return 1; }
    const size_t px = (size_t)W * H, maxf = dbde_hip_max_frame_bytes(W, H);
    const size_t slot = slots ? (maxf + 255) / 256 * 256 : 0;
    const size_t cap = slots ? (size_t)(B - 1) * slot + maxf : (size_t)B * maxf;
    uint8_t *img, *out, *buf;
    uint64_t *offs, *sizes;
    unsigned long long *diff;
    CK(hipMalloc(&img, px * B + 64));
    CK(hipMalloc(&out, px * B + 64));
    CK(hipMalloc(&buf, cap + 128));
    CK(hipMalloc(&offs, 8 * (size_t)B));
    CK(hipMalloc(&sizes, 8 * (size_t)B));
    CK(hipMalloc(&diff, 8));
    CK(hipMemsetAsync(diff, 0, 8, s));
    CK(hipMemsetAsync(out, 0xEE, px * B, s));
    if (dbde_hip_synth_frames(c, mode, 0xDBDE2016ull, 0, B, W, H, img)) { fprintf(stderr, "synth: %s\n", dbde_hip_last_error(c)); return 1; }
    const char *only = getenv("ABBENCH_ONLY");   // "enc" / "dec": the timed loop runs one direction only (the other ran in the warm-up)
    bool timed = false;
    auto step = [&]() -> int {
        if (!(timed && only && !strcmp(only, "dec")))
            if (dbde_hip_encode_frames(c, img, W, H, B, 0, nullptr, nullptr, buf + 32, cap, slot, offs, sizes)) return 1;
        if (!(timed && only && !strcmp(only, "enc")))
            if (dbde_hip_decode_frames(c, buf + 32, cap, offs, W, H, B, out, nullptr)) return 1;
        return 0;
    };
    for (int i = 0; i < 2; i++) if (step()) { fprintf(stderr, "step: %s\n", dbde_hip_last_error(c)); return 1; }
    if (dbde_hip_sync(c)) { fprintf(stderr, "sync: %s\n", dbde_hip_last_error(c)); return 1; }
    hipLaunchKernelGGL(count_diff, dim3(2048), dim3(256), 0, s, (const uint4 *)img, (const uint4 *)out, px * B / 16, diff);
    unsigned long long hdiff = 0;
    CK(hipMemcpyAsync(&hdiff, diff, 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    uint64_t *hs = (uint64_t *)malloc(8 * (size_t)B);
    CK(hipMemcpy(hs, sizes, 8 * (size_t)B, hipMemcpyDeviceToHost));
    double packed = 0;
    for (int i = 0; i < B; i++) packed += (double)hs[i];
    if (!getenv("ABBENCH_NOTIMING")) dbde_hip_timing_enable(c, 1);   // (the event pairs cost a few us per call: off for latency runs)
    if (getenv("ABBENCH_SFDIAG")) {   // -DDBDE_DIAG builds, small launches (one frame per call): wave 0's timeline of ONE
        // encode launch and ONE decode launch, averaged over the launch's workgroups, relative to the earliest start
        using fn_diag = int (*)(ctx *, uint64_t *);
        fn_diag dr = (fn_diag)dlsym(h, "dbde_hip_diag_read");
        if (!dr) { fprintf(stderr, "no dbde_hip_diag_read\n"); return 1; }
        dbde_hip_timing_enable(c, 0);
        for (int i = 0; i < 20; i++) if (step()) return 1;
        if (dbde_hip_sync(c)) return 1;
        uint64_t d[16];
        dr(c, d);
        auto show = [&](const char *what, const char *const *names) {
            if (!d[10]) { fprintf(stderr, "sfdiag %s: no record (not a -DDBDE_DIAG build, or not the small / fused kernels)\n", what); return; }
            const double n = (double)d[10], t0 = (double)(~d[0]);
            fprintf(stderr, "sfdiag %s: %llu workgroups, start spread %.2f us (mean start +%.2f), last end +%.2f us; wave 0 reaches (mean, us after the first start):",
                    what, (unsigned long long)d[10], ((double)d[12] - t0) / 100.0, ((double)d[13] / n - t0) / 100.0, ((double)d[11] - t0) / 100.0);
            for (int i = 1; i < 10; i++) fprintf(stderr, " %s %.2f", names[i], ((double)d[i] / n - t0) / 100.0);
            fprintf(stderr, "\n");
        };
        static const char *en[10] = {"", "ticket", "pixels", "stats+AGG", "packed", "prefix", "barrier", "stores issued", "stores retired", "cleanup"};
        static const char *dn[10] = {"", "depths summed", "records read", "validated", "DMA issued", "DMA landed", "scan", "unpacked", "stores issued", "stores retired"};
        for (int rep = 0; rep < 3; rep++) {
            if (dbde_hip_encode_frames(c, img, W, H, B, 0, nullptr, nullptr, buf + 32, cap, slot, offs, sizes)) return 1;
            if (dbde_hip_sync(c)) return 1;
            dr(c, d);
            show("encode", en);
            if (dbde_hip_decode_frames(c, buf + 32, cap, offs, W, H, B, out, nullptr)) return 1;
            if (dbde_hip_sync(c)) return 1;
            dr(c, d);
            show("decode", dn);
        }
        return 0;
    }
    double ms[4] = {0, 0, 0, 0};
    uint64_t n[4] = {0, 0, 0, 0};
    dbde_hip_timing_read(c, ms, n, 1);
    if (getenv("ABBENCH_DIAG")) {   // counters of the warm-up launches: read and dropped
        using fn_diag = int (*)(ctx *, uint64_t *);
        fn_diag dr = (fn_diag)dlsym(h, "dbde_hip_diag_read");
        uint64_t junk[16];
        if (dr) dr(c, junk);
        using fn_trace0 = int (*)(ctx *, uint64_t *, size_t);
        fn_trace0 trd0 = (fn_trace0)dlsym(h, "dbde_hip_diag_trace_read");
        static uint64_t junk_tr[16 * 1024];
        if (trd0) trd0(c, junk_tr, 16 * 1024);
    }
    timed = true;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; i++) if (step()) return 1;
    if (dbde_hip_sync(c)) { fprintf(stderr, "sync: %s\n", dbde_hip_last_error(c)); return 1; }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ms[3] = 0; n[3] = 0;
    dbde_hip_timing_read(c, ms, n, 1);
    if (getenv("ABBENCH_DIAG")) {   // -DDBDE_DIAG builds: in-kernel cycle counters of the encoder (sums over the timed steps)
        using fn_diag = int (*)(ctx *, uint64_t *);
        fn_diag dr = (fn_diag)dlsym(h, "dbde_hip_diag_read");
        uint64_t d[16] = {0};
        if (dr && dr(c, d) == 0 && d[8]) {
            const double wg = (double)d[8];   // workgroup runs summed over steps
            fprintf(stderr, "diag[%s %s]: per workgroup-run: kernel %.0f cyc, wait_inc %.0f cyc (%.1f %%, %.1f waits, %.2f polls each, %.0f %% answered at the first poll), wave0 at barrier %.0f cyc (%.1f %%) | "
                    "scanner per launch: %.0f rounds, %.1f %% idle, %.1f records/round, %.0f cyc\n", tag, content,
                    d[2] / wg, d[0] / wg, 100.0 * d[0] / d[2], d[1] / wg, (double)d[9] / (d[1] ? d[1] : 1), 100.0 * d[10] / (d[1] ? d[1] : 1), d[7] / wg, 100.0 * d[7] / d[2],
                    (double)d[3] / steps, 100.0 * d[4] / (d[3] ? d[3] : 1), (double)d[6] / (d[3] ? d[3] : 1), (double)d[5] / steps);
            using fn_trace = int (*)(ctx *, uint64_t *, size_t);
            fn_trace trd = (fn_trace)dlsym(h, "dbde_hip_diag_trace_read");
            if (trd && steps == 1) {   // per-workgroup timeline of the one timed launch: min / mean / max of each point, us after the first entry
                static uint64_t tr[16 * 1024];
                if (trd(c, tr, 16 * 1024) == 0) {
                    uint64_t t0 = ~0ull; int n = 0;
                    for (int g = 0; g < 1024; g++) if (tr[16 * g + 9]) { n++; if (tr[16 * g] < t0) t0 = tr[16 * g]; }
                    static const char *nm[10] = {"entry", "mode agreed", "step 0 done", "step 1 done", "step 2 done", "step 3 done", "arrival number back", "poll starts", "last prefetch", "exit"};
                    fprintf(stderr, "trace[%s]: %d encoding workgroups;", tag, n);
                    for (int i = 0; i < 10; i++) {
                        double mn = 1e30, mx = 0, sum = 0; int k = 0;
                        for (int g = 0; g < 1024; g++) if (tr[16 * g + 9] && tr[16 * g + i]) { const double v = (double)(tr[16 * g + i] - t0) / 100.0; mn = v < mn ? v : mn; mx = v > mx ? v : mx; sum += v; k++; }
                        if (k) fprintf(stderr, " %s %.1f/%.1f/%.1f", nm[i], mn, sum / k, mx);
                    }
                    double smn = 1e30, smx = 0, cmn = 1e30, cmx = 0;
                    for (int g = 0; g < 1024; g++) if (tr[16 * g + 9]) { const double a = (double)tr[16 * g + 10], b = (double)tr[16 * g + 11]; smn = a < smn ? a : smn; smx = a > smx ? a : smx; cmn = b < cmn ? b : cmn; cmx = b > cmx ? b : cmx; }
                    fprintf(stderr, " | steps %.0f..%.0f, chunks %.0f..%.0f per workgroup (min/mean/max us)\n", smn, smx, cmn, cmx);
                }
            }
            if (d[11] && steps == 1) {   // one timed launch: when its workgroups started and the last one left (10 ns wall clock)
                const double t0 = (double)(~d[11]);
                fprintf(stderr, "diag[%s]: workgroups start within %.1f us, the last leaves %.1f us after the first start; a pair of steps takes "
                        "%.2f us early in the launch (pairs 2-9), %.2f us later (eight pairs from -DDBDE_DIAG_LATE on, default 34; wall clock)\n", tag,
                        ((double)d[12] - t0) / 100.0, ((double)d[14] - t0) / 100.0, d[15] / wg / 8.0 / 100.0, d[13] / wg / 8.0 / 100.0);
            }
        }
    }
    const double alg = (double)px * B + packed;
    const double enc = ms[0] / (n[0] ? n[0] : 1), idx = ms[1] / (n[1] ? n[1] : 1), dec = ms[2] / (n[2] ? n[2] : 1);
    printf("{\"tag\": \"%s\", \"W\": %d, \"H\": %d, \"frames\": %d, \"content\": \"%s\", \"layout\": \"%s\", \"steps\": %d, "
           "\"enc_ms\": %.4f, \"idx_ms\": %.4f, \"dec_ms\": %.4f, \"enc_frac\": %.4f, \"dec_frac\": %.4f, "
           "\"wall_ms_per_step\": %.4f, \"fps\": %.1f, \"packed_over_raw\": %.4f, \"diff_dwords\": %llu}\n",
           tag, W, H, B, content, slots ? "slots" : "concat", steps, enc, idx, dec, alg / (enc * 1e-3) / 8e12,
           alg / (dec * 1e-3) / 8e12, wall / steps * 1e3, B * steps / wall, packed / ((double)px * B), hdiff);
    fflush(stdout);
    dbde_hip_destroy(c);
    return hdiff ? 3 : 0;
}
