// mempattern.hip -- what the codec's access SHAPES cost by themselves (no codec arithmetic).
// One 256-thread workgroup per 32 KB strip (= 8 image rows of 4096 bytes = one chunk of 512 tiles), the
// decoder's launch shape.  Write-only and read-only kernels, same bytes, different order:
//   fill   : instruction i of the workgroup covers bytes [4K*i, 4K*(i+1)) of the strip (what torch.fill_ does)
//   rows   : the codec's shape: a wave instruction covers 1 KB of ONE image row; the wave's 8 instructions
//            walk the 8 rows (stride 4096)
//   wrows  : a wave owns two whole rows (8 instructions x 1 KB, contiguous 8 KB per wave)
// each with plain and non-temporal accesses.  Build: hipcc --offload-arch=gfx950 -O3 profiles/mempattern.hip -o profiles/mempattern
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int PAT, bool NT>
__global__ __launch_bounds__(256) void wr(uint8_t *dst, uint32_t v) {
    uint8_t *base = dst + (size_t)blockIdx.x * 32768u;
    const uint32_t t = threadIdx.x, w = t >> 6, l = t & 63;
    u32x4 q = {v, v + t, v, v};
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) {
        uint32_t off;
        if (PAT == 0) off = (i * 256u + t) * 16u;
        else if (PAT == 1) off = i * 4096u + (w * 64u + l) * 16u;
        else off = (2u * w + (i >> 2)) * 4096u + ((i & 3u) * 64u + l) * 16u;
        u32x4 *p = reinterpret_cast<u32x4 *>(base + off);
        if (NT) __builtin_nontemporal_store(q, p); else *p = q;
    }
}

template <int PAT, bool NT>
__global__ __launch_bounds__(256) void rd(const uint8_t *src, uint32_t *sink) {
    const uint8_t *base = src + (size_t)blockIdx.x * 32768u;
    const uint32_t t = threadIdx.x, w = t >> 6, l = t & 63;
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) {
        uint32_t off;
        if (PAT == 0) off = (i * 256u + t) * 16u;
        else if (PAT == 1) off = i * 4096u + (w * 64u + l) * 16u;
        else off = (2u * w + (i >> 2)) * 4096u + ((i & 3u) * 64u + l) * 16u;
        const u32x4 *p = reinterpret_cast<const u32x4 *>(base + off);
        u32x4 q = NT ? __builtin_nontemporal_load(p) : *p;
        acc ^= q[0] ^ q[1] ^ q[2] ^ q[3];
    }
    if (acc == 0x12345u) sink[0] = acc;
}

// read one buffer, write another (the decoder's mix at noise8: 1 byte read per byte written), rows shape
template <int PAT, bool NT>
__global__ __launch_bounds__(256) void cp(const uint8_t *src, uint8_t *dst) {
    const uint8_t *sb = src + (size_t)blockIdx.x * 32768u;
    uint8_t *db = dst + (size_t)blockIdx.x * 32768u;
    const uint32_t t = threadIdx.x, w = t >> 6, l = t & 63;
    u32x4 q[8];
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) {
        const u32x4 *p = reinterpret_cast<const u32x4 *>(sb + (i * 256u + t) * 16u);
        q[i] = NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) {
        uint32_t off;
        if (PAT == 0) off = (i * 256u + t) * 16u;
        else if (PAT == 1) off = i * 4096u + (w * 64u + l) * 16u;
        else off = (2u * w + (i >> 2)) * 4096u + ((i & 3u) * 64u + l) * 16u;
        u32x4 *p = reinterpret_cast<u32x4 *>(db + off);
        if (NT) __builtin_nontemporal_store(q[i], p); else *p = q[i];
    }
}

// Generic geometry (BASELINE configs[3], 1921x1081: every image row starts at an odd address): the decoder's
// launch shape again (256 threads, 512 consecutive tiles, two per lane), bytes addressed as the codec does.
//   MODE 0: two unaligned 8-byte accesses per image row and lane (one per tile)   <- round-1 kernels
//   MODE 1: one unaligned 16-byte access per row and lane (both tiles when they are neighbours)
//   MODE 2: the same on a 1920-wide, 16-byte aligned image (what alignment alone is worth)
template <int MODE, bool WRITE>
__global__ __launch_bounds__(256) void generic(uint8_t *img, int W, int H, uint32_t w, uint32_t T, uint32_t cpf, uint32_t *sink) {
    const uint32_t f = blockIdx.x / cpf, cf = blockIdx.x - f * cpf;
    uint8_t *base = img + (size_t)f * (size_t)W * (size_t)H;
    const uint32_t t0 = cf * 512u + 2u * threadIdx.x;
    if (t0 >= T) return;
    const uint32_t ty = t0 / w, tx = t0 - ty * w;
    uint32_t acc = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int yy = 8 * (int)ty + r;
        yy = yy < H ? yy : H - 1;
        size_t x0 = 8u * tx;
        if (x0 + 16 > (size_t)W) x0 = (size_t)W - 16;      // stay inside the row (edge fix-up is ALU work)
        uint8_t *p = base + (size_t)yy * (size_t)W + x0;
        if (MODE == 3) {   // WRITE only: lane j stores the ALIGNED block that follows its address (own tail + next lane's head);
                           // lanes that start / end a run (wave edge, image-row edge) store their partial block bytewise
            const uintptr_t a = reinterpret_cast<uintptr_t>(p);
            const uint32_t s = (uint32_t)(a & 15u);
            const uint32_t lane = threadIdx.x & 63u;
            const bool row_start = tx == 0u, row_end = tx + 2u >= w;
            u32x4 v = {t0, t0, t0, t0};
            if (s == 0u) { *reinterpret_cast<u32x4 *>(p) = v; }
            else {
                if (!(lane == 63u || row_end)) *reinterpret_cast<u32x4 *>(p + (16u - s)) = v;       // aligned: a - s + 16
                else for (uint32_t b = 16u - s; b < 16u; b++) p[b] = (uint8_t)t0;                    // tail bytes of the run
                if (lane == 0u || row_start) for (uint32_t b = 0; b < 16u - s; b++) p[b] = (uint8_t)t0;   // head bytes of the run
            }
        } else if (MODE == 0) {
            if (WRITE) { uint64_t v = t0; __builtin_memcpy(p, &v, 8); __builtin_memcpy(p + 8, &v, 8); }
            else { uint64_t a, b; __builtin_memcpy(&a, p, 8); __builtin_memcpy(&b, p + 8, 8); acc ^= (uint32_t)(a ^ b) ^ (uint32_t)((a ^ b) >> 32); }
        } else {
            if (WRITE) { u32x4 v = {t0, t0, t0, t0}; __builtin_memcpy(p, &v, 16); }
            else { u32x4 v; __builtin_memcpy(&v, p, 16); acc ^= v[0] ^ v[1] ^ v[2] ^ v[3]; }
        }
    }
    if (!WRITE && acc == 0x12345u) sink[0] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char **argv) {
    const bool full = argc > 1;   // any argument: also the aligned fill/rows/wrows patterns
    const size_t bytes = 12ull << 30;
    const unsigned blocks = (unsigned)(bytes / 32768);
    uint8_t *a, *b;
    uint32_t *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch, double bytes_moved) {
        launch(); launch();
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 5; i++) launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-22s %8.3f ms  %7.1f GB/s\n", name, ms / 5, bytes_moved / (ms / 5 * 1e-3) / 1e9);
        fflush(stdout);
    };
#define W(P, N) timeit(N ? "write pat" #P " nt" : "write pat" #P " plain", [&] { hipLaunchKernelGGL((wr<P, N>), dim3(blocks), dim3(256), 0, 0, a, 7u); }, (double)bytes)
#define R(P, N) timeit(N ? "read  pat" #P " nt" : "read  pat" #P " plain", [&] { hipLaunchKernelGGL((rd<P, N>), dim3(blocks), dim3(256), 0, 0, a, sink); }, (double)bytes)
#define C(P, N) timeit(N ? "copy  pat" #P " nt" : "copy  pat" #P " plain", [&] { hipLaunchKernelGGL((cp<P, N>), dim3(blocks), dim3(256), 0, 0, a, b); }, 2.0 * bytes)
    if (full) {
    W(0, false); W(0, true); W(1, false); W(1, true); W(2, false); W(2, true);
    R(0, false); R(0, true); R(1, false); R(1, true); R(2, false); R(2, true);
    C(0, false); C(0, true); C(1, false); C(1, true); C(2, false); C(2, true);
    }
    const int widths[] = {1921, 1922, 1924, 1928, 1936, 1920};
    for (int W : widths) {
        const int H = 1081;
        const uint32_t w = (W + 7) / 8, h = (H + 7) / 8, T = w * h, cpf = (T + 511) / 512;
        const int frames = (int)(bytes / ((size_t)W * H)) - 1;
        const double moved = (double)frames * W * H;
        const unsigned gb = (unsigned)frames * cpf;
        char name[64];
        snprintf(name, sizeof name, "W=%d read  2x8B", W);
        timeit(name, [&] { hipLaunchKernelGGL((generic<0, false>), dim3(gb), dim3(256), 0, 0, a, W, H, w, T, cpf, sink); }, moved);
        snprintf(name, sizeof name, "W=%d read  1x16B", W);
        timeit(name, [&] { hipLaunchKernelGGL((generic<1, false>), dim3(gb), dim3(256), 0, 0, a, W, H, w, T, cpf, sink); }, moved);
        snprintf(name, sizeof name, "W=%d write 2x8B", W);
        timeit(name, [&] { hipLaunchKernelGGL((generic<0, true>), dim3(gb), dim3(256), 0, 0, a, W, H, w, T, cpf, sink); }, moved);
        snprintf(name, sizeof name, "W=%d write 1x16B", W);
        timeit(name, [&] { hipLaunchKernelGGL((generic<1, true>), dim3(gb), dim3(256), 0, 0, a, W, H, w, T, cpf, sink); }, moved);
        snprintf(name, sizeof name, "W=%d write aligned blocks + edges", W);
        timeit(name, [&] { hipLaunchKernelGGL((generic<3, true>), dim3(gb), dim3(256), 0, 0, a, W, H, w, T, cpf, sink); }, moved);
    }
    return 0;
}
