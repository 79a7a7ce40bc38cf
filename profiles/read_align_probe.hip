// read_align_probe.hip -- what the ALIGNMENT of the encoder's pixel loads costs by itself (round 4, config 4: 1921 wide).
// The any-geometry encoder's access shape: a 512-thread workgroup per chunk of 512 tile pairs that never leave a tile row,
// one 16-byte non-temporal load per image row and lane, eight rows per lane.  Same bytes, the address rounded down to
//   MODE 0: nothing (natural position: odd addresses on odd rows)      MODE 1: 2 bytes      MODE 2: 4 bytes (dword)
//   MODE 3: 16 bytes (the lane's aligned block)                        MODE 4: the WAVE's first address rounded to 128 (whole lines per wave)
// (rounded loads fetch slightly different bytes -- the point is the rate and FETCH_SIZE, not the data).
// Build: hipcc --offload-arch=gfx950 -O3 profiles/read_align_probe.hip -o profiles/read_align_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_u;

template <int MODE, bool NT>
__global__ __launch_bounds__(512) void rd(const uint8_t *img, int W, int H, uint32_t lpr, uint32_t h, uint32_t cpf, uint32_t *sink) {
    const uint32_t f = blockIdx.x / cpf, cf = blockIdx.x - f * cpf;
    const uint8_t *base = img + (size_t)f * (size_t)W * (size_t)H;
    const uint32_t pair = cf * 512u + threadIdx.x;
    uint32_t ty = pair / lpr, j = pair - ty * lpr;
    if (ty >= h) { ty = 0; j = 0; }
    uint32_t acc = 0;
    u32x4 q[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int yy = 8 * (int)ty + r;
        yy = yy < H ? yy : H - 1;
        size_t x0 = 16u * j;
        if (x0 + 16 > (size_t)W) x0 = (size_t)W - 16;
        uintptr_t a = reinterpret_cast<uintptr_t>(base + (size_t)yy * (size_t)W + x0);
        if (MODE == 1) a &= ~(uintptr_t)1;
        if (MODE == 2) a &= ~(uintptr_t)3;
        if (MODE == 3) a &= ~(uintptr_t)15;
        if (MODE == 4) {   // shift the whole wave so that its first lane starts a cache line
            const uintptr_t a0 = (uintptr_t)__shfl((unsigned long long)a, 0, 64);
            a -= a0 & 127u;
        }
        q[r] = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_u *>(a)) : *reinterpret_cast<const u32x4_u *>(a);
    }
#pragma unroll
    for (int r = 0; r < 8; r++) acc ^= q[r][0] ^ q[r][1] ^ q[r][2] ^ q[r][3];
    if (acc == 0x12345u) sink[0] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char **argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 1921, H = argc > 2 ? atoi(argv[2]) : 1081;
    const int frames = argc > 3 ? atoi(argv[3]) : 2048;
    const uint32_t w = (W + 7) / 8, h = (H + 7) / 8, lpr = (w + 1) / 2, cpf = (h * lpr + 511) / 512;
    const size_t bytes = (size_t)W * H * frames;
    uint8_t *a;
    uint32_t *sink;
    CK(hipMalloc(&a, bytes + 4096)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, bytes + 4096));
    a += 128;   // (room for the rounded-down first load)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned gb = (unsigned)frames * cpf;
    auto timeit = [&](const char *name, auto launch) {
        launch(); launch();
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 10; i++) launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("W=%d x%d %-28s %8.3f ms  %7.1f GB/s\n", W, frames, name, ms / 10, (double)bytes / (ms / 10 * 1e-3) / 1e9);
        fflush(stdout);
    };
#define R(M, N, S) timeit(S, [&] { hipLaunchKernelGGL((rd<M, N>), dim3(gb), dim3(512), 0, 0, a, W, H, lpr, h, cpf, sink); })
    R(0, true, "natural nt"); R(1, true, "round 2 nt"); R(2, true, "round 4 nt"); R(3, true, "round 16 nt"); R(4, true, "wave at 128 nt");
    R(0, false, "natural plain"); R(2, false, "round 4 plain"); R(3, false, "round 16 plain"); R(4, false, "wave at 128 plain");
    return 0;
}
