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

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main() {
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
    W(0, false); W(0, true); W(1, false); W(1, true); W(2, false); W(2, true);
    R(0, false); R(0, true); R(1, false); R(1, true); R(2, false); R(2, true);
    C(0, false); C(0, true); C(1, false); C(1, true); C(2, false); C(2, true);
    return 0;
}
