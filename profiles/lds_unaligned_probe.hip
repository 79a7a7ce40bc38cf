// lds_unaligned_probe.hip -- does the LDS of gfx950 (ROCm's default "unaligned" access mode) take 8- and 16-byte
// accesses at ANY byte address, and at what price?  Correctness first (every offset 0..15, checked bytewise),
// then a throughput loop: aligned vs odd-address ds_write_b64 / ds_read_b64 / ds_read_b128.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void probe(uint32_t *out) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
    const uint32_t l = threadIdx.x;
    uint32_t bad_w64 = 0, bad_r64 = 0, bad_r128 = 0, bad_w128 = 0;
    for (uint32_t off = 0; off < 16; off++) {
        for (uint32_t i = l; i < 4096; i += 64) lds[i] = 0xEE;
        __syncthreads();
        // each lane writes 8 bytes at 32*l + off
        const uint32_t a = 32u * l + off;
        const uint64_t v = 0x0807060504030201ull + 0x1010101010101010ull * (l & 7);
        asm volatile("ds_write_b64 %0, %1\n s_waitcnt lgkmcnt(0)" ::"v"(a), "v"(v) : "memory");
        __syncthreads();
        for (uint32_t b = 0; b < 32; b++) {
            const uint8_t want = (b >= off && b < off + 8) ? (uint8_t)(v >> (8 * (b - off))) : 0xEE;
            if (lds[32u * l + b] != want) bad_w64 |= 1u << off;
        }
        __syncthreads();
        for (uint32_t i = l; i < 4096; i += 64) lds[i] = (uint8_t)(i * 7 + 3);
        __syncthreads();
        uint64_t r;
        asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
        for (uint32_t b = 0; b < 8; b++) if ((uint8_t)(r >> (8 * b)) != (uint8_t)((a + b) * 7 + 3)) bad_r64 |= 1u << off;
        uint32_t q0, q1, q2, q3;
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        u4 q;
        asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(a) : "memory");
        q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3];
        const uint32_t qq[4] = {q0, q1, q2, q3};
        for (uint32_t b = 0; b < 16; b++) if ((uint8_t)(qq[b >> 2] >> (8 * (b & 3))) != (uint8_t)((a + b) * 7 + 3)) bad_r128 |= 1u << off;
        __syncthreads();
        for (uint32_t i = l; i < 4096; i += 64) lds[i] = 0xEE;
        __syncthreads();
        u4 wv = {0x04030201u + l, 0x08070605u, 0x0c0b0a09u, 0x100f0e0du};
        asm volatile("ds_write_b128 %0, %1\n s_waitcnt lgkmcnt(0)" ::"v"(a), "v"(wv) : "memory");
        __syncthreads();
        for (uint32_t b = 0; b < 32; b++) {
            const uint32_t ww[4] = {wv[0], wv[1], wv[2], wv[3]};
            const uint8_t want = (b >= off && b < off + 16) ? (uint8_t)(ww[(b - off) >> 2] >> (8 * ((b - off) & 3))) : 0xEE;
            if (lds[32u * l + b] != want) bad_w128 |= 1u << off;
        }
        __syncthreads();
    }
    atomicOr(&out[0], bad_w64); atomicOr(&out[1], bad_r64); atomicOr(&out[2], bad_r128); atomicOr(&out[3], bad_w128);
}

template <int OP, int OFF>
__global__ __launch_bounds__(256) void rate(uint32_t *out, int iters) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[65536];
    const uint32_t t = threadIdx.x;
    uint32_t a = (t * 16u + (uint32_t)OFF) & 0xFFFFu;
    uint64_t v = t;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t ak = (((a + 4096u * k) & 0xFFF0u) + (uint32_t)OFF) & 0xFFFFu;
            if (OP == 0) asm volatile("ds_write_b64 %0, %1" ::"v"(ak), "v"(v) : "memory");
            if (OP == 1) { uint64_t r; asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(ak) : "memory"); acc[0] ^= (uint32_t)r; }
            if (OP == 2) { u4 r; asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(ak) : "memory"); acc ^= r; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (acc[0] == 0x1234567u) out[8] = acc[1];
}

int main() {
    uint32_t *d, h[16] = {0};
    hipMalloc(&d, 64); hipMemset(d, 0, 64);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("offsets (bit i = byte offset i) that gave WRONG bytes: ds_write_b64 %04x  ds_read_b64 %04x  ds_read_b128 %04x  ds_write_b128 %04x\n", h[0], h[1], h[2], h[3]);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RATE(OP, OFF, name) { hipLaunchKernelGGL((rate<OP, OFF>), dim3(1024), dim3(256), 0, 0, d, 200); hipDeviceSynchronize(); hipEventRecord(e0, 0); \
    hipLaunchKernelGGL((rate<OP, OFF>), dim3(1024), dim3(256), 0, 0, d, 2000); hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
    printf("%-28s %8.3f ms\n", name, ms); }
    RATE(0, 0, "ds_write_b64 aligned"); RATE(0, 1, "ds_write_b64 +1"); RATE(0, 4, "ds_write_b64 +4"); RATE(0, 5, "ds_write_b64 +5");
    RATE(0, 8, "ds_write_b64 +8"); RATE(0, 9, "ds_write_b64 +9 (crosses 16)"); RATE(0, 12, "ds_write_b64 +12 (crosses 16)"); RATE(0, 15, "ds_write_b64 +15");
    RATE(1, 0, "ds_read_b64 aligned"); RATE(1, 1, "ds_read_b64 +1"); RATE(1, 4, "ds_read_b64 +4"); RATE(1, 9, "ds_read_b64 +9 (crosses 16)"); RATE(1, 12, "ds_read_b64 +12");
    RATE(2, 0, "ds_read_b128 aligned"); RATE(2, 1, "ds_read_b128 +1"); RATE(2, 8, "ds_read_b128 +8"); RATE(2, 4, "ds_read_b128 +4");
    return 0;
}
