// valu_rate_probe.hip -- issue cost of the 64-bit shifts the encoder's bit funnel is made of, against 32-bit ALU ops
// (gfx950; one workgroup of 256 threads per CU x 4, dependent chains x 8 independent streams per lane).
// Build: hipcc --offload-arch=gfx950 -O3 profiles/valu_rate_probe.hip -o profiles/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t *out, uint32_t s, int iters) {
    uint64_t a[8];
    uint32_t b[8], c[8];
    const uint64_t mask = __builtin_amdgcn_read_exec() ^ (uint64_t)s;
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 0x9E3779B97F4A7C15ull + i; b[i] = threadIdx.x * 2654435761u + i; c[i] = b[i] * 3u; }
    if (MODE == 16 || MODE == 21) asm volatile("s_mov_b64 vcc, %0" ::"s"(mask) : "vcc");
    for (int it = 0; it < iters; it++) {
        if (MODE == 19) asm volatile("v_cmp_gt_u32_e32 vcc, %1, %0" :: "v"(b[0]), "v"(s) : "vcc");
        if (MODE == 20) asm volatile("s_mov_b64 vcc, %0" ::"s"(mask) : "vcc");
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(a[i]) : "v"(s));
            if (MODE == 1) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(a[i]) : "v"(s));
            if (MODE == 2) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(b[i]) : "v"(s));
            if (MODE == 3) asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(b[i]) : "v"(s));
            if (MODE == 4) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(b[i]) : "v"(s));
            if (MODE == 5) asm volatile("v_lshl_add_u64 %0, %0, 0, %0" : "+v"(a[i]));
            if (MODE == 6) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(a[i]) : "v"(s) : "vcc");
            if (MODE == 7) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(b[i]) : "v"(s));
            if (MODE == 8) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(s), "s"(mask));
            if (MODE == 12) asm volatile("v_lshlrev_b32 %0, %1, %0\n s_nop 0" : "+v"(b[i]) : "v"(s));
            if (MODE == 13) asm volatile("v_cndmask_b32 %0, %0, %1, %2\n s_nop 0" : "+v"(b[i]) : "v"(s), "s"(mask));
            if (MODE == 16) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(b[i]) : "v"(s));
            if (MODE == 17) asm volatile("v_cmp_gt_u32_e32 vcc, %1, %0\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(b[i]) : "v"(s) : "vcc");
            if (MODE == 18) asm volatile("v_cmp_gt_u32_e64 %2, %1, %0\n v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(b[i]) : "v"(s), "s"(mask));
            if (MODE == 19 || MODE == 20) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(b[i]) : "v"(s));
            if (MODE == 21) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(b[i]) : "v"(s));
            if (MODE == 22) asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %1, vcc" : "+v"(b[i]) : "v"(s) : "vcc");
            if (MODE == 24) asm volatile("v_cmp_gt_u32_e32 vcc, %2, %0\n v_cndmask_b32_e32 %0, %0, %2, vcc\n v_cndmask_b32_e32 %1, %1, %2, vcc" : "+v"(b[i]), "+v"(c[i]) : "v"(s) : "vcc");
            if (MODE == 25) asm volatile("v_cmp_gt_u32_e64 %3, %2, %0\n v_cndmask_b32_e64 %0, %0, %2, %3\n v_cndmask_b32_e64 %1, %1, %2, %3" : "+v"(b[i]), "+v"(c[i]) : "v"(s), "s"(mask));
            if (MODE == 26) asm volatile("v_cmp_gt_u32_e32 vcc, %2, %0\n v_cndmask_b32_e64 %0, %0, %2, vcc\n v_cndmask_b32_e64 %1, %1, %2, vcc" : "+v"(b[i]), "+v"(c[i]) : "v"(s) : "vcc");
            if (MODE == 14) asm volatile("v_bfi_b32 %0, %1, %0, %0" : "+v"(b[i]) : "v"(s));
            if (MODE == 15) asm volatile("v_add_co_u32 %0, %2, %0, %1" : "+v"(b[i]) : "v"(s), "s"(mask));
            if (MODE == 9) asm volatile("v_perm_b32 %0, %0, %1, %0" : "+v"(b[i]) : "v"(s));
            if (MODE == 10) asm volatile("v_lshl_or_b32 %0, %0, %1, %0" : "+v"(b[i]) : "v"(s));
            if (MODE == 11) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(b[i]) : "v"(s));
        }
    }
    uint64_t x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x ^= a[i] ^ b[i] ^ c[i];
    if (x == 0x1234567) out[0] = x;
}

int main() {
    uint64_t *out;
    hipMalloc(&out, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, grid = 256 * 8;
    const char *names[] = {"v_lshlrev_b64", "v_lshrrev_b64", "v_lshlrev_b32", "v_alignbit_b32", "v_dot4_u32_u8", "v_lshl_add_u64", "v_mad_u64_u32", "v_pk_min_u16", "v_cndmask_b32", "v_perm_b32", "v_lshl_or_b32", "v_mul_lo_u32", "lshl_b32 + s_nop", "cndmask + s_nop", "v_bfi_b32", "v_add_co_u32", "cndmask e32 vcc", "cmp+cndmask vcc (2 instr)", "cmp+cndmask sgpr (2 instr)", "v_cmp, 8 cndmask e32", "s_mov vcc, 8 cndmask e32", "cndmask e64 vcc", "v_addc_co chain", "", "cmp + 2 cndmask e32 (3 instr)", "cmp_e64 + 2 cndmask e64 sgpr (3 instr)", "cmp_e32 + 2 cndmask e64 vcc (3 instr)"};
    float base = 0;
#define RUN(M) { hipLaunchKernelGGL(k<M>, dim3(grid), dim3(256), 0, 0, out, 3u, 16); hipDeviceSynchronize(); \
    hipEventRecord(e0, 0); hipLaunchKernelGGL(k<M>, dim3(grid), dim3(256), 0, 0, out, 3u, iters); hipEventRecord(e1, 0); hipEventSynchronize(e1); \
    float ms; hipEventElapsedTime(&ms, e0, e1); if (M == 2) base = ms; printf("%-16s %8.3f ms\n", names[M], ms); }
    RUN(2) RUN(0) RUN(1) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(16) RUN(17) RUN(18) RUN(19) RUN(20) RUN(21) RUN(22) RUN(24) RUN(25) RUN(26)
    printf("(relative to v_lshlrev_b32 = %.3f ms)\n", base);
    return 0;
}
