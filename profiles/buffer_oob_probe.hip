// Probe (hipcc --offload-arch=gfx950 -O2 buffer_oob_probe.hip): how MI355X range-checks raw buffer stores.
// Result on the pool (2026-10): buffer_store_dwordx4 is checked PER DWORD against num_records, at any 4-byte
// aligned base -- a 16-byte store that straddles the end writes exactly the in-range dwords:
//   num_records=24: lanes 0,1 -> bytes [0,24) written, nothing else;   num_records=20 -> [0,20).
// (Kept for the next round: a branch-free, static-count payload store in the encoder can rely on it.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned char *p, int n, int shift) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p + shift, 0, n, 0x00020000);
    u32x4 v; v[0] = 0x11111111u * (threadIdx.x + 1); v[1] = v[0]; v[2] = v[0]; v[3] = v[0];
    __builtin_amdgcn_raw_buffer_store_b128(v, r, threadIdx.x * 16, 0, 2);
}
int main() {
    unsigned char *d; hipMalloc(&d, 256);
    for (int shift : {0, 8}) for (int n : {24, 16, 20, 40}) {
        hipMemset(d, 0xEE, 256);
        hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, d, n, shift);
        unsigned char h[96]; hipMemcpy(h, d, 96, hipMemcpyDeviceToHost);
        printf("shift=%d num_records=%d: ", shift, n);
        for (int i = 0; i < 80; i += 4) printf("%02x", h[i]);
        printf("\n");
    }
    return 0;
}
