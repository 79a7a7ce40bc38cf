// pcie_probe.cpp -- what the host link of this box carries: H2D alone, D2H alone, both at once; pinned and pageable host
// memory; 12.6 MB transfers (one 4096x3072 frame).  The ceiling the drop-in shim's host-pointer API is measured against.
// Build: hipcc --offload-arch=gfx950 -O2 profiles/pcie_probe.cpp -o profiles/pcie_probe -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double run(bool pinned, int n_h2d, int n_d2h, size_t bytes, int reps) {
    const int n = n_h2d + n_d2h;
    std::vector<void *> h(n), d(n);
    std::vector<hipStream_t> s(n);
    for (int i = 0; i < n; i++) {
        if (pinned) (void)hipHostMalloc(&h[i], bytes, hipHostMallocDefault); else { h[i] = malloc(bytes); memset(h[i], 1, bytes); }
        (void)hipMalloc(&d[i], bytes);
        (void)hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
    }
    auto work = [&](int i) {
        (void)hipSetDevice(0);
        for (int r = 0; r < reps; r++) {
            if (i < n_h2d) (void)hipMemcpyAsync(d[i], h[i], bytes, hipMemcpyHostToDevice, s[i]);
            else (void)hipMemcpyAsync(h[i], d[i], bytes, hipMemcpyDeviceToHost, s[i]);
            (void)hipStreamSynchronize(s[i]);
        }
    };
    for (int i = 0; i < n; i++) work(i);   // warm-up, serial
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int i = 0; i < n; i++) th.emplace_back(work, i);
    for (auto &t : th) t.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int i = 0; i < n; i++) { if (pinned) (void)hipHostFree(h[i]); else free(h[i]); (void)hipFree(d[i]); (void)hipStreamDestroy(s[i]); }
    return (double)bytes * reps * n / dt / 1e9;
}

int main() {
    const size_t bytes = 12582912;
    for (int pinned = 1; pinned >= 0; pinned--)
        for (auto c : {std::pair<int, int>{1, 0}, {0, 1}, {1, 1}, {2, 2}, {4, 4}, {8, 8}, {4, 0}, {0, 4}})
            printf("%s  h2d x%d  d2h x%d : %6.1f GB/s total\n", pinned ? "pinned  " : "pageable", c.first, c.second,
                   run(pinned != 0, c.first, c.second, bytes, 40));
    return 0;
}
