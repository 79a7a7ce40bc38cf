import torch, time
n = 12884901888
x = torch.empty(n, dtype=torch.uint8, device="cuda")
y = torch.empty(n, dtype=torch.uint8, device="cuda")
xi = x.view(torch.int64); yi = y.view(torch.int64)
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
a = t(lambda: xi.fill_(3)); print("fill  %.3f ms  %.0f GB/s" % (a*1e3, n/a/1e9))
a = t(lambda: yi.copy_(xi)); print("copy  %.3f ms  %.0f GB/s (r+w)" % (a*1e3, 2*n/a/1e9))
a = t(lambda: xi.sum()); print("sum   %.3f ms  %.0f GB/s" % (a*1e3, n/a/1e9))
a = t(lambda: torch.add(xi, 1, out=yi)); print("add   %.3f ms  %.0f GB/s (r+w)" % (a*1e3, 2*n/a/1e9))
