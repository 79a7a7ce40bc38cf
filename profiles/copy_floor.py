#!/usr/bin/env python3
"""What a plain device copy of one 4096x3072 frame costs per launch (the floor for one frame per call): run under
rocprofv3 --kernel-trace and read the copy kernel's duration."""
import time, torch
x = torch.randint(0, 255, (3072, 4096), dtype=torch.uint8, device="cuda")
y = torch.empty_like(x)
for _ in range(20): y.copy_(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(400): y.copy_(x)
torch.cuda.synchronize()
print(f"copy 12.6 MB: {(time.perf_counter() - t0) / 400 * 1e6:.2f} us per launch, pipelined")
