#!/usr/bin/env python3
"""BASELINE configs[1] taken literally: ONE 4096x3072 frame per encode+decode call, device-resident.
Run under `rocprofv3 --kernel-trace --stats` to see where the time of a single round trip goes."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dbde_video_cpp_amd as dv  # noqa: E402

W, H = 4096, 3072
content = sys.argv[1] if len(sys.argv) > 1 else "noise8"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
codec = dv.Codec(0)
img = codec.synth_frames(content, 0xDBDE2016, 0, 1, W, H)
buf, lead, cap = codec.alloc_stream(W, H, 1)
out = torch.empty_like(img)
offs = torch.empty(1, dtype=torch.int64, device=img.device)
sizes = torch.empty(1, dtype=torch.int64, device=img.device)
res = torch.empty((1, 4), dtype=torch.int64, device=img.device)


def step():
    codec.encode_frames(img, W, H, 1, buf, lead, cap, offsets=offs, nbytes=sizes)
    codec.decode_frames(buf, lead, cap, offs, W, H, 1, images=out, results=res)


for _ in range(20):
    step()
codec.sync()
t0 = time.perf_counter()
for _ in range(reps):
    step()
codec.sync()
dt = (time.perf_counter() - t0) / reps
t0 = time.perf_counter()
for _ in range(reps):
    step()
    codec.sync()
dl = (time.perf_counter() - t0) / reps
assert torch.equal(out, img)
print(f"{content}: pipelined {dt*1e6:.1f} us per round trip ({1/dt:.0f} frames/s); "
      f"with a sync per frame {dl*1e6:.1f} us (latency)")
