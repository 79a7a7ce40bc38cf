#!/usr/bin/env python3
"""Throughput of the other BASELINE.json shapes (configs 3 and 4) and small batches.
Run on the GPU box: python profiles/shape_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dbde_video_cpp_amd as dv

codec = dv.Codec(0)
cases = [(4096, 3072, 64, "mixed"), (4096, 3072, 1, "mixed"), (4096, 3072, 4, "mixed"), (2048, 2048, 192, "mixed"),
         (2048, 2048, 192, "noise8"), (1921, 1081, 384, "mixed"), (1921, 1081, 384, "noise8"), (1920, 1080, 384, "mixed"),
         (640, 480, 2048, "mixed"), (10, 10, 4096, "mixed")]
for (W, H, n, content) in cases:
    imgs = codec.synth_frames(content, 1, 0, n, W, H)
    buf, lead, cap = codec.alloc_stream(W, H, n)
    out = torch.empty_like(imgs)
    offs = torch.empty(n, dtype=torch.int64, device=imgs.device)
    sizes = torch.empty(n, dtype=torch.int64, device=imgs.device)
    for _ in range(2):
        codec.encode_frames(imgs, W, H, n, buf, lead, cap, offsets=offs, nbytes=sizes)
        codec.decode_frames(buf, lead, cap, offs, W, H, n, images=out)
    codec.sync()
    assert torch.equal(out, imgs)
    packed = int(sizes.sum().item())
    codec.timing(True); codec.timing_read()
    K = 10
    t0 = time.perf_counter()
    for _ in range(K):
        codec.encode_frames(imgs, W, H, n, buf, lead, cap, offsets=offs, nbytes=sizes)
        codec.decode_frames(buf, lead, cap, offs, W, H, n, images=out)
    codec.sync()
    wall = (time.perf_counter() - t0) / K
    tk = codec.timing_read(); codec.timing(False)
    alg = n * W * H + packed
    e, d, i = tk["encode"][0] / K, tk["decode"][0] / K, tk["decode_index"][0] / K
    print(f"{W}x{H} n={n:5d} {content:7s} enc {e*1e3:8.1f} us {alg/e/1e6:7.0f} GB/s | dec {d*1e3:8.1f} us {alg/d/1e6:7.0f} GB/s | idx {i*1e3:6.1f} us | wall/step {wall*1e6:8.1f} us  {n/wall:10.0f} frames/s", flush=True)
    del imgs, buf, out
