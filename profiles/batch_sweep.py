#!/usr/bin/env python3
"""Encode/decode rates across batch sizes and both layouts (4096x3072). GPU box: python profiles/batch_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dbde_video_cpp_amd as dv
W, H = 4096, 3072
codec = dv.Codec(0)
for content in ("noise8", "mixed"):
    for n in (1, 4, 16, 64, 128, 256, 512, 1024):
        imgs = codec.synth_frames(content, 1, 0, n, W, H)
        out = torch.empty_like(imgs)
        for layout in ("concat", "slots"):
            slot = 0 if layout == "concat" else ((dv.max_frame_bytes(W, H) + 255) // 256) * 256
            buf, lead, cap = codec.alloc_stream(W, H, n, slot_stride=slot)
            offs = torch.empty(n, dtype=torch.int64, device=imgs.device)
            sizes = torch.empty(n, dtype=torch.int64, device=imgs.device)
            for _ in range(2):
                codec.encode_frames(imgs, W, H, n, buf, lead, cap, offsets=offs, nbytes=sizes, slot_stride=slot)
                codec.decode_frames(buf, lead, cap, offs, W, H, n, images=out)
            codec.sync()
            assert torch.equal(out, imgs)
            packed = int(sizes.sum().item())
            codec.timing(True); codec.timing_read()
            K = max(3, min(50, 2048 // n))
            for _ in range(K):
                codec.encode_frames(imgs, W, H, n, buf, lead, cap, offsets=offs, nbytes=sizes, slot_stride=slot)
                codec.decode_frames(buf, lead, cap, offs, W, H, n, images=out)
            codec.sync()
            tk = codec.timing_read(); codec.timing(False)
            alg = n * W * H + packed
            e, d, i = tk["encode"][0] / K, tk["decode"][0] / K, tk["decode_index"][0] / K
            print(f"{content:7s} n={n:5d} {layout:6s} enc {e*1e3:9.1f} us {alg/e/1e6:6.0f} GB/s | dec {d*1e3:9.1f} us {alg/d/1e6:6.0f} GB/s | idx {i*1e3:6.1f} us", flush=True)
            del buf
        del imgs, out
