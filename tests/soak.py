#!/usr/bin/env python3
"""Soak: many full-size batches (random content / batch size / layout), each round-tripped on the GPU
and spot-checked against the oracle's bytes.  Looks for rare, timing-dependent corruption in the
persistent encoder (hand-placed waits, cross-workgroup hand-off), which short parity tests could miss.

    python tests/soak.py [--rounds 40] [--seed 1]   (test infrastructure: the oracle is its checker; logs: profiles/soak_r0N.log)
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dbde_video_cpp_amd as dv  # noqa: E402
from oracle_ffi import Oracle  # noqa: E402  (checker only)

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
codec, ora = dv.Codec(0), Oracle()
# (the last six: tiny-frame kernels, T <= 64; with n in {1, 3} the large shapes take the fused index + decode launch)
shapes = [(4096, 3072), (4096, 3072), (2048, 2048), (1921, 1081), (1920, 1080), (1001, 999), (1366, 768), (641, 481), (1928, 1080),
          (720, 1280), (1440, 900), (1080, 1920), (1360, 768),   # 16-byte rows: direct 16-byte stores / staged per chunk; 8-byte rows staged
          (72, 72), (96, 96), (128, 128), (9, 600), (130, 121),   # 65 .. 272 tiles: whole frames per workgroup (encode_mid / decode_mid) and just above
          (160, 120), (176, 144), (200, 152), (104, 100), (224, 200),   # round 4: 65 .. 640 tiles with 8-byte rows: encode_frames_kernel
          (2999, 2001), (1923, 1083), (1935, 1080),                     # round 4: odd rows on dword-aligned fetches (kInRaw4) / last pair of 15 columns (natural)
          (1081, 1921), (1009, 700),                                    # ... pairs dealt linearly (68 / 64 pairs per tile row); the others above: one wave per row segment (kInRow)
          (1921, 1081), (1001, 999), (64, 64), (8, 8), (61, 59), (33, 31), (512, 8), (24, 16),
          (32, 32), (40, 24), (16, 16), (80, 80), (120, 120), (128, 120), (72, 72), (64, 64)]   # round 4, second half: the persistent small-frame kernels (decode_mid_kernel, encode_group_kernel)
t0 = time.time()
frames_done = 0
for r in range(a.rounds):
    W, H = shapes[int(rng.integers(0, len(shapes)))]
    per = W * H
    n = int(rng.choice([1, 3, 17, 64, 200, 512, 1024]))
    if per <= 128 * 128:
        n = int(rng.choice([1, 7, 1000, 30000, 200000]))     # small frames: many per workgroup, persistent workgroups walk many groups
    n = max(1, min(n, int(9e9 // per)))
    content = str(rng.choice(["noise8", "mixed", "smooth", "flat"]))
    concat = bool(rng.integers(0, 2)) and n * 8 * dv.tiles(W, H) < 2**32
    first = int(rng.integers(0, 1 << 40))
    imgs = codec.synth_frames(content, 0xDBDE2016 + r, first, n, W, H)
    slot = 0 if concat else ((dv.max_frame_bytes(W, H) + 255) // 256) * 256
    buf, lead, cap = codec.alloc_stream(W, H, n, slot_stride=slot)
    for rep in range(3):   # same inputs three times: a race would not repeat itself identically
        offs, sizes = codec.encode_frames(imgs, W, H, n, buf, lead, cap, first_index=first, slot_stride=slot)
        back, res = codec.decode_frames(buf, lead, cap, offs, W, H, n)
        codec.sync()
        assert torch.equal(back, imgs), (r, rep, W, H, n, content, concat)
        if rep == 0:
            host_o, host_s = offs.cpu().numpy(), sizes.cpu().numpy()
            ref_sum = int(buf[lead:lead + int(host_o[-1] + host_s[-1])].to(torch.int64).sum().item()) if concat else None
        elif concat:
            assert int(buf[lead:lead + int(host_o[-1] + host_s[-1])].to(torch.int64).sum().item()) == ref_sum
    for f in sorted(set([0, n // 2, n - 1])):
        want = ora.pack_frame(first + f, imgs[f].cpu().numpy(), W, H)
        got = buf[lead + int(host_o[f]): lead + int(host_o[f] + host_s[f])].cpu().numpy()
        assert got.tobytes() == want.tobytes(), (r, f, W, H, n, content, concat)
    frames_done += 3 * n
    del imgs, buf, back, res
    print(f"round {r:3d}: {W}x{H} n={n:5d} {content:7s} {'concat' if concat else 'slots '} ok", flush=True)
# DBDE16 (extension): the persistent encoder's static / ticket chunk ids and the record sums, against itself
# (round trip) and, for one frame per round, against the DBDE16 oracle's bytes
import ctypes as C  # noqa: E402
from oracle_ffi import ORACLE_SO  # noqa: E402
from test_oracle_u16 import pack16, u8p, u16p  # noqa: E402
o16 = C.CDLL(ORACLE_SO)
o16.dbde16_oracle_max_frame_bytes.restype = C.c_size_t
o16.dbde16_oracle_max_frame_bytes.argtypes = [C.c_int, C.c_int]
o16.dbde16_oracle_pack_frame.restype = C.c_size_t
o16.dbde16_oracle_pack_frame.argtypes = [C.c_uint64, u16p, C.c_int, C.c_int, u8p]
u16_rounds = max(a.rounds // 4, 1)
for r in range(u16_rounds):
    # (aligned widths with enough chunks take the persistent encoder, PIX = 2; 1000x1003: T % 8 != 0, unaligned minima)
    # (round 4: widths off 8 pixels with enough chunks take its any-geometry instance)
    W, H = [(2048, 1536), (1921, 1081), (4096, 3072), (640, 480), (1000, 1003), (1024, 768), (1001, 1003), (1003, 517)][int(rng.integers(0, 8))]
    n = int(rng.choice([1, 3, 16, 40, 64]))
    kind = str(rng.choice(["full", "mixed", "small"]))
    d = rng.integers(0, 17, size=(n, (H + 7) // 8, (W + 7) // 8))
    dd = np.repeat(np.repeat(d, 8, axis=1), 8, axis=2)[:, :H, :W]
    noise = rng.integers(0, 65536, size=(n, H, W))
    if kind == "full":
        img = noise
    elif kind == "mixed":
        img = np.minimum(20000, 65535 - ((1 << dd) - 1)) + (noise & ((1 << dd) - 1))
    else:
        img = noise >> 11
    imgs_h = np.ascontiguousarray(img.astype(np.uint16))
    imgs = torch.from_numpy(imgs_h.view(np.int16)).cuda()
    maxf = int(codec.L.dbde16_hip_max_frame_bytes(W, H))
    concat = bool(rng.integers(0, 2))
    slot = 0 if concat else ((maxf + 255) // 256) * 256
    cap = n * maxf if concat else (n - 1) * slot + maxf
    buf = torch.empty(32 + cap + 64, dtype=torch.uint8, device="cuda")
    for rep in range(3):
        offs, sizes = codec.encode_frames16(imgs, W, H, n, buf, 32, cap, slot_stride=slot)
        back, res = codec.decode_frames16(buf, 32, cap, offs, W, H, n)
        codec.sync()
        assert torch.equal(back, imgs), ("u16", r, rep, W, H, n, kind, concat)
    o, sz = offs.cpu().numpy(), sizes.cpu().numpy()
    f = n // 2
    want = pack16(o16, imgs_h[f], f)
    assert buf[32 + int(o[f]): 32 + int(o[f] + sz[f])].cpu().numpy().tobytes() == want.tobytes(), ("u16", r, W, H, n, kind)
    frames_done += 3 * n
    del imgs, buf, back
    print(f"u16 round {r:3d}: {W}x{H} n={n:3d} {kind:6s} {'concat' if concat else 'slots '} ok", flush=True)
print(f"soak ok: {a.rounds} rounds (+{u16_rounds} DBDE16), {frames_done} frame round trips, {time.time() - t0:.0f} s")
