#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points (dbde_hip_pack_frame / dbde_hip_unpack_frame:
the reference's own signatures, host buffers in and out).  Never the headline value -- bench.py
measures device-resident batches -- but this is what a link-time drop-in user sees per call.

    python profiles/host_api_bench.py [--content mixed] [--reps 20]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dbde_video_cpp_amd as dv  # noqa: E402

W, H = 4096, 3072
ap = argparse.ArgumentParser()
ap.add_argument("--content", default="mixed")
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()

codec = dv.Codec(0)
img = codec.synth_frames(a.content, 0xDBDE2016, 0, 1, W, H).cpu().numpy().reshape(-1).copy()
out = np.empty(dv.max_frame_bytes(W, H) + 64, np.uint8)
back = np.empty(W * H, np.uint8)
L, h = codec.L, codec.h


def pack():
    return L.dbde_hip_pack_frame(h, 5, img.ctypes.data, W, H, out.ctypes.data)


def unpack():
    cur = C.c_void_p(out.ctypes.data)
    L.dbde_hip_unpack_frame(h, C.byref(cur), W, H, back.ctypes.data)
    return cur.value - out.ctypes.data


n = pack(); unpack()                      # warm-up (staging buffers)
t0 = time.perf_counter()
for _ in range(a.reps):
    pack()
tp = (time.perf_counter() - t0) / a.reps
t0 = time.perf_counter()
for _ in range(a.reps):
    unpack()
tu = (time.perf_counter() - t0) / a.reps
assert (back == img).all()
print(json.dumps({"content": a.content, "frame": f"{W}x{H}", "packed_bytes": int(n),
                  "pack_frame_ms": round(tp * 1e3, 3), "unpack_frame_ms": round(tu * 1e3, 3),
                  "round_trip_frames_per_s": round(1.0 / (tp + tu), 1),
                  "host_bytes_moved_per_round_trip": int(2 * (W * H + n)),
                  "effective_host_GBps": round(2 * (W * H + n) / (tp + tu) / 1e9, 2),
                  "note": "pageable host buffers, one frame per call, H2D + kernels + D2H + sync inside each call"}))
