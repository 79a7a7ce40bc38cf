#!/usr/bin/env python3
"""File ends of the path: frames/s of the batched .dbde writer and reader
with images resident in HBM and the file on a memory-backed filesystem (so the figure shows
the PCIe + pipeline cost, not a disk).  Not the headline metric: bench.py stays device-resident.

    python profiles/file_io_bench.py [--frames 128] [--batch 16] [--content mixed] [--dir /dev/shm]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dbde_video_cpp_amd as dv  # noqa: E402

W, H = 4096, 3072

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=128)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--content", default="mixed")
ap.add_argument("--dir", default="/dev/shm")
a = ap.parse_args()

codec = dv.Codec(0)
n = a.frames
imgs = codec.synth_frames(a.content, 0xDBDE2016, 0, n, W, H)
path = os.path.join(a.dir, f"dbde_io_bench_{os.getpid()}.dbde")
out = {"frames": n, "batch": a.batch, "content": a.content, "dir": a.dir}
try:
    for rep in range(2):   # first pass warms allocations and page cache
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with codec.open_writer(path, W, H, batch_frames=a.batch) as w:
            w.put(imgs, n)
        tw = time.perf_counter() - t0
        size = os.path.getsize(path)
        back = torch.empty((a.batch, H, W), dtype=torch.uint8, device=imgs.device)
        got, ok = 0, True
        t0 = time.perf_counter()
        with codec.open_reader(path, batch_frames=a.batch) as r:
            while True:
                im, hd = r.next(images=back)
                if not hd:
                    break
                if rep == 1:
                    ok = ok and torch.equal(im, imgs[got:got + len(hd)])
                got += len(hd)
        tr = time.perf_counter() - t0
    # the last timing includes the equality checks; time the reader once more without them
    t0 = time.perf_counter()
    with codec.open_reader(path, batch_frames=a.batch) as r:
        while r.next(images=back)[1]:
            pass
    tr = time.perf_counter() - t0
    out.update({"file_bytes": size, "write_frames_per_s": round(n / tw, 1), "write_file_GBps": round(size / tw / 1e9, 2),
                "read_frames_per_s": round(n / tr, 1), "read_file_GBps": round(size / tr / 1e9, 2),
                "frames_read": got, "round_trip_identical": bool(ok)})
finally:
    if os.path.exists(path):
        os.remove(path)
print(json.dumps(out))
