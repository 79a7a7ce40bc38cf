#!/usr/bin/env python3
"""Generate tests/golden/*.{npz,json} by running the REAL reference (oracle/_ref/libdbde_ref.so).

Run in the build container (where /root/reference exists and `make -C oracle` has produced
oracle/_ref/libdbde_ref.so):

    python tests/golden/make_golden.py

The outputs are data only -- inputs and the reference's outputs -- and are committed; the
reference itself never travels.  Inputs come from three places:
  * the README's worked 10x10 example (reference README.md:73-84),
  * the 8x16 image of the reference's own known-answer test (dbde_util_test.cpp:135-144),
  * the counter-based generators of oracle/synth.c (seeded, reproducible anywhere).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_ffi import Oracle, Reference  # noqa: E402

SEED = 0xDBDE2016

README_10x10 = np.array([
    25, 27, 23, 29, 22, 24, 29, 23, 25, 24,
    22, 24, 21, 25, 22, 27, 28, 21, 27, 26,
    25, 26, 22, 29, 25, 20, 28, 23, 26, 25,
    19, 23, 25, 21, 28, 19, 22, 25, 25, 27,
    27, 25, 30, 28, 25, 23, 27, 26, 24, 24,
    31, 30, 31, 28, 29, 26, 24, 25, 27, 26,
    30, 28, 32, 25, 28, 27, 28, 27, 26, 26,
    29, 31, 31, 32, 29, 29, 25, 22, 24, 25,
    31, 34, 33, 31, 30, 29, 28, 28, 26, 26,
    34, 34, 35, 35, 33, 28, 29, 28, 26, 26], np.uint8).reshape(10, 10)

KAT_8x16 = np.array([
    0, 1, 9, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
    8, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17,
    4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19,
    6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21,
    7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22,
    5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 21,
    3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20,
    1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 17, 19], np.uint8).reshape(8, 16)

SMALL_SIZES = [(1, 1), (5, 3), (8, 8), (9, 17), (16, 24), (31, 33), (64, 64), (10, 10), (17, 250),
               (1, 16), (16, 1), (40, 72), (23, 129)]  # (H, W)
MODES = {"noise8": 0, "mixed": 1, "flat": 2, "smooth": 3}

BIG = [  # BASELINE.json configs 2, 3, 4 at the bench seeds: hashes only
    ("cfg2_4096x3072", 3072, 4096), ("cfg3_2048x2048", 2048, 2048), ("cfg4_1921x1081", 1081, 1921),
    # shapes whose kernel forms differ from the configs' (round 3): portrait HD, 1366x768, 16-byte rows that are not
    # whole cache lines, frames of 81 / 144 tiles
    ("shape_1080x1920", 1920, 1080), ("shape_1366x768", 768, 1366), ("shape_1440x900", 900, 1440),
    ("shape_720x1280", 1280, 720), ("shape_72x72", 72, 72), ("shape_96x96", 96, 96),
    # frames of 300 .. 1200 tiles (round 4: whole frames per workgroup up to ~700 tiles, staged loads and stores)
    ("shape_160x120", 120, 160), ("shape_176x144", 144, 176), ("shape_320x240", 240, 320),
    # frames of 64 and 256 tiles: the two ends of the small-frame decoder's old and new ranges (round 4, second half)
    ("shape_64x64", 64, 64), ("shape_128x128", 128, 128)]
# bench.py --gpus N: rank r round-trips frames r * 1024 .. r * 1024 + 1023 of the headline shape; every rank is held to
# the reference's SHA-256 of ITS frames 0 and 3 (rank 0's are the cfg2 entries above)
RANK_FRAMES = [r * 1024 + d for r in range(1, 8) for d in (0, 3)]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def depth_ramp_frame(H, W):
    """Every tile column gets a different range so depths 0..8 all occur, edges included."""
    img = np.zeros((H, W), np.uint8)
    for y in range(H):
        for x in range(W):
            d = (x // 8 + y // 8) % 9
            span = (1 << d) - 1
            img[y, x] = (17 * (x // 8) + 5 * (y // 8)) % (256 - span) + ((x * 7 + y * 13) % (span + 1))
    return img


def main():
    if not Reference.available():
        sys.exit("oracle/_ref/libdbde_ref.so missing: run `make -C oracle` where /root/reference exists")
    ref, ora = Reference(), Oracle()
    arrays, manifest = {}, {"seed": SEED, "frames": [], "big": [], "headers": {}, "malformed": []}

    def add_frame(name, img, index):
        H, W = img.shape
        packed = ref.pack_frame(index, img, W, H)
        n, fh, back = ref.unpack_frame(packed, W, H)
        assert n == len(packed) and (back == img).all() and fh == (2, index, 0)
        arrays[name + ".image"] = img
        arrays[name + ".packed"] = packed
        manifest["frames"].append({"name": name, "H": H, "W": W, "index": index,
                                   "packed_bytes": int(len(packed))})

    add_frame("readme_10x10", README_10x10, 7)
    add_frame("kat_8x16", KAT_8x16, 1)
    for (H, W) in SMALL_SIZES:
        for mname, mode in MODES.items():
            img = ora.synth_frame(mode, SEED, H * 1000 + W, W, H)
            add_frame(f"synth_{mname}_{H}x{W}", img, H * 1000 + W)
    for (H, W) in [(8, 72), (19, 83), (72, 8), (26, 26)]:
        add_frame(f"ramp_{H}x{W}", depth_ramp_frame(H, W), 42)

    # tile-level: the reference test's four demos (dbde_util_test.cpp:219-299) use the README
    # image with row 3 col 3 = 41; record codes + payloads (printed, un-asserted there).
    demo = README_10x10.copy()
    demo[3, 3] = 41
    flat = demo.reshape(-1)
    tiles = []
    for (off, rm, dm) in [(0, 8, 8), (8, 2, 8), (80, 8, 2), (88, 2, 2)]:
        if rm == 8 and dm == 8:
            code, payload, _ = ref.pack_8x8(flat, off, 10)
        else:
            code, payload, _ = ref.pack_8x8_partial(flat, off, 10, rm, dm)
        tiles.append({"off": off, "rm": rm, "dm": dm, "code": int(code), "payload": payload.tobytes().hex()})
    arrays["demo_10x10.image"] = demo
    manifest["tile_demos"] = tiles

    # depth boundaries: range -> depth
    bounds = []
    for rng in [0, 1, 2, 3, 4, 7, 8, 15, 16, 31, 32, 63, 64, 127, 128, 255]:
        t = np.zeros(64, np.uint8)
        t[5] = rng
        code, _, _ = ref.pack_8x8(t, 0, 8)
        bounds.append([rng, int(code >> 8)])
    manifest["depth_bounds"] = bounds

    # headers (trap T1: elapsed_ns travels as a double)
    fhs = []
    for (u, idx, el) in [(2, 0, 0), (2, 5, 1000000007), (2, 2**64 - 1, 2**53 + 1), (2, 123456789012345, 2**63),
                         (3, 1, 1), (2, 77, 999999999999999999)]:
        wire = ref.pack_frame_header(u, idx, el)
        n, back = ref.unpack_frame_header(wire)
        fhs.append({"in": [u, idx, el], "wire": wire.tobytes().hex(), "advance": int(n),
                    "out": [int(x) for x in back]})
    manifest["headers"]["frame"] = fhs
    vhs = []
    for (u, h, w, hz) in [(3, 8, 16, 1.0), (3, 3072, 4096, 59.94), (3, 1081, 1921, 1000.0), (4, 1, 2, 0.5)]:
        wire = ref.pack_video_header(u, h, w, hz)
        n, back = ref.unpack_video_header(wire)
        vhs.append({"in": [u, h, w, hz], "wire": wire.tobytes().hex(), "advance": int(n),
                    "out": [int(back[0]), int(back[1]), int(back[2]), float(back[3])]})
    manifest["headers"]["video"] = vhs

    # malformed streams (traps T9): what the reference returns
    base = ref.pack_frame(9, README_10x10, 10, 10)
    T = 4
    for (label, pos, delta) in [("nb", 20, 1), ("nm", 20 + 4 + T, 1), ("n64", 20 + 8 + 2 * T, 1),
                                ("n64_minus", 20 + 8 + 2 * T, -1), ("frame_field", 0, 1)]:
        bad = base.copy()
        bad[pos] = (int(bad[pos]) + delta) % 256
        n, fh, img = ref.unpack_frame(bad, 10, 10, fill=0xEE)
        n_img, img2 = ref.unpack_image(bad[20:], 10, 10, fill=0xEE)
        manifest["malformed"].append({"label": label, "pos": pos, "delta": delta, "advance": int(n),
                                      "u64s": int(fh[0]), "index": int(fh[1]),
                                      "image_untouched": bool((img == 0xEE).all()),
                                      "unpack_image_ret": int(n_img),
                                      "image_sha": sha(img)})
    arrays["malformed_base.packed"] = base

    # big configs: hashes only (images come from synth at test time)
    for (name, H, W) in BIG:
        for mname in ("noise8", "mixed"):
            for frame in (0, 3):
                img = ora.synth_frame(MODES[mname], SEED, frame, W, H)
                packed = ref.pack_frame(frame, img, W, H)
                T_ = ((W + 7) // 8) * ((H + 7) // 8)
                depths = packed[24:24 + T_]
                hist = np.bincount(depths, minlength=9).tolist()
                manifest["big"].append({"name": name, "mode": mname, "frame": frame, "H": H, "W": W,
                                        "image_sha": sha(img), "packed_sha": sha(packed),
                                        "packed_bytes": int(len(packed)), "depth_hist": hist})

    for mname in ("noise8", "mixed"):
        for frame in RANK_FRAMES:
            W, H = 4096, 3072
            img = ora.synth_frame(MODES[mname], SEED, frame, W, H)
            packed = ref.pack_frame(frame, img, W, H)
            hist = np.bincount(packed[24:24 + (W // 8) * (H // 8)], minlength=9).tolist()
            manifest["big"].append({"name": "cfg2_rank_frames", "mode": mname, "frame": frame, "H": H, "W": W,
                                    "image_sha": sha(img), "packed_sha": sha(packed),
                                    "packed_bytes": int(len(packed)), "depth_hist": hist})

    np.savez_compressed(os.path.join(HERE, "frames.npz"), **arrays)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("wrote", len(arrays), "arrays;", len(manifest["frames"]), "frames;", len(manifest["big"]), "big hashes")


if __name__ == "__main__":
    main()
