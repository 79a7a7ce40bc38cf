"""DBDE16 throughput (extension, parity unpinned): n frames of 4096x3072 U16, per-tile random depth 0..16."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dbde_video_cpp_amd as dv

W, H, n = 4096, 3072, 128
SLOTS = False
if len(sys.argv) > 1 and sys.argv[1] == "frames":   # many small frames, one slot each
    W, H, n, SLOTS = 1024, 768, 2048, True
CROP = None
if "odd" in sys.argv:    # a width that is no multiple of 8 pixels: the any-geometry form of the persistent encoder (round 4)
    CROP = (4091, 3070)
codec = dv.Codec(0)
g = torch.Generator(device="cuda").manual_seed(1)
d = torch.randint(0, 17, (n, H // 8, W // 8), device="cuda", generator=g)
dd = d.repeat_interleave(8, 1).repeat_interleave(8, 2)
mask = (torch.ones_like(dd) << dd) - 1
noise = torch.randint(0, 65536, (n, H, W), device="cuda", generator=g) & mask
base = torch.randint(0, 32768, (n, H // 8, W // 8), device="cuda", generator=g).repeat_interleave(8, 1).repeat_interleave(8, 2)
imgs = torch.minimum(base, 65535 - mask).add_(noise).to(torch.int32).to(torch.int16).contiguous()   # two's complement bits = the U16 pixels
if "full" in sys.argv:    # every tile of depth 16 (worst case: payload = raw)
    imgs = torch.randint(-32768, 32768, (n, H, W), device="cuda", generator=g, dtype=torch.int16)
if "d12" in sys.argv:     # 12-bit sensor noise in 16-bit pixels: every tile of depth 12
    imgs = torch.randint(0, 4096, (n, H, W), device="cuda", generator=g, dtype=torch.int16)
del d, dd, mask, noise, base
if CROP:
    W, H = CROP
    imgs = imgs[:, :H, :W].contiguous()
maxf = int(codec.L.dbde16_hip_max_frame_bytes(W, H))
slot = ((maxf + 255) // 256) * 256 if SLOTS else 0
cap = (n - 1) * slot + maxf if SLOTS else n * maxf
buf = torch.empty(32 + cap + 64, dtype=torch.uint8, device="cuda")
out = torch.empty_like(imgs)
for _ in range(2):
    offs, sizes = codec.encode_frames16(imgs, W, H, n, buf, 32, cap, slot_stride=slot)
    codec.decode_frames16(buf, 32, cap, offs, W, H, n, images=out)
codec.sync()
assert torch.equal(out, imgs)
packed = int(sizes.sum().item())
codec.timing(True); codec.timing_read(reset=True)
steps = 10
t0 = time.perf_counter()
for _ in range(steps):
    offs, sizes = codec.encode_frames16(imgs, W, H, n, buf, 32, cap, slot_stride=slot)
    codec.decode_frames16(buf, 32, cap, offs, W, H, n, images=out)
codec.sync()
dt = (time.perf_counter() - t0) / steps
tk = codec.timing_read()
raw = n * W * H * 2
enc, idx, dec = tk["encode"][0] / steps, tk["decode_index"][0] / steps, tk["decode"][0] / steps
print(f"DBDE16 {n} x {W}x{H}: packed/raw {packed/raw:.3f}; encode (single pass, decoupled look-back) {enc:.3f} ms = "
      f"{(raw + packed)/enc/1e6:.0f} GB/s algorithmic ({(raw+packed)/enc/1e6/8000:.3f} of 8 TB/s); decode {dec:.3f} ms + index {idx:.3f} ms = "
      f"{(raw + packed)/dec/1e6:.0f} GB/s ({(raw+packed)/dec/1e6/8000:.3f}); round trip {n/dt:.0f} frames/s")
if os.environ.get("U16_DIAG"):   # -DDBDE_DIAG builds: wave 0's phase times of the encoder, per workgroup
    import ctypes as C
    d = (C.c_uint64 * 16)()
    codec.L.dbde_hip_diag_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    codec.L.dbde_hip_diag_read(codec.h, d)
    codec.sync()
    wg = max(d[6], 1)
    names = ["ticket", "load+reduce+publish", "pack", "look-back", "barrier", "store"]
    wd = max(d[12], 1)
    print("dec16 per workgroup (us): " + ", ".join(f"{nm} {d[8 + i] / wd / 100:.2f}" for i, nm in enumerate(["fetch issue", "barrier", "LDS reads", "unpack+store issue"])) + f"  [{wd} workgroups]")
    print("enc16 per workgroup (us): " + ", ".join(f"{nm} {d[i] / wg / 100:.2f}" for i, nm in enumerate(names)) + f"  [{wg} workgroups]")
