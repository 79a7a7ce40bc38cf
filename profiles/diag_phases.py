#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the persistent encoder (DIAG build, s_memtime stamps).
Run on the GPU box: DBDE_HIP_EXPERIMENT=4 python profiles/diag_phases.py [content]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dbde_video_cpp_amd as dv
content = sys.argv[1] if len(sys.argv) > 1 else "noise8"
W, H, B = 4096, 3072, 64
codec = dv.Codec(0)
imgs = codec.synth_frames(content, 0xDBDE2016, 0, B, W, H)
buf, lead, cap = codec.alloc_stream(W, H, B)
for _ in range(3):
    codec.encode_frames(imgs, W, H, B, buf, lead, cap)
codec.sync()
out = (C.c_uint64 * 32)()
L = dv.lib()
L.dbde_hip_debug_read_diag.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
L.dbde_hip_debug_read_diag(codec.h, out)
N = 5
for _ in range(N):
    codec.encode_frames(imgs, W, H, B, buf, lead, cap)
codec.sync()
L.dbde_hip_debug_read_diag(codec.h, out)
names = ["mailbox+ticket", "issue loads", "barrier", "store prev", "pack", "rotate/loadwait", "stats+AGG", "offsets(prev)"]
chunks = N * B * 192
for base, who in ((0, "wave0"), (8, "wave7")):
    tot = sum(out[base + i] for i in range(8))
    print(who, "cycles per chunk (s_memtime ticks):", round(tot / chunks, 1))
    for i, n in enumerate(names[:8]):
        print(f"   {n:16s} {out[base+i]/chunks:9.1f}  {100*out[base+i]/max(tot,1):5.1f}%")

print("scanner: rounds per launch", out[16] / N, "empty", out[17] / N, "load-wait cycles/round", out[18] / max(out[16], 1),
      "process cycles/round", out[19] / max(out[16], 1), "records/non-empty round", N * B * 192 / max(out[16] - out[17], 1))
print("slow-path spins per launch:", out[20] / N, "of", 511 * 24, "block-iterations")
