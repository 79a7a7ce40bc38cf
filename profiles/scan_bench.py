"""Times dbde_hip_index_stream_async on BASELINE configs[2]'s stream (1000 frames of 2048x2048 mixed) for the
segment / workgroup counts given in the environment (DBDE_HIP_SCAN_SEGS, DBDE_HIP_SCAN_WGS)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dbde_video_cpp_amd as dv
W, H, n = 2048, 2048, 1000
codec = dv.Codec(0)
imgs = codec.synth_frames("mixed", 0xDBDE2016, 0, n, W, H)
buf, lead, cap = codec.alloc_stream(W, H, n)
offs, sizes = codec.encode_frames(imgs, W, H, n, buf, lead, cap)
codec.sync()
total = int((offs[-1] + sizes[-1]).item())
found = torch.empty(n, dtype=torch.int64, device="cuda")
cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
for _ in range(3):
    codec.index_stream_async(buf, lead, total, W, H, n, found, cnt)
codec.sync()
assert torch.equal(found, offs) and int(cnt.item()) == n
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    codec.index_stream_async(buf, lead, total, W, H, n, found, cnt)
b.record(); torch.cuda.synchronize()
print(f"segs={os.environ.get('DBDE_HIP_SCAN_SEGS','-')} wgs={os.environ.get('DBDE_HIP_SCAN_WGS','-')}: {a.elapsed_time(b)/20*1e3:.1f} us per walk of {n} frames")
