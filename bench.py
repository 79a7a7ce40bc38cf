#!/usr/bin/env python3
"""bench.py -- DBDE encode+decode round trip on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames B] [--content noise8|mixed]

One step = one pass of the hot path over one batch: encode B distinct synthetic 4096x3072 U8
frames (resident in HBM) into DBDE frames, then decode them back to images.  Default layout:
one fixed-stride slot per frame -- the reference's own semantics, dbde_pack_frame packs each
frame into its own target -- which lets the encoder give every workgroup whole frames;
--concat writes one concatenated stream instead (a ready .dbde body; chunk offsets then come
from an in-launch scan).  B*W*H is far beyond the 256 MiB Infinity Cache, so every step
streams from HBM.  Default B = 1024 (BASELINE config 5 gives each GPU 1250 frames).
N > 1: launched by torch.distributed.run, one rank per GPU; frames are sharded by blocks
(rank g owns frames [g*B, (g+1)*B)), no collective on the data path (weak scaling).  The
RCCL gather of the compressed stream to rank 0 is run and timed separately ("gather").

Rank 0 prints ONE JSON line.  `value` = frames all ranks round-tripped per second.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import dbde_video_cpp_amd as dv  # noqa: E402

W, H = 4096, 3072
SEED = 0xDBDE2016
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def cpu_baseline(images_host, n_avail, budget_s=12.0):
    """Reference (oracle/_ref, kind 'reference') or oracle port timed on the host cores:
    every thread round-trips its own frames; bounded to about `budget_s` seconds."""
    from oracle_ffi import Oracle, Reference
    impl, kind = (Reference(), "reference") if Reference.available() else (Oracle(), "port")
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16, n_avail))
    # calibrate on one frame, one thread
    t1 = impl.time_roundtrip(images_host[0:1], 1, W, H, 1)[0]
    reps = max(1, int(budget_s / max(t1, 1e-4)))
    results = [None] * cores

    def work(k):
        results[k] = impl.time_roundtrip(images_host[k:k + 1], 1, W, H, reps)

    t0 = time.time()
    th = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
    [t.start() for t in th]
    [t.join() for t in th]
    wall = time.time() - t0
    bad = sum(r[3] for r in results)
    frames = cores * reps
    return {"value": round(frames / wall, 2), "unit": "frames/s", "cores": cores, "kind": kind,
            "sample": f"{cores} threads x {reps} round trips of one 4096x3072 frame each ({frames} total, "
                      f"{wall:.1f} s wall)",
            "single_thread_frames_per_s": round(1.0 / t1, 2), "mismatched_pixels": int(bad)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=1024, help="frames per step per GPU")
    ap.add_argument("--content", default="noise8", choices=["noise8", "mixed", "smooth", "flat"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the one-frame-per-call leg (profiling runs)")
    ap.add_argument("--concat", action="store_true", help="one concatenated stream instead of one slot per frame")
    ap.add_argument("--no-check", action="store_true", help="(experiments) skip the round-trip parity gate")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rehearsal on a one-GPU box: DBDE_BENCH_REHEARSAL=1 puts every rank on cuda:0 over gloo
        # (exercises the launch contract, sharding, barrier and max-over-ranks; not a measurement)
        rehearsal = os.environ.get("DBDE_BENCH_REHEARSAL") == "1"
        if rehearsal:
            local = 0
        torch.cuda.set_device(local)
        if rehearsal:
            dist.init_process_group("gloo")
            args.no_gather = True
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
        local = 0
    dev = torch.device("cuda", local)
    codec = dv.Codec(local)
    B = args.frames
    T = (W // 8) * (H // 8)

    # ---- inputs resident in HBM before the timed region ------------------------------------
    imgs = codec.synth_frames(args.content, SEED, rank * B, B, W, H)
    slot = 0 if args.concat else ((dv.max_frame_bytes(W, H) + 255) // 256) * 256
    buf, lead, cap = codec.alloc_stream(W, H, B, slot_stride=slot)
    out = torch.empty_like(imgs)
    offs = torch.empty(B, dtype=torch.int64, device=dev)
    sizes = torch.empty(B, dtype=torch.int64, device=dev)
    res = torch.empty((B, 4), dtype=torch.int64, device=dev)
    stream_cap = cap

    def step():
        codec.encode_frames(imgs, W, H, B, buf, lead, cap, first_index=rank * B, offsets=offs, nbytes=sizes, slot_stride=slot)
        codec.decode_frames(buf, lead, stream_cap, offs, W, H, B, images=out, results=res)

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    codec.sync()
    # parity gate on the measured configuration: round trip identical, sizes consistent
    assert args.no_check or torch.equal(out, imgs), "round trip mismatch"
    s_h = sizes.cpu().numpy()
    packed_bytes = int(s_h.sum())

    codec.timing(True)
    codec.timing_read(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    tk = codec.timing_read(reset=True)
    codec.timing(False)
    codec.sync()

    t_all = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
    dt_max = float(t_all.item())
    frames_total = world * B * args.steps
    fps = frames_total / dt_max

    # ---- roofline of the kernels (HIP events on the codec's stream, inside the timed region) --
    raw = B * W * H
    enc_ms = tk["encode"][0] / max(tk["encode"][1], 1)
    dec_ms = tk["decode"][0] / max(tk["decode"][1], 1)
    idx_ms = tk["decode_index"][0] / max(tk["decode_index"][1], 1)
    alg = raw + packed_bytes                      # encode reads raw, writes packed; decode the reverse
    enc_gbps = alg / (enc_ms * 1e-3) / 1e9
    dec_gbps = alg / (dec_ms * 1e-3) / 1e9
    fw_min = int(os.environ.get("DBDE_HIP_FRAMEWISE_MIN", "0"))
    enc_name = "dbde::encode_framewise_kernel" if (slot and fw_min and B >= fw_min) else "dbde::encode_kernel"
    dom = (enc_name, enc_ms, enc_gbps) if enc_ms >= dec_ms else ("dbde::decode_kernel", dec_ms, dec_gbps)
    roofline = {"bound": "hbm", "kernel": dom[0], "achieved": round(dom[2], 1), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(dom[2] / HBM_PEAK_GBPS, 4), "traffic": None,
                "algorithmic_bytes_per_launch": alg, "launch_ms": round(dom[1], 4),
                "encode": {"ms": round(enc_ms, 4), "GBps": round(enc_gbps, 1), "frac": round(enc_gbps / HBM_PEAK_GBPS, 4)},
                "decode": {"ms": round(dec_ms, 4), "GBps": round(dec_gbps, 1), "frac": round(dec_gbps / HBM_PEAK_GBPS, 4),
                           "index_ms": round(idx_ms, 4)},
                "round_trip_frac": round(2 * alg / ((enc_ms + dec_ms + idx_ms) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
    traffic_file = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(traffic_file):
        try:
            t = json.load(open(traffic_file)).get(args.content, {})
            # per-launch HBM bytes from the rocprofv3 PMC passes (profiles/*_summary.txt), same batch
            roofline["traffic"] = t.get(dom[0]) if t.get("frames_per_launch") == B else None
        except Exception:
            pass

    # ---- RCCL gather of the compressed stream to rank 0 (not on the round-trip path) ----------
    gather = None
    if dist is not None and not args.no_gather:
        from dbde_video_cpp_amd import distributed as dd
        if slot:   # the gathered stream is the concatenated form: re-encode this rank's block that way once
            codec.encode_frames(imgs, W, H, B, buf, lead, cap, first_index=rank * B, offsets=offs, nbytes=sizes)
            codec.sync()
        seg = buf[lead:lead + packed_bytes]
        recv = None
        if rank == 0:
            recv = torch.empty(world * (cap + 64), dtype=torch.uint8, device=dev)
        dd.gather_stream(seg, packed_bytes, dst=0, out=recv)      # warm-up (connection setup)
        fence()
        tg0 = time.perf_counter()
        reps = 2
        for _ in range(reps):
            stream, allsz = dd.gather_stream(seg, packed_bytes, dst=0, out=recv)
        fence()
        tg = (time.perf_counter() - tg0) / reps
        del recv, stream
        gather = {"ms": round(tg * 1e3, 3), "bytes": sum(allsz),
                  "GBps_into_root": round((sum(allsz) - allsz[0]) / tg / 1e9, 1),
                  "frames_per_s_if_serialised": round(world * B / (dt_max / args.steps + tg), 1),
                  "note": "variable-length gather of the compressed stream to rank 0 over RCCL; measured "
                          "separately, not inside the round-trip steps"}

    # ---- configs[1] literally: ONE frame per call (launch-bound; reported beside the batched value) ----
    single = None
    if rank == 0 and world == 1 and not args.no_single:
        one, ob, oo = imgs[:1], buf, out[:1]
        def step1():
            codec.encode_frames(one, W, H, 1, ob, lead, cap, first_index=0, offsets=offs[:1], nbytes=sizes[:1], slot_stride=slot)
            codec.decode_frames(ob, lead, stream_cap, offs[:1], W, H, 1, images=oo, results=res[:1])
        for _ in range(20):
            step1()
        codec.sync()
        reps1 = 300
        t1 = time.perf_counter()
        for _ in range(reps1):
            step1()
        codec.sync()
        d1 = (time.perf_counter() - t1) / reps1
        single = {"frames_per_step": 1, "us_per_round_trip": round(d1 * 1e6, 2), "frames_per_s": round(1.0 / d1, 1),
                  "identical": bool(torch.equal(oo, one)),
                  "note": "one 4096x3072 frame per encode+decode call, device-resident, back to back: bounded by "
                          "kernel launch and start-up latency, not by HBM"}

    if rank == 0:
        line = {
            "metric": "frames/s + raw-pixel GB/s, 4096x3072 U8 encode+decode round-trip",
            "value": round(fps, 1), "unit": "frames/s",
            "raw_pixel_GBps": round(fps * W * H / 1e9, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1] shape: 4096x3072 U8 frames, {args.content}, "
                                   f"{B} distinct frames per step per GPU, device-resident in and out, "
                                   + ("one concatenated stream" if args.concat else "one slot per frame"),
                       "frames_per_step_per_gpu": B, "content": args.content,
                       "layout": "concat" if args.concat else "slots",
                       "packed_over_raw": round(packed_bytes / raw, 4), "parallelism": f"frames sharded x{world}"},
            "roofline": roofline,
        }
        if single:
            line["single_frame"] = single
        if gather:
            line["gather"] = gather
        if not args.no_cpu and world == 1:
            line["cpu_baseline"] = cpu_baseline(imgs[:16].cpu().numpy(), 16)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
