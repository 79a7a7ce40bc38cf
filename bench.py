#!/usr/bin/env python3
"""bench.py -- DBDE encode+decode round trip on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5] [--content noise8|mixed|smooth|flat]
                    [--frames B] [--concat] [--only] [--no-cpu] [--no-single] [--no-gather]

One step = one pass of the hot path over one batch: encode B distinct synthetic U8 frames (resident in
HBM) into DBDE frames, then decode them back to images.  `value` (frames/s) is always the configs[1]
workload -- 4096x3072, noise8, B = 1024 frames per step per GPU, one slot per frame -- timed over exactly K
steps between barriers.  B*W*H is far beyond the 256 MiB Infinity Cache, so every step streams from HBM.

The same run then times, with the same method (HIP events on the codec's stream + host wall clock),
  contents : mixed (depths 0..8 uniform: the path that really bit-packs) and smooth, same shape
  configs  : "3" 1000 frames of 2048x2048 mixed as ONE concatenated stream, decoded as a .dbde reader would:
                 from the offsets the device stream scanner finds (dbde_hip_index_stream_async);
             "4" 1921x1081 mixed (every row unaligned, edge tiles on two sides: constant-pad path)
  single_frame : configs[1] literally, one frame per encode+decode call
  cpu_baseline : the reference (oracle/_ref) on ALL host cores
`--only` keeps just the headline leg (profiling runs); `--config 3|4|5` makes that workload the headline.
`--config 5`: the 10,000-frame stream, each rank walking its block of frames in batches through the
streaming driver (frames produced on the fly, never all resident), with the RCCL gather of batch k
overlapped with the encode of batch k+1; reported with and without the gather.

N > 1: `python bench.py --gpus N` starts N ranks itself (torch.distributed.run, one per GPU, before any GPU
call); launched under torchrun it uses the ranks it is given.  Frames are sharded by contiguous blocks
(rank g owns frames [g*B, (g+1)*B) of each step): no collective on the data path, weak scaling.
Rank 0 prints ONE JSON line.  `value` = frames all ranks round-tripped per second.
"""
import argparse
import copy
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SEED = 0xDBDE2016
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
CONFIGS = {              # BASELINE.json configs[1..4]
    2: dict(W=4096, H=3072, frames=1024, content="noise8", layout="slots", name="configs[1]: 4096x3072 U8 frames"),
    3: dict(W=2048, H=2048, frames=1000, content="mixed", layout="concat+scan",
            name="configs[2]: 1000-frame 2048x2048 stream, mixed depths 0-8"),
    4: dict(W=1921, H=1081, frames=2048, content="mixed", layout="slots",
            name="configs[3]: 1921x1081 frames (edge tiles, constant-pad path)"),
    5: dict(W=4096, H=3072, frames=10000, content="noise8", layout="stream",
            name="configs[4]: 10000-frame 4096x3072 stream sharded by frame blocks"),
}


def kernels_fingerprint():
    """Identity of the kernel sources a profile belongs to (profiles/hbm_traffic.json carries it)."""
    h = hashlib.sha256()
    for f in ("dbde_kernels.hip", "dbde_kernels.h", "dbde_bits.h"):
        h.update(open(os.path.join(ROOT, "dbde-video-cpp_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def spawn_ranks(n, argv):
    """`bench.py --gpus N` without a launcher: become the launcher.  Nothing in this process has touched
    the GPU (no torch.cuda call, no codec), and the ranks are CHILD processes -- never an exec."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


def host_cores():
    """(cores this process may really use, cores visible): the affinity mask capped by the cgroup CPU quota
    (a GPU box hands a 1-GPU job a share of the host, e.g. 16 of 256 hardware threads)."""
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(period)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, q // period)
        except Exception:
            pass
    return (min(visible, quota) if quota else visible), visible


def progress(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


_REAL_STDOUT = None


def quiet_stdout():
    """Libraries below us write banners to file descriptor 1 (RCCL prints its version block there when a communicator is
    created).  The contract is ONE JSON line on stdout: everything else written to fd 1 goes to stderr from here on, and
    emit() writes the line to the real stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(line):
    out = _REAL_STDOUT or sys.stdout
    out.write(json.dumps(line) + "\n")
    out.flush()


def cpu_baseline(make_frames, W, H, budget_s=8.0):
    """Reference (oracle/_ref, kind 'reference') or oracle port on ALL the host cores this job may use: every
    thread round-trips its own frame until the budget has passed (the reference's own timing loop brackets
    dbde_pack_frame / dbde_unpack_frame the same way, dbde_util_test.cpp:326-349).  `make_frames(content, n)` ->
    host frames.  noise8 (all depth 8: the reference's SIMD fast path) is `value`; `mixed` (depths 0..8: its scalar
    bit loops, dbde_util.cpp:87-100,234-242) is reported beside it.  Single-thread figures are HOT: one untimed
    round trip (first-touch page faults, caches), then the mean of >= 5."""
    from oracle_ffi import Oracle, Reference
    impl, kind = (Reference(), "reference") if Reference.available() else (Oracle(), "port")
    cores, visible = host_cores()
    n_distinct = min(cores, 32)

    def leg(content, budget):
        images_host = make_frames(content, n_distinct)
        impl.time_roundtrip(images_host[0:1], 1, W, H, 1)                 # warm-up, untimed
        reps = 5
        t1 = impl.time_roundtrip(images_host[0:1], 1, W, H, reps)[0] / reps
        chunk = max(1, int(0.5 / max(t1, 1e-4)))                         # round trips per call (about half a second)
        done, bad = [0] * cores, [0] * cores
        deadline = time.time() + budget

        def work(k):
            j = k % n_distinct
            while time.time() < deadline:
                r = impl.time_roundtrip(images_host[j:j + 1], 1, W, H, chunk)
                done[k] += chunk
                bad[k] += r[3]

        t0 = time.time()
        th = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
        [t.start() for t in th]
        [t.join() for t in th]
        wall = time.time() - t0
        return {"value": round(sum(done) / wall, 2), "single_thread_frames_per_s": round(1.0 / t1, 2),
                "single_thread_reps": reps, "round_trips": sum(done), "seconds": round(wall, 1),
                "mismatched_pixels": int(sum(bad))}

    n8 = leg("noise8", budget_s)
    mx = leg("mixed", budget_s * 0.75)
    return {"value": n8["value"], "unit": "frames/s", "cores": cores, "kind": kind, "host_cores_visible": visible,
            "sample": f"{cores} threads (every core this job may use; {visible} visible) round-tripping one "
                      f"{W}x{H} noise8 frame each for {n8['seconds']} s ({n8['round_trips']} round trips); "
                      f"single thread: mean of {n8['single_thread_reps']} round trips after one untimed",
            "single_thread_frames_per_s": n8["single_thread_frames_per_s"], "mismatched_pixels": n8["mismatched_pixels"],
            "mixed": {"value": mx["value"], "single_thread_frames_per_s": mx["single_thread_frames_per_s"],
                      "seconds": mx["seconds"], "round_trips": mx["round_trips"], "mismatched_pixels": mx["mismatched_pixels"],
                      "note": "depths 0..8 uniform: the reference's scalar bit-packing path"}}


def golden_sha(W, H, content):
    """{frame: (sha256, bytes)} of the reference's own packed frames for this shape / content, from the committed
    fixture (tests/golden/manifest.json "big": made by tests/golden/make_golden.py with the real reference).  Data,
    not code: no oracle is imported here."""
    try:
        m = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))
    except Exception:
        return {}
    return {e["frame"]: (e["packed_sha"], e["packed_bytes"]) for e in m.get("big", [])
            if e["W"] == W and e["H"] == H and e["mode"] == content}


EXIT_EXCHANGE_FAILED = 3   # a rank that gave up on a stuck or failed exchange leaves with this code, never with 0


class Watchdog:
    """The exchange legs are the one part of a multi-rank run that can hang (a collective nobody answers).  If the
    block does not finish in `seconds`, `on_timeout()` runs (rank 0 prints the line it has) and the process leaves
    NON-ZERO: it has touched the GPU and failed, and the launcher (torchrun / spawn_ranks) must see that.  The process
    ends there -- it is never restarted or replaced."""

    def __init__(self, seconds, on_timeout):
        self.t = threading.Timer(seconds, self._fire)
        self.t.daemon = True
        self.on_timeout = on_timeout

    def _fire(self):
        try:
            self.on_timeout()
        finally:
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(EXIT_EXCHANGE_FAILED)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


class Bench:
    def __init__(self, args, dv, torch, dist, world, rank, local):
        self.args, self.dv, self.torch, self.dist = args, dv, torch, dist
        self.world, self.rank, self.local = world, rank, local
        self.dev = torch.device("cuda", local)
        self.codec = dv.Codec(local)
        self.failed_exchange = False

    def fence(self):
        self.torch.cuda.synchronize(self.dev)
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def case(self, W, H, B, content, layout, steps, warmup, check=True):
        """Times `steps` encode+decode passes over B frames.  layout: 'slots' | 'concat' | 'concat+scan'
        (decode from the offsets the device stream scanner finds, not from the encoder's)."""
        torch, dv, codec, rank = self.torch, self.dv, self.codec, self.rank
        if rank == 0:
            progress(f"case {W}x{H} x{B} {content} {layout}: {steps} steps")
        imgs = codec.synth_frames(content, SEED, rank * B, B, W, H)
        slot = ((dv.max_frame_bytes(W, H) + 255) // 256) * 256 if layout == "slots" else 0
        buf, lead, cap = codec.alloc_stream(W, H, B, slot_stride=slot)
        out = torch.empty_like(imgs)
        offs = torch.empty(B, dtype=torch.int64, device=self.dev)
        sizes = torch.empty(B, dtype=torch.int64, device=self.dev)
        res = torch.empty((B, 4), dtype=torch.int64, device=self.dev)
        scan = layout == "concat+scan"
        packed_bytes = [0]
        found = torch.empty(B, dtype=torch.int64, device=self.dev) if scan else None

        count = torch.zeros(1, dtype=torch.int32, device=self.dev) if scan else None

        def step():
            codec.encode_frames(imgs, W, H, B, buf, lead, cap, first_index=rank * B, offsets=offs, nbytes=sizes,
                                slot_stride=slot)
            if scan:
                # a reader's view: only the bytes and their total length are known.  The device finds the frame
                # starts (dbde_hip_index_stream_async: the speculative segment-parallel walk) and the decode,
                # enqueued behind it, takes its offsets from there
                total = packed_bytes[0] or cap
                codec.index_stream_async(buf, lead, total, W, H, B, found, count)
                codec.decode_frames(buf, lead, total, found, W, H, B, images=out, results=res)
            else:
                codec.decode_frames(buf, lead, cap, offs, W, H, B, images=out, results=res)

        if scan:   # the stream length a reader would know (file size): one encode to learn it
            codec.encode_frames(imgs, W, H, B, buf, lead, cap, first_index=rank * B, offsets=offs, nbytes=sizes)
            codec.sync()
            packed_bytes[0] = int((offs[-1] + sizes[-1]).item())
        for _ in range(warmup):
            step()
        codec.sync()
        sha_ok = None
        if check:   # parity gate on the measured configuration
            assert torch.equal(out, imgs), f"round trip mismatch ({W}x{H} {content} {layout})"
            # ... and not on round-trip identity alone (an encoder / decoder pair wrong in the same way would pass it):
            # the packed bytes of frames 0 and 3 against the SHA-256 of the REFERENCE's output for the same frames
            # EVERY rank is held to it: rank r's frames are r*B .. r*B + B - 1, the fixture holds frames r*1024 and
            # r*1024 + 3 of the headline shape for r = 0..7 (tests/golden/make_golden.py RANK_FRAMES)
            want = {fr - rank * B: v for fr, v in golden_sha(W, H, content).items() if 0 <= fr - rank * B < B}
            if want:
                o_h, s_h = offs.cpu().numpy(), sizes.cpu().numpy()
                sha_ok = True
                for f, (sha, nbytes) in want.items():
                    got = buf[lead + int(o_h[f]):lead + int(o_h[f]) + int(s_h[f])].cpu().numpy().tobytes()
                    sha_ok = sha_ok and len(got) == nbytes and hashlib.sha256(got).hexdigest() == sha
                assert sha_ok, f"rank {rank}: packed frames differ from the reference's ({W}x{H} {content})"
            if scan:
                assert torch.equal(found, offs), "stream scanner offsets differ from the encoder's"
                assert int(count.item()) == B, "stream scanner lost frames"
        packed = int(sizes.sum().item())
        # per rank: 1 = checked against the reference's SHA-256 and equal, -1 = no fixture for this rank's frames
        # (a mismatch has already raised above)
        sha_ranks = [sha_ok]
        if self.dist is not None:
            flag = torch.tensor([1 if sha_ok else -1], dtype=torch.int32, device=self.dev)
            allf = [torch.zeros_like(flag) for _ in range(self.world)]
            self.dist.all_gather(allf, flag)
            sha_ranks = [True if int(x.item()) == 1 else None for x in allf]

        codec.timing(True)
        codec.timing_read(reset=True)
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.fence()
        dt = time.perf_counter() - t0
        tk = codec.timing_read(reset=True)
        codec.timing(False)
        codec.sync()
        t_all = torch.tensor([dt], dtype=torch.float64, device=self.dev)
        if self.dist is not None:
            self.dist.all_reduce(t_all, op=self.dist.ReduceOp.MAX)
        dt_max = float(t_all.item())

        per_step = lambda k: tk[k][0] / steps            # kernel time of one step (a step may launch a kernel several times)
        enc_ms, idx_ms, dec_ms, scan_ms = per_step("encode"), per_step("decode_index"), per_step("decode"), per_step("scan")
        raw = B * W * H
        alg = raw + packed                              # encode reads raw, writes packed; decode the reverse
        gbps = lambda ms: alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        r = {"W": W, "H": H, "frames_per_step_per_gpu": B, "content": content, "layout": layout,
             "frames_per_s": round(self.world * B * steps / dt_max, 1),
             "raw_pixel_GBps": round(self.world * B * steps / dt_max * W * H / 1e9, 1),
             "ms_per_step": round(dt_max / steps * 1e3, 4), "steps": steps,
             "packed_over_raw": round(packed / raw, 4), "algorithmic_bytes_per_launch": alg,
             "encode": {"ms": round(enc_ms, 4), "GBps": round(gbps(enc_ms), 1), "frac": round(gbps(enc_ms) / HBM_PEAK_GBPS, 4)},
             "decode": {"ms": round(dec_ms, 4), "GBps": round(gbps(dec_ms), 1), "frac": round(gbps(dec_ms) / HBM_PEAK_GBPS, 4),
                        "index_ms": round(idx_ms, 4)},
             "round_trip_frac": round(2 * alg / (dt_max / steps) / 1e9 / HBM_PEAK_GBPS, 4),
             "identical": bool(check), "packed_sha_ok": sha_ok, "packed_sha_ok_ranks": sha_ranks}
        if scan:
            r["decode"]["scan_ms"] = round(scan_ms, 4)
            r["decode"]["note"] = ("frame starts found on the device (speculative segment-parallel walk, exact by "
                                   "construction), then one decode launch; scan_ms is inside the step")
        r["_dt_max"], r["_packed"] = dt_max, packed
        del imgs, buf, out
        return r

    def case16(self, W=4096, H=3072, n=128, steps=10):
        """DBDE16 (the higher-bit-depth extension of SURVEY 8f rank 4; parity unpinned: the reference defines no such
        format): n frames of U16 pixels, per-tile depth uniform in 0..16, encode + decode per step, round trip checked
        for identity.  Reported beside the headline, never part of `value`."""
        torch, codec = self.torch, self.codec
        progress(f"DBDE16 {W}x{H} x{n}: {steps} steps")
        g = torch.Generator(device="cuda").manual_seed(SEED & 0x7FFFFFFF)
        d = torch.randint(0, 17, (n, H // 8, W // 8), device="cuda", generator=g, dtype=torch.int32)
        dd = d.repeat_interleave(8, 1).repeat_interleave(8, 2)
        mask = (torch.ones_like(dd) << dd) - 1
        noise = torch.randint(0, 65536, (n, H, W), device="cuda", generator=g, dtype=torch.int32) & mask
        base = torch.randint(0, 32768, (n, H // 8, W // 8), device="cuda", generator=g, dtype=torch.int32).repeat_interleave(8, 1).repeat_interleave(8, 2)
        imgs = torch.minimum(base, 65535 - mask).add_(noise).to(torch.int16).contiguous()   # two's complement bits = the U16 pixels
        del d, dd, mask, noise, base
        cap = n * int(codec.L.dbde16_hip_max_frame_bytes(W, H))
        buf = torch.empty(32 + cap + 64, dtype=torch.uint8, device="cuda")
        out = torch.empty_like(imgs)
        for _ in range(2):
            offs, sizes = codec.encode_frames16(imgs, W, H, n, buf, 32, cap)
            codec.decode_frames16(buf, 32, cap, offs, W, H, n, images=out)
        codec.sync()
        identical = bool(torch.equal(out, imgs))
        packed = int(sizes.sum().item())
        codec.timing(True)
        codec.timing_read(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            offs, sizes = codec.encode_frames16(imgs, W, H, n, buf, 32, cap)
            codec.decode_frames16(buf, 32, cap, offs, W, H, n, images=out)
        codec.sync()
        dt = (time.perf_counter() - t0) / steps
        tk = codec.timing_read()
        codec.timing(False)
        raw = n * W * H * 2
        enc, dec = tk["encode"][0] / steps, tk["decode"][0] / steps
        return {"workload": f"{n} frames of {W}x{H} U16, per-tile depth uniform in 0..16, concatenated", "parity": "unpinned (extension)",
                "frames_per_s": round(n / dt, 1), "packed_over_raw": round(packed / raw, 4), "identical": identical,
                "encode": {"ms": round(enc, 4), "frac": round((raw + packed) / enc / 1e6 / HBM_PEAK_GBPS, 4)},
                "decode": {"ms": round(dec, 4), "frac": round((raw + packed) / dec / 1e6 / HBM_PEAK_GBPS, 4),
                           "index_ms": round(tk["decode_index"][0] / steps, 4)}}

    def single_frame(self, W, H, content, reps=300):
        """configs[1] literally: ONE frame per encode+decode call, device-resident, back to back."""
        torch, dv, codec = self.torch, self.dv, self.codec
        progress(f"single frame {W}x{H} {content}")
        one = codec.synth_frames(content, SEED, 0, 1, W, H)
        buf, lead, cap = codec.alloc_stream(W, H, 1)
        out = torch.empty_like(one)
        offs = torch.zeros(1, dtype=torch.int64, device=self.dev)
        sizes = torch.empty(1, dtype=torch.int64, device=self.dev)
        res = torch.empty((1, 4), dtype=torch.int64, device=self.dev)

        def step1():
            codec.encode_frames(one, W, H, 1, buf, lead, cap, first_index=0, offsets=offs, nbytes=sizes)
            codec.decode_frames(buf, lead, cap, offs, W, H, 1, images=out, results=res)
        for _ in range(20):
            step1()
        codec.sync()
        t1 = time.perf_counter()
        for _ in range(reps):
            step1()
        codec.sync()
        d1 = (time.perf_counter() - t1) / reps
        alg = 2 * (W * H + int(sizes[0].item()))
        return {"frames_per_step": 1, "us_per_round_trip": round(d1 * 1e6, 2), "frames_per_s": round(1.0 / d1, 1),
                "GBps": round(alg / d1 / 1e9, 1), "identical": bool(torch.equal(out, one)),
                "note": "one frame per encode+decode call, device-resident, back to back; the 25 MB working set "
                        "stays in the 256 MiB Infinity Cache, so this is a latency figure, not an HBM one"}

    def stream(self, W, H, n_total, batch, content, gather_mode):
        """configs[4]: every rank walks its contiguous block of the n_total frames through the streaming
        driver; returns kernels-only and with-gather columns."""
        from dbde_video_cpp_amd import distributed as dd
        from dbde_video_cpp_amd.streaming import RoundTripStream
        torch, dv = self.torch, self.dv
        lo, hi = dd.shard_frames(n_total, self.rank, self.world)
        side = torch.cuda.Stream(self.dev)
        src_codec = dv.Codec(self.local, stream=side)            # frame source: its own stream

        def source(first, n, out):
            src_codec.synth_frames(content, SEED, first, n, W, H, out=out)

        cols = {}
        native = None
        if gather_mode == "native":
            # the C-ABI gather (csrc/dbde_gather.cpp): its own RCCL communicator; the rendezvous token is made on rank 0
            # and handed to the other ranks through the process group that is already up
            uid = torch.zeros(dv.GATHER_ID_BYTES, dtype=torch.uint8, device=self.dev)
            if self.rank == 0:
                uid.copy_(torch.from_numpy(dv.gather_unique_id()))
            if self.dist is not None:
                self.dist.broadcast(uid, src=0)
            err = None
            try:
                native = dv.Gather(self.codec, uid.cpu().numpy(), self.world, self.rank, root=0)
            except Exception as e:
                err = e
            # the outcome is AGREED before anyone moves on: a rank that failed alone would otherwise take the fallback
            # (or skip the leg) while its peers sit in the native collective -- mismatched collectives, then the watchdog
            if self.agree_failed(err is not None):
                if native is not None:
                    native.close()
                raise RuntimeError(f"native gather unavailable on at least one rank (this rank: {err!r})")
        try:
            return self._stream_columns(W, H, n_total, batch, content, gather_mode, native, source, side, lo, hi, cols), (lo, hi)
        finally:   # also on the exception path: an open Gather's destructor would wait on its comm stream
            if native is not None:
                native.close()
            src_codec.close()

    def agree_failed(self, failed):
        """True on every rank if `failed` is true on any (one small all-reduce over the process group that is up)."""
        if self.dist is None:
            return bool(failed)
        t = self.torch.tensor([1 if failed else 0], dtype=self.torch.int32, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return bool(int(t.item()))

    def _stream_columns(self, W, H, n_total, batch, content, gather_mode, native, source, side, lo, hi, cols):
        from dbde_video_cpp_amd.streaming import RoundTripStream
        torch = self.torch
        for name, g in (("resident_ring", None), ("kernels_only", None), ("with_gather", gather_mode)):
            if name == "with_gather" and g is None:
                continue
            if self.rank == 0:
                progress(f"stream {W}x{H} {n_total} frames in batches of {batch}, {content}, {name}")
            if name == "resident_ring":   # the same driver on two resident batches: no frame source beside the codec
                rts = RoundTripStream(self.codec, W, H, batch)
                for k in range(2):
                    self.codec.synth_frames(content, SEED, lo + k * batch, batch, W, H, out=rts.inp[k])
            else:
                rts = RoundTripStream(self.codec, W, H, batch, source=source, source_stream=side, gather=g,
                                      native=native if g == "native" else None, world=self.world, rank=self.rank)
            self.fence()
            rounds = -(-(-(-n_total // self.world)) // batch)      # batches of the largest rank block
            err = None
            try:
                r = rts.run(lo, hi - lo, self.world, self.rank, rounds=rounds)
            except Exception as e:
                if not g:
                    raise
                err = e
            # agreed across ranks as well: the leg is taken or dropped by all of them
            if g and self.agree_failed(err is not None):
                cols[name] = {"error": (f"{type(err).__name__}: {err}" if err is not None else "failed on another rank")[:400]}
                self.failed_exchange = True   # the exchange failed: the other columns stand
                break
            t = torch.tensor([r["seconds"]], dtype=torch.float64, device=self.dev)
            if self.dist is not None:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            sec = float(t.item())
            cols[name] = {"frames_per_s": round(n_total / sec, 1), "seconds": round(sec, 4),
                          "batches_per_rank": r["batches"]}
            if name == "resident_ring":
                cols[name]["note"] = "two resident input batches re-used: the codec and the pipeline alone"
            if name == "kernels_only":
                cols[name]["note"] = ("every frame distinct, produced on the fly by a generator kernel on a side stream; "
                                      "the generator's own HBM writes (one image per frame) compete with the codec")
            if g:
                cols[name]["gathered_bytes"] = r["gathered_bytes"]
                cols[name]["GBps_into_root"] = round((r["gathered_bytes"] - r["packed_bytes"]) / sec / 1e9, 1)
                cols[name]["exchange"] = {"native": "C-ABI dbde_hip_gather_* over librccl (ncclAllGather of the counts, grouped "
                                                    "ncclSend/ncclRecv; rank 0 encodes into its window)",
                                          "nccl": "torch.distributed (RCCL) all_gather + batch_isend_irecv",
                                          "host": "gloo rehearsal through pinned host memory"}[g]
            del rts
        return cols


def dry_run(args, world, rank, dist):
    """Launch-contract rehearsal without a GPU (tests/test_bench_contract.py): N ranks, sharding, barrier,
    max-over-ranks and the gloo gather of a fabricated stream.  Measures nothing and says so."""
    import torch
    from dbde_video_cpp_amd import distributed as dd
    lo, hi = dd.shard_frames(args.frames * world, rank, world)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    gathered = None
    line = {"metric": "frames/s + raw-pixel GB/s, 4096x3072 U8 encode+decode round-trip", "value": 0.0,
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": None, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "none (dry run: launch contract only, no GPU work, not a measurement)",
            "dry_run": True, "frames_of_rank0": [lo, hi], "gathered_bytes": None,
            "config": {"workload": "dry run"}}
    if dist is not None:
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        line["ms_per_step"] = round(float(t.item()) / args.steps * 1e3, 4)
        finished = copy.deepcopy(line)

        def give_up():
            if rank == 0:
                finished["gather"] = {"error": "the exchange leg did not finish (a collective was never answered); "
                                               f"the process leaves with code {EXIT_EXCHANGE_FAILED}"}
                emit(finished)

        # the exchange leg as the real run has it: under the watchdog.  DBDE_BENCH_DRY_FAIL=skip_gather makes the last
        # rank walk past the gather, so that the others wait for a collective nobody answers (tests/test_bench_contract.py)
        with Watchdog(float(os.environ.get("DBDE_BENCH_WATCHDOG_S", "300")), give_up):
            if not (os.environ.get("DBDE_BENCH_DRY_FAIL") == "skip_gather" and rank == world - 1):
                seg = torch.full((100 + 7 * rank,), rank, dtype=torch.uint8)
                stream, sizes = dd.gather_stream(seg, seg.numel(), dst=0)
                gathered = sum(sizes)
            else:
                time.sleep(3600)   # (the launcher ends this rank when rank 0 has left)
        line["gathered_bytes"] = gathered
    else:
        line["ms_per_step"] = round(float(t.item()) / args.steps * 1e3, 4)
    if rank == 0:
        emit(line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)   # 60 x 8.4 ms: half a second of timed GPU work for the headline
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5], help="BASELINE.json configs[config-1] as the headline")
    ap.add_argument("--frames", type=int, default=0, help="frames per step per GPU (default: the config's)")
    ap.add_argument("--content", default=None, choices=["noise8", "mixed", "smooth", "flat"])
    ap.add_argument("--concat", action="store_true", help="one concatenated stream instead of one slot per frame")
    ap.add_argument("--batch", type=int, default=250, help="frames per batch of the streaming driver (config 5)")
    ap.add_argument("--only", action="store_true", help="headline leg only (profiling runs)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--gather", default="native", choices=["native", "nccl", "none"],
                    help="exchange step of the streaming legs: the C-ABI's RCCL gather (default) or torch.distributed's")
    ap.add_argument("--no-single", action="store_true", help="skip the one-frame-per-call leg")
    ap.add_argument("--no-check", action="store_true", help="(experiments) skip the round-trip parity gate")
    ap.add_argument("--dry-run", action="store_true", help="launch contract only: no GPU, no codec (CPU tests)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")

    quiet_stdout()
    import torch
    dist = None
    # rehearsal on a one-GPU box: DBDE_BENCH_REHEARSAL=1 puts every rank on cuda:0 over gloo (exercises the
    # launch contract, sharding, barrier, max-over-ranks and the staged gather pipeline; not a measurement)
    # (DBDE_BENCH_REHEARSAL=nccl: the same with the RCCL backend, where the runtime accepts two ranks on one device)
    rehearsal = os.environ.get("DBDE_BENCH_REHEARSAL") in ("1", "nccl")
    rehearsal_nccl = os.environ.get("DBDE_BENCH_REHEARSAL") == "nccl"
    # DBDE_BENCH_FORCE_DIST=1: build the process group even for one rank (a one-GPU box can then run the RCCL
    # init / barrier / all-reduce / size all-gather calls of the N > 1 path; RANK, WORLD_SIZE=1, MASTER_* from the env)
    if world > 1 or os.environ.get("DBDE_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:   # DBDE_BENCH_FORCE_DIST without a launcher: a one-rank rendezvous on the loopback interface
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1")):
                os.environ.setdefault(k, v)
            if "MASTER_PORT" not in os.environ:
                sk = socket.socket()
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
                sk.close()
        if args.dry_run:
            dist.init_process_group("gloo")
        elif rehearsal:
            local = 0
            torch.cuda.set_device(0)
            if rehearsal_nccl:
                dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
            else:
                dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            import datetime
            # collectives that hang abort after two minutes instead of ten: the gather leg below is guarded
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=120))
        assert dist.get_world_size() == args.gpus
    if args.dry_run:
        args.frames = args.frames or 16
        dry_run(args, world, rank, dist)
        if dist is not None:
            dist.destroy_process_group()
        return
    if world == 1:
        torch.cuda.set_device(0)
        local = 0

    import dbde_video_cpp_amd as dv
    b = Bench(args, dv, torch, dist, world, rank, local)
    cfg = dict(CONFIGS[args.config])
    W, H = cfg["W"], cfg["H"]
    content = args.content or cfg["content"]
    gather_failed = False
    # the exchange step: "native" = the C-ABI gather (dbde_hip_gather_*, librccl), "nccl" = the same through
    # torch.distributed, "host" = gloo rehearsal.  One rank needs no process group for the native form.
    if args.no_gather or args.gather == "none":
        gather_mode = None
    elif rehearsal and not rehearsal_nccl:
        gather_mode = "host"
    elif args.gather == "nccl":
        gather_mode = "nccl" if dist is not None else None
    else:
        gather_mode = "native"

    line = {"metric": "frames/s + raw-pixel GB/s, 4096x3072 U8 encode+decode round-trip", "unit": "frames/s",
            "n_gpus": dist.get_world_size() if dist is not None else 1, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "vs_baseline": None, "dtype": "u8", "data": "synthetic"}
    if rehearsal:
        line["rehearsal"] = "all ranks on cuda:0 over gloo: launch-contract check, NOT a scaling measurement"

    if args.config == 5:
        # ---- the 10,000-frame stream: strong scaling over frame blocks, streaming driver -------------------
        n_total = args.frames or cfg["frames"]
        cols, (lo, hi) = b.stream(W, H, n_total, args.batch, content, gather_mode)
        head = cols["with_gather"] if "frames_per_s" in cols.get("with_gather", {}) else cols["kernels_only"]
        gather_failed = b.failed_exchange
        line.update({"value": head["frames_per_s"], "raw_pixel_GBps": round(head["frames_per_s"] * W * H / 1e9, 1),
                     "ms_per_step": round(head["seconds"] / max(head["batches_per_rank"], 1) * 1e3, 4),
                     "steps": head["batches_per_rank"], "warmup": 0, "scaling": "strong",
                     "config": {"workload": f"BASELINE {cfg['name']}: {n_total} frames {W}x{H} {content}, rank blocks of "
                                            f"{hi - lo}, batches of {args.batch} through the streaming driver (frames "
                                            "produced on the fly on a side stream, two input and two stream slots)",
                                "content": content, "batch": args.batch, "parallelism": f"frame blocks x{world}"},
                     "stream": cols,
                     "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None,
                                  "traffic": None, "note": "streaming run: see the default run for the kernels' roofline"}})
    else:
        B = args.frames or cfg["frames"]
        layout = "concat" if args.concat else cfg["layout"]
        r = b.case(W, H, B, content, layout, args.steps, args.warmup, check=not args.no_check)
        enc, dec = r["encode"], r["decode"]
        dom = ("dbde::encode_kernel", enc) if enc["ms"] >= dec["ms"] else ("dbde::decode_kernel", dec)
        roofline = {"bound": "hbm", "kernel": dom[0], "achieved": dom[1]["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(dom[1]["GBps"] / HBM_PEAK_GBPS, 4), "traffic": None, "traffic_source": None,
                    "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"], "launch_ms": dom[1]["ms"],
                    "encode": enc, "decode": dec, "round_trip_frac": r["round_trip_frac"]}
        # HBM bytes per launch from the rocprofv3 PMC passes (profiles/*_summary.txt): replayed from the
        # committed record ONLY when it was taken on these kernel sources and this workload, else null
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                rec = json.load(open(tf))
                t = rec.get(f"cfg{args.config}_{content}", {})
                if rec.get("kernels_sha") == kernels_fingerprint() and t.get("frames_per_launch") == B and t.get(dom[0]):
                    roofline["traffic"] = t[dom[0]]
                    roofline["traffic_source"] = (f"replayed: rocprofv3 --pmc passes of {rec.get('tag')} taken at revision "
                                                  f"{rec.get('revision', '?')}"
                                                  + (f", re-keyed to these sources by profiles/rekey.py (instruction streams identical, isa {rec.get('isa_sha')})"
                                                     if rec.get("rekeyed") else " on these kernel sources"))
            except Exception:
                pass
        line.update({"value": r["frames_per_s"], "raw_pixel_GBps": r["raw_pixel_GBps"], "ms_per_step": r["ms_per_step"],
                     "scaling": "weak",
                     "config": {"workload": f"BASELINE {cfg['name']}, {content}, {B} distinct frames per step per GPU, "
                                            f"device-resident in and out, layout {layout}",
                                "frames_per_step_per_gpu": B, "content": content, "layout": layout, "W": W, "H": H,
                                "packed_over_raw": r["packed_over_raw"], "parallelism": f"frames sharded x{world}"},
                     "roofline": roofline})
        strip = lambda d: {k: v for k, v in d.items() if not k.startswith("_")}
        if not args.only:
            sub_steps = max(3, args.steps // 2)
            # ---- the bit-packing path and low-entropy content at the headline shape ----------------------------
            line["contents"] = {}
            for c in ("mixed", "smooth"):
                if c != content:
                    line["contents"][c] = strip(b.case(W, H, B, c, layout, sub_steps, 2))
            # ---- the other single-GPU configs, same method -----------------------------------------------------
            line["configs"] = {}
            for k in (2, 3, 4):
                if k != args.config and world == 1:
                    c = CONFIGS[k]
                    line["configs"][str(k)] = strip(b.case(c["W"], c["H"], c["frames"], c["content"], c["layout"], sub_steps, 2))
                    line["configs"][str(k)]["workload"] = c["name"]
            if world == 1 and args.config == 2:
                # configs[3] again with as many bytes per launch as the headline (BASELINE gives that config no frame
                # count; the 2048-frame figure above is the one earlier rounds quoted): the persistent encoder's ramp and
                # tail weigh 10 % on a 1.3 ms launch and 3 % on a 4 ms one
                c = CONFIGS[4]
                line["configs"]["4_headline_bytes"] = strip(b.case(c["W"], c["H"], 6144, c["content"], c["layout"], max(3, sub_steps // 2), 1))
                line["configs"]["4_headline_bytes"]["workload"] = c["name"] + ", 6144 frames per launch (12.8 GB of pixels, as configs[1])"
            if world == 1 and args.config == 2:
                # shapes whose kernel forms differ from the configs' (DESIGN.md 4.2): same method, fewer steps, each gated on
                # the reference's SHA-256 like the configs (tests/golden: made with the real reference)
                line["shapes"] = {}
                for (sw, sh_, sn, note) in ((1080, 1920, 2048, "portrait HD: 8-byte rows, staged decode at 79 % chunk fill"),
                                            (1366, 768, 4096, "odd rows: any-geometry encoder, 192-thread staged decode"),
                                            (1440, 900, 2048, "16-byte rows, not whole cache lines: direct 16-byte stores"),
                                            (720, 1280, 4096, "16-byte rows, whole-tile-row chunks: direct or staged per chunk"),
                                            (64, 64, 262144, "64 tiles: whole frames per wave (encode) / per 256-thread persistent workgroup (decode_mid_kernel, round 4)"),
                                            (72, 72, 262144, "81 tiles: whole frames per workgroup (encode: staged through LDS; decode: persistent, software-pipelined, round 4)"),
                                            (128, 128, 65536, "256 tiles: the largest frames of the small-frame decoder"),
                                            (160, 120, 65536, "300 tiles: three whole frames per 512-thread workgroup on the encode side (round 4)"),
                                            (320, 240, 16384, "1200 tiles: above the whole-frame forms, two chunks per frame")):
                    try:
                        r_ = strip(b.case(sw, sh_, sn, "mixed", "slots", max(3, sub_steps // 3), 1))
                        r_["workload"] = f"{sn} frames of {sw}x{sh_}, mixed, slots; {note}"
                        line["shapes"][f"{sw}x{sh_}"] = r_
                    except Exception as e:   # reported, never fatal to the line
                        line["shapes"][f"{sw}x{sh_}"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        # ---- gather pipeline (N > 1): BASELINE configs[4] -- the 10,000-frame 4096x3072 stream in rank blocks, batch k's
        # compressed bytes travelling to rank 0 while batch k+1 is encoded.  The headline and the contents above are
        # complete at this point; this leg is the one part that cannot be rehearsed on a one-GPU box (rank-to-rank RCCL
        # traffic), so it runs under a watchdog: if a collective is never answered, rank 0 prints the line it has.
        if gather_mode and world > 1 and not args.only:
            c5 = CONFIGS[5]

            # what the line holds BEFORE the leg starts: the watchdog fires on a timer thread while this thread may be in
            # the middle of adding to `line`, so it prints this finished copy, never the live dictionary
            finished = copy.deepcopy(line)

            def give_up():
                if rank == 0:
                    finished["gather"] = {"error": f"the exchange leg did not finish within {watchdog_s:.0f} s (a collective was never answered); "
                                                   f"the process leaves with code {EXIT_EXCHANGE_FAILED}"}
                    emit(finished)

            watchdog_s = float(os.environ.get("DBDE_BENCH_WATCHDOG_S", "300"))
            with Watchdog(watchdog_s, give_up):
                for mode in ([gather_mode, "nccl"] if gather_mode == "native" else [gather_mode]):
                    try:
                        cols, _ = b.stream(c5["W"], c5["H"], c5["frames"], args.batch, c5["content"], mode)
                        gather_failed = b.failed_exchange
                        line["gather"] = {"workload": f"BASELINE {c5['name']}: {c5['frames']} frames in rank blocks, batches of {args.batch}",
                                          "resident_ring": cols.get("resident_ring"), "kernels_only": cols.get("kernels_only"),
                                          "with_gather": cols.get("with_gather"),
                                          "note": "streaming driver: variable-length gather of each batch's compressed bytes to "
                                                  "rank 0 overlapped with the next batch's encode+decode; root ingress over xGMI "
                                                  "(7 links) bounds the with_gather column on incompressible content"}
                        break
                    except Exception as e:   # reported, not fatal: `value` does not depend on it
                        line.setdefault("gather_errors", []).append(f"{mode}: {type(e).__name__}: {e}"[:400])
                        if mode == "nccl" or gather_mode != "native":
                            line["gather"] = {"error": line["gather_errors"][-1]}
                            gather_failed = True
        if rank == 0 and world == 1 and not args.only:
            try:
                line["dbde16"] = b.case16()
            except Exception as e:   # an extension: reported, never fatal to the line
                line["dbde16"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if rank == 0 and world == 1 and not args.no_single and not args.only:
            line["single_frame"] = b.single_frame(4096, 3072, "noise8")
            line["single_frame"]["mixed_us_per_round_trip"] = b.single_frame(4096, 3072, "mixed")["us_per_round_trip"]

    if rank == 0:
        if not args.no_cpu and world == 1 and not args.only:
            progress("cpu baseline")
            mk = lambda content, n: b.codec.synth_frames(content, SEED, 0, n, 4096, 3072).cpu().numpy()
            line["cpu_baseline"] = cpu_baseline(mk, 4096, 3072)
        emit(line)
    if dist is not None:
        if gather_failed:    # the communicator may be unusable: leave without the farewell collective -- and NOT with 0:
            sys.stdout.flush()   # the line (printed above, with the error in it) comes from a run whose exchange failed
            sys.stderr.flush()
            os._exit(EXIT_EXCHANGE_FAILED)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
