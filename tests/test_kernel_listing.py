"""CPU, compile only: static checks on the gfx950 listing of the kernels (`make asm`, no GPU).

The persistent encoder keeps the NEXT chunk's eight 16-byte pixel loads in flight while it reduces the
CURRENT chunk, whose pixels were requested one step earlier.  Hardware retires a wave's vector-memory
operations through one in-order counter, and the compiler places `s_waitcnt vmcnt(N)` from a STATIC count of
what was issued after the value it needs.  Two things can silently go wrong when the kernel is edited:

  * the loads of a step are issued conditionally (or behind a branch merge): the compiler can no longer
    count them and waits for everything -> the software pipeline overlaps nothing (that
    cost 5 points of roofline before it was found);
  * a statically countable number of stores lands between a chunk's loads and their use: the waits become
    LOOSER than the number of loads (vmcnt(23) ...), and the kernel then depends on stores and loads
    retiring strictly in issue order.  An experiment of round 1 that did exactly that produced wrong payload
    bytes once in a few hundred chunks.

Both show in the listing: in the steady-state loop every group of eight pixel loads must be followed by the
waits vmcnt(15), vmcnt(14), ..., vmcnt(8) -- exactly the eight loads just issued stay outstanding, nothing
else is assumed about younger operations.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dbde-video-cpp_amd", "csrc")


@pytest.fixture(scope="module")
def listing():
    r = subprocess.run(["make", "-s", "-C", CSRC, "asm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(os.path.join(CSRC, "dbde_kernels.s")).read()


def function_body(listing, mangled):
    m = re.search(r"^%s:[^\n]*\n(.*?)\n\.Lfunc_end" % re.escape(mangled), listing, re.S | re.M)
    assert m, f"{mangled} not in the listing"
    return m.group(1).splitlines()


def pixel_load_groups(lines):
    """[(index of the first load, index of the last, [vmcnt immediates up to the next barrier, repeats dropped])] for
    every group of non-temporal 16-byte loads (a step's pixel fetch; address arithmetic may sit between the loads, a
    label, a branch or a vector-memory wait ends a group)."""
    idx = [i for i, ln in enumerate(lines)
           if ln.strip().startswith("global_load_dwordx4") and ln.strip().endswith("nt")]
    groups, cur = [], []
    for i in idx:
        if cur:
            between = [ln.strip() for ln in lines[cur[-1] + 1:i]]
            if i - cur[-1] > 64 or any(b.startswith((".LBB", "s_barrier", "s_cbranch", "s_branch")) or
                                       re.match(r"s_waitcnt.*vmcnt", b) for b in between):
                groups.append(cur)
                cur = []
        cur.append(i)
    if cur:
        groups.append(cur)
    out = []
    for g in groups:
        assert len(g) == 8, f"a pixel fetch of {len(g)} loads at listing line {g[0]}: expected 8 per step"
        waits = []
        for ln in lines[g[-1] + 1:]:
            t = ln.strip()
            if t.startswith("s_barrier"):
                break
            m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", t)
            if m and (not waits or waits[-1] != int(m.group(1))):
                waits.append(int(m.group(1)))
        out.append((g[0], g[-1], waits))
    return out


PERSISTENT = ["_ZN4dbde13encode_kernelILi0ELb1ELi1EEEvNS_9EncParamsE", "_ZN4dbde13encode_kernelILi0ELb0ELi1EEEvNS_9EncParamsE",
              "_ZN4dbde13encode_kernelILi1ELb1ELi1EEEvNS_9EncParamsE", "_ZN4dbde13encode_kernelILi1ELb0ELi1EEEvNS_9EncParamsE",
              # any geometry, dword-aligned fetches (round 4)
              "_ZN4dbde13encode_kernelILi3ELb1ELi1EEEvNS_9EncParamsE", "_ZN4dbde13encode_kernelILi3ELb0ELi1EEEvNS_9EncParamsE",
              # ... one wave per segment of a tile row (round 4)
              "_ZN4dbde13encode_kernelILi4ELb1ELi1EEEvNS_9EncParamsE", "_ZN4dbde13encode_kernelILi4ELb0ELi1EEEvNS_9EncParamsE",
              # DBDE16 through the same kernel (PIX = 2: one 16-bit tile per lane, the same eight 16-byte loads per step)
              "_ZN4dbde13encode_kernelILi0ELb1ELi2EEEvNS_9EncParamsE", "_ZN4dbde13encode_kernelILi0ELb0ELi2EEEvNS_9EncParamsE",
              # ... and its any-geometry form (round 4)
              "_ZN4dbde13encode_kernelILi1ELb1ELi2EEEvNS_9EncParamsE", "_ZN4dbde13encode_kernelILi1ELb0ELi2EEEvNS_9EncParamsE"]


@pytest.mark.parametrize("mangled", PERSISTENT)
def test_persistent_encoder_pixel_waits_are_exactly_the_loads_in_flight(listing, mangled):
    """Aligned widths (encode_kernel<0,*,1>), any geometry (<1,*,1>: BASELINE configs[3]'s kernel), DBDE16 (<0,*,2>)."""
    lines = function_body(listing, mangled)
    groups = pixel_load_groups(lines)
    # the prologue's groups come first: the first chunk's fetch meets a barrier (drained behind it); round 4's prologue
    # fetches the first chunk a second time when the launch falls back to tickets (a group whose first wait is vmcnt(0)
    # for the mode flag, not part of the loop).  The loop's two unrolled steps are the LAST two groups.
    assert 2 <= len(groups) <= 4, groups
    assert all(not w or w[0] == 0 for _, _, w in groups[:-2]), f"unexpected load group in front of the loop: {groups}"
    steady = groups[-2:]
    for first_load, last, w in steady:
        assert w[:8] == [15, 14, 13, 12, 11, 10, 9, 8], (
            f"pixel waits are {w[:8]}: looser than 15..8 means younger stores are being counted on to retire in "
            f"order; tighter means the prefetch no longer overlaps the statistics")
        # the straight-line block that computes the addresses and issues the loads (everything after the last
        # label) must not wait on vector memory: a vmcnt there means every wave drains its previous stores
        # before it may prefetch (the fake loop edge described in encode_kernel)
        first = last
        while first > 0 and not lines[first - 1].startswith(".LBB"):
            first -= 1
        block = [ln.strip() for ln in lines[first:last + 1]]
        assert sum(b.startswith("global_load_dwordx4") for b in block) == 8, "load group is not one straight-line block"
        assert not [b for b in block if re.match(r"s_waitcnt.*vmcnt", b)], \
            f"vector-memory wait in front of the prefetch: {[b for b in block if b.startswith('s_waitcnt')]}"


@pytest.mark.parametrize("mangled", ["_ZN4dbde19encode_small_kernelILi0ELb1EEEvNS_9EncParamsE",
                                     "_ZN4dbde19encode_small_kernelILi0ELb0EEEvNS_9EncParamsE",
                                     "_ZN4dbde19encode_small_kernelILi1ELb1EEEvNS_9EncParamsE",
                                     "_ZN4dbde19encode_small_kernelILi1ELb0EEEvNS_9EncParamsE"])
def test_small_encoder_waits_for_its_own_loads_only(listing, mangled):
    """One workgroup per chunk, no prefetch: the eight pixel loads are consumed in order with vmcnt(7) ... vmcnt(0) --
    nothing older is outstanding, nothing younger is counted on."""
    groups = pixel_load_groups(function_body(listing, mangled))
    assert len(groups) == 1, groups
    assert groups[0][2][:8] == [7, 6, 5, 4, 3, 2, 1, 0], groups[0][2]


def test_dbde16_encoder_waits(listing):
    """enc16_kernel: a chunk's eight tile-row loads are waited for before any of the chunk's stores is issued -- the
    first vector-memory wait behind them never allows more than the seven younger loads to be outstanding."""
    text = open(os.path.join(CSRC, "dbde16_kernels.s")).read()
    lines = function_body(text, "_ZN6dbde1612enc16_kernelENS_8Params16E")
    idx = [i for i, ln in enumerate(lines) if ln.strip().startswith("global_load_dwordx4")]
    assert len(idx) >= 8
    first_group = idx[:8]
    assert not any(ln.strip().startswith(("global_store", "buffer_store")) for ln in lines[first_group[0]:first_group[-1]])
    for ln in lines[first_group[-1] + 1:]:
        m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", ln.strip())
        if m:
            assert int(m.group(1)) <= 7, ln
            break
        assert not ln.strip().startswith(("global_store", "buffer_store")), "a store before the chunk's loads were waited for"


def test_no_scratch_and_expected_occupancy(listing):
    """The hot kernels must not spill and must keep the residency the design assumes: encoder <= 128 VGPRs
    (2 workgroups of 8 waves per CU); decoder: LDS (<= 40 KB) allows 4 workgroups = 4 waves per SIMD, registers must
    not be what limits it (<= 96 leaves a fifth wave)."""
    meta = {}
    md = listing[listing.index("amdhsa.kernels:"):]
    for entry in re.split(r"\n  - \.", md)[1:]:   # one YAML list item per kernel
        get = lambda k: re.search(r"\.?%s:\s+(\S+)" % k, entry).group(1)
        meta[get("name")] = {"vgpr": int(get("vgpr_count")), "scratch": int(get("private_segment_fixed_size")),
                             "lds": int(get("group_segment_fixed_size"))}
    for name in PERSISTENT + ["_ZN4dbde19encode_small_kernelILi0ELb1EEEvNS_9EncParamsE",
                              "_ZN4dbde19encode_small_kernelILi1ELb1EEEvNS_9EncParamsE"]:
        enc = meta[name]
        assert enc["scratch"] == 0 and enc["vgpr"] <= 128 and enc["lds"] <= 80 * 1024, (name, enc)
    dec = [v for k, v in meta.items() if k.startswith("_ZN4dbde13decode_kernelILi")]
    assert dec and all(d["scratch"] == 0 and d["vgpr"] <= 96 and d["lds"] <= 40 * 1024 for d in dec), dec
    # the small-frame decoder (256-thread instances are the ones launched): six persistent workgroups per CU
    for staged in ("1", "0"):
        mid = meta["_ZN4dbde17decode_mid_kernelILi256ELb%sEEEvNS_9DecParamsE" % staged]
        assert mid["scratch"] == 0 and mid["vgpr"] <= 80 and mid["lds"] <= 24 * 1024, mid


@pytest.mark.parametrize("mangled", ["_ZN4dbde17decode_mid_kernelILi256ELb1EEEvNS_9DecParamsE",
                                     "_ZN4dbde17decode_mid_kernelILi256ELb0EEEvNS_9DecParamsE"])
def test_small_frame_decoder_keeps_its_prefetch_in_flight(listing, mangled):
    """decode_mid_kernel walks groups of frames in a software-pipelined loop: the depth / minimum bytes and the three I32
    fields of the NEXT group and the offset of the one after it are requested at the top of an iteration and first
    looked at behind the current group's payload loads.  Inside the loop: no vector-memory wait between those requests
    and the scan's barrier (the scheduler once moved their consumers up to the loads; __syncthreads() would drain them
    too), ONE wait behind the payload pieces, and none behind the last store of an iteration (a wait there would be a
    wait for the stores)."""
    lines = [ln.strip() for ln in function_body(listing, mangled)]
    head = next(i for i, ln in enumerate(lines) if "Loop Header: Depth=1" in ln)
    body = lines[head:]
    ub = [i for i, ln in enumerate(body) if ln.startswith("global_load_ubyte")]
    assert len(ub) >= 2, "the prefetch of the next group's depth / minimum bytes is gone"
    bar = next(i for i, ln in enumerate(body) if ln.startswith("s_barrier") and i > ub[1])
    assert not [ln for ln in body[ub[0]:bar] if re.match(r"s_waitcnt.*vmcnt", ln)], body[ub[0]:bar]
    pieces = [i for i, ln in enumerate(body) if ln.startswith("global_load_dwordx4") and i > bar]
    assert len(pieces) >= 4
    first_wait = next(i for i, ln in enumerate(body) if re.match(r"s_waitcnt.*vmcnt", ln) and i > pieces[0])
    assert first_wait > pieces[3] and "vmcnt(0)" in body[first_wait]
    # the loop's latch (the block that falls into the header again, behind an iteration's stores) waits for nothing
    latch = max(i for i in range(head) if lines[i].startswith(".LBB"))
    assert "in Loop: Header=" in lines[latch], lines[latch]
    assert not [ln for ln in lines[latch:head] if re.match(r"s_waitcnt.*vmcnt", ln)], lines[latch:head]
