"""CPU: which kernel form a batch call runs (dbde_hip_encode_plan / dbde_hip_decode_plan: the very functions
dbde_hip_encode_frames / dbde_hip_decode_frames call, pure host arithmetic).  Pins the choices DESIGN.md 4.1 / 4.2
describe, the BASELINE configs first, so that a change of a threshold shows up here and not only in a benchmark."""
import pytest

import dbde_video_cpp_amd as dv


@pytest.fixture(scope="module", autouse=True)
def built():
    dv.build()


PERSISTENT, SMALL, TINY, MID, FRAMES, GROUP = 0, 1, 2, 3, 4, 5
DIRECT, STAGED, TILES = 0, 1, 2
TABLE, SELF, FUSED = 0, 1, 2


def test_baseline_configs():
    # configs[1]: 4096x3072, 1024 frames per step
    e = dv.encode_plan(4096, 3072, 1024, slot_stride=12976384)
    assert (e["kernel"], e["input_mode"], e["threads"], e["chunk_tiles"], e["chunks_per_frame"]) == (PERSISTENT, 0, 512, 1024, 192)
    d = dv.decode_plan(4096, 3072, 1024)
    assert (d["kernel"], d["image_mode"], d["index_mode"], d["threads"], d["chunk_tiles"], d["chunks_per_frame"]) == (0, DIRECT, TABLE, 256, 512, 384)
    # ... taken literally: ONE frame per call -> one workgroup per chunk / fused index + decode
    assert dv.encode_plan(4096, 3072, 1)["kernel"] == SMALL
    d1 = dv.decode_plan(4096, 3072, 1)
    assert (d1["kernel"], d1["index_mode"], d1["n_chunks"]) == (0, FUSED, 384)
    # configs[2]: 2048x2048 x 1000, concatenated
    assert dv.encode_plan(2048, 2048, 1000)["kernel"] == PERSISTENT
    assert dv.decode_plan(2048, 2048, 1000)["image_mode"] == DIRECT
    # configs[3]: 1921x1081 -- any-geometry encoder, staged decode on chunks of two whole tile rows (2 x 241 tiles)
    e = dv.encode_plan(1921, 1081, 2048, slot_stride=2168320)
    # (round 4: dword-aligned fetches -- input mode 3 -- a wave owns 63 tile pairs: 8 x 63 x 2 tiles per chunk, 136 x 121 pairs)
    assert (e["kernel"], e["input_mode"], e["aligned_out"], e["chunk_tiles"], e["chunks_per_frame"]) == (PERSISTENT, 4, 1, 976, 34)   # 121 pairs per tile row = segments of 61 + 60, one wave each
    # rows that are dword aligned already, or whose last pair holds more than 13 columns, keep natural-position fetches
    assert dv.encode_plan(1928, 1080, 2048)["input_mode"] == 1      # even addresses read at the full rate as they are
    assert dv.encode_plan(1366, 768, 2048)["input_mode"] == 1
    assert dv.encode_plan(1928, 1080, 2048, image_address=1)["input_mode"] == 4
    e = dv.encode_plan(1081, 1921, 2048)                           # 68 pairs per tile row would half fill two waves: dealt linearly
    assert (e["input_mode"], e["chunk_tiles"]) == (3, 1008)
    e = dv.encode_plan(1001, 1001, 2048)                           # 63 pairs: one full wave per tile row
    assert (e["input_mode"], e["chunk_tiles"], e["chunks_per_frame"]) == (4, 1008, 16)
    assert dv.encode_plan(1935, 1080, 2048)["input_mode"] == 1      # last pair: 15 columns
    d = dv.decode_plan(1921, 1081, 2048)
    assert (d["image_mode"], d["threads"], d["chunk_tiles"], d["chunks_per_frame"]) == (STAGED, 256, 482, 68)


@pytest.mark.parametrize("W,H,mode,threads,chunk_tiles", [
    (1920, 1080, DIRECT, 256, 512),     # whole cache lines per wave
    (1280, 720, DIRECT, 256, 512),
    (1440, 900, DIRECT, 256, 512),      # 16-byte rows, 70 % chunk fill: direct 16-byte stores from plain chunks
    (1600, 900, DIRECT, 256, 512),      # 78 %
    (720, 1280, STAGED, 256, 450),      # 16-byte rows, 88 % fill: whole tile rows (5 x 90), per-chunk direct / staged
    (1360, 768, STAGED, 256, 510),
    (1080, 1920, STAGED, 256, 405),     # 8-byte rows, 79 % fill (portrait HD)
    (1928, 1080, STAGED, 256, 482),
    (1366, 768, STAGED, 192, 342),      # odd rows, 2 x 171 tiles: 67 % of 512 slots, 89 % of 384 -> the 192-thread workgroup
    (3000, 2000, STAGED, 192, 375),
    (2999, 2001, STAGED, 192, 375),
    (2200, 1000, TILES, 256, 512),      # 8-byte rows, 275 tiles: 54 % / 72 % fill -> tile by tile
    (1001, 999, STAGED, 256, 504),
])
def test_decode_forms(W, H, mode, threads, chunk_tiles):
    d = dv.decode_plan(W, H, 4096)
    assert (d["kernel"], d["image_mode"], d["threads"], d["chunk_tiles"]) == (0, mode, threads, chunk_tiles), d


def test_unaligned_image_base_changes_the_form():
    assert dv.decode_plan(4096, 3072, 64, image_address=0)["image_mode"] == DIRECT
    # a base that is not a multiple of 128: a wave's 1 KB no longer covers whole cache lines -> chunks of whole tile rows
    # (one row of 512 tiles fills the workgroup), staged or, per chunk, direct 16-byte stores
    for addr in (64, 16, 8, 1):
        d = dv.decode_plan(4096, 3072, 64, image_address=addr)
        assert (d["image_mode"], d["chunk_tiles"], d["chunks_per_frame"]) == (STAGED, 512, 384), (addr, d)
    assert dv.encode_plan(4096, 3072, 64, image_address=0)["input_mode"] == 0
    assert dv.encode_plan(4096, 3072, 64, image_address=1)["input_mode"] == 1
    assert dv.encode_plan(4096, 3072, 64, out_address=0)["aligned_out"] == 1
    assert dv.encode_plan(4096, 3072, 64, out_address=4)["aligned_out"] == 0


@pytest.mark.parametrize("W,H,T,enc,dec,threads", [
    (8, 8, 1, TINY, MID, 256), (64, 64, 64, GROUP, MID, 256),   # (decode: one kernel from single-tile frames up to 256 tiles)
    (32, 32, 16, GROUP, MID, 256), (16, 16, 4, GROUP, MID, 256), (16, 8, 2, TINY, MID, 256), (20, 20, 9, GROUP, MID, 256), (60, 60, 64, GROUP, MID, 256), (18, 18, 9, TINY, MID, 256),
    # 8-byte aligned rows, frames and buffers whole 16-byte blocks (round 4): whole frames per workgroup, staged through LDS
    # (encode; the decode side keeps decode_mid_kernel -- persistent, software-pipelined from the second half of round 4 --
    # up to 256 tiles and the chunk kernels above: the staged whole-frame decoder measured no faster and is an experiment switch)
    (72, 72, 81, GROUP, MID, 256),           # three frames per 256-thread persistent workgroup (95 %)
    (96, 96, 144, FRAMES, MID, 256),         # 3 frames in 512 slots (84 %; 7 in 1024 would be 98 %: measured slower, eight-wave barriers)
    (128, 128, 256, FRAMES, MID, 256),       # 2 frames in 512 slots
    (160, 120, 300, FRAMES, 0, 256),         # one frame in 512 slots (59 %; 3 in 1024: 88 %, measured equal or slower)
    (176, 144, 396, FRAMES, 0, 256),         # one frame in 512 slots = two in 1024 (77 %): the smaller workgroup
    (320, 240, 1200, PERSISTENT, 0, 512),    # above 640 tiles: the chunk kernels
    # rows that are not 8-byte aligned keep the one-tile-per-lane forms
    (75, 70, 90, MID, MID, 1024), (100, 100, 169, GROUP, MID, 256), (102, 100, 169, MID, MID, 512), (136, 128, 272, FRAMES, 0, 256), (200, 168, 525, FRAMES, 0, 512),
])
def test_small_frames(W, H, T, enc, dec, threads):
    slot = ((32 + 66 * T + 255) // 256) * 256
    e = dv.encode_plan(W, H, 100000, slot_stride=slot)
    d = dv.decode_plan(W, H, 100000)
    assert e["kernel"] == enc and d["kernel"] == dec, (e, d)
    if enc in (TINY, MID, FRAMES, GROUP):
        assert e["threads"] == threads
    if dec == MID:
        assert d["threads"] == 256          # the persistent mid decoder: 256-thread workgroups at every fill
    # an image base that is not a multiple of 16 bytes: the staged form does not apply
    if enc == FRAMES:
        assert dv.encode_plan(W, H, 100000, slot_stride=slot, image_address=8)["kernel"] in (TINY, MID, PERSISTENT)
    if enc == GROUP:   # (rows and bases of 4-byte multiples suffice there)
        assert dv.encode_plan(W, H, 100000, slot_stride=slot, image_address=8)["kernel"] == GROUP
        assert dv.encode_plan(W, H, 100000, slot_stride=slot, image_address=2)["kernel"] in (TINY, MID, PERSISTENT)
        assert dv.decode_plan(W, H, 100000, image_address=8)["kernel"] in (MID, 0)
    # concatenated frames need each other's sizes: the chunk kernels
    assert dv.encode_plan(W, H, 100000, slot_stride=0)["kernel"] in (PERSISTENT, SMALL)


def test_mid_size_frames_that_would_store_tile_by_tile_take_the_small_frame_decoder():
    """257 .. 768 tiles, rows that are not 8-byte aligned, whole-tile-row chunks that do not fit the staged image: one frame
    per 512- or 1024-thread workgroup of decode_mid_kernel instead of two chunks per frame + the index kernel (measured);
    not where the chunks stage their pixels, not at 1024 tiles."""
    for (W, H) in [(130, 121), (150, 150), (180, 180), (220, 215)]:
        d = dv.decode_plan(W, H, 100000)
        assert d["kernel"] == MID and d["threads"] in (512, 1024), (W, H, d)
    assert dv.decode_plan(250, 250, 100000)["kernel"] == 0           # 1024 tiles: the chunk decoder
    assert dv.decode_plan(300, 200, 100000)["kernel"] == 0           # staged chunks (13 tile rows of 38 tiles)
    assert dv.decode_plan(160, 120, 100000)["kernel"] == 0           # 16-byte rows: direct stores


def test_index_forms_follow_the_batch():
    assert dv.decode_plan(200, 123, 13)["index_mode"] == SELF                 # few small frames: workgroups index themselves
    assert dv.decode_plan(1024, 768, 6)["index_mode"] == FUSED                # 144 chunks fit the device: fused
    assert dv.decode_plan(1024, 768, 6, n_cu=8)["index_mode"] == TABLE        # ... not a device of 8 CUs
    assert dv.decode_plan(1024, 768, 4096)["index_mode"] == TABLE
    assert dv.encode_plan(1024, 768, 6)["kernel"] == SMALL and dv.encode_plan(1024, 768, 4096)["kernel"] == PERSISTENT
    assert dv.encode_plan(1024, 768, 6, resident_workgroups=9)["kernel"] == PERSISTENT


def test_bad_arguments():
    for bad in ((0, 8, 1), (8, 0, 1), (8, 8, 0)):
        with pytest.raises(ValueError):
            dv.decode_plan(*bad)
        with pytest.raises(ValueError):
            dv.encode_plan(*bad)
