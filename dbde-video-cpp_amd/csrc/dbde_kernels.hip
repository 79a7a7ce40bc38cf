// dbde_kernels.hip -- DBDE encode / decode kernels for MI355X (gfx950, wave64).
//
// Replaces the hot loops of the reference: dbde_pack_image / dbde_pack_8x8[_partial]
// (dbde_util.cpp:22-180) and dbde_unpack_image / dbde_unpack_8x8[_partial] (:216-328).
//
// Work decomposition (both directions): one 256-thread workgroup per CHUNK = 512 consecutive
// tiles in stream order; each lane owns two adjacent 8x8 tiles (2 x 64 B = 32 VGPRs).
//   * image side: one 16-byte access per lane per image row -> every wave instruction moves
//     1 KiB of one image row (coalesced);
//   * stream side: the chunk's payload is one contiguous byte range; it is staged in LDS and
//     moved with 16-byte-per-lane accesses, never with per-tile narrow accesses;
//   * min/max and the depth are computed inside the lane (packed-u16 VALU, no shuffles);
//   * tile offsets inside a chunk: wave-level shuffle scan + 4-entry cross-wave scan;
//   * chunk offsets inside the frame / launch: ENCODE uses a single-pass decoupled look-back
//     (relaxed agent-scope 8-byte records, dynamic ticket order, bounded spins); DECODE reads
//     them from the index kernel, which also performs the reference's validation.
// Integer byte/bit work: no MFMA.  Roofline: HBM.
#include "dbde_kernels.h"

#include "dbde_bits.h"

#ifndef DBDE_POLL
#define DBDE_POLL 0   // where the encoder polls its mailbox (0 = round-1 order, A/B builds only)
#endif
#ifndef DBDE_LINE_ALIGNED_STORES
#define DBDE_LINE_ALIGNED_STORES 1   // A/B switch: 0 = payload stores start at the wave's first 16-byte boundary
#endif
#ifndef DBDE_UNALIGNED_OUT_WORDS
#define DBDE_UNALIGNED_OUT_WORDS 0   // A/B switch: 1 = frames at any alignment leave word by word (8-byte stores at whatever address)
#endif
#ifndef DBDE_NT
#define DBDE_NT 1   // non-temporal hint on the streamed-once traffic (pixels, payload, decoded images)
#endif

namespace dbde {

// -DDBDE_DIAG builds, one launch at a time (profiles/abbench, ABBENCH_SFDIAG): wave 0's wall-clock time (10 ns) at up
// to ten points of a workgroup's life, summed over the launch's workgroups, plus the earliest and latest start and the
// latest end -- where one frame per call spends its microseconds.
#ifndef DBDE_DIAG_LATE
#define DBDE_DIAG_LATE 34   // -DDBDE_DIAG: first of the eight "late" step pairs whose duration is recorded (early: pairs 2-9)
#endif
#ifdef DBDE_DIAG
#define SF_DECL uint64_t sfd[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define SF_MARK(i) do { if (threadIdx.x == 0) sfd[i] = wall_clock64(); } while (0)
#define SF_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#define SF_FLUSH(diag, last)                                                                          \
    do {                                                                                              \
        if (threadIdx.x == 0 && (diag)) {                                                             \
            atomicMax(&(diag)[0], ~(unsigned long long)sfd[0]);                                       \
            for (int i_ = 1; i_ < 10; i_++) atomicAdd(&(diag)[i_], (unsigned long long)sfd[i_]);      \
            atomicAdd(&(diag)[10], 1ull);                                                             \
            atomicMax(&(diag)[11], (unsigned long long)sfd[last]);                                    \
            atomicMax(&(diag)[12], (unsigned long long)sfd[0]);                                       \
            atomicAdd(&(diag)[13], (unsigned long long)sfd[0]);                                       \
        }                                                                                             \
    } while (0)
#else
#define SF_DECL
#define SF_MARK(i)
#define SF_DRAIN()
#define SF_FLUSH(diag, last)
#endif

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));   // native vector for the nontemporal builtins

typedef unsigned long long u64a;   // type of the look-back records

// ---------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------

__device__ __forceinline__ uint64_t load_u64_any(const uint8_t *p) {   // any byte alignment
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ void store_u64_any(uint8_t *p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
__device__ __forceinline__ void store_u32_bytes(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}
__device__ __forceinline__ uint32_t load_u32_bytes(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
__device__ __forceinline__ uint64_t load_u64_bytes(const uint8_t *p) {
    return (uint64_t)load_u32_bytes(p) | ((uint64_t)load_u32_bytes(p + 4) << 32);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Wave-wide inclusive scan with DPP row shifts / row broadcasts (gfx9 wave64 idiom): no LDS.
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x) {
    uint32_t t = x;
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x111, 0xF, 0xF, false);   // row_shr:1
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x112, 0xF, 0xF, false);   // row_shr:2
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x114, 0xF, 0xF, false);   // row_shr:4
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x118, 0xF, 0xF, false);   // row_shr:8
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1,3
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2,3
    return t;
}

// Inclusive scan of one value per lane across the 256-thread block.
// Returns this lane's inclusive prefix; block_total = sum over the block.
template <int NW = 4>
__device__ __forceinline__ uint32_t block_scan_incl(uint32_t v, uint32_t *s_wave_tot, int lane, int wave,
                                                    uint32_t &block_total) {
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t n = __shfl_up(incl, o, 64);
        if (lane >= o) incl += n;
    }
    if (lane == 63) s_wave_tot[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) {
        const uint32_t tk = s_wave_tot[k];
        base += k < wave ? tk : 0u;
        tot += tk;
    }
    block_total = tot;
    return base + incl;
}

// LDS staging images are XOR-swizzled at 16-byte slot granularity so that the regular lane
// strides of uniform-depth content (lane stride = 2*depth qwords; 128 B for depth 8) do not
// pile onto one bank.  Both are involutions that only permute slots inside an aligned group.
__device__ __forceinline__ uint32_t swz8(uint32_t slot) { return slot ^ ((slot >> 3) & 7u); }      // 128-B groups
__device__ __forceinline__ uint32_t swz16(uint32_t slot) { return slot ^ ((slot >> 4) & 15u); }    // 256-B groups
// qword index -> swizzled qword index (the half inside the slot is kept)
__device__ __forceinline__ uint32_t swzq8(uint32_t q) { return (swz8(q >> 1) << 1) | (q & 1u); }
#ifndef DBDE_ENC_SWZ_ALL
#define DBDE_ENC_SWZ_ALL 0   // A/B switch: 1 = the encoder's payload image swizzled for every wave (round-2 start)
#endif
__device__ __forceinline__ uint32_t swzq16(uint32_t q) { return (swz16(q >> 1) << 1) | (q & 1u); }

// Any of the four depth bytes of x above the format's maximum (8, or 16 for DBDE16)?  Non-zero if so.
__device__ __forceinline__ uint32_t depth_bytes_bad(uint32_t x, uint32_t min_bytes) {
    return min_bytes == 1u ? ((x & 0xF0F0F0F0u) | (((x & 0x0F0F0F0Fu) + 0x77777777u) & 0x80808080u))
                           : ((x & 0xE0E0E0E0u) | (((x & 0x1F1F1F1Fu) + 0x6F6F6F6Fu) & 0x80808080u));
}

// x86-64 `(uint64_t)double` as g++ compiles it (dbde_util.cpp:334): cvttsd2si below 2^63,
// else cvttsd2si(v - 2^63) ^ 2^63; out-of-range and NaN give the "integer indefinite".
__device__ __forceinline__ uint64_t f64_to_u64_x86(double v) {
    const double two63 = 9223372036854775808.0;
    bool high = v >= two63;             // false for NaN, as comisd/jae falls through
    double a = high ? v - two63 : v;
    uint64_t r;
    if (a >= -two63 && a < two63) r = (uint64_t)(long long)a;   // truncates toward zero
    else r = 0x8000000000000000ull;                              // includes NaN
    return high ? (r ^ 0x8000000000000000ull) : r;
}

// Does [off, off + len) lie inside [0, extent)?  Written so that it cannot wrap: `off` may be anything a caller left
// in an offsets array or a cursor (a stale -1, 2^63, ...), and `off + len <= extent` would accept offsets near 2^64.
__device__ __forceinline__ bool in_extent(uint64_t off, uint64_t len, uint64_t extent) {
    return off <= extent && len <= extent - off;
}

// n / d with d fixed for the launch: m = floor(2^32 / d) comes from the host (div_magic_of), the quotient estimate
// mulhi(n, m) is the true one or one below it (n * m / 2^32 > n / d - 1 for every n < 2^32), one correction.  A u32
// division is 30-40 vector instructions on this part and the chunk geometry needs two or three per lane and step.
__device__ __forceinline__ uint32_t div_magic(uint32_t n, uint32_t d, uint32_t m, uint32_t &rem) {
    uint32_t q = __umulhi(n, m);
    uint32_t r = n - q * d;
    const bool fix = r >= d;
    q += fix ? 1u : 0u;
    r -= fix ? d : 0u;
    rem = r;
    return q;
}

// ---------------------------------------------------------------------------------------
// ENCODE
// ---------------------------------------------------------------------------------------

// Clamp-to-edge load of one (possibly partial) tile: restates the constant padding of
// dbde_pack_8x8_partial (dbde_util.cpp:105-135).
__device__ __forceinline__ void load_tile_generic(const uint8_t *img, int W, int H, uint32_t w, uint32_t t,
                                                  uint32_t (&v)[16]) {
    uint32_t ty = t / w, tx = t - ty * w;
    int x0 = 8 * (int)tx;
    int rm = W - x0 < 8 ? W - x0 : 8;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int yy = 8 * (int)ty + r;
        yy = yy < H ? yy : H - 1;
        const uint8_t *row = img + (size_t)yy * (size_t)W + x0;
        uint64_t q;
        if (rm == 8) {
            q = load_u64_any(row);
        } else if (W >= 8) {
            // right margin: the 8 bytes that END at the last valid pixel are inside the image; shift
            // the valid ones down and repeat the last one (dbde_util.cpp:116-128) -- one load, not rm
            q = load_u64_any(row + rm - 8) >> (8 * (8 - rm));
            const uint64_t last = (q >> (8 * (rm - 1))) & 0xFFull;
            q |= (last * 0x0101010101010101ull) << (8 * rm);
        } else {
            q = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                int kk = k < rm ? k : rm - 1;
                q |= (uint64_t)row[kk] << (8 * k);
            }
        }
        v[2 * r] = (uint32_t)q;
        v[2 * r + 1] = (uint32_t)(q >> 32);
    }
}

// Chunk offsets: a central in-order scan instead of a distributed look-back.
//
// Every chunk's payload-word count is published by its workers as an AGG record, in parallel
// and without waiting on anything.  ONE wave in the whole launch (the scanner: the scout wave of
// whichever workgroup started first) walks the records in chunk order, 512 per round with all
// loads in flight at once, turns them into inclusive prefixes with a wave scan, and overwrites
// each record with its INC form.  A chunk's scout only polls its own 8-byte record.  This
// moves >100 chunks/us with a few KB of polling per round, where a decoupled look-back (every
// chunk re-reading a window of predecessors until it meets an INC) advanced only one window
// per hop latency and flooded the fabric with polls.
//
// Records (8 B, relaxed agent-scope atomics, the record is its own flag):
//   AGG : [63:62] = 1, [31:0] payload words of the chunk
//   INC : [63:62] = 2, [61:32] payload words of the frame up to and including the chunk,
//                      [31:0]  payload words of the launch up to and including it (mod 2^32)
// PIX = bytes per pixel: 1 = DBDE, 2 = DBDE16 (U16 minima: nm = 2T, 32 + 3T bytes in front of the payload).
template <bool ALIGNED_OUT, int PIX = 1>
__device__ __forceinline__ void write_frame_fields(const EncParams &p, uint32_t f, uint32_t cf, uint32_t inf_incl,
                                                     uint32_t frame_start_glob) {
    const uint64_t meta = 32ull + (uint64_t)(1 + PIX) * p.T;
    const uint64_t frame_base = p.slot_stride ? (uint64_t)f * p.slot_stride
                                              : (uint64_t)f * meta + 8ull * (uint64_t)frame_start_glob;
    uint8_t *fb = p.out + frame_base;
    if (cf == 0u) {   // frame header + the two I32 T fields (dbde_util.cpp:140-143, 182-196)
        const uint64_t index = p.indices ? p.indices[f] : p.first_index + f;
        const uint64_t el = p.elapsed_ns ? p.elapsed_ns[f] : 0ull;
        const uint64_t elbits = (uint64_t)__double_as_longlong(__ull2double_rn(el));   // trap T1: F64 on the wire
        if (ALIGNED_OUT) {
            uint32_t *h = reinterpret_cast<uint32_t *>(fb);
            h[0] = 2u;
            h[1] = (uint32_t)index; h[2] = (uint32_t)(index >> 32);
            h[3] = (uint32_t)elbits; h[4] = (uint32_t)(elbits >> 32);
            h[5] = p.T;
            *reinterpret_cast<uint32_t *>(fb + 24 + p.T) = (uint32_t)PIX * p.T;
        } else {
            store_u32_bytes(fb, 2u);
            store_u32_bytes(fb + 4, (uint32_t)index); store_u32_bytes(fb + 8, (uint32_t)(index >> 32));
            store_u32_bytes(fb + 12, (uint32_t)elbits); store_u32_bytes(fb + 16, (uint32_t)(elbits >> 32));
            store_u32_bytes(fb + 20, p.T);
            store_u32_bytes(fb + 24 + p.T, (uint32_t)PIX * p.T);
        }
        if (p.frame_offsets) p.frame_offsets[f] = frame_base;
    }
    if (cf == p.chunks_per_frame - 1u) {   // I32 n64 (dbde_util.cpp:144-146,179)
        if (ALIGNED_OUT) *reinterpret_cast<uint32_t *>(fb + meta - 4ull) = inf_incl;
        else store_u32_bytes(fb + meta - 4ull, inf_incl);
        if (p.frame_bytes) p.frame_bytes[f] = meta + 8ull * inf_incl;
    }
}

#ifndef DBDE_SCAN_LOADS
#define DBDE_SCAN_LOADS 8
#endif
constexpr int kScanLoads = DBDE_SCAN_LOADS;   // records per lane per round (64 x this many per round)

template <bool ALIGNED_OUT>
__device__ __forceinline__ void scanner_loop(const EncParams &p, int lane) {
    __builtin_amdgcn_s_setprio(3);
    uint32_t F = 0;                    // first record not yet converted
    uint32_t carry_glob = 0;           // launch-wide payload words before record F
    uint32_t frame_start_glob = 0;     // launch-wide payload words before the frame that contains record F
    uint64_t t_progress = wall_clock64();
#ifdef DBDE_DIAG
    const uint64_t dg_t0 = __builtin_amdgcn_s_memtime();
    uint64_t dg_rounds = 0, dg_idle = 0, dg_conv_calls = 0;
#endif
    while (F < p.n_chunks) {
        u64a w[kScanLoads];
#pragma unroll
        for (int j = 0; j < kScanLoads; j++) {
            const uint32_t idx = F + 64u * (uint32_t)j + (uint32_t)lane;
            w[j] = idx < p.n_chunks ? __hip_atomic_load(&p.state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
        uint32_t done = 0;
        bool stop = false;
#pragma unroll
        for (int j = 0; j < kScanLoads; j++) {
            if (!stop) {
                const uint32_t base = F + 64u * (uint32_t)j;
                const uint32_t idx = base + (uint32_t)lane;
                const bool ready = (uint32_t)(w[j] >> 62) == 1u;
                const uint64_t rdy = __ballot(ready);
                const uint32_t cnt = rdy == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~rdy);   // leading ready lanes
                if (cnt != 0u) {
                    const bool act = (uint32_t)lane < cnt;
                    const uint32_t t = act ? (uint32_t)w[j] : 0u;
                    const uint32_t incl = wave_scan_incl(t);
                    const uint32_t g_incl = carry_glob + incl, g_excl = g_incl - t;
                    // launch-wide prefix at the start of this lane's frame: inside the window it is
                    // the exclusive prefix of the lane holding the frame's first chunk
                    uint32_t cf_lane;
                    (void)div_magic(idx, p.chunks_per_frame, p.magic_cpf, cf_lane);
                    const uint32_t fstart = idx - cf_lane;                 // first chunk of the lane's frame
                    const uint32_t from_lane = __shfl(g_excl, fstart >= base ? (int)(fstart - base) : 0, 64);
                    const uint32_t gs = fstart >= base ? from_lane : frame_start_glob;
                    const uint32_t inf_incl = g_incl - gs;
                    if (act) {
                        const u64a rec = kStInc | ((u64a)(inf_incl & 0x3FFFFFFFu) << 32) | (u64a)g_incl;
                        __hip_atomic_store(&p.state[idx], rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    // carries for the record after the last converted one
                    const uint32_t last = cnt - 1u;
                    carry_glob = __builtin_amdgcn_readlane(g_incl, last);
                    const uint32_t nidx = base + cnt;
                    uint32_t ncf;
                    (void)div_magic(nidx, p.chunks_per_frame, p.magic_cpf, ncf);
                    const uint32_t nf_start = nidx - ncf;
                    // frame of the next record starts at nf_start: its prefix is known if nf_start <= nidx
                    if (nf_start == nidx) frame_start_glob = carry_glob;
                    else if (nf_start >= base) frame_start_glob = __builtin_amdgcn_readlane(g_excl, nf_start - base);
                    done += cnt;
                }
                if (cnt < 64u) stop = true;
            }
        }
#ifdef DBDE_DIAG
        dg_rounds++;
        dg_idle += done ? 0 : 1;
        dg_conv_calls += done;
#endif
        if (done) {
            F += done;
            t_progress = wall_clock64();
        } else {
            if (wall_clock64() - t_progress > 200000000ull) {   // 2 s without progress: give up
                if (lane == 0) atomicOr(p.sticky, 1u);
                return;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    }
#ifdef DBDE_DIAG
    if (lane == 0) {
        atomicAdd(&p.diag[3], dg_rounds);
        atomicAdd(&p.diag[4], dg_idle);
        atomicAdd(&p.diag[5], __builtin_amdgcn_s_memtime() - dg_t0);
        atomicAdd(&p.diag[6], dg_conv_calls);
    }
#endif
}

// A workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vector memory (vmcnt(0): loads and stores
// share one counter on this part), i.e. it waits for every pixel the wave has requested -- in the encoder's prologue that is
// the launch's first 32 MB burst, 6-12 us, and the claim mode was being waited for BEHIND it.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// One uncached dword through the SCALAR unit (glc: past the scalar cache): its result does not queue behind the wave's
// outstanding vector loads, which return in issue order.
__device__ __forceinline__ uint32_t scalar_load_uncached(const uint32_t *ptr) {
    uint32_t v;
    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ptr) : "memory");
    return v;
}

// Scout side: wait until the scanner has converted this chunk's own record.
__device__ __forceinline__ bool wait_inc(const u64a *state, uint32_t c, uint32_t &inf_incl, uint32_t &glob_incl,
                                         uint32_t *n_polls = nullptr) {
    const uint64_t t_start = wall_clock64();
    for (uint32_t spins = 0;; spins++) {
        if (n_polls) *n_polls = spins + 1u;
        const u64a w = __hip_atomic_load(&state[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t st = (uint32_t)(w >> 62);
        if (st == 2u) {
            inf_incl = (uint32_t)((w >> 32) & 0x3FFFFFFFull);
            glob_incl = (uint32_t)w;
            // The record has done its work (the scanner never looks below the record it is converting, nobody else
            // reads this one): it is handed back clean, so that the next launch finds zeros without a memset in front
            // of it -- that memset and its boundary were 8-10 us of every encode launch.
            __hip_atomic_store(const_cast<u64a *>(&state[c]), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
        if (st == 3u) return false;
        if (spins > 64u && (wall_clock64() - t_start) > 300000000ull) return false;   // 3 s at 100 MHz
        __builtin_amdgcn_s_sleep(8);
    }
}

// ---------------------------------------------------------------------------------------
// ENCODE kernel: persistent, software-pipelined
// ---------------------------------------------------------------------------------------
// A workgroup = 8 waves (two per SIMD, so 2 workgroups per CU are resident), alive for the
// whole launch, walking chunks (1024 consecutive tiles) in ticket order.  (Chunks twice the
// decoder's size halve the rate of ticket draws, which all hit one address.)  Three chunks are in
// flight per workgroup:
//     nxt : image loads in flight (16 B per lane per row, registers R1), issued at the top
//     cur : pixels in R0 -> min/max, depth, wave scan, AGG record, pack into the wave's LDS region
//     prev: packed payload in LDS -> stored once its mailbox (INC record) has been read
// Order inside an iteration (what makes it fast):
//   1. lane 0 polls prev's mailbox and draws the next ticket BEFORE any load of this iteration
//      is issued: a wave's vector-memory results return in issue order, so a poll issued behind
//      the prefetch loads would wait for them.  prev's AGG was published a whole iteration ago,
//      so the scanner has normally answered already.
//   2. all waves issue nxt's image loads (consumed at the end of the iteration).
//   3. statistics of cur; the last wave to finish publishes cur's AGG record immediately.
//   4. ONE workgroup barrier: wave totals, prev's offsets and the next chunk id become visible.
//   5. each wave stores its own contiguous part of prev (LDS -> global, 16 B per lane) and packs
//      cur over it.  The LDS image is private to the wave, so this needs no workgroup barrier.
// Forward progress: chunks are claimed in ticket order by running workgroups, AGG records are
// published without waiting on anything, the scanner is a running workgroup by construction:
// the smallest unfinished chunk can always finish.
constexpr int kEncWaves = kEncChunkTiles / 128;                  // two tiles per lane
// control words of a persistent launch (EncParams::ctrl, a set of kCtrlWords u32, zero when the launch starts)
// Sixteen tail-ticket counters (workgroup b uses counter b % 16), each in a 256-byte slot of its own; the verdict, the
// scanner's role and the ticket counter of the fallback behind them.
constexpr uint32_t kEncGroups = 16, kCtrlSlot = DBDE_CTRL_SLOT_WORDS;   // (slot stride in u32)
constexpr uint32_t kCtrlTail = 0, kCtrlVerdict = kEncGroups * kCtrlSlot /* 0 = open, 1 = static, 2 = tickets */,
                   kCtrlScanner = kCtrlVerdict + 1u /* the scanner's role: 0 = free */, kCtrlTickets = kCtrlVerdict + 2u;
static_assert(kCtrlTickets + 2u <= kEncCtrlWords, "control words do not fit their set");
constexpr int kEncThreads = 64 * kEncWaves;
constexpr uint32_t kWaveWords = 128 * 8;                         // 1024 U64 = 8 KiB per wave

struct EncShared {
    uint64_t pay[kEncWaves][kWaveWords + 64];   // payload image of each wave (swzq8-swizzled when the wave is all depth 8) + a trash word per lane
    uint32_t tot[2][kEncWaves];              // [parity][wave] payload words of cur per wave
    uint32_t lb[2][4];                       // [parity] {in-frame prefix, launch prefix, ok, next chunk id}
    uint32_t acc[2];                         // [parity] arrivals << 24 | sum of the waves' totals
    uint32_t boot[2];                        // role / first two chunk ids
    uint32_t claim[2];                       // ticket mode: the first chunk's id
#ifdef DBDE_DIAG
    uint64_t dg[2];
#endif
};

struct ChunkRef {
    uint32_t c, f, cf, t0;     // t0: stream index of the lane's first tile
    uint32_t ty, tx;           // its tile row and column
    bool valid, hasA, hasB;
    bool loads;                // the lane fetches pixels at (ty, tx): hasA, or (kInRaw4, kInRow) the lane behind the wave's pairs, which only feeds the last of them
    uint32_t seg_j0;           // kInRow (wave-uniform): first pair of the wave's segment of tile row ty
    bool seg_last;             // ... and whether the segment ends the tile row
};

template <int PIX = 1>
__device__ __forceinline__ ChunkRef chunk_ref(const EncParams &p, uint32_t c, int tidw) {
    ChunkRef k;
    c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);   // wave-uniform at every call site: frame, chunk-in-frame (and kInRow's tile row) in scalar registers
    k.c = c;
    k.valid = c < p.n_chunks;
    uint32_t cf = 0;
    k.f = k.valid ? div_magic(c, p.chunks_per_frame, p.magic_cpf, cf) : 0u;
    k.cf = k.valid ? cf : 0u;
    k.seg_j0 = 0u;
    k.seg_last = false;
    if (PIX == 2) {                         // DBDE16: ONE tile per lane (a tile row is 16 bytes), 512 consecutive tiles
        k.t0 = k.cf * (kEncChunkTiles / 2u) + (uint32_t)tidw;
        k.hasA = k.hasB = k.valid && k.t0 < p.T;
        k.ty = div_magic(k.t0, p.w, p.magic_w, k.tx);
        k.loads = k.hasA;
    } else if (p.lanes_per_row == 0u) {     // plain: 1024 consecutive tiles
        k.t0 = k.cf * kEncChunkTiles + 2u * (uint32_t)tidw;
        k.hasA = k.valid && k.t0 < p.T;
        k.hasB = k.valid && k.t0 + 1u < p.T;
        k.ty = div_magic(k.t0, p.w, p.magic_w, k.tx);
        k.loads = k.hasA;
    } else if (p.seg_per_row != 0u) {       // kInRow: a wave = one segment of ONE tile row; everything but the lane's column is scalar
        const uint32_t lane = (uint32_t)tidw & 63u;
        const uint32_t seg = k.cf * (uint32_t)kEncWaves + (uint32_t)__builtin_amdgcn_readfirstlane(tidw >> 6);
        uint32_t sidx;
        const uint32_t ty = div_magic(seg, p.seg_per_row, p.magic_seg, sidx);
        const uint32_t len = p.seg_q + (sidx < p.seg_rem ? 1u : 0u);
        k.seg_j0 = sidx * p.seg_q + (sidx < p.seg_rem ? sidx : p.seg_rem);
        k.seg_last = sidx + 1u == p.seg_per_row;
        const uint32_t j = k.seg_j0 + lane;
        const bool row = k.valid && ty < p.h;
        k.ty = ty;
        k.tx = 2u * j;
        k.t0 = ty * p.w + k.tx;
        k.hasA = row && lane < len;
        k.hasB = k.hasA && k.tx + 1u < p.w;
        k.loads = row && lane <= len && j < p.lanes_per_row;   // (lane `len`: the next segment's first pair, for the sake of lane len - 1)
    } else {                                // the lane's pair: tile row = pair / lanes_per_row
        // pairs_per_wave = 64: 512 consecutive pairs per chunk.  63 (kInRaw4): a wave OWNS 63 consecutive pairs and its
        // 64th lane fetches the pair after them -- the first pair of the next wave -- for the sake of lane 62 alone
        const uint32_t ppw = p.pairs_per_wave, lane = (uint32_t)tidw & 63u;
        const uint32_t pair = k.cf * (ppw * (uint32_t)(kEncChunkTiles / 128u)) + ((uint32_t)tidw >> 6) * ppw + lane;
        uint32_t j;
        k.ty = div_magic(pair, p.lanes_per_row, p.magic_lpr, j);
        k.tx = 2u * j;
        k.t0 = k.ty * p.w + k.tx;
        k.loads = k.valid && k.ty < p.h;
        k.hasA = k.loads && lane < ppw;
        k.hasB = k.hasA && k.tx + 1u < p.w;
    }
    return k;
}

// IN_MODE: how a chunk's pixels are fetched.
//   kInFast : W % 16 == 0 and a 16-byte aligned base: one aligned 16-byte load per image row and lane.
//   kInRaw  : any geometry with W >= 16: ONE (unaligned) 16-byte load per image row and lane as well -- the lane's
//             two tiles are neighbours in one tile row by construction (chunk_ref).  Near the right edge
//             the 16 bytes that END at the row's last pixel are fetched instead (always inside the image) and
//             shifted into place, padded (dbde_util.cpp:116-132) or dropped when they are consumed.  Unaligned
//             reads cost 8 %; two 8-byte loads per row (round 1) cost twice the vector-memory instructions.
//   kInBytes: images narrower than 16 pixels (byte by byte).
// A template parameter, not a run-time branch: the number of loads a step issues must be static for them to
// stay in flight.
//   kInRaw4 : kInRaw with every fetch moved down to a DWORD boundary (round 4).  A 16-byte load at an odd address runs at
//             0.87 of the rate of one at ANY even address (profiles/read_align_probe.hip, 1921 wide: 5.99 against
//             6.89 TB/s; rows that are merely 2-byte aligned already read at the full rate), and odd widths put seven
//             of every eight image rows at odd addresses.  A lane fetches the 16 bytes from `address & ~3`; the b = address & 3
//             bytes it is then short of at the top are the first bytes of the NEXT lane's fetch (same image row, same b):
//             one DPP move (wave_shl:1) and four v_alignbyte_b32 per image row put the lane's 16 bytes back together when
//             the registers are consumed (load_fixup_generic).  The wave's last lane has no next lane, so a wave owns 63
//             pairs and its 64th lane fetches the pair after them (chunk_ref).  The last pair of a tile row is followed
//             by the next tile row's first pair, whose bytes are not its continuation: the host takes this form only
//             when the bytes such a lane is short of lie behind column W (EncPlan), where they are padding anyway.
//             Image rows whose fetches must not move -- the batch's last image row, whose last fetch is moved LEFT to
//             end with the buffer, and a first row that would start in front of it -- are fetched as kInRaw does, by
//             the whole wave (a wave-uniform vote, taken again when the registers are consumed).
//   kInRow  : kInRaw4 with the lanes dealt differently: a wave works on one SEGMENT of one tile row (chunk_ref) -- the
//             tile row is cut into ceil(pairs / 63) nearly equal segments -- so the image row, the segment's start and the
//             shift b are the same for all its lanes: they live in scalar registers, a fetch is `scalar base + 16 * lane`
//             (no vector arithmetic per image row), b needs no per-lane work when the registers are consumed, and "does this
//             wave hold a row end / the batch's last fetch" are scalar tests instead of votes.  Idle lanes are what it
//             costs: the host takes it when the segments fill at least 90 % of their waves (1921 wide: 121 pairs = 61 + 60
//             of 128 lanes) and kInRaw4's linear deal otherwise.
constexpr int kInFast = 0, kInRaw = 1, kInBytes = 2, kInRaw4 = 3, kInRow = 4;

// kInRow: the wave's place in fetch row r -- byte offset in the frame of its segment's start, and by how many bytes (0..3) the row's
// fetches are moved down to a dword boundary.  All scalar.  Rows that must stay where they are: the batch's last image row
// in the wave that holds its moved-left last fetch (`pinned`), and the batch's very first fetch when it would start in
// front of the caller's buffer.
struct RowSeg {
    uint32_t ty, j0, base_lo;
    bool pin, first;
    __device__ __forceinline__ uint32_t row(const EncParams &p, int r, uint32_t &off, bool &pinned) const {
        uint32_t yy = 8u * ty + (uint32_t)r;
        yy = yy < (uint32_t)p.H ? yy : (uint32_t)p.H - 1u;       // rows below the image repeat its last row
        off = yy * (uint32_t)p.W + 16u * j0;
        pinned = pin && yy == (uint32_t)p.H - 1u;
        return (pinned || (first && r == 0)) ? 0u : (base_lo + off) & 3u;
    }
};
__device__ __forceinline__ RowSeg row_seg(const EncParams &p, const ChunkRef &k, const uint8_t *img) {
    RowSeg s;
    // (wave-uniform by construction; said again where the compiler has lost track of it across the pipeline's loop)
    const bool row = k.valid && k.ty < p.h;                      // (a wave behind the frame's last tile row fetches like its first)
    s.ty = (uint32_t)__builtin_amdgcn_readfirstlane((int)(row ? k.ty : 0u));
    s.j0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(row ? k.seg_j0 : 0u));
    s.base_lo = (uint32_t)reinterpret_cast<uintptr_t>(img);
    s.pin = __builtin_amdgcn_readfirstlane((int)(row && k.seg_last && k.f == p.last_frame)) != 0;
    s.first = k.f == 0u && s.ty == 0u && s.j0 == 0u && (s.base_lo & 3u) != 0u;
    return s;
}

// kInRaw4: which of a chunk's eight fetch rows must stay where they are, for the whole wave (bit r: row r) -- the batch's
// last image row when the wave holds its moved-left last fetch, and the batch's very first fetch when moving it down
// would start in front of the caller's buffer.  One vote per chunk in the common case (nobody has such a fetch); the
// answer is wave-uniform, taken in front of the loads (no control flow between them, tests/test_kernel_listing.py).
__device__ __forceinline__ uint32_t raw4_natural_rows(const EncParams &p, const ChunkRef &k, uint32_t ty, uint32_t tx, bool at_end) {
    const bool first = k.f == 0u && ty == 0u && tx == 0u && ((uint32_t)reinterpret_cast<uintptr_t>(p.images) & 3u) != 0u;
    uint32_t rows = 0u;
    if (__any((int)(at_end || first))) {
        const int rclamp = p.H - 1 - 8 * (int)ty;     // rows from here on repeat the image's last row
        const uint32_t m = (at_end && rclamp <= 7 ? (0xFFu << (rclamp < 0 ? 0 : rclamp)) & 0xFFu : 0u) | (first ? 1u : 0u);
#pragma unroll
        for (int r = 0; r < 8; r++) rows |= __any((int)((m >> r) & 1u)) ? 1u << r : 0u;
    }
    return __builtin_amdgcn_readfirstlane(rows);
}

// The eight fetch rows of a kInRaw4 lane: where each fetch would naturally start (byte offset in the frame; a frame has
// fewer than 2^31 pixels when this form is taken (EncPlan), so 32 bits and ONE multiplication per chunk) and by how many bytes (0..3) it is moved down to a
// dword boundary.  Worked out twice, when the loads are issued and when their registers are consumed, from the same
// inputs: the kernel sits at its VGPR limit and carries nothing from one to the other.
struct Raw4Rows {
    uint32_t off0, off_max, base_lo, nat_rows;
    int dx_end;
    __device__ __forceinline__ uint32_t row(const EncParams &p, int r, uint32_t &off) const {
        uint32_t o = off0 + (uint32_t)r * (uint32_t)p.W;
        o = o < off_max ? o : off_max;                           // rows below the image repeat its last row
        const bool nat = (nat_rows >> r) & 1u;                   // (wave-uniform)
        if (nat) o = (uint32_t)((int)o + (o == off_max ? dx_end : 0));   // the moved-left fetch of the batch's last row
        off = o;
        return nat ? 0u : (base_lo + o) & 3u;
    }
};
__device__ __forceinline__ Raw4Rows raw4_rows(const EncParams &p, const ChunkRef &k, const uint8_t *img, uint32_t ty, uint32_t tx, bool at_end) {
    Raw4Rows rw;
    rw.nat_rows = raw4_natural_rows(p, k, ty, tx, at_end);
    rw.off0 = 8u * ty * (uint32_t)p.W + 8u * tx;
    rw.off_max = (uint32_t)(p.H - 1) * (uint32_t)p.W + 8u * tx;
    rw.base_lo = (uint32_t)reinterpret_cast<uintptr_t>(img);
    rw.dx_end = at_end ? p.W - 16 - 8 * (int)tx : 0;
    return rw;
}

// PIX == 2 (DBDE16, kInFast only): the lane's 16 bytes of a row are ONE tile's eight U16 pixels -- va holds the
// left half (pixels 0..3) of each row, vb the right half; frame_pixels counts bytes there.
template <int IN_MODE, bool ZERO = true, int PIX = 1>
__device__ __forceinline__ void load_chunk(const EncParams &p, const ChunkRef &k, uint32_t (&va)[16],
                                           uint32_t (&vb)[16]) {
    static_assert(PIX == 1 || IN_MODE == kInFast || IN_MODE == kInRaw, "DBDE16: aligned rows, or fetches where they lie (U16 rows always start at even addresses)");
    if (ZERO) {   // (callers that never read registers of tile-less lanes skip this)
#pragma unroll
        for (int i = 0; i < 16; i++) { va[i] = 0; vb[i] = 0; }
    }
    const uint8_t *img = p.images + (size_t)k.f * p.frame_pixels;
    if (IN_MODE == kInRow) {
        // scalar base + one vector offset that is the same in all eight rows: 16 * lane -- but 0 for lanes that fetch nothing
        // (they read the segment's first bytes), and where the batch's last fetch must END with the buffer in a pinned row
        const RowSeg sg = row_seg(p, k, img);
        const uint32_t lane_off = k.loads ? 8u * k.tx - 16u * sg.j0 : 0u;
        const bool is_end = k.loads && (k.tx >> 1) + 1u == p.lanes_per_row;
        const uint32_t end_off = is_end ? (uint32_t)p.W - 16u - 16u * sg.j0 : lane_off;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint32_t off;
            bool pinned;
            const uint32_t b = sg.row(p, r, off, pinned);
            const uint8_t *sbase = img + (ptrdiff_t)(int32_t)(off - b);     // (signed: up to 3 bytes in front of the frame)
            const uint8_t *src = sbase + (size_t)(pinned ? end_off : lane_off);
            typedef u32x4_t __attribute__((aligned(1))) u32x4_unaligned;
            const u32x4_t q = DBDE_NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_unaligned *>(src))
                                      : *reinterpret_cast<const u32x4_unaligned *>(src);
            va[2 * r] = q[0]; va[2 * r + 1] = q[1];
            vb[2 * r] = q[2]; vb[2 * r + 1] = q[3];
        }
    } else if (IN_MODE == kInFast || IN_MODE == kInRaw || IN_MODE == kInRaw4) {
        constexpr bool RAW = IN_MODE == kInRaw || IN_MODE == kInRaw4;
        // No branch around the loads: a lane without tiles reads tile 0 of frame 0 and nobody looks
        // at the result.  With a conditional issue the compiler cannot know how many loads are in
        // flight and makes the statistics of the CURRENT chunk wait for these as well.
        const uint32_t ty = k.loads ? k.ty : 0u, tx = k.loads ? k.tx : 0u;
        const uint32_t x0 = (PIX == 2 ? 16u : 8u) * tx;   // byte column
        // kInRaw: the 16 bytes of a row's last lane run into the next image row -- harmless, those bytes are replaced by
        // the constant padding (load_fixup_generic) -- except in the last image row of the batch, where they would pass
        // the end of the caller's buffer: there, and only there, the fetch is moved left to END at the row's last pixel
        const uint32_t row_bytes = (uint32_t)(PIX * p.W);
        const bool at_end = RAW && k.f == p.last_frame && x0 + 16u > row_bytes;
        Raw4Rows rw;
        if (IN_MODE == kInRaw4) rw = raw4_rows(p, k, img, ty, tx, at_end);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint8_t *src;
            if (IN_MODE == kInRaw4) {
                uint32_t off;
                const uint32_t b = rw.row(p, r, off);
                // (signed: the first fetch of a frame starts up to 3 bytes in front of it, in the frame before)
                src = img + (ptrdiff_t)(int32_t)(off - b);
            } else {
                int yy = 8 * (int)ty + r;
                yy = yy < p.H ? yy : p.H - 1;   // bottom padding = repeat the last row
                const uint32_t xr = at_end && yy == p.H - 1 ? row_bytes - 16u : x0;
                src = img + (size_t)yy * (size_t)(PIX * p.W) + xr;
            }
            u32x4_t q;
            if (IN_MODE == kInFast) {
                q = DBDE_NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(src))
                            : *reinterpret_cast<const u32x4_t *>(src);
            } else {   // any byte alignment (kInRaw4: dword aligned but for pinned rows): global memory takes it, the compiler must be told
                typedef u32x4_t __attribute__((aligned(1))) u32x4_unaligned;
                q = DBDE_NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_unaligned *>(src))
                            : *reinterpret_cast<const u32x4_unaligned *>(src);
            }
            va[2 * r] = q[0]; va[2 * r + 1] = q[1];
            vb[2 * r] = q[2]; vb[2 * r + 1] = q[3];
        }
    } else {   // images narrower than two tiles: byte by byte
        if (k.hasA) load_tile_generic(img, p.W, p.H, p.w, k.t0, va);
        if (k.hasB) load_tile_generic(img, p.W, p.H, p.w, k.t0 + 1u, vb);
    }
}

// Second half of the kInRaw load (see load_chunk): applied to the registers of `k` when they are used.  Only the
// last lane(s) of an image row have anything to do, so the work sits behind wave-uniform tests:
//   * the constant padding of a tile row's last tile(s): the last valid pixel repeated (dbde_util.cpp:116-128) --
//     branch-free on the two dwords of a tile row, per-lane masks (valid columns rm == 8 changes nothing);
//   * in the last image row of the batch the fetch had been moved left (load_chunk): shifted back first.
template <int IN_MODE, int PIX = 1>
__device__ __forceinline__ void load_fixup_generic(const EncParams &p, const ChunkRef &k, uint32_t (&va)[16],
                                                   uint32_t (&vb)[16]) {
    if (IN_MODE != kInRaw && IN_MODE != kInRaw4 && IN_MODE != kInRow) return;
    // (PIX == 2, DBDE16: ONE tile of 16 bytes per lane and row -- W below counts the BYTES of an image row)
    const uint32_t x0 = (PIX == 2 ? 16u : 8u) * k.tx, W = (uint32_t)(PIX * p.W);
    if (IN_MODE == kInRow) {   // as kInRaw4 below, with the shift of each row in a scalar register
        const RowSeg sg = row_seg(p, k, p.images + (size_t)k.f * p.frame_pixels);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint32_t off;
            bool pinned;
            const uint32_t b = sg.row(p, r, off, pinned);
            const uint32_t t0 = va[2 * r], t1 = va[2 * r + 1], t2 = vb[2 * r], t3 = vb[2 * r + 1];
            const uint32_t n0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t0, 0x130, 0xF, 0xF, false);   // wave_shl:1
            va[2 * r] = __builtin_amdgcn_alignbyte(t1, t0, b);
            va[2 * r + 1] = __builtin_amdgcn_alignbyte(t2, t1, b);
            vb[2 * r] = __builtin_amdgcn_alignbyte(t3, t2, b);
            vb[2 * r + 1] = __builtin_amdgcn_alignbyte(n0, t3, b);
        }
    }
    if (IN_MODE == kInRaw4) {
        // every fetch had been moved down to a dword boundary (load_chunk): the lane's 16 bytes are bytes b .. b + 15 of its
        // own four dwords followed by the next lane's first one (wave_shl:1 -- lane 63 receives nothing and owns no tile)
        const uint32_t ty = k.loads ? k.ty : 0u, tx = k.loads ? k.tx : 0u;
        const Raw4Rows rw = raw4_rows(p, k, p.images + (size_t)k.f * p.frame_pixels, ty, tx, k.f == p.last_frame && 8u * tx + 16u > W);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint32_t off;
            const uint32_t b = rw.row(p, r, off);
            const uint32_t t0 = va[2 * r], t1 = va[2 * r + 1], t2 = vb[2 * r], t3 = vb[2 * r + 1];
            const uint32_t n0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t0, 0x130, 0xF, 0xF, false);   // wave_shl:1
            va[2 * r] = __builtin_amdgcn_alignbyte(t1, t0, b);
            va[2 * r + 1] = __builtin_amdgcn_alignbyte(t2, t1, b);
            vb[2 * r] = __builtin_amdgcn_alignbyte(t3, t2, b);
            vb[2 * r + 1] = __builtin_amdgcn_alignbyte(n0, t3, b);
        }
    }
    // the lane that holds a tile row's last tile: tile A when w is odd (the pair's second tile does not exist), else tile B
    const bool last_lane = k.hasA && k.tx + (PIX == 2 ? 1u : 2u) >= p.w;
    const bool at_end = last_lane && k.f == p.last_frame && x0 + 16u > W;
    if (__any((int)at_end)) {   // (once per launch, in the workgroup that holds the batch's last tile row)
        const uint32_t sh = x0 + 16u - W;            // bytes the fetch was moved left: 1..15
#pragma unroll
        for (int r = 0; r < 8; r++) {
            int yy = 8 * (int)k.ty + r;
            yy = yy < p.H ? yy : p.H - 1;
            if (at_end && yy == p.H - 1) {
                uint64_t lo = ((uint64_t)va[2 * r + 1] << 32) | va[2 * r], hi = ((uint64_t)vb[2 * r + 1] << 32) | vb[2 * r];
                if (sh >= 8u) { lo = hi >> (8u * (sh - 8u)); hi = 0; }
                else { lo = (lo >> (8u * sh)) | (hi << (64u - 8u * sh)); hi >>= 8u * sh; }
                va[2 * r] = (uint32_t)lo; va[2 * r + 1] = (uint32_t)(lo >> 32);
                vb[2 * r] = (uint32_t)hi; vb[2 * r + 1] = (uint32_t)(hi >> 32);
            }
        }
    }
    // The constant padding.  Only a tile row's LAST tile can be partial, it has the same W - 8 (w - 1) valid columns in
    // every tile row, and it is the lane's tile A or B by the parity of w: the masks are launch constants (scalar), one
    // tile is patched, not two (round 3 worked out per-lane masks for both tiles: twice the work in every wave that holds
    // a row end -- more than half of them at 241 tiles across).
    if (PIX == 2) {   // the row's last tile: rm valid U16 columns, the last of them repeated (oracle/dbde16_oracle.c)
        const uint32_t rm = (uint32_t)p.W - 8u * (p.w - 1u);          // 1..8
        if (rm == 8u || !__any((int)last_lane)) return;
        const uint32_t src = (rm - 1u) >> 1, hi_half = (rm - 1u) & 1u;    // dword and half of the last valid pixel (launch constants)
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint32_t d[4] = {va[2 * r], va[2 * r + 1], vb[2 * r], vb[2 * r + 1]};
            const uint32_t sv = src == 0u ? d[0] : src == 1u ? d[1] : src == 2u ? d[2] : d[3];
            const uint32_t f = ((hi_half ? sv >> 16 : sv) & 0xFFFFu) * 0x00010001u;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // columns 2j, 2j + 1 of the tile row: both valid, the first only, neither
                const uint32_t keep = rm >= 2u * (uint32_t)j + 2u ? 0xFFFFFFFFu : rm == 2u * (uint32_t)j + 1u ? 0x0000FFFFu : 0u;
                const uint32_t kk = last_lane ? keep : 0xFFFFFFFFu;
                d[j] = (d[j] & kk) | (f & ~kk);
            }
            va[2 * r] = d[0]; va[2 * r + 1] = d[1]; vb[2 * r] = d[2]; vb[2 * r + 1] = d[3];
        }
        return;
    }
    const uint32_t rm = W - 8u * (p.w - 1u);          // 1..8
    if (rm == 8u || !__any((int)last_lane)) return;
    const uint32_t m0 = rm >= 4u ? 0xFFFFFFFFu : (1u << (8u * rm)) - 1u;                 // valid bytes of the low dword
    const uint32_t m1 = rm <= 4u ? 0u : (1u << (8u * (rm - 4u))) - 1u;                   // ... of the high dword (rm < 8)
    const bool from_hi = rm > 4u;                                                        // where the last valid pixel is
    const uint32_t sh = 8u * ((rm - 1u) & 3u);
    const uint32_t k0 = last_lane ? m0 : 0xFFFFFFFFu, k1 = last_lane ? m1 : 0xFFFFFFFFu;   // (other lanes keep their bytes)
    auto patch = [&](uint32_t (&v)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t f = (((from_hi ? v[2 * r + 1] : v[2 * r]) >> sh) & 0xFFu) * 0x01010101u;   // the last valid pixel, repeated (dbde_util.cpp:116-128)
            v[2 * r] = (v[2 * r] & k0) | (f & ~k0);
            v[2 * r + 1] = (v[2 * r + 1] & k1) | (f & ~k1);
        }
    };
    if (p.w & 1u) patch(va);
    else patch(vb);
}

// One tile row -> 8*d-bit integer with two v_dot4_u32_u8 per 4 pixels (weights 1, 2^d);
// d == 8 (weights do not fit a byte) keeps the bytes as they are.
__device__ __forceinline__ uint64_t pack_row_dot(uint32_t lo, uint32_t hi, uint32_t d, uint32_t w_lo, uint32_t w_hi,
                                                 bool is8) {
    uint32_t g0 = __builtin_amdgcn_udot4(lo, w_lo, 0u, false) | (__builtin_amdgcn_udot4(lo, w_hi, 0u, false) << (2u * d));
    uint32_t g1 = __builtin_amdgcn_udot4(hi, w_lo, 0u, false) | (__builtin_amdgcn_udot4(hi, w_hi, 0u, false) << (2u * d));
    g0 = is8 ? lo : g0;
    g1 = is8 ? hi : g1;
    return (uint64_t)g0 | ((uint64_t)g1 << (4u * d));
}

// Subtract the minimum, pack each row to 8*d bits, concatenate rows into d U64 words of the
// wave's LDS region starting at word q.  Straight-line: EVERY row stores the word it is filling -- a word that
// is not complete yet is simply stored again by the next row with more bits in it -- so there is no branch on
// "did this row complete a word" (sixteen exec-mask regions per lane and step in round 1).  A tile without
// payload (d == 0) stores into the lane's own trash word behind the image instead.
__device__ __forceinline__ void pack_tile(const uint32_t (&v)[16], uint32_t mn, uint32_t d, uint64_t *pay, uint32_t q,
                                          uint32_t trash) {
    const uint32_t m4 = mn * 0x01010101u;   // every byte >= mn: no borrow crosses a byte
    const uint32_t w_lo = 1u | ((1u << d) << 8), w_hi = w_lo << 16;
    const bool is8 = d >= 8u;
    const uint32_t nb = 8u * d;
    uint32_t qq = d ? q : trash;
    uint64_t acc = 0;
    uint32_t fill = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint64_t row = pack_row_dot(v[2 * r] - m4, v[2 * r + 1] - m4, d, w_lo, w_hi, is8);
        const uint64_t merged = acc | (row << fill);
        pay[DBDE_ENC_SWZ_ALL ? swzq8(qq) : qq] = merged;
        const uint32_t nf = fill + nb;
        const bool emit = nf >= 64u;
        const uint64_t spill = (row >> 1) >> (63u - fill);   // the bits of row beyond the word (0 when fill == 0)
        acc = emit ? spill : merged;
        fill = nf & 63u;
        qq += emit ? 1u : 0u;
    }
}

// Depth 8 everywhere in the wave: the payload is the min-subtracted bytes, row by row.
__device__ __forceinline__ void pack_tile_d8(const uint32_t (&v)[16], uint32_t mn, uint64_t *pay, uint32_t q) {
    const uint32_t m4 = mn * 0x01010101u;
#pragma unroll
    for (int r = 0; r < 8; r++)
        pay[swzq8(q + (uint32_t)r)] = (uint64_t)(v[2 * r] - m4) | ((uint64_t)(v[2 * r + 1] - m4) << 32);
}

// ---- DBDE16 (PIX == 2; format: oracle/dbde16_oracle.c) -- the tile of a lane is va (left halves of the rows) + vb ----
__device__ __forceinline__ void tile_minmax16(const uint32_t (&va)[16], const uint32_t (&vb)[16], uint32_t &mn, uint32_t &mx) {
    uint32_t lo = pk_min_u16(va[0], vb[0]), hi = pk_max_u16(va[0], vb[0]);
#pragma unroll
    for (int i = 1; i < 16; i++) {
        lo = pk_min_u16(pk_min_u16(lo, va[i]), vb[i]);
        hi = pk_max_u16(pk_max_u16(hi, va[i]), vb[i]);
    }
    mn = (lo & 0xFFFFu) < (lo >> 16) ? (lo & 0xFFFFu) : (lo >> 16);
    mx = (hi & 0xFFFFu) > (hi >> 16) ? (hi & 0xFFFFu) : (hi >> 16);
}
// Four pixels (two dwords of two U16, already minus the minimum) -> the 4*d-bit integer p0 | p1<<d | p2<<2d | p3<<3d;
// a pair fits 32 bits (d <= 16).
__device__ __forceinline__ uint64_t pack4x16(uint32_t a, uint32_t b, uint32_t d) {
    const uint32_t lo = (a & 0xFFFFu) | ((a >> 16) << d);
    const uint32_t hi = (b & 0xFFFFu) | ((b >> 16) << d);
    return (uint64_t)lo | ((uint64_t)hi << (2u * d));
}
// The tile's d payload words into the wave's LDS region from word q on: the straight-line funnel of pack_tile over
// sixteen half rows of 4*d <= 64 bits (a tile row is the 8*d-bit integer at byte r*d of the tile's payload).
__device__ __forceinline__ void pack_tile16(const uint32_t (&va)[16], const uint32_t (&vb)[16], uint32_t mn, uint32_t d,
                                            uint64_t *pay, uint32_t q, uint32_t trash) {
    const uint32_t mn2 = mn * 0x00010001u;   // every 16-bit half >= mn: no borrow crosses a half
    const uint32_t nb = 4u * d;
    uint32_t qq = d ? q : trash;
    uint64_t acc = 0;
    uint32_t fill = 0;
#pragma unroll
    for (int h = 0; h < 16; h++) {
        const int r = h >> 1;
        const uint64_t bits = (h & 1) ? pack4x16(vb[2 * r] - mn2, vb[2 * r + 1] - mn2, d)
                                      : pack4x16(va[2 * r] - mn2, va[2 * r + 1] - mn2, d);
        const uint64_t merged = acc | (bits << fill);
        pay[qq] = merged;
        const uint32_t nf = fill + nb;
        const bool emit = nf >= 64u;
        acc = emit ? ((bits >> 1) >> (63u - fill)) : merged;
        fill = nf & 63u;
        qq += emit ? 1u : 0u;
    }
}

// Depth 16 everywhere in the wave: the payload is the min-subtracted pixels, half row by half row; lanes write at a
// 128-byte stride, so the image is swizzled exactly as for pack_tile_d8.
__device__ __forceinline__ void pack_tile16_d16(const uint32_t (&va)[16], const uint32_t (&vb)[16], uint32_t mn, uint64_t *pay, uint32_t q) {
    const uint32_t mn2 = mn * 0x00010001u;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        pay[swzq8(q + 2u * (uint32_t)r)] = (uint64_t)(va[2 * r] - mn2) | ((uint64_t)(va[2 * r + 1] - mn2) << 32);
        pay[swzq8(q + 2u * (uint32_t)r + 1u)] = (uint64_t)(vb[2 * r] - mn2) | ((uint64_t)(vb[2 * r + 1] - mn2) << 32);
    }
}

template <int PIX = 1>
__device__ __forceinline__ uint64_t frame_base_of(const EncParams &p, uint32_t f, uint32_t inf, uint32_t glob) {
    const uint64_t meta = 32ull + (uint64_t)(1 + PIX) * p.T;
    return p.slot_stride ? (uint64_t)f * p.slot_stride : (uint64_t)f * meta + 8ull * (uint64_t)(uint32_t)(glob - inf);
}

// Worker-side stores of one finished chunk: the per-tile depth/min bytes of this lane and the
// wave's contiguous payload range (LDS -> global, 16 B per lane).
template <bool ALIGNED_OUT, int PIX = 1>
__device__ __forceinline__ void store_wave_part(const EncParams &p, const ChunkRef &k, uint32_t meta4, uint32_t wbase,
                                                uint32_t wtot, uint32_t inf, uint32_t glob, const uint64_t *pay,
                                                int lane, bool swz) {
    const uint64_t meta = 32ull + (uint64_t)(1 + PIX) * p.T;
    uint8_t *fb = p.out + frame_base_of<PIX>(p, k.f, inf, glob);
    uint8_t *depth_arr = fb + 24;
    uint8_t *min_arr = fb + 28 + p.T;
    const uint32_t dA = meta4 & 0xFFu, dB = (meta4 >> 8) & 0xFFu, mnA = (meta4 >> 16) & 0xFFu, mnB = meta4 >> 24;
    if (PIX == 2) {   // one tile per lane: meta4 = depth | minimum << 16 (ALIGNED_OUT: fb % 8 == 0, T % 8 == 0 -> U16-aligned minima)
        if (k.hasA) {
            depth_arr[k.t0] = (uint8_t)dA;
            uint8_t *m = min_arr + 2ull * k.t0;
            if (ALIGNED_OUT) *reinterpret_cast<uint16_t *>(m) = (uint16_t)(meta4 >> 16);
            else { m[0] = (uint8_t)(meta4 >> 16); m[1] = (uint8_t)(meta4 >> 24); }
        }
    } else if (ALIGNED_OUT) {   // fb % 8 == 0 and T % 4 == 0: t0 is even, so both arrays are 2-aligned here
        if (k.hasB) {
            *reinterpret_cast<uint16_t *>(depth_arr + k.t0) = (uint16_t)(dA | (dB << 8));
            *reinterpret_cast<uint16_t *>(min_arr + k.t0) = (uint16_t)(mnA | (mnB << 8));
        } else if (k.hasA) {
            depth_arr[k.t0] = (uint8_t)dA;
            min_arr[k.t0] = (uint8_t)mnA;
        }
    } else {
        if (k.hasA) { depth_arr[k.t0] = (uint8_t)dA; min_arr[k.t0] = (uint8_t)mnA; }
        if (k.hasB) { depth_arr[k.t0 + 1] = (uint8_t)dB; min_arr[k.t0 + 1] = (uint8_t)mnB; }
    }
    uint8_t *dst = fb + meta + 8ull * ((uint64_t)inf + wbase);
    if (ALIGNED_OUT) {
        const uint32_t q0 = (uint32_t)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1u);   // 1: dst is 8 mod 16
        const uint32_t lead = q0 < wtot ? q0 : wtot;
        if (lead && lane == 0) *reinterpret_cast<uint64_t *>(dst) = pay[0];
        const uint32_t rest = wtot - lead;
        const uint32_t npairs = rest >> 1;
        // 16-byte pieces from here on.  The wave's range starts wherever the tiles in front of it ended: the pieces in
        // front of the first 128-byte boundary (h of them, 0..7) leave with the LAST h lanes of a first, partial
        // instruction, so that every other store instruction of the wave covers whole cache lines (a 1 KB wave store
        // that straddles lines runs at little more than half the rate of an aligned one, profiles/mempattern.hip)
#if DBDE_LINE_ALIGNED_STORES
        const uint32_t h = (uint32_t)((128u - (uint32_t)(reinterpret_cast<uintptr_t>(dst + 8u * lead) & 127u)) & 127u) >> 4;
        for (int base = h ? (int)h - 64 : 0; base < (int)npairs; base += 64) {
            const int i = base + lane;
            if (i < 0 || i >= (int)npairs) continue;
#else
        for (uint32_t i = lane; i < npairs; i += 64u) {
#endif
            const uint32_t q = lead + 2u * (uint32_t)i;
            ulonglong2 v2;
            v2.x = pay[swz ? swzq8(q) : q];
            v2.y = pay[swz ? swzq8(q + 1u) : q + 1u];
            if (DBDE_NT) {
                u32x4_t o;
                o[0] = (uint32_t)v2.x; o[1] = (uint32_t)(v2.x >> 32); o[2] = (uint32_t)v2.y; o[3] = (uint32_t)(v2.y >> 32);
                __builtin_nontemporal_store(o, reinterpret_cast<u32x4_t *>(dst + 8ull * q));
            } else {
                *reinterpret_cast<ulonglong2 *>(dst + 8ull * q) = v2;
            }
        }
        if ((rest & 1u) && lane == 63) {
            const uint32_t q = wtot - 1u;
            *reinterpret_cast<uint64_t *>(dst + 8ull * q) = pay[swz ? swzq8(q) : q];
        }
    } else {
#if DBDE_UNALIGNED_OUT_WORDS   // (round 1-4a form: one 8-byte store per word at whatever address it has)
        for (uint32_t q = lane; q < wtot; q += 64u) store_u64_any(dst + 8ull * q, pay[swz ? swzq8(q) : q]);
#else
        // Any alignment of the frame (tile counts that are no multiple of 4, odd slot strides or bases): per-lane 8-byte stores
        // at 2 or 4 mod 8 cost the encoder 12-15 % (1008x1000 0.63 against 1008x1008 0.72, profiles/r04b_unaligned.sh).  The
        // wave's byte range leaves as ALIGNED 16-byte blocks instead, whole cache lines per store instruction as above; a
        // block is bytes r .. r + 15 of three consecutive payload words (r = how far the range's blocks sit off its words,
        // the same for the whole wave: four v_alignbyte), the partial block at either end takes its bytes in 8 / 4 / 2 / 1
        // pieces (the neighbouring wave or frame field owns the rest of it).
        const uint32_t nbytes = 8u * wtot;
        if (nbytes != 0u) {
            const uint32_t sft = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u);
            uint8_t *blk0 = dst - sft;                                   // 16-byte aligned
            const uint32_t n_blocks = (sft + nbytes + 15u) >> 4;
            const uint32_t r = (16u - sft) & 7u;
            auto word = [&](uint32_t q) -> uint64_t { return pay[swz ? swzq8(q) : q]; };
            const uint32_t h = (uint32_t)((128u - (uint32_t)(reinterpret_cast<uintptr_t>(blk0) & 127u)) & 127u) >> 4;
            for (int base = h ? (int)h - 64 : 0; base < (int)n_blocks; base += 64) {
                const int j = base + lane;
                if (j < 0 || j >= (int)n_blocks) continue;
                const int off = 16 * j - (int)sft;                       // payload byte the block starts with (< 0: the first block)
                const bool first = off < 0, last = off + 16 > (int)nbytes;
                const uint32_t qa = first ? 0u : (uint32_t)off >> 3;
                const uint32_t rr = first ? 0u : r;
                // (words behind the wave's last one are read and not used: they lie inside the workgroup's payload images)
                // (two words, the block's low half, then the third: six dwords held at once made the instance spill)
                u32x4_t o;
                const bool low = rr < 4u;
                {
                    const uint64_t wa = word(qa), wb = word(qa + 1u);
                    const uint32_t d0 = (uint32_t)wa, d1 = (uint32_t)(wa >> 32), d2 = (uint32_t)wb, d3 = (uint32_t)(wb >> 32);
                    o[0] = __builtin_amdgcn_alignbyte(low ? d1 : d2, low ? d0 : d1, rr);
                    o[1] = __builtin_amdgcn_alignbyte(low ? d2 : d3, low ? d1 : d2, rr);
                    o[2] = low ? d2 : d3;       // (the low operand of the next one)
                    o[3] = d3;
                }
                {
                    const uint64_t wc = word(qa + 2u);
                    const uint32_t d4 = (uint32_t)wc, d5 = (uint32_t)(wc >> 32);
                    const uint32_t d3 = o[3];
                    o[2] = __builtin_amdgcn_alignbyte(low ? d3 : d4, o[2], rr);
                    o[3] = __builtin_amdgcn_alignbyte(low ? d4 : d5, low ? d3 : d4, rr);
                }
                if (!first && !last) {
                    if (DBDE_NT) __builtin_nontemporal_store(o, reinterpret_cast<u32x4_t *>(blk0 + 16ll * j));
                    else *reinterpret_cast<u32x4_t *>(blk0 + 16ll * j) = o;
                } else {
                    // o holds the range's bytes from the block's first valid one on: n of them go to where that one belongs
                    uint8_t *to = first ? dst : blk0 + 16ll * j;
                    uint32_t n = (first ? 16u - sft : 16u);
                    const uint32_t left = nbytes - (first ? 0u : (uint32_t)off);
                    n = n < left ? n : left;
                    uint64_t lo = ((uint64_t)o[1] << 32) | o[0];
                    const uint64_t hi = ((uint64_t)o[3] << 32) | o[2];
                    if (n & 8u) { store_u64_any(to, lo); to += 8; lo = hi; }
                    if (n & 4u) { const uint32_t x = (uint32_t)lo; __builtin_memcpy(to, &x, 4); to += 4; lo >>= 32; }
                    if (n & 2u) { const uint16_t x = (uint16_t)lo; __builtin_memcpy(to, &x, 2); to += 2; lo >>= 16; }
                    if (n & 1u) *to = (uint8_t)lo;
                }
            }
        }
#endif
    }
}

template <int IN_MODE, bool ALIGNED_OUT, int PIX = 1>
__global__ __launch_bounds__(kEncThreads, 4) void encode_kernel(EncParams p) {
    __shared__ __attribute__((aligned(16))) EncShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef DBDE_DIAG
    const uint64_t dg_entry = (uint64_t)wall_clock64();
    if (tid == 0) {   // when the launch's workgroups start (wall clock, 10 ns): earliest as a max of the complement
        const unsigned long long t = dg_entry;
        atomicMax(&p.diag[11], ~t);
        atomicMax(&p.diag[12], t);
    }
#endif

    // Roles, ranks and the claim mode -- settled without a single read-modify-write on the way to the first pixels.
    // What the prologue costs was read off the in-kernel timeline (-DDBDE_DIAG, profiles/abbench): round 3 -- arrivals on one
    // counter, then 511 pollers on one word until all had been seen -- held every launch for 22 us with one chunk of loads in
    // flight; 16 group counters in slots of their own still 17 us, because a workgroup's FIRST read-modify-write of a launch
    // comes back after 8 us on average (14 at worst) whatever it addresses, and the chunk id waited for it.  Now:
    //   * workgroup b > 0 ENCODES with rank b - 1: it announces itself with a plain store (arrive_flags[b], tagged with the
    //     launch's epoch: never cleared) and fetches chunk `rank` at once -- nothing to wait for;
    //   * workgroup 0 CHECKS and then SCANS: one wave sweeps the arrival flags; when every one of the G encoders has been
    //     seen running it settles the launch's mode with a CAS -- STATIC strides (chunk = rank + k G, no atomics: only safe
    //     when every encoding workgroup runs) -- and its 512 threads TELL everybody (one mode flag per workgroup, epoch-tagged
    //     as well; a workgroup polls only its OWN flag).  It then claims the scanner's role and runs the in-order scan;
    //   * a mode flag that has not come after 20 us (oversubscribed device: somebody -- workgroup 0, perhaps -- is not
    //     running) ends in the same CAS with the other verdict, TICKETS: every chunk id, the first included, is then drawn
    //     from one counter by a workgroup that is running -- dense in draw order, forward progress whatever the dispatch
    //     order (a single-address ticket per chunk costs 6-13 % at full speed, which is why it is the fallback).  Whoever
    //     decides so tells the others and, when workgroup 0 has not announced itself, claims the scanner's role in its
    //     place (the role is one CAS: exactly one workgroup scans, and it is one that runs).  A workgroup 0 that finds the
    //     role taken encodes like the others.
    const uint32_t G = gridDim.x - 1u;              // workgroups that encode
    const uint32_t n_grp = gridDim.x < kEncGroups ? gridDim.x : kEncGroups, grp = blockIdx.x % kEncGroups;   // (tail tickets)
    const uint32_t tag = p.launch_epoch << 2;
    uint32_t *const tickets = p.ctrl + kCtrlTickets;
    if (tid == 0) {
        // (a context whose sticky failure word is set has records in an unknown state: its launches do nothing until
        // dbde_hip_sync has reported the failure and the host has cleared the workspace.  A plain load: the word was last
        // written by an EARLIER kernel, so the caches are good for it)
        const uint32_t dead = *p.sticky;
        if ((p.flags & 512u) && blockIdx.x == 0u) {   // (tests: workgroup 0 turns up 60 us late -- the others give up waiting, one of them scans)
            const uint64_t t_late = wall_clock64();
            while (wall_clock64() - t_late < 6000ull) __builtin_amdgcn_s_sleep(8);
        }
        if (!dead) __hip_atomic_store(&p.arrive_flags[blockIdx.x], tag | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh.boot[0] = dead;
        sh.boot[1] = 0u;
        sh.acc[0] = 0; sh.acc[1] = 0;
    }
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(sh.boot[0]) != 0u) return;
    const uint32_t rank = blockIdx.x - 1u;          // (workgroup 0 has none)
    ChunkRef cur = chunk_ref<PIX>(p, blockIdx.x ? rank : 0xFFFFFFFFu, tid);
    uint32_t r0a[16], r0b[16], r1a[16], r1b[16];
    if (blockIdx.x != 0u) load_chunk<IN_MODE, true, PIX>(p, cur, r0a, r0b);          // in flight while the mode arrives
    lds_barrier();     // sh.boot is reused below (NOT __syncthreads: nobody waits for the pixels here)
    if (blockIdx.x == 0u) {
        // ---- the checker: have all G encoders announced themselves?  (or has somebody given up waiting) ----
        if (wave == 0) {
            uint32_t tell = 0u;
            const uint64_t t0 = wall_clock64();
            for (;;) {
                uint32_t seen = 0;
                for (uint32_t k = 1u + (uint32_t)lane; k <= G; k += 64u)
                    seen += __hip_atomic_load(&p.arrive_flags[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (tag | 1u) ? 1u : 0u;
                seen = wave_sum(seen);
                if (seen == G) {
                    const uint32_t verdict = (p.flags & 1u) ? 2u : 1u;
                    uint32_t old = 0;
                    if (lane == 0) old = atomicCAS(&p.ctrl[kCtrlVerdict], 0u, verdict);
                    old = __builtin_amdgcn_readfirstlane(old);
                    tell = old ? 0u : verdict;
                    break;
                }
                if (__hip_atomic_load(&p.ctrl[kCtrlVerdict], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;   // somebody timed out
                if (wall_clock64() - t0 > 300000000ull) { if (lane == 0) atomicOr(p.sticky, 1u); break; }                 // 3 s: give up
                __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0) sh.boot[1] = tell;
        }
        __syncthreads();
        {
            const uint32_t tell = __builtin_amdgcn_readfirstlane(sh.boot[1]);
            if (tell)
                for (uint32_t i = (uint32_t)tid; i < gridDim.x; i += (uint32_t)kEncThreads)
                    __hip_atomic_store(&p.mode_flags[i], tag | tell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) sh.boot[0] = atomicCAS(&p.ctrl[kCtrlScanner], 0u, 1u) == 0u ? 1u : 0u;
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(sh.boot[0]) != 0u) {
            if (wave == 0) {
                // the control words of the NEXT launch (the host alternates between two sets) are cleared here: no memset
                if (lane < (int)kEncGroups) p.ctrl_next[kCtrlTail + lane * kCtrlSlot] = 0u;
                if (lane < 8) p.ctrl_next[kCtrlVerdict + lane] = 0u;
                scanner_loop<ALIGNED_OUT>(p, lane);
            }
            return;
        }
        // the role is taken (a workgroup that gave up waiting for this one): encode like the others, on tickets
    }
    if (tid == 0) {
        uint32_t mode = 0u, tell = 0u, scan = 0u;
        const uint64_t t0 = wall_clock64();
#ifdef DBDE_DIAG
        sh.dg[0] = t0;
        sh.dg[1] = t0;   // (first chunk's loads issued, barrier passed: the poll starts)
#endif
        for (;;) {
            const uint32_t f = scalar_load_uncached(&p.mode_flags[blockIdx.x]);   // (not behind the pixel loads in flight)
            if ((f & ~3u) == tag && (f & 3u)) { mode = f & 3u; break; }
            if (blockIdx.x == 0u || wall_clock64() - t0 > 2000ull) {   // 20 us: not everybody is running
                const uint32_t old = atomicCAS(&p.ctrl[kCtrlVerdict], 0u, 2u);
                mode = old ? old : 2u;             // (a verdict that was on its way is as good as the flag)
                if (!old) {
                    tell = 2u;
                    // nobody scans unless workgroup 0 runs: has it announced itself?  if not, this workgroup takes the role
                    if (__hip_atomic_load(&p.arrive_flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (tag | 1u))
                        scan = atomicCAS(&p.ctrl[kCtrlScanner], 0u, 1u) == 0u ? 1u : 0u;
                }
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        sh.boot[1] = mode | (tell << 8) | (scan << 16);
        if (mode == 1u) {
            sh.boot[0] = rank + G;
        } else if (!scan) {   // tickets: the chunk fetched above is dropped, ids are dense in draw order from 0 (the second draw: below)
            sh.claim[0] = atomicAdd(tickets, 1u);
        }
    }
    lds_barrier();
    const bool static_mode = (__builtin_amdgcn_readfirstlane(sh.boot[1]) & 0xFFu) == 1u;
    if (((__builtin_amdgcn_readfirstlane(sh.boot[1]) >> 8) & 0xFFu) != 0u)   // this workgroup settled it by time-out: everybody is told
        for (uint32_t i = (uint32_t)tid; i < gridDim.x; i += (uint32_t)kEncThreads)
            __hip_atomic_store(&p.mode_flags[i], tag | 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((__builtin_amdgcn_readfirstlane(sh.boot[1]) >> 16) != 0u) {          // ... and it scans, in workgroup 0's place
        if (wave == 0) {
            if (lane < (int)kEncGroups) p.ctrl_next[kCtrlTail + lane * kCtrlSlot] = 0u;
            if (lane < 8) p.ctrl_next[kCtrlVerdict + lane] = 0u;
            scanner_loop<ALIGNED_OUT>(p, lane);
        }
        return;
    }
    if (!static_mode) {
        cur = chunk_ref<PIX>(p, __builtin_amdgcn_readfirstlane(sh.claim[0]), tid);
        load_chunk<IN_MODE, true, PIX>(p, cur, r0a, r0b);
        // The second ticket is drawn BEHIND the first chunk's fetch, not with the first: two tickets drawn at once are
        // neighbouring chunk ids, and the in-order prefix then chains the workgroups one behind the other (chunk 2 a + 1 is
        // published only after its owner has waited for chunk 2 a's prefix, which needs 2 a - 1, ...: measured, 100 us per
        // step).  By now the workgroups that run have all drawn their first.  (Ids stay increasing inside a workgroup: one
        // that held a smaller id for later would wait for itself.)
        if (tid == 0) sh.boot[0] = atomicAdd(tickets, 1u);
        lds_barrier();
    }
    // Static strides leave the launch's tail to chance: workgroups do not run at the same speed (the first leaves 30-40 us
    // before the last on a 1-4 ms launch).  The last kTailRounds rounds of chunk ids -- everything from s_static on, the
    // same boundary for every workgroup -- are therefore drawn as tickets even in static mode: a ticket belongs to a
    // running workgroup, and a fast workgroup simply draws more of them.  Sixteen counters again, one per group: group g
    // hands out the ids s_static + 16 t + g, so the 32 workgroups of a group level out among themselves and no counter
    // sees more than 32 draws per round (511 draws per round on one counter take longer than the round).
    // (A workgroup's first two chunks, rank and rank + G, are always static: the boundary lies at 2 G or above.)
    constexpr uint32_t kTailRounds = 3;
    const uint32_t full_rounds = p.n_chunks / G;
    const uint32_t s_static = full_rounds >= kTailRounds + 2u ? (full_rounds - kTailRounds) * G : 0xFFFFFFFFu;
    ChunkRef nxt = chunk_ref<PIX>(p, __builtin_amdgcn_readfirstlane(sh.boot[0]), tid);
    ChunkRef prev = chunk_ref<PIX>(p, 0xFFFFFFFFu, tid);
    uint64_t *pay = sh.pay[wave];
    uint32_t prev_meta = 0, prev_wbase = 0, prev_wtot = 0, prev_total = 0;
    bool prev_swz = false;   // prev's payload image is swizzled (its wave was all depth 8)
#ifdef DBDE_DIAG
    uint64_t dg_wait = 0, dg_nwait = 0, dg_bar = 0, dg_npoll = 0, dg_first = 0;
    const uint64_t dg_k0 = __builtin_amdgcn_s_memtime();
    // per-workgroup timeline (wall clock, 10 ns): [0] entry, [1] mode agreed, [2..7] end of steps 0..5, [8] first step that
    // had no chunk left to prefetch, [9] exit, [10] steps run, [11] chunks packed
    uint64_t tr[12] = {dg_entry, (uint64_t)wall_clock64(), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

    // One pipeline step.  `ca/cb` hold the pixels of cur (loaded one step ago), `na/nb` receive
    // those of nxt; the caller alternates the two register sets instead of copying them, so the
    // loads issued here are not waited for until the NEXT step's statistics.
    // A step never leaves early: once the workgroup has run out of chunks (or a look-back timed out) the
    // remaining step of the pair runs empty -- dummy loads, no record, no stores -- so that the loop below
    // has ONE exit, behind step 1.  With an exit between the two steps the compiler's wait-count analysis
    // sees an edge from the end of step 0 back to the loop header, believes the registers step 0 has just
    // requested may still be in flight when step 0 starts again, and protects their reuse as address
    // temporaries with vmcnt(0): every step then waited for ALL of its previous stores to be acknowledged
    // before it issued its prefetch (tests/test_kernel_listing.py pins the absence of that wait).
    auto step = [&](const uint32_t par, uint32_t (&ca)[16], uint32_t (&cb)[16], uint32_t (&na)[16],
                    uint32_t (&nb)[16]) __attribute__((always_inline)) -> void {

        // ---- mailbox of prev (its prefixes, needed by the stores behind the barrier) and the next ticket ----
        auto mailbox = [&]() __attribute__((always_inline)) {
            if (tid == 0) {
                uint32_t inf = 0, glob = 0, ok = 1u;
                if (prev.valid) {
                    uint32_t inf_incl = 0, glob_incl = 0;
#ifdef DBDE_DIAG
                    const uint64_t dg_w0 = __builtin_amdgcn_s_memtime();
#endif
#ifdef DBDE_DIAG
                    uint32_t dg_polls = 0;
                    ok = wait_inc(p.state, prev.c, inf_incl, glob_incl, &dg_polls) ? 1u : 0u;
                    dg_wait += __builtin_amdgcn_s_memtime() - dg_w0;
                    dg_nwait++;
                    dg_npoll += dg_polls;
                    dg_first += dg_polls == 1u ? 1u : 0u;
#else
                    ok = wait_inc(p.state, prev.c, inf_incl, glob_incl) ? 1u : 0u;
#endif
                    inf = inf_incl - prev_total;
                    glob = glob_incl - prev_total;
                    if (!ok) atomicOr(p.sticky, 1u);
                }
                uint32_t tnew = 0xFFFFFFFFu;   // id of the chunk after nxt
                if (nxt.valid) {
                    if (!static_mode) tnew = atomicAdd(tickets, 1u);
                    else if (nxt.c + G < s_static) tnew = nxt.c + G;
                    else tnew = s_static + atomicAdd(&p.ctrl[kCtrlTail + grp * kCtrlSlot], 1u) * n_grp + grp;
                }
                sh.lb[par][0] = inf;
                sh.lb[par][1] = glob;
                sh.lb[par][2] = ok;
                sh.lb[par][3] = tnew;
            }
        };
        auto load_nxt = [&]() __attribute__((always_inline)) { load_chunk<IN_MODE, true, PIX>(p, nxt, na, nb); };
        // ---- statistics of cur (dbde_util.cpp:30-68), offsets inside the wave, AGG ---------------------------
        uint32_t mnA, mxA, mnB, mxB, dA, dB, incl, wtot;
        auto statistics = [&]() __attribute__((always_inline)) {
            if (PIX == 2) {   // one 16-bit tile per lane: "A" is the tile, "B" stays empty
                if (IN_MODE == kInRaw) {
                    cur = chunk_ref<PIX>(p, cur.c, tid);
                    load_fixup_generic<IN_MODE, PIX>(p, cur, ca, cb);
                }
                tile_minmax16(ca, cb, mnA, mxA);
                dA = cur.hasA ? depth_of_range(mxA - mnA) : 0u;
                mnB = 0u; mxB = 0u; dB = 0u;
            } else {
                // the any-geometry forms sit at the register limit: the lane's place in the chunk is worked out again here
                // (ten VALU) instead of being carried from the step that issued the loads
                if (IN_MODE == kInRaw4 || IN_MODE == kInRaw || IN_MODE == kInRow) cur = chunk_ref<PIX>(p, cur.c, tid);
                load_fixup_generic<IN_MODE>(p, cur, ca, cb);
                tile_minmax(ca, mnA, mxA);
                tile_minmax(cb, mnB, mxB);
                dA = cur.hasA ? depth_of_range(mxA - mnA) : 0u;
                dB = cur.hasB ? depth_of_range(mxB - mnB) : 0u;
            }
            incl = wave_scan_incl(dA + dB);
            wtot = __builtin_amdgcn_readlane(incl, 63);
            if (lane == 0) {
                sh.tot[par][wave] = wtot;
                // The last wave to get here publishes the chunk's AGG record at once: it never waits
                // for a barrier or for the scanner.
                const uint32_t old = atomicAdd(&sh.acc[par], (1u << 24) | wtot);
                if ((old >> 24) == (uint32_t)(kEncWaves - 1)) {
                    sh.acc[par] = 0;   // next used two iterations from now
                    if (cur.valid) {
                        const uint32_t total = (old & 0xFFFFFFu) + wtot;
                        __hip_atomic_store(&p.state[cur.c], kStAgg | (u64a)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        };
        // Order (DBDE_POLL, A/B builds).  The shipped order is 0: lane 0 polls prev's mailbox FIRST -- prev's AGG was
        // published a whole step ago, so the scanner has normally answered -- then every wave issues its prefetch and
        // reduces cur.  Vector-memory results return in issue order, so a poll issued behind the prefetch would wait
        // for the pixels.  The alternatives (1: poll after the statistics; 2: wave 0 reduces cur first, polls, and only
        // then prefetches) measured equal or worse in round 2 (the prefetch then arrives late); the wait
        // that remains is the in-order prefix itself, not the position of the poll.
#if DBDE_POLL == 0
        mailbox();
        load_nxt();
        statistics();
#elif DBDE_POLL == 1
        load_nxt();
        statistics();
        mailbox();
#else
        if (wave == 0) {
            statistics();
            mailbox();
            load_nxt();
        } else {
            load_nxt();
            statistics();
        }
#endif
#ifdef DBDE_DIAG
        const uint64_t dg_b0 = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();   // ---- 4. the one workgroup barrier ----
#ifdef DBDE_DIAG
        dg_bar += __builtin_amdgcn_s_memtime() - dg_b0;
#endif
        const uint32_t inf = sh.lb[par][0], glob = sh.lb[par][1];
        // wave-uniform on purpose: a divergent exit test makes the compiler route every loop exit through one
        // flag-guarded block that also falls back into the loop header; that fake edge made it treat the
        // registers of the chunk two steps back as still in flight and wait for ALL outstanding stores
        // (vmcnt(0)) before issuing a step's pixel loads (tests/test_kernel_listing.py pins the fix).
        const uint32_t lb_ok = __builtin_amdgcn_readfirstlane(sh.lb[par][2]);
        const uint32_t next_id = __builtin_amdgcn_readfirstlane(sh.lb[par][3]);
        if (!lb_ok) {   // look-back time-out (sticky word already set): drop everything, finish empty
            prev.valid = false;
            cur.valid = cur.hasA = cur.hasB = false;
            nxt.valid = nxt.hasA = nxt.hasB = false;
            cur.c = nxt.c = 0xFFFFFFFFu;
        }
        uint32_t wbase = 0, cur_total = 0;
#pragma unroll
        for (int k = 0; k < kEncWaves; k++) {
            const uint32_t tk = sh.tot[par][k];
            wbase += k < wave ? tk : 0u;
            cur_total += tk;
        }

        // ---- 5. prev: LDS -> global; cur: pack over it (wave-private region) -----------------------
        if (prev.valid) {
            store_wave_part<ALIGNED_OUT, PIX>(p, prev, prev_meta, prev_wbase, prev_wtot, inf, glob, pay, lane, prev_swz);
            // what depends only on prefixes -- frame header, the I32 fields, per-frame offset and size -- is
            // written by the workgroup that holds the frame's first / last chunk (one lane, a few stores).  The
            // scanner used to do this; with 64 or fewer chunks per frame it then met a frame boundary in every
            // window of records and its rounds took twice as long (profiles/r02 diag).
            if (tid == 64 * (kEncWaves - 1) && (prev.cf == 0u || prev.cf == p.chunks_per_frame - 1u))
                write_frame_fields<ALIGNED_OUT, PIX>(p, prev.f, prev.cf, inf + prev_total, glob - inf);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // all depth 8: lanes write at a 128-byte stride, the image is swizzled; otherwise the offsets are as irregular
        // as the depths and the swizzle is only address arithmetic (mixed encode 0.68 -> 0.72 without it)
        // (DBDE16: depth 16 in every tile of the wave)
        const bool all8 = PIX == 2 ? (bool)__builtin_amdgcn_readfirstlane(__all(dA == 16u || !cur.hasA))
                                   : (bool)__builtin_amdgcn_readfirstlane(__all((dA == 8u || !cur.hasA) && (dB == 8u || !cur.hasB)));
        if (wtot != 0u) {
            const uint32_t offA = incl - (dA + dB), offB = offA + dA;
            if (PIX == 2) {
                if (!all8) pack_tile16(ca, cb, mnA, dA, pay, offA, kWaveWords + (uint32_t)lane);
                else if (cur.hasA) pack_tile16_d16(ca, cb, mnA, pay, offA);
            } else if (all8) {
                if (cur.hasA) pack_tile_d8(ca, mnA, pay, offA);
                if (cur.hasB) pack_tile_d8(cb, mnB, pay, offB);
            } else {
                pack_tile(ca, mnA, dA, pay, offA, kWaveWords + (uint32_t)lane);
                pack_tile(cb, mnB, dB, pay, offB, kWaveWords + (uint32_t)lane);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

#ifdef DBDE_DIAG
        {
            const uint64_t t_now = (uint64_t)wall_clock64();
#pragma unroll
            for (int i_ = 0; i_ < 4; i_++) if (tr[10] == (uint64_t)i_) tr[2 + i_] = t_now;
            if (!nxt.valid && !tr[8]) tr[8] = t_now;
        }
        tr[10]++;
        tr[11] += cur.valid ? 1u : 0u;
#endif
        // ---- rotate the pipeline -------------------------------------------------------------------
        prev = cur;
        prev_meta = PIX == 2 ? dA | (mnA << 16) : dA | (dB << 8) | (mnA << 16) | (mnB << 24);
        prev_wbase = wbase;
        prev_wtot = wtot;
        prev_swz = all8 || DBDE_ENC_SWZ_ALL;
        prev_total = cur_total;
        cur = nxt;
        nxt = chunk_ref<PIX>(p, lb_ok ? next_id : 0xFFFFFFFFu, tid);
    };
#ifdef DBDE_DIAG
    uint32_t dg_pair = 0;
    uint64_t dg_tp = wall_clock64(), dg_early = 0, dg_late = 0;   // (the constant 100 MHz clock: s_memtime follows the shader clock)
#endif
    do {
        step(0u, r0a, r0b, r1a, r1b);
        step(1u, r1a, r1b, r0a, r0b);
#ifdef DBDE_DIAG
        {   // how long a pair of steps takes early in the launch (pairs 2..9) and later (eight pairs from DBDE_DIAG_LATE on), in 10 ns
            const uint64_t t = wall_clock64();
            if (dg_pair >= 2u && dg_pair < 10u) dg_early += t - dg_tp;
            if (dg_pair >= (unsigned)DBDE_DIAG_LATE && dg_pair < (unsigned)DBDE_DIAG_LATE + 8u) dg_late += t - dg_tp;
            dg_tp = t;
            dg_pair++;
        }
#endif
    } while (cur.valid || prev.valid);
#ifdef DBDE_DIAG
    if (tid == 0) { atomicAdd(&p.diag[15], dg_early); atomicAdd(&p.diag[13], dg_late); }
    if (tid == 0) {
        atomicAdd(&p.diag[0], dg_wait);
        atomicAdd(&p.diag[1], dg_nwait);
        atomicAdd(&p.diag[2], __builtin_amdgcn_s_memtime() - dg_k0);
        atomicAdd(&p.diag[7], dg_bar);
        atomicAdd(&p.diag[8], 1ull);
        atomicAdd(&p.diag[9], dg_npoll);
        atomicAdd(&p.diag[10], dg_first);
        atomicMax(&p.diag[14], wall_clock64());   // ... and when the last one leaves
        tr[9] = (uint64_t)wall_clock64();
        if (blockIdx.x < 1024u)
            for (int i = 0; i < 12; i++) p.diag[16 + 16 * blockIdx.x + i] = tr[i];
            p.diag[16 + 16 * blockIdx.x + 6] = sh.dg[0];
            p.diag[16 + 16 * blockIdx.x + 7] = sh.dg[1];
    }
#endif
}

static int in_mode_of(const EncParams &p, bool fast_in) {
    return fast_in ? kInFast : (p.lanes_per_row ? (p.seg_per_row ? kInRow : p.pairs_per_wave == 63u ? kInRaw4 : kInRaw) : kInBytes);
}

// DBDE16 through the same persistent kernel (PIX = 2): W % 8 == 0 and a 16-byte aligned base, 512 tiles per chunk,
// EncParams::frame_pixels in BYTES.  Other geometries and small launches stay with dbde16_kernels.hip.
hipError_t launch_encode16_fast(const EncParams &p, bool fast_in, bool aligned_out, hipStream_t s) {
    dim3 block(kEncThreads);
    dim3 grid(p.n_chunks + 1u < p.grid_blocks ? p.n_chunks + 1u : p.grid_blocks);
    if (!fast_in) {   // any width from 8 pixels on, any (U16-aligned) base: fetches where they lie
        if (aligned_out) hipLaunchKernelGGL((encode_kernel<kInRaw, true, 2>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((encode_kernel<kInRaw, false, 2>), grid, block, 0, s, p);
    } else if (aligned_out) hipLaunchKernelGGL((encode_kernel<kInFast, true, 2>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((encode_kernel<kInFast, false, 2>), grid, block, 0, s, p);
    return hipGetLastError();
}

hipError_t launch_encode(const EncParams &p, bool fast_in, bool aligned_out, hipStream_t s) {
    dim3 block(kEncThreads);
    // resident capacity, scanner included (so that all encoding workgroups can be co-resident)
    dim3 grid(p.n_chunks + 1u < p.grid_blocks ? p.n_chunks + 1u : p.grid_blocks);
    switch (in_mode_of(p, fast_in) * 2 + (aligned_out ? 1 : 0)) {
        case kInFast * 2 + 1: hipLaunchKernelGGL((encode_kernel<kInFast, true>), grid, block, 0, s, p); break;
        case kInFast * 2 + 0: hipLaunchKernelGGL((encode_kernel<kInFast, false>), grid, block, 0, s, p); break;
        case kInRaw * 2 + 1: hipLaunchKernelGGL((encode_kernel<kInRaw, true>), grid, block, 0, s, p); break;
        case kInRaw * 2 + 0: hipLaunchKernelGGL((encode_kernel<kInRaw, false>), grid, block, 0, s, p); break;
        case kInBytes * 2 + 1: hipLaunchKernelGGL((encode_kernel<kInBytes, true>), grid, block, 0, s, p); break;
        case kInRaw4 * 2 + 1: hipLaunchKernelGGL((encode_kernel<kInRaw4, true>), grid, block, 0, s, p); break;
        case kInRaw4 * 2 + 0: hipLaunchKernelGGL((encode_kernel<kInRaw4, false>), grid, block, 0, s, p); break;
        case kInRow * 2 + 1: hipLaunchKernelGGL((encode_kernel<kInRow, true>), grid, block, 0, s, p); break;
        case kInRow * 2 + 0: hipLaunchKernelGGL((encode_kernel<kInRow, false>), grid, block, 0, s, p); break;
        default: hipLaunchKernelGGL((encode_kernel<kInBytes, false>), grid, block, 0, s, p); break;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// ENCODE, small launches: one workgroup per chunk, no scanner
// ---------------------------------------------------------------------------------------
// One frame per call (BASELINE configs[1] taken literally) is a chain of latencies, not a bandwidth problem:
// with the persistent encoder a single 4096x3072 frame took 14.5 us, most of it the hand-off
// AGG -> scanner round -> INC -> poll.  When the launch has no more chunks than the device holds workgroups,
// every workgroup takes ONE chunk -- chunk id = workgroup id -- and works out its own prefix: it publishes its word count
// and sums the counts of the chunks in front of it.  A record is {launch epoch 32 | words 32}: records of older
// launches carry older epochs (and the persistent encoder's AGG / INC records have bits 63:62 set, which no epoch
// has), so nothing is drawn, cleared or reset per launch -- round 2's arrival ticket in front of the pixel loads and
// its "last workgroup out clears the records" counter were 1.35 + 1.0 us of a 10.5 us launch (in-kernel timeline,
// -DDBDE_DIAG).  Forward progress does not rest on dispatch order or co-residency: a record that has not appeared
// after 30 us is computed by the waiting wave itself from the pixels (chunk_words_by_wave) -- every spin ends.
struct EncSharedSmall {
    uint64_t pay[kEncWaves][kWaveWords + 64];
    uint32_t tot[kEncWaves];
    uint32_t pre[kEncWaves][2];   // per wave: payload words in front of the chunk, in its frame / in the launch (partial sums)
    uint32_t acc;
};

// Payload words of chunk j, by one wave, from the pixels (the fallback of encode_small_kernel's record wait).
__device__ __forceinline__ uint32_t chunk_words_by_wave(const EncParams &p, uint32_t j, int lane) {
    uint32_t sum = 0;
    for (int tw = lane; tw < kEncThreads; tw += 64) {
        const ChunkRef k2 = chunk_ref(p, j, tw);
        const uint8_t *img = p.images + (size_t)k2.f * p.frame_pixels;
        uint32_t v[16], mn, mx;
        if (k2.hasA) { load_tile_generic(img, p.W, p.H, p.w, k2.t0, v); tile_minmax(v, mn, mx); sum += depth_of_range(mx - mn); }
        if (k2.hasB) { load_tile_generic(img, p.W, p.H, p.w, k2.t0 + 1u, v); tile_minmax(v, mn, mx); sum += depth_of_range(mx - mn); }
    }
    return wave_sum(sum);
}

template <int IN_MODE, bool ALIGNED_OUT>
__global__ __launch_bounds__(kEncThreads, 4) void encode_small_kernel(EncParams p) {
    __shared__ __attribute__((aligned(16))) EncSharedSmall sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    SF_DECL;
    SF_MARK(0);
    if (tid == 0) sh.acc = 0;
    __syncthreads();
    const uint32_t c = blockIdx.x;
    SF_MARK(1);
    const ChunkRef k = chunk_ref(p, c, tid);
    uint32_t ra[16], rb[16];
    load_chunk<IN_MODE>(p, k, ra, rb);
    SF_DRAIN();
    SF_MARK(2);
    load_fixup_generic<IN_MODE>(p, k, ra, rb);
    uint32_t mnA, mxA, mnB, mxB;
    tile_minmax(ra, mnA, mxA);
    tile_minmax(rb, mnB, mxB);
    const uint32_t dA = k.hasA ? depth_of_range(mxA - mnA) : 0u;
    const uint32_t dB = k.hasB ? depth_of_range(mxB - mnB) : 0u;
    const uint32_t incl = wave_scan_incl(dA + dB);
    const uint32_t wtot = __builtin_amdgcn_readlane(incl, 63);
    const u64a tag = (u64a)p.small_epoch << 32;
    if (lane == 0) {
        sh.tot[wave] = wtot;
        const uint32_t old = atomicAdd(&sh.acc, (1u << 24) | wtot);
        // last wave in: publish at once (flags bit 6, tests: odd chunks keep silent, as if they were not running yet)
        if ((old >> 24) == (uint32_t)(kEncWaves - 1) && !((p.flags & 64u) && (c & 1u)))
            __hip_atomic_store(&p.state[c], tag | (u64a)((old & 0xFFFFFFu) + wtot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    SF_MARK(3);
    // pack while the record travels (wave-private LDS region, wave-local offsets)
    uint64_t *pay = sh.pay[wave];
    const bool all8 = __builtin_amdgcn_readfirstlane(__all((dA == 8u || !k.hasA) && (dB == 8u || !k.hasB)));
    if (wtot != 0u) {
        const uint32_t offA = incl - (dA + dB), offB = offA + dA;
        if (all8) {
            if (k.hasA) pack_tile_d8(ra, mnA, pay, offA);
            if (k.hasB) pack_tile_d8(rb, mnB, pay, offB);
        } else {
            pack_tile(ra, mnA, dA, pay, offA, kWaveWords + (uint32_t)lane);
            pack_tile(rb, mnB, dB, pay, offB, kWaveWords + (uint32_t)lane);
        }
    }
    SF_MARK(4);
    // the chunk's prefixes: sum of the word counts in front of it, in its frame and in the launch.  The records are read
    // in groups of 64, group g by wave g mod 8: a single 4096x3072 frame (192 chunks) is one memory round trip for every
    // workgroup, where one wave walking the groups in turn paid up to three
    {
        const uint32_t fstart = k.f * p.chunks_per_frame;
        const uint64_t t_start = wall_clock64();
        uint32_t inf = 0, glob = 0;
        for (uint32_t base = 64u * (uint32_t)wave; base < c; base += 64u * (uint32_t)kEncWaves) {
            const uint32_t j = base + (uint32_t)lane;
            u64a w = 0;
            bool got = j >= c;
            for (;;) {
                if (!got) { w = __hip_atomic_load(&p.state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); got = (w >> 32) == (tag >> 32); }
                if (__all((int)got)) break;
                if (wall_clock64() - t_start > 3000ull) {   // 30 us: whoever has not published may not be running yet
                    uint64_t miss = __ballot((int)!got);
                    while (miss) {
                        const uint32_t kk = (uint32_t)__builtin_ctzll(miss);
                        miss &= miss - 1ull;
                        const uint32_t words = chunk_words_by_wave(p, base + kk, lane);
                        if ((uint32_t)lane == kk) { w = words; got = true; }
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            const uint32_t v = j < c ? (uint32_t)w : 0u;
            glob += v;
            inf += j >= fstart ? v : 0u;
        }
        inf = wave_sum(inf);
        glob = wave_sum(glob);
        if (lane == 0) { sh.pre[wave][0] = inf; sh.pre[wave][1] = glob; }
    }
    SF_MARK(5);
    __syncthreads();
    SF_MARK(6);
    uint32_t inf = 0, glob = 0;
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (int q = 0; q < kEncWaves; q++) {
        const uint32_t tk = sh.tot[q];
        wbase += q < wave ? tk : 0u;
        total += tk;
        inf += sh.pre[q][0];
        glob += sh.pre[q][1];
    }
    if (k.valid) {
        store_wave_part<ALIGNED_OUT>(p, k, dA | (dB << 8) | (mnA << 16) | (mnB << 24), wbase, wtot, inf, glob, pay, lane, all8 || DBDE_ENC_SWZ_ALL);
        if (tid == 64 * (kEncWaves - 1) && (k.cf == 0u || k.cf == p.chunks_per_frame - 1u))
            write_frame_fields<ALIGNED_OUT>(p, k.f, k.cf, inf + total, glob - inf);
    }
    SF_MARK(7);
#ifdef DBDE_DIAG
    SF_DRAIN();
    SF_MARK(8);
    SF_MARK(9);
    SF_FLUSH(p.diag, 9);
#endif
}

hipError_t launch_encode_small(const EncParams &p, bool fast_in, bool aligned_out, hipStream_t s) {
    dim3 block(kEncThreads), grid(p.n_chunks);
    switch (in_mode_of(p, fast_in) * 2 + (aligned_out ? 1 : 0)) {
        case kInFast * 2 + 1: hipLaunchKernelGGL((encode_small_kernel<kInFast, true>), grid, block, 0, s, p); break;
        case kInFast * 2 + 0: hipLaunchKernelGGL((encode_small_kernel<kInFast, false>), grid, block, 0, s, p); break;
        case kInRaw * 2 + 1: hipLaunchKernelGGL((encode_small_kernel<kInRaw, true>), grid, block, 0, s, p); break;
        case kInRaw * 2 + 0: hipLaunchKernelGGL((encode_small_kernel<kInRaw, false>), grid, block, 0, s, p); break;
        case kInBytes * 2 + 1: hipLaunchKernelGGL((encode_small_kernel<kInBytes, true>), grid, block, 0, s, p); break;
        case kInRaw4 * 2 + 1: hipLaunchKernelGGL((encode_small_kernel<kInRaw4, true>), grid, block, 0, s, p); break;
        case kInRaw4 * 2 + 0: hipLaunchKernelGGL((encode_small_kernel<kInRaw4, false>), grid, block, 0, s, p); break;
        case kInRow * 2 + 1: hipLaunchKernelGGL((encode_small_kernel<kInRow, true>), grid, block, 0, s, p); break;
        case kInRow * 2 + 0: hipLaunchKernelGGL((encode_small_kernel<kInRow, false>), grid, block, 0, s, p); break;
        default: hipLaunchKernelGGL((encode_small_kernel<kInBytes, false>), grid, block, 0, s, p); break;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// ENCODE, tiny frames (T <= 64 tiles), one slot per frame: one tile per lane, several frames per wave
// ---------------------------------------------------------------------------------------
// The mirror of the small-frame decoder, decode_mid_kernel (the reference's randomized test packs 1024 single-tile frames,
// dbde_util_test.cpp:66-96): a 1024-tile chunk per frame leaves the lanes empty and a record round trip per frame is all
// the kernels above would do.  A lane owns ONE tile (clamp-to-edge load: the constant padding of dbde_util.cpp:105-135),
// a wave 64 / T whole frames; the tile's word offset in its frame and the frame's n64 are a segmented wave scan; the
// tile's d payload words leave the lane as they are completed (the bit funnel of dbde_bits.h).  Frames in slots depend
// on nothing outside the frame, so there is no workspace and nothing to wait for; concatenated tiny frames keep the
// general path.
__global__ __launch_bounds__(256) void encode_tiny_kernel(EncParams p) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t T = p.T, fpw = 64u / T;
    const uint32_t fl = lane / T, t = lane - fl * T;
    const uint32_t f = (blockIdx.x * 4u + wave) * fpw + fl;
    const bool active = fl < fpw && f < p.n_chunks;          // (n_chunks carries the frame count here)
    uint32_t v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = 0;
    if (active) load_tile_generic(p.images + (size_t)f * p.frame_pixels, p.W, p.H, p.w, t, v);
    uint32_t mn, mx;
    tile_minmax(v, mn, mx);
    const uint32_t d = active ? depth_of_range(mx - mn) : 0u;
    const uint32_t incl = wave_scan_incl(d);
    const uint32_t first = fl * T;
    const uint32_t base = (uint32_t)__shfl((int)(incl - d), (int)(first < 64u ? first : 0u), 64);
    const uint32_t upto = (uint32_t)__shfl((int)incl, (int)(first + T - 1u < 64u ? first + T - 1u : 63u), 64);
    if (!active) return;
    const uint32_t prefix = incl - d - base, total = upto - base;
    uint8_t *fb = p.out + (uint64_t)f * p.slot_stride;
    fb[24u + t] = (uint8_t)d;
    fb[28u + T + t] = (uint8_t)mn;
    uint8_t *dst = fb + 32ull + 2ull * T + 8ull * prefix;
    const uint32_t mn4 = mn * 0x01010101u;   // every byte >= mn: no borrow crosses a byte
    Funnel fn;
    fn.reset();
#pragma unroll
    for (int r = 0; r < 8; r++) {
        uint64_t word;
        if (d != 0u && fn.push(pack_row(v[2 * r] - mn4, v[2 * r + 1] - mn4, d), 8u * d, word)) {
            store_u64_any(dst, word);
            dst += 8;
        }
    }
    if (t == 0u) write_frame_fields<false>(p, f, 0u, total, 0u);   // header, the I32 fields, per-frame offset and size
}

hipError_t launch_encode_tiny(const EncParams &p, uint32_t n_frames, hipStream_t s) {
    EncParams q = p;
    q.n_chunks = n_frames;
    q.chunks_per_frame = 1u;     // write_frame_fields: the one "chunk" is the frame's first and last
    const uint32_t per_wg = 4u * (64u / p.T);
    hipLaunchKernelGGL(encode_tiny_kernel, dim3((n_frames + per_wg - 1u) / per_wg), dim3(256), 0, s, q);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// ENCODE, frames between the tiny path and one chunk (64 < T <= 512 tiles: 72 .. 180 pixels a side), one slot per frame
// ---------------------------------------------------------------------------------------
// encode_tiny_kernel one level up: a lane still owns ONE tile, a WORKGROUP of 256 / 512 / 1024 threads holds as many
// whole frames as fit (the host takes the size they fill best); the segmented scan goes through LDS.  The persistent and
// small encoders give such a frame a whole 1024-tile chunk: 72x72 (T = 81) ran at 0.12 of peak, 96x96 at 0.18.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void encode_mid_kernel(EncParams p) {
    constexpr int NW = THREADS / 64;
    __shared__ uint32_t s_tot[NW];
    __shared__ uint32_t s_incl[THREADS];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t T = p.T, fpw = (uint32_t)THREADS / T;
    const uint32_t fl = tid / T, t = tid - fl * T;
    const uint32_t f = blockIdx.x * fpw + fl;
    const bool active = fl < fpw && f < p.n_chunks;          // (n_chunks carries the frame count here)
    uint32_t v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = 0;
    if (active) load_tile_generic(p.images + (size_t)f * p.frame_pixels, p.W, p.H, p.w, t, v);
    uint32_t mn, mx;
    tile_minmax(v, mn, mx);
    const uint32_t d = active ? depth_of_range(mx - mn) : 0u;
    uint32_t block_total;
    const uint32_t incl = block_scan_incl<NW>(d, s_tot, (int)lane, (int)wave, block_total);
    s_incl[tid] = incl;
    __syncthreads();
    if (!active) return;
    const uint32_t first = fl * T;
    const uint32_t base = first ? s_incl[first - 1u] : 0u;
    const uint32_t prefix = incl - d - base, total = s_incl[first + T - 1u] - base;
    uint8_t *fb = p.out + (uint64_t)f * p.slot_stride;
    fb[24u + t] = (uint8_t)d;
    fb[28u + T + t] = (uint8_t)mn;
    uint8_t *dst = fb + 32ull + 2ull * T + 8ull * prefix;
    const uint32_t mn4 = mn * 0x01010101u;   // every byte >= mn: no borrow crosses a byte
    Funnel fn;
    fn.reset();
#pragma unroll
    for (int r = 0; r < 8; r++) {
        uint64_t word;
        if (d != 0u && fn.push(pack_row(v[2 * r] - mn4, v[2 * r + 1] - mn4, d), 8u * d, word)) {
            store_u64_any(dst, word);
            dst += 8;
        }
    }
    if (t == 0u) write_frame_fields<false>(p, f, 0u, total, 0u);   // header, the I32 fields, per-frame offset and size
}

// Threads of the workgroup that whole frames of T tiles (64 < T <= 512) fill best: 256, 512 or 1024.
#ifndef DBDE_MID_DECODE_MAX_THREADS
#define DBDE_MID_DECODE_MAX_THREADS 256   // largest workgroup of the (persistent) mid decoder: 256 threads measured best or equal at every fill (96x96, one frame of 144 tiles per workgroup: 0.50 / 0.64 against 0.455 / 0.52 with seven frames on 1024 threads, one workgroup per CU)
#endif
uint32_t mid_threads_for(uint32_t T, uint32_t max_threads) {
    uint32_t best = max_threads, best_used = 0u;
    for (uint32_t th = 256u; th <= max_threads; th *= 2u) {
        if (th < T) continue;
        const uint32_t used = (th / T) * T;
        if ((uint64_t)used * best > (uint64_t)best_used * th) { best = th; best_used = used; }
    }
    return best;
}

#ifndef DBDE_MID_ENCODE_MAX_THREADS
#define DBDE_MID_ENCODE_MAX_THREADS 1024
#endif
uint32_t mid_encode_threads_for(uint32_t T) {
    return mid_threads_for(T, T <= (unsigned)DBDE_MID_ENCODE_MAX_THREADS ? (unsigned)DBDE_MID_ENCODE_MAX_THREADS : 1024u);
}
uint32_t mid_decode_threads_for(uint32_t T) {
    return mid_threads_for(T, T <= (unsigned)DBDE_MID_DECODE_MAX_THREADS ? (unsigned)DBDE_MID_DECODE_MAX_THREADS : 1024u);
}

hipError_t launch_encode_mid(const EncParams &p, uint32_t n_frames, hipStream_t s) {
    EncParams q = p;
    q.n_chunks = n_frames;
    q.chunks_per_frame = 1u;     // write_frame_fields: the one "chunk" is the frame's first and last
    const uint32_t th = mid_encode_threads_for(p.T), per_wg = th / p.T;
    const dim3 grid((n_frames + per_wg - 1u) / per_wg);
    if (th == 256u) hipLaunchKernelGGL(encode_mid_kernel<256>, grid, dim3(256), 0, s, q);
    else if (th == 512u) hipLaunchKernelGGL(encode_mid_kernel<512>, grid, dim3(512), 0, s, q);
    else hipLaunchKernelGGL(encode_mid_kernel<1024>, grid, dim3(1024), 0, s, q);
    return hipGetLastError();
}

// Resident workgroups per CU of the encoder (occupancy query; LDS- and VGPR-bound).
int encode_blocks_per_cu() {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, encode_kernel<kInFast, true>, kEncThreads, 0) != hipSuccess || n < 1)
        n = 1;
    return n;
}

// ---------------------------------------------------------------------------------------
// DECODE: index + validation (dbde_util.cpp:295-303), then the tile kernel
// ---------------------------------------------------------------------------------------

// One workgroup per frame.  Sums the depth bytes per chunk (16-byte aligned loads, bytes
// outside the array masked off), scans the chunk sums, validates the three I32 fields and
// parses the frame header.
__global__ __launch_bounds__(1024) void decode_index_kernel(IdxParams p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_sum[];   // [chunks_per_frame]
    __shared__ uint32_t s_part[16];
    __shared__ uint32_t s_flags;

    const uint32_t f = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t T = p.T, cpf = p.chunks_per_frame;
    const uint64_t off = p.frame_offsets[f];
    const uint64_t meta_t = (1ull + p.min_bytes) * T;   // bytes of the depth and minimum arrays
    const uint64_t need = 32ull + meta_t;   // header + metadata must lie inside the stream
    const bool in_range = in_extent(off, need, p.stream_bytes);
    const uint8_t *fb = p.stream + off;

    for (uint32_t k = tid; k < cpf; k += blockDim.x) s_sum[k] = 0;
    if (tid == 0) s_flags = 0;
    __syncthreads();

    uint32_t bad_depth = 0;
    if (in_range) {
        const uint8_t *darr = fb + 24;
        const uint32_t head = (uint32_t)(reinterpret_cast<uintptr_t>(darr) & 15u);   // bytes before the array in piece 0
        const uint8_t *a_lo = darr - head;                                          // 16-byte aligned
        const uint32_t npieces = (head + T + 15u) >> 4;
        // chunk of a piece's first tile, kept incrementally (whole-row form with >= 16 tiles per chunk: a piece
        // then meets at most one chunk boundary); everything else divides
        const bool incr = p.geom.pieces == 1u && p.geom.ct >= 16u;
        const uint32_t step = 16u * blockDim.x, q_step = incr ? step / p.geom.ct : 0u, r_step = incr ? step - q_step * p.geom.ct : 0u;
        uint32_t k_run = 0, rem_run = 0;
        bool have_run = false;
        for (uint32_t i = tid; i < npieces; i += blockDim.x) {
            uint4 q;
            if (a_lo + 16ull * (i + 1u) <= p.stream + p.stream_bytes) {
                q = *reinterpret_cast<const uint4 *>(a_lo + 16ull * i);
            } else {   // tiny frame at the very end of the extent (T < 7): the piece would cross it
                uint32_t wq[4] = {0, 0, 0, 0};
                for (uint32_t b = 0; b < 16u; b++) {
                    const uint8_t *src = a_lo + 16ull * i + b;
                    if (src < p.stream + p.stream_bytes) wq[b >> 2] |= (uint32_t)*src << (8u * (b & 3u));
                }
                q = make_uint4(wq[0], wq[1], wq[2], wq[3]);
            }
            uint32_t wv[4] = {q.x, q.y, q.z, q.w};
            // tile positions this piece covers: [lo, hi) inside [0, T)
            const long long pos0 = 16ll * i - (long long)head;
            const uint32_t lo = pos0 < 0 ? 0u : (uint32_t)pos0;
            const uint32_t hi = pos0 + 16 > (long long)T ? T : (uint32_t)(pos0 + 16);
            // keep only the bytes inside [lo, hi)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const long long pj = pos0 + 4 * j;
                uint32_t mask = 0xFFFFFFFFu;
                if (pj < 0) mask = (pj <= -4) ? 0u : (0xFFFFFFFFu << (8 * (int)(-pj)));
                if (pj + 4 > (long long)T) {
                    const long long keep = (long long)T - pj;   // bytes to keep
                    mask &= keep <= 0 ? 0u : (keep >= 4 ? 0xFFFFFFFFu : (0xFFFFFFFFu >> (8 * (int)(4 - keep))));
                }
                wv[j] &= mask;
                bad_depth |= depth_bytes_bad(wv[j], p.min_bytes);
            }
            uint32_t k_first, k_last;
            if (incr && pos0 >= 0) {
                if (!have_run) { k_run = lo / p.geom.ct; rem_run = lo - k_run * p.geom.ct; have_run = true; }
                k_first = k_run;
                k_last = rem_run + (hi - 1u - lo) >= p.geom.ct ? k_run + 1u : k_run;
                k_run += q_step; rem_run += r_step;
                if (rem_run >= p.geom.ct) { rem_run -= p.geom.ct; k_run++; }
            } else {
                k_first = dec_chunk_of(p.geom, lo);
                k_last = dec_chunk_of(p.geom, hi - 1u);
            }
            if (k_first == k_last || (incr && pos0 >= 0)) {
                // one chunk, or (every 32nd piece when the array sits 8 mod 16) ONE boundary inside the piece:
                // bytes below the boundary tile go to k_first, the rest to the next chunk -- no branches
                const long long bnd = k_first == k_last ? (long long)T : (long long)(k_first + 1u) * p.geom.ct;
                uint32_t sum_lo = 0, sum_hi = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const long long nb = bnd - (pos0 + 4 * j);   // bytes of dword j below the boundary
                    const uint32_t m = nb >= 4 ? 0xFFFFFFFFu : (nb <= 0 ? 0u : (0xFFFFFFFFu >> (8 * (int)(4 - nb))));
                    sum_lo += __builtin_amdgcn_sad_u8(wv[j] & m, 0u, 0u);
                    sum_hi += __builtin_amdgcn_sad_u8(wv[j] & ~m, 0u, 0u);
                }
                if (sum_lo) atomicAdd(&s_sum[k_first], sum_lo);
                if (sum_hi) atomicAdd(&s_sum[k_first + 1u], sum_hi);
            } else {                   // several boundaries possible (tiny chunks, wide frames in pieces): byte by byte
                for (uint32_t pos = lo; pos < hi; pos++) {
                    const uint32_t b = (uint32_t)((long long)pos - pos0);
                    const uint32_t v = (wv[b >> 2] >> (8u * (b & 3u))) & 0xFFu;
                    if (v) atomicAdd(&s_sum[dec_chunk_of(p.geom, pos)], v);
                }
            }
        }
    }
    if (bad_depth) atomicOr(&s_flags, 1u);
    __syncthreads();

    // exclusive scan of the chunk sums: each thread owns a contiguous segment
    const uint32_t seg = (cpf + blockDim.x - 1u) / blockDim.x;
    const uint32_t k0 = (uint32_t)tid * seg;
    uint32_t local = 0;
    for (uint32_t k = k0; k < k0 + seg && k < cpf; k++) local += s_sum[k];
    uint32_t incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t n = __shfl_up(incl, o, 64);
        if (lane >= o) incl += n;
    }
    if (lane == 63) s_part[wave] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); k++) {
        const uint32_t v = s_part[k];
        if (k < wave) base += v;
        total += v;
    }
    uint32_t run = base + incl - local;
    uint32_t *co = p.chunk_off + (size_t)f * (cpf + 1u);
    for (uint32_t k = k0; k < k0 + seg && k < cpf; k++) {
        co[k] = run;
        run += s_sum[k];
    }
    if (tid == 0) co[cpf] = total;

    if (tid == 0) {
        bool ok = in_range;
        uint32_t field = 0;
        uint64_t index = 0, elapsed = 0, consumed = 20;
        if (in_extent(off, 20, p.stream_bytes)) {
            field = load_u32_bytes(fb);
            index = load_u64_bytes(fb + 4);
            elapsed = f64_to_u64_x86(__longlong_as_double((long long)load_u64_bytes(fb + 12)));
        }
        if (ok) {
            const int32_t nb = (int32_t)load_u32_bytes(fb + 20);
            const int32_t nm = (int32_t)load_u32_bytes(fb + 24 + T);
            const int32_t n64 = (int32_t)load_u32_bytes(fb + 28 + meta_t);
            ok = nb == (int32_t)T && nm == (int32_t)(T * p.min_bytes) && n64 == (int32_t)total && !(s_flags & 1u);
            // the payload itself must also be inside the stream
            if (ok && !in_extent(off, need + 8ull * total, p.stream_bytes)) ok = false;
            if (ok) consumed = need + 8ull * total;
        }
        p.frame_ok[f] = ok ? 1u : 0u;
        if (p.results) {
            FrameResultDev *r = reinterpret_cast<FrameResultDev *>(p.results) + f;
            r->u64s = (field == 2u && ok) ? 2u : 0xFFFFFFFFu;   // dbde_util.cpp:335,342
            r->pad_ = 0;
            r->index = index;
            r->elapsed_ns = elapsed;
            r->consumed = consumed;
        }
    }
}

// ---------------------------------------------------------------------------------------
// The same index for SMALL batches: one workgroup per frame leaves the device idle when there
// are few frames (16.9 us for a single 4096x3072 frame), so a frame is cut into p.split
// pieces.  A piece sums its chunks (one wave per chunk, CT/64 bytes per lane, DPP reduction,
// no LDS atomics) and publishes the sums in place of the offsets; the LAST piece of a frame to
// finish (per-frame arrival counter, self-resetting) turns them into the exclusive prefix and
// does the validation and the header exactly as decode_index_kernel does.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_index_split_kernel(IdxParams p) {
    __shared__ uint32_t s_part[4];
    __shared__ uint32_t s_last;

    const uint32_t f = blockIdx.x / p.split, piece = blockIdx.x - f * p.split;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t T = p.T, cpf = p.chunks_per_frame;
    const uint64_t off = p.frame_offsets[f];
    const uint64_t meta_t = (1ull + p.min_bytes) * T;
    const uint64_t need = 32ull + meta_t;
    const bool in_range = in_extent(off, need, p.stream_bytes);
    const uint8_t *fb = p.stream + off;
    uint32_t *co = p.chunk_off + (size_t)f * (cpf + 1u);

    // ---- chunk sums of this piece ----
    const uint32_t per = (cpf + p.split - 1u) / p.split;
    const uint32_t k_begin = piece * per, k_end = k_begin + per < cpf ? k_begin + per : cpf;
    uint32_t bad = 0;
    for (uint32_t k = k_begin + (uint32_t)wave; k < k_end; k += 4u) {
        uint32_t sum = 0;
        if (in_range) {
            const uint8_t *darr = fb + 24;
            // the chunk's tiles [c_lo, c_hi): at most 512, eight depth bytes per lane
            const uint32_t c_lo = dec_chunk_begin(p.geom, k), c_hi = dec_chunk_begin(p.geom, k + 1u);
#pragma unroll
            for (uint32_t j = 0; j < 8u; j += 4u) {
                const uint32_t q = c_lo + (uint32_t)lane * 8u + j;
                if (q < c_hi) {   // the 4 bytes at q lie inside the frame (the min array follows the depths)
                    uint32_t x;
                    __builtin_memcpy(&x, darr + q, 4);
                    const uint32_t keep = c_hi - q;   // tiles of this chunk left
                    if (keep < 4u) x &= 0xFFFFFFFFu >> (8u * (4u - keep));
                    bad |= depth_bytes_bad(x, p.min_bytes);
                    sum += __builtin_amdgcn_sad_u8(x, 0u, 0u);
                }
            }
        }
        const uint32_t tot = wave_scan_incl(sum);
        if (lane == 63) __hip_atomic_store(&co[k], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (__any((int)(bad != 0u)) && lane == 0) atomicOr(&p.frame_flag[f], 1u);

    // ---- arrival: the last piece of the frame finishes the job ----
    // The sums were written through to memory (agent-scope atomic stores); once every wave has
    // seen its stores acknowledged, the arrival can be counted.  No fence: an agent-scope
    // release on this part writes back the whole L2, which costs more than the kernel.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(&p.frame_ctr[f], 1u) == p.split - 1u ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;

    const uint32_t seg = (cpf + 255u) / 256u;
    const uint32_t k0 = (uint32_t)tid * seg;
    uint32_t local = 0;
    for (uint32_t k = k0; k < k0 + seg && k < cpf; k++)
        local += __hip_atomic_load(&co[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t incl = wave_scan_incl(local);
    if (lane == 63) s_part[wave] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t v = s_part[k];
        base += k < wave ? v : 0u;
        total += v;
    }
    uint32_t run = base + incl - local;
    for (uint32_t k = k0; k < k0 + seg && k < cpf; k++) {
        const uint32_t v = __hip_atomic_load(&co[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        co[k] = run;
        run += v;
    }
    if (tid == 0) {
        co[cpf] = total;
        const uint32_t flags = __hip_atomic_load(&p.frame_flag[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        p.frame_flag[f] = 0;   // clean for the next launch
        p.frame_ctr[f] = 0;
        bool ok = in_range;
        uint32_t field = 0;
        uint64_t index = 0, elapsed = 0, consumed = 20;
        if (in_extent(off, 20, p.stream_bytes)) {
            field = load_u32_bytes(fb);
            index = load_u64_bytes(fb + 4);
            elapsed = f64_to_u64_x86(__longlong_as_double((long long)load_u64_bytes(fb + 12)));
        }
        if (ok) {
            const int32_t nb = (int32_t)load_u32_bytes(fb + 20);
            const int32_t nm = (int32_t)load_u32_bytes(fb + 24 + T);
            const int32_t n64 = (int32_t)load_u32_bytes(fb + 28 + meta_t);
            ok = nb == (int32_t)T && nm == (int32_t)(T * p.min_bytes) && n64 == (int32_t)total && !(flags & 1u);
            if (ok && !in_extent(off, need + 8ull * total, p.stream_bytes)) ok = false;
            if (ok) consumed = need + 8ull * total;
        }
        p.frame_ok[f] = ok ? 1u : 0u;
        if (p.results) {
            FrameResultDev *r = reinterpret_cast<FrameResultDev *>(p.results) + f;
            r->u64s = (field == 2u && ok) ? 2u : 0xFFFFFFFFu;   // dbde_util.cpp:335,342
            r->pad_ = 0;
            r->index = index;
            r->elapsed_ns = elapsed;
            r->consumed = consumed;
        }
    }
}

hipError_t launch_decode_index(const IdxParams &p, int n_frames, hipStream_t s) {
    const size_t lds = (size_t)p.chunks_per_frame * sizeof(uint32_t);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(decode_index_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (p.split > 1u) {
        hipLaunchKernelGGL(decode_index_split_kernel, dim3((uint32_t)n_frames * p.split), dim3(256), 0, s, p);
        return hipGetLastError();
    }
    // one 16-byte piece of the depth array per thread and pass: small frames get small workgroups (1024 threads for a
    // 256x256 frame -- 65 pieces -- made the index pass cost 70 % of the decode it serves)
    const uint32_t pieces = (p.T + 15u) / 16u + 1u;
    uint32_t threads = (pieces + 63u) / 64u * 64u;
    threads = threads < 64u ? 64u : (threads > 1024u ? 1024u : threads);
    hipLaunchKernelGGL(decode_index_kernel, dim3(n_frames), dim3(threads), lds, s, p);
    return hipGetLastError();
}

// Physical byte address of logical byte address A in the 16-byte-slot swizzled LDS image
// (slot ^= (slot >> 4) & 15, i.e. the permutation stays inside a 256-byte group).
__device__ __forceinline__ uint32_t swz_byte16(uint32_t A) { return A ^ ((A >> 4) & 0xF0u); }

// Row r of a tile is the 8*d-bit integer at byte (r*d) of the tile payload.  Rows start on
// byte boundaries, so the 8 bytes that hold a row come out of the two enclosing qwords with
// byte alignment ops (v_alignbyte) instead of 64-bit shifts; the two 4d-bit halves are split
// with v_alignbit and expanded with v_bfe.  The minimum is added byte-wise with wrap-around, as
// the reference's _mm_add_epi8 does (dbde_util.cpp:245-277).
#ifndef DBDE_DEC_SWZ_ALL
#define DBDE_DEC_SWZ_ALL 0   // A/B switch: 1 = the decoder's payload image swizzled for every chunk (round-2 start)
#endif
#ifndef DBDE_DEC_SWZ_REGULAR
#define DBDE_DEC_SWZ_REGULAR 1   // A/B switch: 0 = swizzle for all-depth-8 chunks only
#endif
#ifndef DBDE_UNPACK_PLAIN
// byte k of x <- low byte of (g >> sh): one SDWA shift writes the field where it belongs and leaves the other bytes
__device__ __forceinline__ void put_byte1(uint32_t &x, uint32_t sh, uint32_t g) {
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(x) : "v"(sh), "v"(g));
}
__device__ __forceinline__ void put_byte2(uint32_t &x, uint32_t sh, uint32_t g) {
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(x) : "v"(sh), "v"(g));
}
__device__ __forceinline__ void put_byte3(uint32_t &x, uint32_t sh, uint32_t g) {
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(x) : "v"(sh), "v"(g));
}
#endif

template <bool SWZ>
__device__ __forceinline__ void unpack_tile_from_lds(const uint8_t *s_img, uint32_t byte_base, uint32_t d,
                                                     uint32_t mn, uint32_t (&v)[16]) {
    const uint32_t mn4 = mn * 0x01010101u;
    const bool d8 = d >= 8u;
    uint32_t a = byte_base;
#ifdef DBDE_UNPACK_PLAIN
    const uint32_t m1 = ((1u << d) - 1u) * 0x00010001u;            // d-bit fields in 16-bit lanes
    const uint32_t m2 = (1u << (2u * d)) - 1u;                     // 2d <= 16
    const uint32_t m4 = d >= 8u ? 0xFFFFFFFFu : ((1u << (4u * d)) - 1u);
#else
    // four d-bit fields of a 4d-bit group -> the low d bits of four bytes: byte k takes the low byte of (group >> k*d)
    // (three SDWA shifts), what lies above the field goes with one mask; the minimum is added byte-wise with the
    // carries cut at bit 7 (add_bytes), the three masks folded into per-tile constants
    const uint32_t md = ((1u << d) - 1u) * 0x01010101u;            // d = 8: all ones
    const uint32_t md7 = md & 0x7F7F7F7Fu, mdh = md & 0x80808080u;
    const uint32_t mnlo = mn4 & 0x7F7F7F7Fu, mnhi = mn4 & 0x80808080u;
    const uint32_t d2 = 2u * d, d3 = 3u * d;
#endif
#pragma unroll
    for (int r = 0; r < 8; r++) {
        uint32_t t0, t1, t2;   // the three dwords that hold bytes [a, a+8)
        if (SWZ) {             // 16-byte slots are permuted: the two enclosing qwords, then the right three of four
            const uint32_t A0 = a & ~7u;
            const uint2 lo = *reinterpret_cast<const uint2 *>(s_img + swz_byte16(A0));
            const uint2 hi = *reinterpret_cast<const uint2 *>(s_img + swz_byte16(A0 + 8u));
            const bool up = (a & 4u) != 0u;
            t0 = up ? lo.y : lo.x; t1 = up ? hi.x : lo.y; t2 = up ? hi.y : hi.x;
        } else {               // linear image: straight from the dword address
            const uint32_t *q = reinterpret_cast<const uint32_t *>(s_img + (a & ~3u));
            t0 = q[0]; t1 = q[1]; t2 = q[2];
        }
        const uint32_t r_lo = __builtin_amdgcn_alignbyte(t1, t0, a);   // bytes [a, a+4)
        const uint32_t r_hi = __builtin_amdgcn_alignbyte(t2, t1, a);   // bytes [a+4, a+8)
#ifdef DBDE_UNPACK_PLAIN
        const uint32_t g_lo = r_lo & m4;
        const uint32_t g_hi = (d8 ? r_hi : __builtin_amdgcn_alignbit(r_hi, r_lo, 4u * d)) & m4;
        // 4d-bit field -> two 2d-bit fields in 16-bit lanes -> four d-bit fields in bytes
        const uint32_t f_lo = (g_lo & m2) | (__builtin_amdgcn_ubfe(g_lo, 2u * d, 2u * d) << 16);
        const uint32_t f_hi = (g_hi & m2) | (__builtin_amdgcn_ubfe(g_hi, 2u * d, 2u * d) << 16);
        const uint32_t x = (f_lo & m1) | (((f_lo >> d) & m1) << 8);
        const uint32_t y = (f_hi & m1) | (((f_hi >> d) & m1) << 8);
        v[2 * r] = add_bytes(x, mn4);
        v[2 * r + 1] = add_bytes(y, mn4);
#else
        const uint32_t g_hi = d8 ? r_hi : __builtin_amdgcn_alignbit(r_hi, r_lo, 4u * d);   // (the shift count is taken modulo 32)
        uint32_t x = r_lo, y = g_hi;
        put_byte1(x, d, r_lo); put_byte2(x, d2, r_lo); put_byte3(x, d3, r_lo);
        put_byte1(y, d, g_hi); put_byte2(y, d2, g_hi); put_byte3(y, d3, g_hi);
        v[2 * r] = (((x & md7) + mnlo) ^ (x & mdh)) ^ mnhi;
        v[2 * r + 1] = (((y & md7) + mnlo) ^ (y & mdh)) ^ mnhi;
#endif
        a += d;
    }
}

// Depth-8 tile whose payload starts 8-byte aligned in the image: a row is one qword.
__device__ __forceinline__ void unpack_tile_d8_from_lds(const uint8_t *s_img, uint32_t byte_base, uint32_t mn,
                                                        uint32_t (&v)[16]) {
    const uint32_t mn4 = mn * 0x01010101u;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint2 q = *reinterpret_cast<const uint2 *>(s_img + swz_byte16(byte_base + 8u * (uint32_t)r));
        v[2 * r] = add_bytes(q.x, mn4);
        v[2 * r + 1] = add_bytes(q.y, mn4);
    }
}

// Workgroup -> chunk.  Workgroups are dealt to the 8 XCDs round-robin (workgroup b runs on XCD
// b % 8), and each XCD has its own L2: with the identity mapping every XCD touches every
// eighth 32 KB piece of the stream and of the image.  Here the launch is cut into groups of
// 8 * kXcdRun chunks and, inside a group, XCD x takes kXcdRun CONSECUTIVE chunks (0.5 MB of
// image, as much stream): measured 5-6 % faster on the full-size decode, flat for runs of
// 8..256 chunks, slower below 8.  All XCDs still move through the buffers together.  A
// trailing partial group keeps the identity mapping.
constexpr uint32_t kXcdRun = 16;
__device__ __forceinline__ uint32_t xcd_local_chunk(uint32_t b, uint32_t n_chunks) {
    constexpr uint32_t kGroup = 8u * kXcdRun;
    const uint32_t grp = b / kGroup, rem = b % kGroup;
    if ((grp + 1u) * kGroup > n_chunks) return b;
    return grp * kGroup + (rem & 7u) * kXcdRun + (rem >> 3);
}

// LDS image of a chunk's payload: up to 15 bytes of alignment shift + 64 B per tile + one qword
// of over-read, rounded up to whole 256-byte swizzle groups.  The same memory later holds the chunk's
// decoded pixels on the staged (kImgLinear) path: 8 * rows * pitch <= 32 KiB (dec_geometry) + slack.
// THREADS = 256 (512 tile slots) everywhere, plus 192 (384 slots) for whole-tile-row chunks that fill the smaller
// workgroup better (dbde_capi.cpp: 1366 wide = 2 rows of 171 tiles: 67 % of 512, 89 % of 384); its image is three
// quarters the size, so six of them are resident per CU.
template <int THREADS = 256>
struct DecLds {
    static constexpr int kThreads = THREADS;
    static constexpr int kWaves = THREADS / 64;
    // (and, on the staged-image path, 8 image-row pieces of 4096 B each shifted by up to 127 B: 8 * 4224 + 128)
    static constexpr uint32_t kSlots = THREADS == 256 ? 2128u : 1600u;   // 34,048 B: four workgroups per CU; 25,600 B: six
    static constexpr int kPieces = (kSlots + kThreads - 1) / kThreads;   // 16-B pieces per thread
};

// The LDS of gfx950 takes 8- and 16-byte accesses at ANY byte address at the price of aligned ones
// (profiles/lds_unaligned_probe.hip: every offset bit-exact, same time).  The compiler does not know that and
// would split an under-aligned access into bytes, so the instruction is spelled out.
__device__ __forceinline__ void lds_store_u64_any(uint8_t *s_base, uint32_t byte_addr, uint32_t lo, uint32_t hi) {
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)s_base + byte_addr;
    asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory");
}
// Exactly n (1..7) bytes of v at any LDS byte address (the last, partial tile of an image row).
__device__ __forceinline__ void lds_store_bytes(uint8_t *s_px, uint32_t a, uint64_t v, uint32_t n) {
    if (n & 4u) { __builtin_memcpy(s_px + a, &v, 4); a += 4u; v >>= 32; }       // (byte stores: the compiler splits them)
    if (n & 2u) { s_px[a] = (uint8_t)v; s_px[a + 1u] = (uint8_t)(v >> 8); a += 2u; v >>= 16; }
    if (n & 1u) s_px[a] = (uint8_t)v;
}

// Write one (possibly partial) tile: only the valid region (dbde_util.cpp:281-289).
__device__ __forceinline__ void store_tile_generic(uint8_t *img, int W, int H, uint32_t w, uint32_t t,
                                                   const uint32_t (&v)[16]) {
    uint32_t ty = t / w, tx = t - ty * w;
    int x0 = 8 * (int)tx;
    int rm = W - x0 < 8 ? W - x0 : 8;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int yy = 8 * (int)ty + r;
        if (yy < H) {
            uint8_t *row = img + (size_t)yy * (size_t)W + x0;
            const uint64_t q = ((uint64_t)v[2 * r + 1] << 32) | v[2 * r];
            if (rm == 8) {
                store_u64_any(row, q);
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (k < rm) row[k] = (uint8_t)(q >> (8 * k));
            }
        }
    }
}

// How decoded pixels reach the image.
//   kImgDirect : W % 16 == 0 and a 16-byte aligned image: every lane's two tiles are one aligned 16-byte
//                store per image row (whole cache lines per wave when W % 128 == 0).
//   kImgLinear : any geometry.  Partial cache-line writes are what makes odd widths slow (unaligned 16-byte
//                stores run at 3.2 TB/s against 5.6 aligned, profiles/mempattern.hip; per-lane alignment is
//                not the issue), so the chunk's pixels are staged in LDS as they lie in the frame -- the
//                chunk's contiguous byte range (whole tile rows: image rows follow each other at pitch W) or
//                one range per image row (a piece of a frame wider than 4096) -- placed so that LDS and
//                global addresses agree mod 16.  Tile rows go in with unaligned 8-byte LDS stores; the range
//                leaves as aligned 16-byte blocks (ds_read_b128 -> global store, no arithmetic), with byte
//                stores only in the first and last block of a range.  Only the valid region is ever written
//                (dbde_util.cpp:281-289): the last tile of an image row stores its W % 8 valid bytes, rows
//                >= H lie behind the copied range.  Used where the tile rows land 8-byte aligned in LDS
//                (W % 8 == 0): 8-byte LDS stores at odd addresses took 5x the time of aligned ones here.
//   kImgTiles  : everything else (odd widths): each tile row is one 8-byte store at its own address.
constexpr int kImgDirect = 0, kImgLinear = 1, kImgTiles = 2;

// SELF_INDEX: launches of a few frames (one frame per call is BASELINE configs[1] taken literally) do without the
// index kernel and its launch boundary: EVERY workgroup reads the frame's whole depth array (T bytes, from L2
// after the first touch), which gives it the reference's validation verdict (dbde_util.cpp:295-303) and the word
// offset of its own chunk.  T bytes per workgroup only pays while frames * chunks * T stays small (dbde_capi.cpp).
// FUSED (kIdxFused): launches of few LARGE frames (one 4096x3072 frame per call: 384 chunks; the whole-array read of
// SELF_INDEX took 38 us there, index kernel + boundary + decode 12.4 us).  The launch's workgroups build the index
// among themselves: each sums ITS chunk's depth bytes (it needs them anyway) and publishes the sum as an 8-byte record
// tagged with the launch's epoch (relaxed agent-scope store to uncached memory; records of older launches carry older
// epochs, so nothing is cleared between launches); wave 0 then reads the records of the frame's chunks -- its prefix,
// the frame's word count for the n64 check (dbde_util.cpp:302-303) and the depth <= 8 verdict are sums over them.
// Co-residency is the normal case (the host takes this form only for launches that fit the device's workgroup slots)
// but nothing depends on it: a record that has not appeared after 30 us is computed by the waiting wave itself from
// the stream (the chunk's 512 depth bytes) -- every spin ends, whatever the dispatch order.
constexpr int kIdxTable = 0, kIdxSelf = 1, kIdxFused = 2;

template <int IMG, int INDEX, int THREADS = 256>
__global__ __launch_bounds__(THREADS) void decode_kernel(DecParams p) {
    constexpr bool SELF_INDEX = INDEX == kIdxSelf;
    typedef DecLds<THREADS> G;
    __shared__ __attribute__((aligned(16))) uint64_t s_in[G::kSlots * 2];
    __shared__ uint32_t s_wave_tot[G::kWaves];
    __shared__ uint32_t s_idx[G::kWaves][4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    SF_DECL;
    SF_MARK(0);
    const uint32_t c = SELF_INDEX ? blockIdx.x : xcd_local_chunk(blockIdx.x, p.n_chunks);
    const uint32_t f = c / p.chunks_per_frame;
    const uint32_t cf = c - f * p.chunks_per_frame;
#ifdef DBDE_DEC_FAKE_INDEX   // (probe: what the index round trip costs -- one slot per frame at abbench's stride, every tile of depth 8)
    const uint64_t foff = (uint64_t)f * (((32ull + 66ull * p.T) + 255ull) / 256ull * 256ull);
#else
    const uint64_t foff = p.frame_offsets[f];
#endif
    const uint8_t *fb = p.stream + foff;
    const uint32_t t_begin = dec_chunk_begin(p.geom, cf);
    const uint32_t n_tiles = dec_chunk_begin(p.geom, cf + 1u) - t_begin;   // <= 512
    uint32_t w_begin, w_end;
    if (SELF_INDEX) {
        const uint32_t T = p.T;
        const uint64_t need = 32ull + 2ull * T;
        const bool in_range = in_extent(foff, need, p.stream_bytes);
        uint32_t s_before = 0, s_mine = 0, s_all = 0, bad = 0;
        if (in_range) {
            // 16-byte aligned pieces of the depth array (bytes outside [0, T) masked off), four in flight per thread
            const uint8_t *darr = fb + 24;
            const uint32_t head = (uint32_t)(reinterpret_cast<uintptr_t>(darr) & 15u);
            const uint8_t *a_lo = darr - head;
            const uint32_t npieces = (head + T + 15u) >> 4;
            const uint8_t *s_end = p.stream + p.stream_bytes;
            auto account = [&](uint32_t i, const uint4 &qv) {
                const uint32_t wv[4] = {qv.x, qv.y, qv.z, qv.w};
                const int pos0 = (int)(16u * i) - (int)head;   // tile position of byte 0 of the piece
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int pj = pos0 + 4 * j;
                    auto below = [&](int limit) -> uint32_t {   // mask of the dword's bytes at tile positions < limit
                        const int n = limit - pj;
                        return n >= 4 ? 0xFFFFFFFFu : (n <= 0 ? 0u : (0xFFFFFFFFu >> (8 * (4 - n))));
                    };
                    const uint32_t in = below((int)T) & ~below(0);
                    const uint32_t x = wv[j] & in;
                    bad |= (x & 0xF0F0F0F0u) | ((x + 0x77777777u) & 0x80808080u);   // any byte > 8
                    const uint32_t m0 = below((int)t_begin), m1 = below((int)(t_begin + n_tiles));
                    s_all += __builtin_amdgcn_sad_u8(x, 0u, 0u);
                    s_before += __builtin_amdgcn_sad_u8(x & m0, 0u, 0u);
                    s_mine += __builtin_amdgcn_sad_u8(x & m1 & ~m0, 0u, 0u);
                }
            };
            auto fetch = [&](uint32_t i) -> uint4 {
                if (i >= npieces) return make_uint4(0, 0, 0, 0);
                if (a_lo + 16ull * (i + 1u) <= s_end) return *reinterpret_cast<const uint4 *>(a_lo + 16ull * i);
                uint32_t wq[4] = {0, 0, 0, 0};   // the piece would cross the end of the readable extent (tiny frames)
                for (uint32_t b = 0; b < 16u; b++) {
                    const uint8_t *src = a_lo + 16ull * i + b;
                    if (src < s_end) wq[b >> 2] |= (uint32_t)*src << (8u * (b & 3u));
                }
                return make_uint4(wq[0], wq[1], wq[2], wq[3]);
            };
            for (uint32_t i = (uint32_t)tid; i < npieces; i += 4u * (uint32_t)G::kThreads) {
                const uint4 q0 = fetch(i), q1 = fetch(i + G::kThreads), q2 = fetch(i + 2u * G::kThreads), q3 = fetch(i + 3u * G::kThreads);
                account(i, q0);
                if (i + G::kThreads < npieces) account(i + G::kThreads, q1);
                if (i + 2u * G::kThreads < npieces) account(i + 2u * G::kThreads, q2);
                if (i + 3u * G::kThreads < npieces) account(i + 3u * G::kThreads, q3);
            }
        }
        s_before = wave_sum(s_before); s_mine = wave_sum(s_mine); s_all = wave_sum(s_all);
        const bool any_bad = __any((int)(bad != 0u));
        if (lane == 0) { s_idx[wave][0] = s_before; s_idx[wave][1] = s_mine; s_idx[wave][2] = s_all; s_idx[wave][3] = any_bad ? 1u : 0u; }
        __syncthreads();
        uint32_t before = 0, mine = 0, total = 0, flag = 0;
#pragma unroll
        for (int k = 0; k < G::kWaves; k++) { before += s_idx[k][0]; mine += s_idx[k][1]; total += s_idx[k][2]; flag |= s_idx[k][3]; }
        bool okf = in_range;
        if (okf) {
            const int32_t nb = (int32_t)load_u32_bytes(fb + 20);
            const int32_t nm = (int32_t)load_u32_bytes(fb + 24 + T);
            const int32_t n64 = (int32_t)load_u32_bytes(fb + 28 + 2ull * T);
            okf = nb == (int32_t)T && nm == (int32_t)T && n64 == (int32_t)total && !flag;
            if (okf && !in_extent(foff, need + 8ull * total, p.stream_bytes)) okf = false;   // the payload must lie inside the stream too
        }
        if (cf == 0u && tid == 0 && p.results) {   // the frame's result record: dbde_unpack_frame's return value
            uint32_t field = 0;
            uint64_t index = 0, elapsed = 0;
            if (in_extent(foff, 20, p.stream_bytes)) {
                field = load_u32_bytes(fb);
                index = load_u64_bytes(fb + 4);
                elapsed = f64_to_u64_x86(__longlong_as_double((long long)load_u64_bytes(fb + 12)));
            }
            FrameResultDev *r = reinterpret_cast<FrameResultDev *>(p.results) + f;
            r->u64s = (field == 2u && okf) ? 2u : 0xFFFFFFFFu;   // dbde_util.cpp:335,342
            r->pad_ = 0;
            r->index = index;
            r->elapsed_ns = elapsed;
            r->consumed = okf ? need + 8ull * total : 20ull;
        }
        if (!okf) return;   // rejected frame: image untouched (dbde_util.cpp:296-303)
        w_begin = before;
        w_end = before + mine;
    } else if (INDEX == kIdxFused) {
        const uint32_t T = p.T, cpf = p.chunks_per_frame;
        const uint64_t need = 32ull + 2ull * T;
        const bool in_range = in_extent(foff, need, p.stream_bytes);
        uint32_t field = 0;
        uint64_t index = 0, elapsed = 0;
        const bool record = cf == 0u && tid == 0 && p.results;
        if (record && in_extent(foff, 20, p.stream_bytes)) {
            field = load_u32_bytes(fb);
            index = load_u64_bytes(fb + 4);
            elapsed = f64_to_u64_x86(__longlong_as_double((long long)load_u64_bytes(fb + 12)));
        }
        FrameResultDev *res = reinterpret_cast<FrameResultDev *>(p.results) + f;
        if (!in_range) {   // every workgroup of the frame sees this by itself: nobody publishes, nobody waits
            if (record) { res->u64s = 0xFFFFFFFFu; res->pad_ = 0; res->index = index; res->elapsed_ns = elapsed; res->consumed = 20ull; }
            return;
        }
        // (the three I32 fields the verdict needs are requested here, with the depth bytes: asked for behind the record
        // exchange they were one more memory round trip, 0.9 us of a 10 us launch)
        const int32_t nb = (int32_t)load_u32_bytes(fb + 20);
        const int32_t nm = (int32_t)load_u32_bytes(fb + 24 + T);
        const int32_t n64 = (int32_t)load_u32_bytes(fb + 28 + 2ull * T);
        // 1. this chunk's depth sum and "a depth above 8" verdict
        const uint8_t *darr = fb + 24;
        const uint32_t te = t_begin + 2u * (uint32_t)tid;
        uint32_t dsum = 0, dbad = 0;
        if (2u * (uint32_t)tid < n_tiles) { const uint32_t d = darr[te]; dsum += d; dbad |= d > 8u ? 1u : 0u; }
        if (2u * (uint32_t)tid + 1u < n_tiles) { const uint32_t d = darr[te + 1u]; dsum += d; dbad |= d > 8u ? 1u : 0u; }
        dsum = wave_sum(dsum);
        const bool wbad = __any((int)dbad);
        if (lane == 0) { s_idx[wave][0] = dsum; s_idx[wave][1] = wbad ? 1u : 0u; }
        __syncthreads();
        SF_MARK(1);
        unsigned long long *rec = p.fuse_rec + (size_t)f * cpf;
        const unsigned long long tag = (unsigned long long)p.fuse_epoch << 32;
        if (tid == 0 && !((p.fuse_flags & 1u) && (cf & 1u))) {   // (fuse_flags bit 0, tests: odd chunks keep silent, as if not running)
            uint32_t mine = 0, mbad = 0;
#pragma unroll
            for (int k = 0; k < G::kWaves; k++) { mine += s_idx[k][0]; mbad |= s_idx[k][1]; }
            __hip_atomic_store(&rec[cf], tag | ((unsigned long long)mbad << 31) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // 2. the records of the frame's chunks, in groups of 64 dealt to the workgroup's waves (group g to wave g mod 4), two
        //    groups per trip with both loads in flight: a 384-chunk frame is ONE memory round trip per wave where a single
        //    wave walking the six groups one after the other paid six
        {
            const uint64_t t_start = wall_clock64();
            uint32_t before = 0, total = 0, flag = 0;
            auto take = [&](const uint32_t base, unsigned long long w) __attribute__((always_inline)) {
                const uint32_t k = base + (uint32_t)lane;
                bool got = k >= cpf || (w >> 32) == (tag >> 32);
                for (;;) {
                    if (__all((int)got)) break;
                    if (wall_clock64() - t_start > 3000ull) {   // 30 us: whoever has not published may not be running yet
                        uint64_t miss = __ballot((int)!got);
                        while (miss) {                           // its chunk's depth bytes, eight per lane, summed here
                            const uint32_t kk = (uint32_t)__builtin_ctzll(miss);
                            miss &= miss - 1ull;
                            const uint32_t c_lo = dec_chunk_begin(p.geom, base + kk), c_hi = dec_chunk_begin(p.geom, base + kk + 1u);
                            uint32_t sm = 0, bd = 0;
#pragma unroll
                            for (uint32_t j = 0; j < 8u; j += 4u) {
                                const uint32_t q = c_lo + (uint32_t)lane * 8u + j;
                                if (q < c_hi) {
                                    uint32_t x;
                                    __builtin_memcpy(&x, darr + q, 4);
                                    const uint32_t keep = c_hi - q;
                                    if (keep < 4u) x &= 0xFFFFFFFFu >> (8u * (4u - keep));
                                    bd |= depth_bytes_bad(x, 1u);
                                    sm += __builtin_amdgcn_sad_u8(x, 0u, 0u);
                                }
                            }
                            sm = wave_sum(sm);
                            const bool anyb = __any((int)(bd != 0u));
                            if ((uint32_t)lane == kk) { w = tag | ((unsigned long long)(anyb ? 1u : 0u) << 31) | sm; got = true; }
                        }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    if (!got) { w = __hip_atomic_load(&rec[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); got = (w >> 32) == (tag >> 32); }
                }
                const uint32_t v = k < cpf ? (uint32_t)w & 0x7FFFFFFFu : 0u;
                total += v;
                before += k < cf ? v : 0u;
                flag |= k < cpf ? (uint32_t)(w >> 31) & 1u : 0u;
            };
            constexpr uint32_t kStride = 64u * (uint32_t)G::kWaves;
            for (uint32_t base = 64u * (uint32_t)wave; base < cpf; base += 2u * kStride) {
                const uint32_t kA = base + (uint32_t)lane, kB = kA + kStride;
                const unsigned long long wA = kA < cpf ? __hip_atomic_load(&rec[kA], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                const unsigned long long wB = kB < cpf ? __hip_atomic_load(&rec[kB], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                take(base, wA);
                if (base + kStride < cpf) take(base + kStride, wB);
            }
            before = wave_sum(before); total = wave_sum(total);
            const bool anyflag = __any((int)flag);
            if (lane == 0) { s_idx[wave][2] = before; s_idx[wave][3] = total | (anyflag ? 0x80000000u : 0u); }   // (a frame holds < 2^31 words)
        }
        SF_MARK(2);
        __syncthreads();
        uint32_t before = 0, total = 0, flag = 0;
#pragma unroll
        for (int k = 0; k < G::kWaves; k++) { before += s_idx[k][2]; total += s_idx[k][3] & 0x7FFFFFFFu; flag |= s_idx[k][3] >> 31; }
        uint32_t mine = 0;
#pragma unroll
        for (int k = 0; k < G::kWaves; k++) mine += s_idx[k][0];
        const bool okf = nb == (int32_t)T && nm == (int32_t)T && n64 == (int32_t)total && !flag &&
                         in_extent(foff, need + 8ull * total, p.stream_bytes);
        if (record) {
            res->u64s = (field == 2u && okf) ? 2u : 0xFFFFFFFFu;   // dbde_util.cpp:335,342
            res->pad_ = 0; res->index = index; res->elapsed_ns = elapsed;
            res->consumed = okf ? need + 8ull * total : 20ull;
        }
        if (!okf) return;   // rejected frame: image untouched (dbde_util.cpp:296-303)
        w_begin = before;
        w_end = before + mine;
        __syncthreads();   // s_idx and s_wave_tot are about to be reused
        SF_MARK(3);
    } else {
        // everything the address arithmetic needs, requested together
#ifdef DBDE_DEC_FAKE_INDEX
        const uint32_t ok = 1u;
        w_begin = 8u * t_begin; w_end = 8u * (t_begin + n_tiles);
#else
        const uint32_t ok = p.frame_ok[f];
        const uint32_t *co = p.chunk_off + (size_t)f * (p.chunks_per_frame + 1u) + cf;
        w_begin = co[0]; w_end = co[1];
#endif
#ifndef DBDE_IDX_SERIAL
        // ... and really together: without this the compiler sinks the loads behind the branch (three dependent
        // scalar round trips before the first payload byte is requested)
        asm volatile("" :: "s"(ok), "s"(w_begin), "s"(w_end), "s"((uint32_t)foff));
#endif
        if (!ok) return;   // rejected frame: image untouched (dbde_util.cpp:296-303)
    }

    const uint32_t t0 = t_begin + 2u * (uint32_t)tid;
    const bool hasA = 2u * (uint32_t)tid < n_tiles, hasB = 2u * (uint32_t)tid + 1u < n_tiles;
    const uint8_t *depth_arr = fb + 24;
    const uint8_t *min_arr = fb + 28 + p.T;

    // ---- issue the payload loads (16 B per lane, source aligned down) BEFORE the depth bytes
    //      are needed: the chunk's extent comes from the index, not from a scan of the depths ----
    const uint32_t chunk_words = w_end - w_begin;
    const uint8_t *src = fb + 32ull + 2ull * p.T + 8ull * w_begin;
    const uint32_t shift = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 15u);
    const uint8_t *asrc = src - shift;
    const uint32_t n16 = (shift + 8u * chunk_words + 15u) >> 4;
    // LDS-DMA (global_load_lds_dwordx4): physical slot i of the image <- logical slot swz16(i)
    // of the stream; the destination is wave-uniform base + lane*16, the permutation stays
    // inside one 256-byte group so the source side remains coalesced.  No staging registers.
    // The image is swizzled only for chunks whose lanes would read it at a regular stride that piles onto few banks:
    // every tile of depth 8 (128 bytes per lane), 4 (64: -11 % unswizzled, measured) or 7 -- the chunk's word count
    // says so before a single depth byte has arrived (a mixed chunk that happens to average 4 or 7 is swizzled for
    // nothing, that is all).  Any other chunk is read at offsets as irregular as its depths, or at strides that
    // measured no slower, and the swizzle would only be address arithmetic in the unpack (6 VALU per tile row:
    // mixed 0.75 -> 0.78, uniform depth 2 / 3 0.72 / 0.75 -> 0.77 / 0.78).
    const bool swz = DBDE_DEC_SWZ_ALL || chunk_words == 8u * n_tiles || (DBDE_DEC_SWZ_REGULAR && (chunk_words == 4u * n_tiles || chunk_words == 7u * n_tiles));
    const uint32_t n16r = (n16 + 15u) & ~15u;
    // Never read past the extent the caller declared (dbde_hip.h: stream_bytes is the READABLE extent): the
    // whole-slot DMA stops before a slot that straddles the end; that slot (the last one of the last chunk
    // of the last frame, at most) is fetched byte by byte below.
    const uint8_t *s_end = p.stream + p.stream_bytes;
    const uint32_t n16_dma = (asrc + 16ull * n16 <= s_end) ? n16 : n16 - 1u;
#pragma unroll
    for (int j = 0; j < G::kPieces; j++) {
        const uint32_t i = (uint32_t)tid + (uint32_t)j * G::kThreads;
        const uint32_t src_slot = swz ? swz16(i) : i;
        if (i < n16r && src_slot < n16_dma) {
            const uint32_t wave_slot0 = (uint32_t)j * G::kThreads + (uint32_t)wave * 64u;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(asrc + 16ull * src_slot),
                (__attribute__((address_space(3))) void *)(&s_in[2u * wave_slot0]), 16, 0, DBDE_NT ? 2 : 0);
        }
    }
    if (n16_dma != n16 && tid < 16) {   // the straddling slot, byte by byte: logical slot L lives at physical swz16(L)
        const uint8_t *b = asrc + 16ull * n16_dma + (uint32_t)tid;
        reinterpret_cast<uint8_t *>(s_in)[16u * (swz ? swz16(n16_dma) : n16_dma) + (uint32_t)tid] = b < s_end ? *b : (uint8_t)0;
    }
    SF_MARK(4);
    uint32_t dA = 0, dB = 0, mA = 0, mB = 0;
    // the lane's two tiles are one u16 load per array where that address is even
    if (hasB && ((reinterpret_cast<uintptr_t>(depth_arr + t0) | reinterpret_cast<uintptr_t>(min_arr + t0)) & 1u) == 0u) {
        const uint32_t d2 = *reinterpret_cast<const uint16_t *>(depth_arr + t0);
        const uint32_t m2 = *reinterpret_cast<const uint16_t *>(min_arr + t0);
        dA = d2 & 0xFFu; dB = d2 >> 8; mA = m2 & 0xFFu; mB = m2 >> 8;
    } else {
        if (hasA) { dA = depth_arr[t0]; mA = min_arr[t0]; }
        if (hasB) { dB = depth_arr[t0 + 1]; mB = min_arr[t0 + 1]; }
    }

    // The payload reaches LDS by DMA (and the tail bytes by ds_write); other waves read those slots after the
    // barrier inside block_scan_incl.  A barrier does not drain vector memory: every wave must have seen its
    // own DMA land (vmcnt) and its LDS writes retire (lgkmcnt) BEFORE it arrives at the barrier.  Stated
    // explicitly instead of relying on where the compiler happens to put its waits.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    SF_MARK(5);
    uint32_t chunk_total;   // equals chunk_words for a validated frame
    const uint32_t incl = block_scan_incl<G::kWaves>(dA + dB, s_wave_tot, lane, wave, chunk_total);   // barrier inside
    SF_MARK(6);
    const uint32_t offA = incl - (dA + dB), offB = offA + dA;

    uint32_t va[16], vb[16];
    const uint8_t *s_img = reinterpret_cast<const uint8_t *>(s_in);
    const uint32_t bA = shift + 8u * offA, bB = shift + 8u * offB;
    // wave-uniform specialisations: all tiles flat (depth 0), or all of depth 8 (rows are whole qwords)
    // all of depth 8, rows whole qwords (lanes without a tile read 64 bytes of slack behind the image and store nothing)
    const bool all8 = swz && (shift & 7u) == 0u && __all((int)((dA == 8u || !hasA) && (dB == 8u || !hasB)));
    const bool flat = __all((int)((dA | dB) == 0u));
    if (flat) {   // flat tiles only: every pixel is its tile's minimum, no payload
#pragma unroll
        for (int i = 0; i < 16; i++) { va[i] = mA * 0x01010101u; vb[i] = mB * 0x01010101u; }
    } else if (all8) {
        unpack_tile_d8_from_lds(s_img, bA, mA, va);
        unpack_tile_d8_from_lds(s_img, bB, mB, vb);
    } else {
        if (swz) {
            unpack_tile_from_lds<true>(s_img, bA, dA, mA, va);
            unpack_tile_from_lds<true>(s_img, bB, dB, mB, vb);
        } else {
            unpack_tile_from_lds<false>(s_img, bA, dA, mA, va);
            unpack_tile_from_lds<false>(s_img, bB, dB, mB, vb);
        }
    }

    SF_MARK(7);
    uint8_t *img = p.images + (size_t)f * p.frame_pixels;
    if (IMG == kImgDirect) {
        if (hasA) {   // W % 16 == 0: w is even and t_begin a multiple of w (or of 512), so both tiles share a tile row
            const uint32_t ty = t0 / p.w, tx = t0 - ty * p.w;
            uint8_t *base = img + (size_t)(8u * tx);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int yy = 8 * (int)ty + r;
                if (yy < p.H) {
                    u32x4_t o;
                    o[0] = va[2 * r]; o[1] = va[2 * r + 1]; o[2] = vb[2 * r]; o[3] = vb[2 * r + 1];
                    // (single-frame launches, whose image nobody reads back: plain stores measured +1.4 us per round trip, the sc0 / sc1 / nt
                    // combinations within 0.5 us of each other -- the 4 us in which these stores retire are the memory side's)
                    if (DBDE_NT) __builtin_nontemporal_store(o, reinterpret_cast<u32x4_t *>(base + (size_t)yy * (size_t)p.W));
                    else *reinterpret_cast<u32x4_t *>(base + (size_t)yy * (size_t)p.W) = o;
                }
            }
        }
        SF_MARK(8);
        SF_DRAIN();
        SF_MARK(9);
        if (INDEX == kIdxFused) SF_FLUSH(p.diag, 9);
        return;
    }

    if (IMG == kImgTiles) {   // any geometry, tile by tile: 8-byte stores at the tile's own (arbitrary) address
        if (hasA) store_tile_generic(img, p.W, p.H, p.w, t0, va);
        if (hasB) store_tile_generic(img, p.W, p.H, p.w, t0 + 1u, vb);
        return;
    }

    // ---- kImgLinear: stage the pixels in LDS, leave as whole cache lines of the chunk's byte range ----------
    const bool whole_rows = p.geom.pieces == 1u;
    const uint32_t ty0 = whole_rows ? t_begin / p.w : cf / p.geom.pieces;
    const uint32_t tx0 = whole_rows ? 0u : (cf - ty0 * p.geom.pieces) * kChunkTiles;
    const uint32_t wspan = whole_rows ? p.w : n_tiles;                     // tiles across in the staged image
    const uint32_t Wu = (uint32_t)p.W, y0 = 8u * ty0;
    const uint32_t tile_rows = whole_rows ? n_tiles / p.w : 1u;
    const uint32_t y_end = y0 + 8u * tile_rows < (uint32_t)p.H ? y0 + 8u * tile_rows : (uint32_t)p.H;
    const uint32_t x_valid = Wu - 8u * tx0 < 8u * wspan ? Wu - 8u * tx0 : 8u * wspan;   // valid bytes of one image row here
    uint8_t *g_first = img + (size_t)y0 * (size_t)p.W + (size_t)(8u * tx0);
    // mod 128, not 16: a wave's 64 x 16 B must cover whole cache lines (a 4-lane group one 64-byte sector),
    // or the stores run at little more than half rate however well each lane is aligned (profiles/mempattern.hip)
    const uint32_t g7 = (uint32_t)(reinterpret_cast<uintptr_t>(g_first) & 127u);
    uint8_t *s_px = reinterpret_cast<uint8_t *>(s_in);
    // Two layouts of the staged image:
    //   a8 (image rows 8-byte aligned: W % 8 == 0 and an 8-byte aligned frame): the chunk's byte range AS IT LIES in the
    //      frame (whole rows: one range, rows at pitch W; pieces of a wider row: a range per image row, LDS pitch 4224),
    //      LDS and global addresses equal mod 128: tile rows go in as aligned 8-byte stores, blocks come out as they are;
    //   otherwise (odd widths, whole rows): tiles cannot be stored where the frame has them -- 8-byte LDS stores at odd
    //      addresses took five times the aligned time in this kernel, and shifting them in registers is issue work that
    //      only paid for incompressible content -- so the image is staged TILE-ALIGNED, image rows at pitch 8 w + 16
    //      (every tile row an aligned ds_write_b64), and the re-alignment moves to the read side of the copy-out, where
    //      it is free: block j of the range's aligned cover is the 16 bytes at (row, column) = divmod(offset, W), one
    //      ds_read_b128 at whatever address that is (the LDS takes any).  A block that runs over the end of an image
    //      row continues in the 16 bytes behind it, which the first two tiles of the NEXT image row have also written
    //      there (2 of w lanes, 8 unaligned stores each).
    const bool a8 = ((g7 | Wu) & 7u) == 0u;
    if (!a8 && !whole_rows) {   // (pieces of an odd-width row wider than 4096: not staged; dbde_capi.cpp does not send them here)
        if (hasA) store_tile_generic(img, p.W, p.H, p.w, t0, va);
        if (hasB) store_tile_generic(img, p.W, p.H, p.w, t0 + 1u, vb);
        return;
    }
    // 16-byte aligned image rows (W % 16 == 0, 16-byte aligned frames): a chunk that is NOT all of depth 8 stores straight
    // from the registers, one aligned 16-byte store per lane and image row.  Its time goes into the unpack (LDS, VALU), and
    // the staging's LDS traffic and barriers cost it more than the partial cache lines at the ends of a wave's rows do
    // (mixed content at 88 % chunk fill: +4..9 %, 1680 wide +9 %); an all-depth-8 chunk is memory-bound and is staged
    // (stored directly it lost 9..13 %).  The chunk's word count tells the two apart for the whole workgroup.
    // (Rows that are not 16-byte aligned gain nothing from the same idea: UNALIGNED 16-byte stores straight from the
    // registers measured 4..13 % below the staged copy-out on 1921, 1928, 1080, 1001 wide mixed frames.)
    if (whole_rows && ((uint32_t)(reinterpret_cast<uintptr_t>(img) & 15u) | (Wu & 15u)) == 0u && chunk_words != 8u * n_tiles) {
        if (hasA) {   // w is even: the lane's two tiles are neighbours in one tile row
            const uint32_t iD = 2u * (uint32_t)tid, rowD = iD / wspan, colD = iD - rowD * wspan;
            uint8_t *base = img + (size_t)(8u * colD);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t yy = y0 + 8u * rowD + (uint32_t)r;
                if (yy < (uint32_t)p.H) {
                    u32x4_t o;
                    o[0] = va[2 * r]; o[1] = va[2 * r + 1]; o[2] = vb[2 * r]; o[3] = vb[2 * r + 1];
                    if (DBDE_NT) __builtin_nontemporal_store(o, reinterpret_cast<u32x4_t *>(base + (size_t)yy * (size_t)p.W));
                    else *reinterpret_cast<u32x4_t *>(base + (size_t)yy * (size_t)p.W) = o;
                }
            }
        }
        return;
    }
    __syncthreads();   // every wave has finished reading the payload image: the memory changes hands
    const uint32_t iA = 2u * (uint32_t)tid;
    const uint32_t rowA = iA / wspan, colA = iA - rowA * wspan;
    const uint32_t rowB = colA + 1u == wspan ? rowA + 1u : rowA, colB = colA + 1u == wspan ? 0u : colA + 1u;
    // bytes of the tile's rows that exist in the image (8, or W % 8 for the last tile of an image row)
    const uint32_t nA = 8u * colA + 8u <= x_valid ? 8u : x_valid - 8u * colA;
    const uint32_t nB = 8u * colB + 8u <= x_valid ? 8u : x_valid - 8u * colB;
    const bool all_rows = y0 + 8u * tile_rows <= (uint32_t)p.H;
    const bool plain = all_rows && __all((int)((!hasA || nA == 8u) && (!hasB || nB == 8u)));   // no edge in this wave
    if (a8) {
        const uint32_t l_pitch = 8u * kChunkTiles + 128u;
        // LDS byte address of image row ry (relative to y0), column byte xb:
        //   whole rows: g7 + ry * W + xb                 (one range: LDS and global agree mod 128 throughout)
        //   pieces    : ((g7 + ry * W) & 127) + ry * l_pitch + xb   (each image row is its own range)
        const uint32_t aA = whole_rows ? g7 + 8u * rowA * Wu + 8u * colA : 8u * colA;   // the tile's row 0, less the row term
        const uint32_t aB = whole_rows ? g7 + 8u * rowB * Wu + 8u * colB : 8u * colB;
        auto row_term = [&](uint32_t r) -> uint32_t {   // wave-uniform
            return whole_rows ? r * Wu : ((g7 + r * Wu) & 127u) + r * l_pitch;
        };
        if (plain) {
            if (hasA) {
#pragma unroll
                for (int r = 0; r < 8; r++) lds_store_u64_any(s_px, aA + row_term((uint32_t)r), va[2 * r], va[2 * r + 1]);
            }
            if (hasB) {
#pragma unroll
                for (int r = 0; r < 8; r++) lds_store_u64_any(s_px, aB + row_term((uint32_t)r), vb[2 * r], vb[2 * r + 1]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t rt = row_term((uint32_t)r);
                if (hasA && y0 + 8u * rowA + (uint32_t)r < y_end) {
                    if (nA == 8u) lds_store_u64_any(s_px, aA + rt, va[2 * r], va[2 * r + 1]);
                    else lds_store_bytes(s_px, aA + rt, ((uint64_t)va[2 * r + 1] << 32) | va[2 * r], nA);
                }
                if (hasB && y0 + 8u * rowB + (uint32_t)r < y_end) {
                    if (nB == 8u) lds_store_u64_any(s_px, aB + rt, vb[2 * r], vb[2 * r + 1]);
                    else lds_store_bytes(s_px, aB + rt, ((uint64_t)vb[2 * r + 1] << 32) | vb[2 * r], nB);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the asm stores are invisible to the compiler's counters
        __syncthreads();
        const uint32_t n_ranges = whole_rows ? 1u : y_end - y0;
        for (uint32_t s = 0; s < n_ranges; s++) {
            uint8_t *g0 = g_first + (size_t)s * Wu;
            const uint32_t bytes = whole_rows ? (y_end - y0) * Wu : x_valid;
            const uint32_t head = (uint32_t)(reinterpret_cast<uintptr_t>(g0) & 127u);
            uint8_t *a0 = g0 - head;                                             // cache-line aligned
            const uint8_t *l0 = s_px + (whole_rows ? 0u : s * l_pitch);          // LDS byte of global byte a0
            const uint32_t n_blocks = (head + bytes + 15u) >> 4;
            for (uint32_t j = (uint32_t)tid; j < n_blocks; j += (uint32_t)G::kThreads) {
                if (16u * j + 16u <= head || 16u * j >= head + bytes) continue;   // (blocks of the first line before the range)
                if (16u * j >= head && 16u * j + 16u <= head + bytes) {
                    const u32x4_t q = *reinterpret_cast<const u32x4_t *>(l0 + 16u * j);
                    if (DBDE_NT) __builtin_nontemporal_store(q, reinterpret_cast<u32x4_t *>(a0 + 16ull * j));
                    else *reinterpret_cast<u32x4_t *>(a0 + 16ull * j) = q;
                } else {       // first / last block of the range: only the bytes that belong to it
#pragma unroll
                    for (uint32_t b = 0; b < 16u; b++) {
                        const uint32_t o = 16u * j + b;
                        if (o >= head && o < head + bytes) a0[o] = l0[o];
                    }
                }
            }
        }
        return;
    }
    // ---- odd widths (whole rows; dbde_capi.cpp sends nothing else here): tile-aligned image, shifted copy-out ----
    // Pass 1, no branch but "has a tile": EVERY tile row goes in as one aligned 8-byte store -- the partial last tile of
    // an image row too (what it writes behind column W lies in the 16 bytes pass 2 rewrites), rows below the image too
    // (they lie behind the range that leaves).  (Round 4 tried the partial tile with its valid bytes only, which makes the
    // two passes' targets disjoint and saves the barrier between them: the byte stores cost the wave that holds a row end
    // more than the barrier costs everybody -- 1921x1081 decode 1.199 -> 1.223 ms, 1366x768 1.334 -> 1.377.)
    const uint32_t P = 8u * wspan + 16u;
    const uint32_t aA = 8u * rowA * P + 8u * colA, aB = 8u * rowB * P + 8u * colB;
    if (hasA) {
#pragma unroll
        for (int r = 0; r < 8; r++) lds_store_u64_any(s_px, aA + (uint32_t)r * P, va[2 * r], va[2 * r + 1]);
    }
    if (hasB) {
#pragma unroll
        for (int r = 0; r < 8; r++) lds_store_u64_any(s_px, aB + (uint32_t)r * P, vb[2 * r], vb[2 * r + 1]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    // Pass 2: the first 16 bytes of every image row once more, behind the end of the image row before it (W >= 16:
    // columns 0 and 1 are whole tiles; the chunk's first image row has no row before it in this chunk).
    if (hasA && colA <= 1u) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t ry = 8u * rowA + (uint32_t)r;
            if (ry != 0u) lds_store_u64_any(s_px, (ry - 1u) * P + Wu + 8u * colA, va[2 * r], va[2 * r + 1]);
        }
    }
    if (hasB && colB <= 1u) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t ry = 8u * rowB + (uint32_t)r;
            if (ry != 0u) lds_store_u64_any(s_px, (ry - 1u) * P + Wu + 8u * colB, vb[2 * r], vb[2 * r + 1]);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const uint32_t bytes = (y_end - y0) * Wu;
        uint8_t *a0 = g_first - g7;                                              // cache-line aligned
        const uint32_t n_blocks = (g7 + bytes + 15u) >> 4;
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t *)s_px;
        // A block's 16 bytes start at any LDS byte address.  An under-aligned ds_read_b128 is legal but took 2-3 times the
        // aligned time in this kernel (0.17 of 1.48 ms), so a block is five 4-byte aligned dwords (ds_read2_b32 x2 +
        // ds_read_b32) and four v_alignbyte_b32 by the address' low two bits.  Four blocks per thread in flight, one wait.
        // The first and the last block of the range are partial: their bytes (1..15, read from the first valid one
        // on) leave as one store per set bit of the count.
        const uint32_t last_n = (g7 + bytes) & 15u;                    // valid bytes of the last block (0: it is whole)
        for (uint32_t j0 = (uint32_t)tid; j0 < n_blocks; j0 += 4u * (uint32_t)G::kThreads) {
            uint64_t d01[4], d23[4];
            uint32_t d4[4], a[4];
            bool live[4];
#pragma unroll
            for (uint32_t k = 0; k < 4u; k++) {
                const uint32_t j = j0 + k * (uint32_t)G::kThreads;
                live[k] = j < n_blocks && 16u * j + 16u > g7;          // holds at least one byte of the range
                const uint32_t o = live[k] && 16u * j > g7 ? 16u * j - g7 : 0u;   // range offset of the first byte looked at
                uint32_t col;
                const uint32_t row = div_magic(o, Wu, p.magic_W, col);
                a[k] = lds0 + row * P + col;
                const uint32_t a4 = a[k] & ~3u;
                asm volatile("ds_read2_b32 %0, %3 offset1:1\n\tds_read2_b32 %1, %3 offset0:2 offset1:3\n\tds_read_b32 %2, %3 offset:16"
                             : "=&v"(d01[k]), "=&v"(d23[k]), "=&v"(d4[k]) : "v"(a4) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d01[0]), "+v"(d23[0]), "+v"(d4[0]), "+v"(d01[1]), "+v"(d23[1]), "+v"(d4[1]),
                         "+v"(d01[2]), "+v"(d23[2]), "+v"(d4[2]), "+v"(d01[3]), "+v"(d23[3]), "+v"(d4[3]) :: "memory");
#pragma unroll
            for (uint32_t k = 0; k < 4u; k++) {
                const uint32_t j = j0 + k * (uint32_t)G::kThreads;
                const uint32_t w0 = (uint32_t)d01[k], w1 = (uint32_t)(d01[k] >> 32), w2 = (uint32_t)d23[k], w3 = (uint32_t)(d23[k] >> 32);
                u32x4_t q;
                q[0] = __builtin_amdgcn_alignbyte(w1, w0, a[k]);
                q[1] = __builtin_amdgcn_alignbyte(w2, w1, a[k]);
                q[2] = __builtin_amdgcn_alignbyte(w3, w2, a[k]);
                q[3] = __builtin_amdgcn_alignbyte(d4[k], w3, a[k]);
                if (!live[k]) continue;
                const bool first = 16u * j < g7, last = 16u * j + 16u > g7 + bytes;
                if (!first && !last) {
                    if (DBDE_NT) __builtin_nontemporal_store(q, reinterpret_cast<u32x4_t *>(a0 + 16ull * j));
                    else *reinterpret_cast<u32x4_t *>(a0 + 16ull * j) = q;
                } else {
                    // q holds the range's bytes from the block's first valid one: n of them go to dst (any alignment)
                    uint8_t *dst = first ? g_first : a0 + 16ull * j;
                    uint32_t n = first ? 16u - (g7 & 15u) : last_n;
                    if (first && last) n = bytes;                       // (a range shorter than one block)
                    uint64_t lo = ((uint64_t)q[1] << 32) | q[0], hi = ((uint64_t)q[3] << 32) | q[2];
                    if (n & 8u) { store_u64_any(dst, lo); dst += 8; lo = hi; }
                    if (n & 4u) { const uint32_t v = (uint32_t)lo; __builtin_memcpy(dst, &v, 4); dst += 4; lo >>= 32; }
                    if (n & 2u) { const uint16_t v = (uint16_t)lo; __builtin_memcpy(dst, &v, 2); dst += 2; lo >>= 16; }
                    if (n & 1u) *dst = (uint8_t)lo;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// DECODE, small frames (T <= DBDE_MID_DECODE_TILES = 256 tiles, from single-tile frames up): whole frames per workgroup
// ---------------------------------------------------------------------------------------
// The reference's own randomized test decodes 1024 single-tile frames (dbde_util_test.cpp:66-96,366); thumbnails are the
// same regime.  A workgroup per 512-tile chunk leaves most tile slots empty there and the index kernel (a workgroup per
// frame) costs more than the decode it serves (64x64 x 262,144 frames: index 2.6 ms, decode 1.1 ms, 0.18 of peak together;
// a 72x72 frame in a chunk: 0.22).  Here a lane owns ONE tile, a WORKGROUP of 256 threads as many whole frames as fit
// (mid_decode_threads_for); the frame's validation (dbde_util.cpp:295-303, plus depth <= 8 and the readable extent) and
// the tile's word offset are a segmented scan over the depth bytes through LDS.  No index kernel, no chunk.
//
// Round 4, second half: the workgroups are PERSISTENT and the groups of frames they walk are software-pipelined.  A group's
// life was a chain of four dependent memory round trips (frame offset -> depth bytes -> the three I32 fields -> payload rows)
// with 24 KB moved per workgroup at the end of it: at eight workgroups per CU that is latency, not bandwidth (72x72: 0.38 of
// peak with the vector ALU a quarter busy).  Now a workgroup holds, while it decodes group g, the depth / minimum bytes and
// the fields of group g + G (requested one iteration earlier, all of them in ONE batch as soon as the offset is known) and
// the offset of group g + 2 G: one exposed round trip per group, the payload rows, and one barrier (the scan's LDS arrays
// alternate between two sets so that no second barrier is needed before they are written again).
struct MidMeta {
    uint32_t d, mn;           // the lane's depth and minimum bytes
    uint32_t nb, nm, n64;     // the frame's three I32 fields (every lane of a frame asks for the same words)
};

#ifndef DBDE_MID_NT
#define DBDE_MID_NT DBDE_NT   // A/B switch: the staged image leaves with non-temporal stores
#endif
#ifndef DBDE_MID_STAGE_ROWS
#define DBDE_MID_STAGE_ROWS 4   // image rows (and bases) that are multiples of this many bytes take the staged form (A/B: 8 = the round-4b start)
#endif
#ifndef DBDE_MID_NO_STAGE
#define DBDE_MID_NO_STAGE 0   // A/B switch: 1 = tile rows stored straight from the registers at every width
#endif
#ifndef DBDE_MID_WAVES
#define DBDE_MID_WAVES 6   // (the register allocator's target; the 256-thread instance comes out at 64 VGPRs = eight waves per SIMD without scratch, asked for eight it spills)
#endif
// STAGED (4-byte aligned image rows: W % 4 == 0 and a 4-byte aligned base; W % 8 == 0: aligned 8-byte LDS stores, else two
// dwords per tile row and one for the row's last tile): the group's frames are ONE contiguous byte range
// of the output (frames follow each other in the batch).  Eight 8-byte stores per lane at a stride of W scatter a wave's
// store over a dozen partial cache lines -- with the payload loads ablated the kernel still took 0.50 of its 0.72 ms on
// 72x72 frames, without the stores 0.34 -- so the tile rows go into an LDS image of that range (aligned ds_write_b64,
// placed so that LDS and global addresses agree mod 16) and leave as aligned 16-byte stores, whole cache lines per wave;
// one more LDS-only barrier per group.
template <int THREADS, bool STAGED>
__global__ __launch_bounds__(THREADS, THREADS == 1024 ? 4 : DBDE_MID_WAVES) void decode_mid_kernel(DecParams p) {   // (1024 threads: one workgroup per CU either way)
    constexpr int NW = THREADS / 64;
    __shared__ uint32_t s_tot[2][NW];
    __shared__ uint32_t s_incl[2][THREADS];
    // a payload slot per lane (five 16-byte pieces: up to 3 bytes of alignment + 64 of payload; the last row's three dwords end
    // at byte 68), and in the same memory, later
    // in a group's life, the image of the group's pixels (STAGED: THREADS / T frames of T tiles, 16 bytes of alignment)
    constexpr uint32_t kMidSlot = 80u;
    __shared__ __attribute__((aligned(16))) uint8_t s_lds[THREADS * kMidSlot];
    __shared__ uint32_t s_ok[THREADS];   // per frame of the group: decoded (as many frames as lanes when T = 1)
    __shared__ uint32_t s_bad[2];        // ... and "a frame of the group was rejected", alternating like the scan's arrays
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t T = p.T, fpw = (uint32_t)THREADS / T;      // frames per workgroup
    const uint32_t fl = tid / T, t = tid - fl * T;            // frame within the workgroup, tile within the frame
    const uint32_t n_frames = p.n_chunks;                     // (n_chunks carries the frame count here)
    const uint32_t n_groups = (n_frames + fpw - 1u) / fpw, G = gridDim.x;
    const bool slot = fl < fpw;
    const uint32_t first = slot ? fl * T : 0u;                // the frame's first lane
    const uint64_t need = 32ull + 2ull * T;
    const uint8_t *s_end = p.stream + p.stream_bytes;
    // where the lane's tile lies in a frame (the same for every group)
    const uint32_t ty = t / p.w, tx = t - ty * p.w;
    const uint32_t Wu = (uint32_t)p.W, x0 = 8u * tx;
    const uint32_t rm = Wu - x0 < 8u ? Wu - x0 : 8u;                                    // valid columns
    const uint32_t rv = (uint32_t)p.H - 8u * ty < 8u ? (uint32_t)p.H - 8u * ty : 8u;    // valid rows
    const uint32_t tile_off = 8u * ty * Wu + x0;                                       // (a frame of <= 1024 tiles)
    const bool want_results = p.results != nullptr;

    auto is_active = [&](uint32_t g) -> bool { return slot && g < n_groups && g * fpw + fl < n_frames; };
    auto get_foff = [&](uint32_t g) -> uint64_t { return is_active(g) ? p.frame_offsets[g * fpw + fl] : 0ull; };
    auto get_meta = [&](uint32_t g, uint64_t foff) -> MidMeta {
        MidMeta m;
        m.d = 0; m.mn = 0; m.nb = 0; m.nm = 0; m.n64 = 0;
        const bool active = is_active(g);
        const uint8_t *fb = p.stream + foff;
        if (active && in_extent(foff, need, p.stream_bytes)) {
            m.d = fb[24u + t]; m.mn = fb[28u + T + t];
            __builtin_memcpy(&m.nb, fb + 20, 4); __builtin_memcpy(&m.nm, fb + 24 + T, 4); __builtin_memcpy(&m.n64, fb + 28 + 2ull * T, 4);
        }
        return m;
    };

    if (STAGED && tid == 0u) { s_bad[0] = 0u; s_bad[1] = 0u; }
    uint32_t g = blockIdx.x;
    uint64_t foff_cur = get_foff(g), foff_nxt = get_foff(g + G);
    // what is carried from one iteration to the next of a group's metadata: depth | minimum << 8, and the frame's n64 field
    // (all ones when nb or nm is not T: no sum of depths equals that)
    auto settle = [&](const MidMeta &x, uint32_t &dm, uint32_t &n64x) {
        dm = x.d | (x.mn << 8);
        n64x = ((int32_t)x.nb == (int32_t)T && (int32_t)x.nm == (int32_t)T) ? x.n64 : 0xFFFFFFFFu;
    };
    uint32_t dm, n64x;
    settle(get_meta(g, foff_cur), dm, n64x);
    for (uint32_t par = 0; g < n_groups; g += G, par ^= 1u) {
        // ---- requests for the groups behind this one (nothing here waits for memory: foff_nxt arrived an iteration ago) ----
        MidMeta m_nxt = get_meta(g + G, foff_nxt);
        uint64_t foff_nn = get_foff(g + 2u * G);
        // ---- this group ----
        const bool active = is_active(g);
        const uint32_t f = g * fpw + fl;
        const uint64_t foff = foff_cur;
        const bool in_range = active && in_extent(foff, need, p.stream_bytes);
        const uint32_t d = dm & 0xFFu, mn = dm >> 8;
        // depth sum and "a depth above 8" count of every frame in one scan: low 20 bits words (<= 1024 * 255), above them flags
        const uint32_t item = in_range ? (d | (d > 8u ? 1u << 20 : 0u)) : 0u;
        const uint32_t w_incl = wave_scan_incl(item);
        s_incl[par][tid] = w_incl;
        if (lane == 63u) s_tot[par][wave] = w_incl;
        lds_barrier();   // (LDS only: __syncthreads() would also wait for the prefetch that has just been asked for)
        if (STAGED && tid == 0u) s_bad[par ^ 1u] = 0u;   // (the previous group's flag: everybody has looked at it by now)
        // inclusive prefix over the workgroup at lane x: the wave's own scan + the totals of the waves in front of it
        auto pref = [&](uint32_t x) -> uint32_t {
            uint32_t a = s_incl[par][x];
#pragma unroll
            for (int k = 0; k < NW - 1; k++) a += (uint32_t)k < (x >> 6) ? s_tot[par][k] : 0u;
            return a;
        };
        // (lanes without a frame take the values of lane 0's: in bounds, never used)
        const uint32_t base = first ? pref(active ? first - 1u : 0u) : 0u;
        const uint32_t upto = pref(active ? first + T - 1u : 0u);
        const uint32_t mine = pref(tid);
        const uint32_t total = (upto - base) & 0xFFFFFu, n_bad = (upto - base) >> 20;
        const uint32_t prefix = (mine - item - base) & 0xFFFFFu;   // payload words of the frame in front of this tile
        const bool ok = in_range && n64x == total && n_bad == 0u && in_extent(foff, need + 8ull * total, p.stream_bytes);
        // the frame's result record (dbde_unpack_frame's return value): its header words are asked for here, in front of
        // the payload rows, and consumed behind them -- the same round trip
        uint32_t h_field = 0;
        uint64_t h_index = 0, h_elapsed = 0;
        const bool record = active && t == 0u && want_results;
        if (record && in_extent(foff, 20, p.stream_bytes)) {
            const uint8_t *fb = p.stream + foff;
            __builtin_memcpy(&h_field, fb, 4); __builtin_memcpy(&h_index, fb + 4, 8); __builtin_memcpy(&h_elapsed, fb + 12, 8);
        }
        // The tile's payload: 8 d contiguous bytes, fetched as ceil(d / 2) 16-byte pieces from its first byte on (eight
        // 8-byte row loads per lane, most of their bytes fetched twice or more, were a third of the kernel's time) and put
        // down in a slot of the lane's own in LDS, from which the rows are cut as the chunk decoder cuts them
        // (unpack_tile_from_lds: three aligned dwords + v_alignbyte per row).  The one wait behind the loads is a wait the
        // whole loop body shares: behind it the prefetched words of the next group have arrived as well (they were asked
        // for earlier), and nothing behind the group's stores waits for memory again.
        // (The pieces start at the dword boundary in front of the payload: frames of an odd tile count have their payload at
        // 2 mod 4, and 16-byte loads from there ran a fifth slower per tile than from dword boundaries -- 72x72, 88x72, 104x72,
        // 88x56 against 80x64, 72x64, 96x64, 80x72, profiles/r04b_parity.sh.  The up to three bytes in front are the end of the
        // frame's minimum array; a tile of even depth may need one piece more, five for depth 8.)
        const uint8_t *pay = p.stream + foff + need + 8ull * prefix;
        const uint32_t poff = (uint32_t)(reinterpret_cast<uintptr_t>(pay) & 3u);
        const uint8_t *psrc = pay - poff;
        const uint32_t npc_all = (8u * d + poff + 15u) >> 4;
#ifdef DBDE_MID_ABLATE_LOADS
        const bool whole = false;
#else
        const bool whole = ok && d != 0u && psrc + 16u * npc_all <= s_end;   // every piece lies inside the readable extent
#endif
        const uint32_t npc = whole ? npc_all : 0u;
        u32x4_t q0 = {0u, 0u, 0u, 0u}, q1 = q0, q2 = q0, q3 = q0, q4 = q0;
        if (npc > 0u) __builtin_memcpy(&q0, psrc, 16);
        if (npc > 1u) __builtin_memcpy(&q1, psrc + 16, 16);
        if (npc > 2u) __builtin_memcpy(&q2, psrc + 32, 16);
        if (npc > 3u) __builtin_memcpy(&q3, psrc + 48, 16);
        if (npc > 4u) __builtin_memcpy(&q4, psrc + 64, 16);
        // First look at anything this iteration asked for.  (Left to itself the scheduler moves the consumers of the
        // prefetched words up to their loads to shorten live ranges -- a wait for the round trip that was to be hidden,
        // seen in the listing -- and a look at them BEHIND the stores would wait for the stores too.)
        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4),
                          "+v"(m_nxt.d), "+v"(m_nxt.mn), "+v"(m_nxt.nb), "+v"(m_nxt.nm), "+v"(m_nxt.n64), "+v"(foff_nn),
                          "+v"(h_field), "+v"(h_index), "+v"(h_elapsed) :: "memory");
        const uint32_t slot = kMidSlot * tid;
        {
            u32x4_t *sl = reinterpret_cast<u32x4_t *>(s_lds + slot);
            if (npc > 0u) sl[0] = q0;
            if (npc > 1u) sl[1] = q1;
            if (npc > 2u) sl[2] = q2;
            if (npc > 3u) sl[3] = q3;
            if (npc > 4u) sl[4] = q4;
        }
#ifndef DBDE_MID_ABLATE_LOADS
        if (ok && d != 0u && !whole) {   // the stream's last bytes: nothing past the extent
            for (uint32_t b = 0; b < 8u * d; b++) s_lds[slot + poff + b] = pay[b];
        }
#endif
        uint32_t v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = 0;
        if (ok) unpack_tile_from_lds<false>(s_lds, slot + poff, d, mn, v);   // (a flat tile: the minimum, whatever the slot holds)
        if (STAGED) lds_barrier();   // the payload slots and the image of the group's pixels share the memory
        if (ok) {   // (a rejected frame's image stays untouched, dbde_util.cpp:296-303)
            uint8_t *dst = p.images + (size_t)f * p.frame_pixels + tile_off;
            // (STAGED: where the tile's first row lies in the image of the group's byte range; the range starts at 0 or 8 mod 16)
            const uint32_t px0 = ((uint32_t)(reinterpret_cast<uintptr_t>(p.images) + (size_t)(g * fpw) * p.frame_pixels) & 15u) +
                                 fl * (uint32_t)p.frame_pixels + tile_off;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t lo = v[2 * r], hi = v[2 * r + 1];
#ifdef DBDE_MID_ABLATE_STORES
                if (lo == 0x12345678u && hi == 0x9ABCDEF0u) {
#else
                if (STAGED) {
                    if ((uint32_t)r < rv) {   // (rows of W % 8 == 0: 8-byte aligned; of W % 8 == 4: two dwords, the row's last tile one)
                        const uint32_t a = px0 + (uint32_t)r * Wu;
                        if ((Wu & 7u) == 0u) lds_store_u64_any(s_lds, a, lo, hi);
                        else {
                            *reinterpret_cast<uint32_t *>(s_lds + a) = lo;
                            if (rm == 8u) *reinterpret_cast<uint32_t *>(s_lds + a + 4u) = hi;
                        }
                    }
                } else if ((uint32_t)r < rv) {   // only the valid region is written (dbde_util.cpp:281-289)
#endif
                    uint8_t *row = dst + (size_t)r * Wu;
                    const uint64_t q = ((uint64_t)hi << 32) | lo;
                    if (rm == 8u) store_u64_any(row, q);
                    else {
#pragma unroll
                        for (uint32_t k = 0; k < 8u; k++)
                            if (k < rm) row[k] = (uint8_t)(q >> (8u * k));
                    }
                }
            }
        }
        if (STAGED) {
            if (active && t == 0u) { s_ok[fl] = ok ? 1u : 0u; if (!ok) s_bad[par] = 1u; }
            lds_barrier();
            const uint32_t P = (uint32_t)p.frame_pixels;
            const uint32_t nfr = n_frames - g * fpw < fpw ? n_frames - g * fpw : fpw;      // frames of this group
            uint8_t *g0 = p.images + (size_t)(g * fpw) * p.frame_pixels;                   // its first byte in the output
            const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(g0) & 15u);         // a multiple of 4
            const uint32_t nbytes = nfr * P;
            if (s_bad[par] == 0u) {   // every frame decoded: aligned 16-byte blocks, the partial one at either end dword by dword
                const uint32_t n_blocks = (sh + nbytes + 15u) >> 4;
                for (uint32_t j = tid; j < n_blocks; j += (uint32_t)THREADS) {
                    uint8_t *dst = g0 - sh + 16ull * j;
                    if (16u * j >= sh && 16u * j + 16u <= sh + nbytes) {
                        const u32x4_t q = *reinterpret_cast<const u32x4_t *>(s_lds + 16u * j);
#ifdef DBDE_MID_ABLATE_STORES
                        if (q[0] != 0x12345678u || q[3] != 0x9ABCDEF1u) continue;
#endif
                        if (DBDE_MID_NT) __builtin_nontemporal_store(q, reinterpret_cast<u32x4_t *>(dst));
                        else *reinterpret_cast<u32x4_t *>(dst) = q;
                    } else {
#pragma unroll
                        for (uint32_t k = 0; k < 4u; k++) {
                            const uint32_t o = 16u * j + 4u * k;
                            if (o >= sh && o < sh + nbytes) *reinterpret_cast<uint32_t *>(dst + 4u * k) = *reinterpret_cast<const uint32_t *>(s_lds + o);
                        }
                    }
                }
            } else {                          // a rejected frame's image stays untouched: 4 bytes at a time, frame by frame
                for (uint32_t u = tid; 4u * u < nbytes; u += (uint32_t)THREADS) {
                    const uint32_t o = 4u * u;
                    if (s_ok[o / P]) *reinterpret_cast<uint32_t *>(g0 + o) = *reinterpret_cast<const uint32_t *>(s_lds + sh + o);
                }
            }
        }
        if (record) {
            FrameResultDev *r = reinterpret_cast<FrameResultDev *>(p.results) + f;
            r->u64s = (h_field == 2u && ok) ? 2u : 0xFFFFFFFFu;   // dbde_util.cpp:335,342
            r->pad_ = 0;
            r->index = h_index;
            r->elapsed_ns = f64_to_u64_x86(__longlong_as_double((long long)h_elapsed));
            r->consumed = ok ? need + 8ull * total : 20ull;
        }
        foff_cur = foff_nxt; foff_nxt = foff_nn;
        settle(m_nxt, dm, n64x);
    }
}

hipError_t launch_decode_mid(const DecParams &p, uint32_t n_frames, uint32_t n_cu, hipStream_t s) {
    DecParams q = p;
    q.n_chunks = n_frames;
    const uint32_t th = mid_decode_threads_for(p.T), per_wg = th / p.T;
    uint32_t groups = (n_frames + per_wg - 1u) / per_wg;
    // 8-byte aligned image rows: pixels staged in LDS, aligned 16-byte stores (decode_mid_kernel<., true>)
    const bool staged = p.W % DBDE_MID_STAGE_ROWS == 0 && (reinterpret_cast<uintptr_t>(p.images) & (DBDE_MID_STAGE_ROWS - 1)) == 0u && !DBDE_MID_NO_STAGE;
    auto go = [&](auto kernel, uint32_t threads) {
        // persistent: as many workgroups as the device holds (n_cu = 0, tests: three, so that small batches walk the
        // pipelined loop too)
        static int resident_per_cu[2][3] = {{0, 0, 0}, {0, 0, 0}};   // (asked once per instance)
        int &per_cu = resident_per_cu[staged ? 1 : 0][threads == 256u ? 0 : (threads == 512u ? 1 : 2)];
        if (per_cu < 1 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)threads, 0) != hipSuccess || per_cu < 1)) per_cu = 1;
        const uint32_t resident = n_cu ? n_cu * (uint32_t)per_cu : 3u;
        hipLaunchKernelGGL(kernel, dim3(groups < resident ? groups : resident), dim3(threads), 0, s, q);
    };
    if (staged) {
        if (th == 256u) go(decode_mid_kernel<256, true>, 256u);
        else if (th == 512u) go(decode_mid_kernel<512, true>, 512u);
        else go(decode_mid_kernel<1024, true>, 1024u);
    } else {
        if (th == 256u) go(decode_mid_kernel<256, false>, 256u);
        else if (th == 512u) go(decode_mid_kernel<512, false>, 512u);
        else go(decode_mid_kernel<1024, false>, 1024u);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Frames of 65 .. 1024 tiles, whole frames per workgroup, STAGED: coalesced on both sides (round 4)
// ---------------------------------------------------------------------------------------
// encode_mid_kernel / decode_mid_kernel fetch a tile with eight strided 8-byte loads per lane and emit its payload with
// per-lane 8-byte stores: the access shape the chunk kernels were built to avoid, 0.37-0.50 of peak, and only up to 256
// (160) tiles; above that a frame took a whole 1024-tile chunk with most lanes idle (160x120: 0.36).  Here a workgroup of
// 256 or 512 threads owns as many whole frames as fit its 512 / 1024 tile slots (two tiles per lane) and
//   * the frames' pixels -- ONE contiguous byte range, frames follow each other in the batch -- travel as whole 16-byte
//     blocks between global memory and an LDS image; tiles are cut out of (put into) that image with aligned 8-byte LDS
//     accesses (taken when rows are 8-byte aligned: W % 8 == 0, frames and base multiples of 16 bytes);
//   * a frame's stream bytes travel as aligned 16-byte blocks too: the encoder assembles every frame in LDS -- header,
//     fields, depth and minimum bytes as they lie in the frame; payload words 8-byte aligned beside them -- and the
//     copy-out shifts the payload into place (five aligned dwords and four v_alignbyte per block, as the staged decoder
//     does); the decoder lands the frame's bytes where LDS and global addresses agree mod 16 and unpacks from there.
// One slot per frame on the encode side (nothing is shared between frames: no workspace, nothing to wait for); the decoder
// takes any frame offsets.  Validation is the reference's (dbde_util.cpp:295-303) plus depth <= 8 and the readable extent.
// 16 bytes at offset o of a frame whose first `meta` = 32 + 2T bytes lie at m and whose payload words lie 8-byte aligned at
// LDS byte address pay0 (pay: the same place as a pointer): whole blocks of the fields' image as they are; payload blocks
// as five aligned dwords shifted into place; the one block that holds the end of the minimum array byte by byte.
__device__ __forceinline__ u32x4_t frame_block(const uint8_t *m, uint32_t pay0, const uint8_t *pay, uint32_t meta, uint32_t o) {
    u32x4_t q;
    if (o + 16u <= meta) {
        q = *reinterpret_cast<const u32x4_t *>(m + o);
    } else if ((meta & 15u) == 0u) {                        // T a multiple of 8: payload blocks are blocks of the LDS image as they are
        // (64x64 encode 0.54 -> 0.58 mixed / 0.57 -> 0.63 incompressible, 128x128 0.61 -> 0.64 / 0.55 -> 0.61; the general path's reads
        // as plain loads instead of the spelled-out three + wait: no difference)
        q = *reinterpret_cast<const u32x4_t *>(pay + (o - meta));
    } else {
        const int po = (int)o - (int)meta;                  // may be negative in the straddling block
        const uint32_t a = (uint32_t)((int)pay0 + po), a4 = a & ~3u;
        uint64_t d01, d23;
        uint32_t d4;
        asm volatile("ds_read2_b32 %0, %3 offset1:1\n\tds_read2_b32 %1, %3 offset0:2 offset1:3\n\tds_read_b32 %2, %3 offset:16\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(d01), "=&v"(d23), "=&v"(d4) : "v"(po < 0 ? pay0 & ~3u : a4) : "memory");
        const uint32_t w0 = (uint32_t)d01, w1 = (uint32_t)(d01 >> 32), w2 = (uint32_t)d23, w3 = (uint32_t)(d23 >> 32);
        q[0] = __builtin_amdgcn_alignbyte(w1, w0, a);
        q[1] = __builtin_amdgcn_alignbyte(w2, w1, a);
        q[2] = __builtin_amdgcn_alignbyte(w3, w2, a);
        q[3] = __builtin_amdgcn_alignbyte(d4, w3, a);
        if (po < 0) {
#pragma unroll
            for (uint32_t i = 0; i < 4u; i++) {
                uint32_t x = 0;
#pragma unroll
                for (uint32_t j = 0; j < 4u; j++) {
                    const uint32_t at = o + 4u * i + j;
                    x |= (uint32_t)(at < meta ? m[at] : pay[at - meta]) << (8u * j);
                }
                q[i] = x;
            }
        }
    }
    return q;
}

template <int THREADS>
struct FramesLds {
    static constexpr uint32_t kCap = 2u * THREADS;                 // tile slots
    static constexpr uint32_t kMaxFrames = kCap / 65u;             // frames of at least 65 tiles
    static constexpr uint32_t kMetaBytes = 2u * kCap + 48u * (kMaxFrames + 1u);   // per frame: 32 + 2 T rounded up to 16
};

template <int THREADS>
__global__ __launch_bounds__(THREADS) void encode_frames_kernel(EncParams p) {
    typedef FramesLds<THREADS> G;
    constexpr int NW = THREADS / 64;
    __shared__ __attribute__((aligned(16))) uint64_t s_px[G::kCap * 8u + THREADS];   // pixels, later payload words (+ a trash word per lane)
    __shared__ __attribute__((aligned(16))) uint8_t s_meta[G::kMetaBytes];
    __shared__ uint32_t s_tot[NW];
    __shared__ uint32_t s_fbase[G::kMaxFrames + 1], s_fend[G::kMaxFrames + 1];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t T = p.T, F = G::kCap / T;
    const uint32_t f0 = blockIdx.x * F;
    const uint32_t nf = p.n_chunks - f0 < F ? p.n_chunks - f0 : F;      // (n_chunks carries the frame count here)
    const uint32_t P = (uint32_t)p.frame_pixels, W = (uint32_t)p.W;
    const uint32_t meta = 32u + 2u * T, mpitch = (meta + 15u) & ~15u;

    // ---- the frames' pixels: one contiguous range, whole 16-byte blocks, by LDS-DMA (global_load_lds_dwordx4): every block
    //      of the workgroup is requested before the first one is waited for (a load -> ds_write loop paid one memory
    //      round trip per iteration, eight of them for six 72x72 frames) ----
    {
        const uint8_t *src = p.images + (size_t)f0 * P;
        const uint32_t n16 = nf * (P >> 4);
        for (uint32_t i = tid; i < n16; i += THREADS)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 16ull * i),
                                             (__attribute__((address_space(3))) void *)(s_px + 2u * (i - lane)), 16, 0, DBDE_NT ? 2 : 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (a barrier does not drain vector memory)
    }
    __syncthreads();

    // ---- two tiles per lane, cut out of the image (rows are 8-byte aligned; rows below the image repeat its last one) ----
    const uint8_t *s_img = reinterpret_cast<const uint8_t *>(s_px);
    uint32_t v[2][16], mn[2], d[2], fl[2], tt[2];
    bool has[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const uint32_t t = 2u * tid + (uint32_t)k;
        has[k] = t < nf * T;
        uint32_t rem;
        fl[k] = has[k] ? div_magic(t, T, p.magic_lpr, rem) : 0u;       // (magic_lpr carries floor(2^32 / T) here)
        tt[k] = has[k] ? rem : 0u;
        uint32_t tx;
        const uint32_t ty = div_magic(tt[k], p.w, p.magic_w, tx);
        const uint32_t base = fl[k] * P + 8u * tx;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint32_t yy = 8u * ty + (uint32_t)r;
            yy = yy < (uint32_t)p.H ? yy : (uint32_t)p.H - 1u;
            const uint2 q = *reinterpret_cast<const uint2 *>(s_img + base + yy * W);
            v[k][2 * r] = q.x; v[k][2 * r + 1] = q.y;
        }
        uint32_t mx;
        tile_minmax(v[k], mn[k], mx);
        d[k] = has[k] ? depth_of_range(mx - mn[k]) : 0u;
    }
    uint32_t block_total;
    const uint32_t incl = block_scan_incl<NW>(d[0] + d[1], s_tot, (int)lane, (int)wave, block_total);   // barrier inside: every tile is in registers
    const uint32_t excl[2] = {incl - d[0] - d[1], incl - d[1]};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (has[k] && tt[k] == 0u) s_fbase[fl[k]] = excl[k];
        if (has[k] && tt[k] == T - 1u) s_fend[fl[k]] = excl[k] + d[k];
    }
    __syncthreads();   // (also: nobody reads the pixel image any more)

    // ---- payload words into the frame's 8-byte aligned LDS area; depth / minimum bytes and the fields where they lie ----
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const uint32_t q = fl[k] * 8u * T + (excl[k] - s_fbase[fl[k]]);
        if (has[k]) pack_tile(v[k], mn[k], d[k], s_px, q, G::kCap * 8u + tid);
        if (has[k]) {
            uint8_t *m = s_meta + fl[k] * mpitch;
            m[24u + tt[k]] = (uint8_t)d[k];
            m[28u + T + tt[k]] = (uint8_t)mn[k];
            if (tt[k] == 0u) {   // header + the three I32 fields (dbde_util.cpp:140-146, 182-196); trap T1: F64 on the wire
                const uint32_t f = f0 + fl[k];
                const uint64_t index = p.indices ? p.indices[f] : p.first_index + f;
                const uint64_t el = p.elapsed_ns ? p.elapsed_ns[f] : 0ull;
                const uint64_t elbits = (uint64_t)__double_as_longlong(__ull2double_rn(el));
                const uint32_t n64 = s_fend[fl[k]] - s_fbase[fl[k]];
                uint32_t *h = reinterpret_cast<uint32_t *>(m);      // (the frame's LDS image starts 16-byte aligned)
                h[0] = 2u; h[1] = (uint32_t)index; h[2] = (uint32_t)(index >> 32);
                h[3] = (uint32_t)elbits; h[4] = (uint32_t)(elbits >> 32); h[5] = T;
                store_u32_bytes(m + 24u + T, T);
                store_u32_bytes(m + 28u + 2u * T, n64);
                if (p.frame_offsets) p.frame_offsets[f] = (uint64_t)f * p.slot_stride;
                if (p.frame_bytes) p.frame_bytes[f] = (uint64_t)meta + 8ull * n64;
            }
        }
    }
    __syncthreads();

    // ---- every frame leaves as aligned 16-byte blocks: exactly its 32 + 2 T + 8 n64 bytes ----
    const uint32_t lds_pay = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t *)reinterpret_cast<const uint8_t *>(s_px);
    for (uint32_t g = 0; g < nf; g++) {
        const uint32_t len = meta + 8u * (s_fend[g] - s_fbase[g]);
        const uint32_t n_blocks = (len + 15u) >> 4;
        uint8_t *out = p.out + (uint64_t)(f0 + g) * p.slot_stride;
        const uint8_t *m = s_meta + g * mpitch;
        const uint32_t pay0 = lds_pay + g * 64u * T;                 // LDS byte address of the frame's payload
        const uint8_t *pay = reinterpret_cast<const uint8_t *>(s_px) + g * 64u * T;
        for (uint32_t b = tid; b < n_blocks; b += THREADS) {
            const uint32_t o = 16u * b;
            const u32x4_t q = frame_block(m, pay0, pay, meta, o);
            uint8_t *dst = out + o;
            if (o + 16u <= len) {
                if (DBDE_NT) __builtin_nontemporal_store(q, reinterpret_cast<u32x4_t *>(dst));
                else *reinterpret_cast<u32x4_t *>(dst) = q;
            } else {   // the frame's last, partial block: its bytes and nothing behind them (lengths are even)
                uint32_t n = len - o;
                uint64_t lo = ((uint64_t)q[1] << 32) | q[0];
                const uint64_t hi = ((uint64_t)q[3] << 32) | q[2];
                if (n & 8u) { *reinterpret_cast<uint64_t *>(dst) = lo; dst += 8; lo = hi; }
                if (n & 4u) { *reinterpret_cast<uint32_t *>(dst) = (uint32_t)lo; dst += 4; lo >>= 32; }
                if (n & 2u) *reinterpret_cast<uint16_t *>(dst) = (uint16_t)lo;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// ENCODE, small frames in slots, 8-byte aligned rows, 1 .. 256 tiles: persistent workgroups, pixels double-buffered (round 4)
// ---------------------------------------------------------------------------------------
// What the decode side gained from walking groups of frames in a software-pipelined loop (decode_mid_kernel), mirrored:
// a lane owns ONE tile, a 256-thread workgroup as many whole frames as fit (256 / T), and the workgroup is persistent.
// The pixels of a group are one contiguous byte range (frames follow each other in the batch) whose address is plain
// arithmetic, so the range of the NEXT group is requested by LDS-DMA into the other of two pixel buffers at the top of an
// iteration and has a whole iteration to arrive: no wave ever waits for a pixel except in the launch's first step.
// Tiles are cut out of the image with aligned 8-byte LDS reads, offsets are a segmented scan (DPP + wave totals), the
// payload words go where the pixels were (the scan's barrier says every tile has been cut), header, fields, depth and
// minimum bytes into a small image of the frame's first 32 + 2T bytes, and every frame leaves as aligned 16-byte blocks
// -- the lanes of a frame store ITS blocks, all frames of the group at once (encode_frames_kernel walks the frames one
// after the other with the whole workgroup: most lanes idle on frames of a few hundred bytes).
// One slot per frame (nothing is shared between frames: no workspace, nothing to wait for).
constexpr uint32_t kGroupThreads = 256u;

// ROWS4: rows and bases of 4-byte multiples (the image shifted to agree with global memory mod 16, head / tail dwords, dword
// tile rows); else rows of 8-byte multiples and frames / bases of whole 16-byte blocks (shift 0 throughout: 3 % faster there)
template <bool ROWS4>
__global__ __launch_bounds__(kGroupThreads, 4) void encode_group_kernel(EncParams p) {
    constexpr uint32_t TH = kGroupThreads, NW = TH / 64u, kWords = TH * 8u;       // 8 payload words (64 pixels) per tile slot
    extern __shared__ __attribute__((aligned(16))) uint8_t s_meta[];               // (256 / T) x the frame's first 32 + 2T bytes, rounded up to 16
    __shared__ __attribute__((aligned(16))) uint64_t s_buf[2][kWords + 64u];      // a group's pixels, later its payload words (+ a trash word per lane of a wave)
    __shared__ uint32_t s_tot[NW];
    __shared__ uint32_t s_incl[TH];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t T = p.T, fpw = TH / T;
    const uint32_t fl = tid / T, t = tid - fl * T;
    const uint32_t n_frames = p.n_chunks, n_groups = (n_frames + fpw - 1u) / fpw, G = gridDim.x;   // (n_chunks carries the frame count here)
    const uint32_t first = fl < fpw ? fl * T : 0u;
    const uint32_t P = (uint32_t)p.frame_pixels, W = (uint32_t)p.W;
    const uint32_t meta = 32u + 2u * T, mpitch = (meta + 15u) & ~15u;
    uint32_t tx;
    const uint32_t ty = div_magic(t, p.w, p.magic_w, tx);
    const uint32_t px = (fl < fpw ? fl * P : 0u) + 8u * tx;
    const uint32_t rm = W - 8u * tx < 8u ? W - 8u * tx : 8u;          // valid columns: 8, or 4 in the last tile of rows of 4 mod 8 bytes
    const bool rows8 = !ROWS4 || ((W & 7u) == 0u && (reinterpret_cast<uintptr_t>(p.images) & 7u) == 0u);   // (uniform) tile rows are aligned 8-byte reads

    // The pixels of group g -> buffer b, nothing waited for.  The image is placed where LDS and global addresses agree mod 16
    // (bases and rows are multiples of 4 bytes, not necessarily of 16): every 16-byte block that lies inside the group's
    // byte range travels as one, the up to three dwords in front of the first and behind the last of them as dwords --
    // nothing outside the range is read.
    auto group_shift = [&](uint32_t g) -> uint32_t {
        return ROWS4 ? (uint32_t)((reinterpret_cast<uintptr_t>(p.images) + (size_t)(g * fpw) * P) & 15u) : 0u;
    };
    auto fetch = [&](uint32_t g, uint32_t b) {
        if (g >= n_groups) return;
        const uint32_t nfr = n_frames - g * fpw < fpw ? n_frames - g * fpw : fpw;
        const uint8_t *src = p.images + (size_t)(g * fpw) * P;
        const uint32_t sh = group_shift(g), total = sh + nfr * P;
        const uint8_t *src0 = src - sh;                              // 16-byte aligned
        const uint32_t j0 = sh ? 1u : 0u, j1 = total >> 4;           // whole blocks [j0, j1)
        for (uint32_t i = tid; j0 + i < j1; i += TH)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src0 + 16ull * (j0 + i)),
                                             (__attribute__((address_space(3))) void *)(&s_buf[b][2u * (j0 + i - lane)]), 16, 0, DBDE_NT ? 2 : 0);
        uint8_t *img = reinterpret_cast<uint8_t *>(s_buf[b]);
        if (ROWS4 && wave == 0u) {
            const uint32_t head = sh ? (16u - sh) >> 2 : 0u;         // dwords in front of the first whole block
            if (lane < head)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 4u * lane),
                                                 (__attribute__((address_space(3))) void *)(img + sh), 4, 0, 0);
            const uint32_t tail = (total & 15u) >> 2;                // ... behind the last one
            if (lane < tail)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src0 + 16ull * j1 + 4u * lane),
                                                 (__attribute__((address_space(3))) void *)(img + 16u * j1), 4, 0, 0);
        }
    };
    auto pref = [&](uint32_t x) -> uint32_t {    // inclusive prefix of the depths over the workgroup at lane x
        uint32_t a = s_incl[x];
#pragma unroll
        for (uint32_t k = 0; k + 1u < NW; k++) a += k < (x >> 6) ? s_tot[k] : 0u;
        return a;
    };

    uint32_t g = blockIdx.x, cur = 0;
    fetch(g, 0u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (a barrier does not drain vector memory)
    lds_barrier();
    for (; g < n_groups; g += G, cur ^= 1u) {
        fetch(g + G, cur ^ 1u);                  // in flight until this group's frames are ready to leave
        const uint32_t sh_cur = group_shift(g);
        const uint32_t nfr = n_frames - g * fpw < fpw ? n_frames - g * fpw : fpw;
        const bool active = fl < nfr;
        const uint32_t f = g * fpw + fl;
        // ---- the lane's tile, cut out of the image (rows are 8-byte aligned; rows below the image repeat its last one) ----
        uint32_t v[16];
        {
            const uint8_t *img = reinterpret_cast<const uint8_t *>(s_buf[cur]);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                uint32_t yy = 8u * ty + (uint32_t)r;
                yy = yy < (uint32_t)p.H ? yy : (uint32_t)p.H - 1u;
                const uint32_t a = sh_cur + px + yy * W;
                if (rows8) {
                    const uint2 q = *reinterpret_cast<const uint2 *>(img + a);
                    v[2 * r] = q.x; v[2 * r + 1] = q.y;
                } else {   // 4-byte aligned rows; the row's last tile of a width of 4 mod 8: the last valid pixel repeated (dbde_util.cpp:116-128)
                    const uint32_t lo = *reinterpret_cast<const uint32_t *>(img + a);
                    v[2 * r] = lo;
                    v[2 * r + 1] = rm == 8u ? *reinterpret_cast<const uint32_t *>(img + a + 4u) : (lo >> 24) * 0x01010101u;
                }

            }
        }
        uint32_t mn, mx;
        tile_minmax(v, mn, mx);
        const uint32_t d = active ? depth_of_range(mx - mn) : 0u;
        const uint32_t w_incl = wave_scan_incl(d);
        s_incl[tid] = w_incl;
        if (lane == 63u) s_tot[wave] = w_incl;
        lds_barrier();                           // (also: every tile is in registers, the image may be written over)
        const uint32_t base = first ? pref(active ? first - 1u : 0u) : 0u;
        const uint32_t n64 = pref(active ? first + T - 1u : 0u) - base;
        const uint32_t prefix = pref(tid) - d - base;
        // ---- payload words into the frame's 8-byte aligned area; depth / minimum bytes and the fields where they lie ----
        if (active) {
            pack_tile(v, mn, d, s_buf[cur], fl * 8u * T + prefix, kWords + lane);
            uint8_t *m = s_meta + fl * mpitch;
            m[24u + t] = (uint8_t)d;
            m[28u + T + t] = (uint8_t)mn;
            if (t == 0u) {   // header + the three I32 fields (dbde_util.cpp:140-146, 182-196); trap T1: F64 on the wire
                const uint64_t index = p.indices ? p.indices[f] : p.first_index + f;
                const uint64_t el = p.elapsed_ns ? p.elapsed_ns[f] : 0ull;
                const uint64_t elbits = (uint64_t)__double_as_longlong(__ull2double_rn(el));
                uint32_t *h = reinterpret_cast<uint32_t *>(m);      // (the frame's image starts 16-byte aligned)
                h[0] = 2u; h[1] = (uint32_t)index; h[2] = (uint32_t)(index >> 32);
                h[3] = (uint32_t)elbits; h[4] = (uint32_t)(elbits >> 32); h[5] = T;
                store_u32_bytes(m + 24u + T, T);
                store_u32_bytes(m + 28u + 2u * T, n64);
                if (p.frame_offsets) p.frame_offsets[f] = (uint64_t)f * p.slot_stride;
                if (p.frame_bytes) p.frame_bytes[f] = (uint64_t)meta + 8ull * n64;
            }
        }
        lds_barrier();
        // The next group's pixels were asked for an iteration ago: waited for HERE, in front of this group's stores (loads and
        // stores share the counter: behind the stores it would be a wait for them too).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ---- every frame leaves as aligned 16-byte blocks, exactly its 32 + 2T + 8 n64 bytes, stored by its own lanes ----
        if (active) {
            const uint32_t len = meta + 8u * n64, n_blocks = (len + 15u) >> 4;
            uint8_t *out = p.out + (uint64_t)f * p.slot_stride;
            const uint8_t *m = s_meta + fl * mpitch;
            const uint8_t *pay = reinterpret_cast<const uint8_t *>(s_buf[cur]) + fl * 64u * T;
            const uint32_t pay0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t *)pay;
            for (uint32_t b = t; b < n_blocks; b += T) {
                const uint32_t o = 16u * b;
                const u32x4_t q = frame_block(m, pay0, pay, meta, o);
                uint8_t *dst = out + o;
                if (o + 16u <= len) {
                    if (DBDE_NT) __builtin_nontemporal_store(q, reinterpret_cast<u32x4_t *>(dst));
                    else *reinterpret_cast<u32x4_t *>(dst) = q;
                } else {   // the frame's last, partial block: its bytes and nothing behind them (lengths are even)
                    const uint32_t n = len - o;
                    uint64_t lo = ((uint64_t)q[1] << 32) | q[0];
                    const uint64_t hi = ((uint64_t)q[3] << 32) | q[2];
                    if (n & 8u) { *reinterpret_cast<uint64_t *>(dst) = lo; dst += 8; lo = hi; }
                    if (n & 4u) { *reinterpret_cast<uint32_t *>(dst) = (uint32_t)lo; dst += 4; lo >>= 32; }
                    if (n & 2u) *reinterpret_cast<uint16_t *>(dst) = (uint16_t)lo;
                }
            }
        }
        lds_barrier();   // the frames have left: both images may be written again; every wave's share of the next pixels has landed
    }
}

hipError_t launch_encode_group(const EncParams &p, uint32_t n_frames, uint32_t n_cu, hipStream_t s) {
    EncParams q = p;
    q.n_chunks = n_frames;
    q.chunks_per_frame = 1u;
    const uint32_t fpw = kGroupThreads / p.T;
    const uint32_t meta_bytes = fpw * ((32u + 2u * p.T + 15u) & ~15u);
    const uint32_t groups = (n_frames + fpw - 1u) / fpw;
    // persistent: as many workgroups as the device holds (n_cu = 0, tests: three, so that small batches walk the loop too)
    // (whole 16-byte blocks everywhere: the instance without the shifted image)
    const bool blocks16 = p.W % 8 == 0 && p.frame_pixels % 16 == 0 && (reinterpret_cast<uintptr_t>(p.images) & 15u) == 0u;
    auto go = [&](auto kernel) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)kGroupThreads, meta_bytes) != hipSuccess || per_cu < 1) per_cu = 1;
        const uint32_t resident = n_cu ? n_cu * (uint32_t)per_cu : 3u;
        hipLaunchKernelGGL(kernel, dim3(groups < resident ? groups : resident), dim3(kGroupThreads), meta_bytes, s, q);
    };
    if (blocks16) go(encode_group_kernel<false>);
    else go(encode_group_kernel<true>);
    return hipGetLastError();
}

// Threads of the workgroup (256: 512 tile slots, 512: 1024).  The larger workgroup only for frames that do not fit the
// smaller one: its barriers span eight waves and two of them fill a CU.  Rounds 3-4a took whichever filled its lanes
// better; measured (profiles/r04b_gain.sh, mixed / incompressible): 90 tiles, eleven frames on 512 threads (97 % full) 0.40 ->
// five on 256 (88 %) 0.50; 110 tiles 0.42 -> 0.49 / 0.415 -> 0.50; 132 tiles (90 % -> 77 %) 0.44 -> 0.49 / 0.41 -> 0.50; 144 tiles
// 0.56 -> 0.60 / 0.51 -> 0.59; 300 tiles (88 % -> 59 %: one frame per workgroup) 0.51 -> 0.51 / 0.48 -> 0.53; never slower.
#ifndef DBDE_FRAMES_512_FROM
#define DBDE_FRAMES_512_FROM 513   // A/B switch: smallest frame (tiles) that takes the 512-thread workgroup
#endif
uint32_t frames_threads_for(uint32_t T) {
    return T >= (unsigned)DBDE_FRAMES_512_FROM ? 512u : 256u;
}

hipError_t launch_encode_frames(const EncParams &p, uint32_t n_frames, hipStream_t s) {
    EncParams q = p;
    q.n_chunks = n_frames;
    q.chunks_per_frame = 1u;
    q.magic_lpr = div_magic_of(p.T);
    const uint32_t th = frames_threads_for(p.T), per_wg = (2u * th) / p.T;
    const dim3 grid((n_frames + per_wg - 1u) / per_wg);
    if (th == 256u) hipLaunchKernelGGL(encode_frames_kernel<256>, grid, dim3(256), 0, s, q);
    else hipLaunchKernelGGL(encode_frames_kernel<512>, grid, dim3(512), 0, s, q);
    return hipGetLastError();
}

// The mirror: as many whole frames as fit the workgroup's tile slots, each frame's bytes landed in LDS where LDS and global
// addresses agree mod 16 (aligned 16-byte blocks; the first and last block of a frame byte by byte where they would reach
// outside the readable extent), validated and unpacked from there, pixels staged in the same memory and written as whole
// 16-byte blocks of the frames' contiguous image range.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void decode_frames_kernel(DecParams p) {
    typedef FramesLds<THREADS> G;
    constexpr int NW = THREADS / 64;
    // stream images of the frames: 32 + 66 T bytes each at most, + 16 of alignment slack and 16 of over-read each; later the pixels
    constexpr uint32_t kBytes = (G::kCap * 66u + (G::kMaxFrames + 1u) * 352u + 255u) & ~255u;   // (+ up to 255 bytes per frame: each image starts a 256-byte swizzle group)
    __shared__ __attribute__((aligned(16))) uint8_t s_buf[kBytes];
    __shared__ uint32_t s_tot[NW];
    __shared__ uint32_t s_fbase[G::kMaxFrames + 1], s_fend[G::kMaxFrames + 1], s_len[G::kMaxFrames + 1], s_ok[G::kMaxFrames + 1], s_at[G::kMaxFrames + 2];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t T = p.T, F = G::kCap / T;
    const uint32_t f0 = blockIdx.x * F;
    const uint32_t nf = p.n_chunks - f0 < F ? p.n_chunks - f0 : F;      // (n_chunks carries the frame count here)
    const uint32_t P = (uint32_t)p.frame_pixels, W = (uint32_t)p.W;
    const uint32_t meta = 32u + 2u * T;
    const uint8_t *s_end = p.stream + p.stream_bytes;

    // ---- how long each frame is (its n64 field), where its image starts in LDS ----
    if (tid < nf) {
        const uint64_t off = p.frame_offsets[f0 + tid];
        uint32_t len = 0, ok = 0;
        if (in_extent(off, meta, p.stream_bytes)) {
            const uint32_t n64 = load_u32_bytes(p.stream + off + 28u + 2u * T);
            if (n64 <= 8u * T && in_extent(off, (uint64_t)meta + 8ull * n64, p.stream_bytes)) { len = meta + 8u * n64; ok = 1u; }
        }
        s_len[tid] = len;       // 0: rejected without looking further (truncated, or a word count no depth array can have)
        s_ok[tid] = ok;
    }
    __syncthreads();
    if (tid == 0) {   // LDS byte where frame g's aligned cover starts: a whole 256-byte swizzle group, room for the shift and the unpack's over-read
        uint32_t at = 0, all8 = 1u;
        for (uint32_t g = 0; g < nf; g++) {
            s_at[g] = at;
            at += (s_len[g] + 15u + 48u + 255u) & ~255u;
            if (s_len[g] && s_len[g] != meta + 64u * T) all8 = 0u;
        }
        s_at[nf] = at;
        s_at[nf + 1u] = all8;
    }
    __syncthreads();
    // Every tile of depth 8: lanes would read the image at a 128-byte stride, all of them on the same banks -- the image is
    // XOR-swizzled at 16-byte granularity then (on the SOURCE side of the DMA, as decode_kernel does); any other content
    // is read at offsets as irregular as its depths and stays linear (the swizzle would only be address arithmetic).
    const bool swz = s_at[nf + 1u] != 0u;
    auto phys = [&](uint32_t a) -> uint32_t { return swz ? swz_byte16(a) : a; };
    // ---- the frames' bytes: aligned 16-byte blocks of global memory, by LDS-DMA, to the same offset mod 16 in LDS ----
    for (uint32_t g = 0; g < nf; g++) {
        const uint32_t len = s_len[g];
        if (!len) continue;
        const uint8_t *src = p.stream + p.frame_offsets[f0 + g];
        const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 15u);
        const uint8_t *a0 = src - sh;
        const uint32_t slot0 = s_at[g] >> 4;                      // (a multiple of 16 slots: groups never straddle frames)
        const uint32_t n16 = (sh + len + 15u) >> 4, n16r = (n16 + 15u) & ~15u;
        for (uint32_t i = tid; i < n16r; i += THREADS) {
            const uint32_t li = swz ? swz16(slot0 + i) - slot0 : i;   // physical slot i of the frame's image <- logical slot li (the permutation is on ABSOLUTE slots and stays inside a group of 16)
            const uint8_t *a = a0 + 16ull * li;
            if (li < n16 && a >= p.stream && a + 16 <= s_end)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)a,
                                                 (__attribute__((address_space(3))) void *)(s_buf + 16u * (slot0 + i - lane)), 16, 0, DBDE_NT ? 2 : 0);
        }
        // the (at most two) blocks that reach outside the readable extent: their inside bytes only
        if (tid < 32u) {
            const uint32_t li = tid < 16u ? 0u : n16 - 1u, b = tid & 15u;
            const uint8_t *a = a0 + 16ull * li;
            if ((a < p.stream || a + 16 > s_end) && (tid < 16u || n16 > 1u))
                s_buf[phys(16u * (slot0 + li) + b)] = (a + b >= p.stream && a + b < s_end) ? a[b] : (uint8_t)0;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- two tiles per lane: depth, minimum; offsets and the reference's validation by a scan over the depths ----
    uint32_t d[2], mnv[2], fl[2], tt[2], fimg[2];
    bool has[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const uint32_t t = 2u * tid + (uint32_t)k;
        has[k] = t < nf * T;
        uint32_t rem;
        fl[k] = has[k] ? div_magic(t, T, p.magic_W, rem) : 0u;        // (magic_W carries floor(2^32 / T) here)
        tt[k] = has[k] ? rem : 0u;
        has[k] = has[k] && s_ok[fl[k]] != 0u;
        const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(p.stream + p.frame_offsets[f0 + fl[k]]) & 15u);
        fimg[k] = s_at[fl[k]] + sh;                                     // LDS byte of the frame's first byte
        d[k] = has[k] ? s_buf[phys(fimg[k] + 24u + tt[k])] : 0u;
        mnv[k] = has[k] ? s_buf[phys(fimg[k] + 28u + T + tt[k])] : 0u;
    }
    // payload words (low 20 bits) and "a depth above 8" count (above them) in one scan
    const uint32_t item0 = d[0] | (d[0] > 8u ? 1u << 20 : 0u), item1 = d[1] | (d[1] > 8u ? 1u << 20 : 0u);
    uint32_t block_total;
    const uint32_t incl = block_scan_incl<NW>(item0 + item1, s_tot, (int)lane, (int)wave, block_total);
    const uint32_t excl[2] = {incl - item0 - item1, incl - item1};
    const uint32_t item[2] = {item0, item1};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (has[k] && tt[k] == 0u) s_fbase[fl[k]] = excl[k];
        if (has[k] && tt[k] == T - 1u) s_fend[fl[k]] = excl[k] + item[k];
    }
    __syncthreads();
    // the verdict of every frame (dbde_util.cpp:295-303, + depth <= 8), its result record
    if (tid < nf) {
        const uint64_t off = p.frame_offsets[f0 + tid];
        bool ok = s_ok[tid] != 0u;
        uint32_t total = 0;
        if (ok) {
            const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(p.stream + off) & 15u);
            const uint32_t fb = s_at[tid] + sh;
            auto rd32 = [&](uint32_t a) -> uint32_t {
                return (uint32_t)s_buf[phys(fb + a)] | ((uint32_t)s_buf[phys(fb + a + 1u)] << 8) | ((uint32_t)s_buf[phys(fb + a + 2u)] << 16) |
                       ((uint32_t)s_buf[phys(fb + a + 3u)] << 24);
            };
            const uint32_t span = s_fend[tid] - s_fbase[tid];
            total = span & 0xFFFFFu;
            const int32_t nb = (int32_t)rd32(20u), nm = (int32_t)rd32(24u + T), n64 = (int32_t)rd32(28u + 2u * T);
            ok = nb == (int32_t)T && nm == (int32_t)T && n64 == (int32_t)total && (span >> 20) == 0u;
        }
        s_ok[tid] = ok ? 1u : 0u;
        if (p.results) {
            uint32_t field = 0;
            uint64_t index = 0, elapsed = 0;
            if (in_extent(off, 20, p.stream_bytes)) {
                const uint8_t *fb = p.stream + off;
                field = load_u32_bytes(fb);
                index = load_u64_bytes(fb + 4);
                elapsed = f64_to_u64_x86(__longlong_as_double((long long)load_u64_bytes(fb + 12)));
            }
            FrameResultDev *r = reinterpret_cast<FrameResultDev *>(p.results) + f0 + tid;
            r->u64s = (field == 2u && ok) ? 2u : 0xFFFFFFFFu;   // dbde_util.cpp:335,342
            r->pad_ = 0;
            r->index = index;
            r->elapsed_ns = elapsed;
            r->consumed = ok ? (uint64_t)meta + 8ull * total : 20ull;
        }
    }
    __syncthreads();

    // ---- unpack (a tile row is the 8 d-bit integer at byte r d of the tile's payload) ----
    uint32_t v[2][16];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        has[k] = has[k] && s_ok[fl[k]] != 0u;
        const uint32_t words = (excl[k] - s_fbase[fl[k]]) & 0xFFFFFu;
        if (has[k]) {
            if (swz) unpack_tile_from_lds<true>(s_buf, fimg[k] + meta + 8u * words, 8u, mnv[k], v[k]);
            else unpack_tile_from_lds<false>(s_buf, fimg[k] + meta + 8u * words, d[k] > 8u ? 8u : d[k], mnv[k], v[k]);
        }
    }
    __syncthreads();   // every tile is in registers: the memory changes hands

    // ---- the pixels into the frames' image (aligned 8-byte rows), then out as whole 16-byte blocks ----
    // (a rejected frame's image stays untouched, dbde_util.cpp:296-303: its blocks are not written)
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (!has[k]) continue;
        const uint32_t ty = tt[k] / p.w, tx = tt[k] - ty * p.w;      // (once per tile: no launch constant kept for it)
        const uint32_t base = fl[k] * P + 8u * tx;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t yy = 8u * ty + (uint32_t)r;
            if (yy < (uint32_t)p.H) *reinterpret_cast<uint2 *>(s_buf + base + yy * W) = make_uint2(v[k][2 * r], v[k][2 * r + 1]);
        }
    }
    __syncthreads();
    for (uint32_t g = 0; g < nf; g++) {
        if (!s_ok[g]) continue;
        u32x4_t *dst = reinterpret_cast<u32x4_t *>(p.images + (size_t)(f0 + g) * P);
        const u32x4_t *src = reinterpret_cast<const u32x4_t *>(s_buf + g * P);
        for (uint32_t i = tid; i < (P >> 4); i += THREADS) {
            if (DBDE_NT) __builtin_nontemporal_store(src[i], dst + i);
            else dst[i] = src[i];
        }
    }
}

hipError_t launch_decode_frames(const DecParams &p, uint32_t n_frames, hipStream_t s) {
    DecParams q = p;
    q.n_chunks = n_frames;
    q.magic_W = div_magic_of(p.T);
    const uint32_t th = frames_threads_for(p.T), per_wg = (2u * th) / p.T;
    const dim3 grid((n_frames + per_wg - 1u) / per_wg);
    if (th == 256u) hipLaunchKernelGGL(decode_frames_kernel<256>, grid, dim3(256), 0, s, q);
    else hipLaunchKernelGGL(decode_frames_kernel<512>, grid, dim3(512), 0, s, q);
    return hipGetLastError();
}

template <int IMG, int THREADS = 256>
static void launch_decode_img(const DecParams &p, int index_mode, dim3 grid, hipStream_t s) {
    const dim3 block(THREADS);
    switch (index_mode) {
        case kIdxTable: hipLaunchKernelGGL((decode_kernel<IMG, kIdxTable, THREADS>), grid, block, 0, s, p); break;
        case kIdxSelf: hipLaunchKernelGGL((decode_kernel<IMG, kIdxSelf, THREADS>), grid, block, 0, s, p); break;
        default: hipLaunchKernelGGL((decode_kernel<IMG, kIdxFused, THREADS>), grid, block, 0, s, p); break;
    }
}
// The workgroup follows the chunk geometry: whole-tile-row chunks of at most 384 tiles (dec_geometry with that capacity)
// run on 192 threads, everything else on 256.
hipError_t launch_decode(const DecParams &p, int img_mode, int index_mode, hipStream_t s) {
    dim3 grid(p.n_chunks);
    switch (img_mode) {
        case kImgDirect: launch_decode_img<kImgDirect>(p, index_mode, grid, s); break;
        case kImgLinear:
            if (p.geom.pieces == 1u && p.geom.ct <= kChunkTilesSmall) launch_decode_img<kImgLinear, kChunkTilesSmall / 2>(p, index_mode, grid, s);
            else launch_decode_img<kImgLinear>(p, index_mode, grid, s);
            break;
        default: launch_decode_img<kImgTiles>(p, index_mode, grid, s); break;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// stream scanner: frame-to-frame hop (reference README.md:12-23: sizes are only in-band)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ld_u32_any(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// One exact hop: the frame at `off` (length in *len) if its fixed part and its payload lie inside the stream.
__device__ __forceinline__ bool frame_at(const uint8_t *stream, uint64_t stream_bytes, uint64_t meta, uint32_t T, uint64_t off, uint64_t *len) {
    if (!in_extent(off, meta, stream_bytes)) return false;
    const uint32_t n64 = ld_u32_any(stream + off + 28 + 2ull * T);
    *len = meta + 8ull * n64;
    return (int32_t)n64 >= 0 && in_extent(off, *len, stream_bytes);
}

// One lane walks the chain (each hop needs the previous frame's word count: a dependent read of HBM, about a
// microsecond); `cursor`, when given, is where the walk starts and where it is left, so a stream can be walked
// a batch at a time while the previous batch is being decoded (dbde_hip_scan_ahead).
__global__ void scan_stream_kernel(const uint8_t *stream, uint64_t stream_bytes, uint32_t T, int max_frames,
                                   uint64_t *offsets, uint32_t *count, uint64_t *cursor) {
    __shared__ int s_n;
    if (blockIdx.x != 0) return;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_s_setprio(3);
        uint64_t off = cursor ? *cursor : 0ull;
        int n = 0;
        const uint64_t meta = 32ull + 2ull * T;
        uint64_t len;
        while (n < max_frames && frame_at(stream, stream_bytes, meta, T, off, &len)) {
            offsets[n++] = off;
            off += len;
        }
        *count = (uint32_t)n;
        if (cursor) *cursor = off;
        s_n = n;
    }
    __syncthreads();
    // entries past the count: an offset no frame can have (every extent check rejects it, see in_extent), so a decode
    // bounded by max_frames instead of the count reports those frames as failed rather than decoding stale offsets
    for (int i = s_n + (int)threadIdx.x; i < max_frames; i += (int)blockDim.x) offsets[i] = ~0ull;
}

hipError_t launch_scan_stream(const uint8_t *stream, uint64_t stream_bytes, uint32_t T, int max_frames,
                              uint64_t *d_offsets, uint32_t *d_count, uint64_t *d_cursor, hipStream_t s) {
    hipLaunchKernelGGL(scan_stream_kernel, dim3(1), dim3(64), 0, s, stream, stream_bytes, T, max_frames,
                       d_offsets, d_count, d_cursor);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// stream scanner, speculative form
// ---------------------------------------------------------------------------------------
// The walk is a chain of dependent HBM reads (0.4 us per frame alone, 1 us beside other traffic): 1000 frames cost
// as much as decoding them.  But next(p) = p + 32 + 2T + 8 * n64(p) is a function of the bytes at p alone, so the
// continuation of the chain from ANY true frame start is what the exact walk would compute there.  The stream is cut
// into K segments; workgroup j looks for the first position at or after its segment's first byte that LOOKS like a
// frame start (u32 2 | ... | u32 T at +20 | u32 T at +24+T -- the fields dbde_unpack_frame_header / dbde_unpack_image
// check, dbde_util.cpp:295-300,335) and walks from there to the first frame start at or past the next segment.  All K
// walks run at once.  Nothing is trusted: the stitch pass follows the exact chain from byte 0 and accepts segment
// j only if the chain ARRIVES at the very position that segment started from (then its positions are the exact
// walk's); a segment that was fooled by payload bytes is simply never arrived at, and the exact walk goes on hop by
// hop until it meets the next segment start that it does arrive at.  Exact by construction, K-fold shorter in the
// usual case.
__global__ __launch_bounds__(1024) void scan_spec_kernel(ScanParams p) {
    __shared__ uint32_t s_last;
    const uint32_t M = p.wg_per_seg, j = blockIdx.x / M, m = blockIdx.x - j * M, tid = threadIdx.x;
    const uint32_t n_seg = gridDim.x / M;
    const uint64_t meta = 32ull + 2ull * p.T;
    const uint64_t q_lo = j == 0 ? 0ull : ((uint64_t)j * p.seg_bytes + p.gran - 1) / p.gran * p.gran;
    const uint64_t q_hi = j + 1 == n_seg ? p.stream_bytes : (((uint64_t)(j + 1) * p.seg_bytes + p.gran - 1) / p.gran * p.gran);
    uint64_t start = 0;
    if (j != 0) {
        // First position >= q_lo that carries the three fields of a frame start.  The segment's M workgroups take
        // the candidate rounds in turn; the best match so far lives in a device word as its COMPLEMENT (atomicMax on
        // a zero-initialised word = min of the positions); a workgroup stops once it is past it (it only moves
        // down, so nothing below the final minimum is ever skipped).  The last workgroup to arrive walks.
        const uint64_t limit = q_lo + 32ull + 66ull * p.T < p.stream_bytes ? q_lo + 32ull + 66ull * p.T : p.stream_bytes;   // a frame starts within one maximal frame
        const uint64_t round_bytes = 16384ull * p.gran;
        for (uint64_t base = q_lo + m * round_bytes; base < limit; base += M * round_bytes) {
            if (base > ~__hip_atomic_load(&p.seg_found_inv[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            uint32_t w[16];
#pragma unroll
            for (uint32_t k = 0; k < 16u; k++) {
                const uint64_t c = base + ((uint64_t)k * 1024u + tid) * p.gran;
                w[k] = in_extent(c, meta, p.stream_bytes) ? ld_u32_any(p.stream + c) : 0u;
            }
#pragma unroll
            for (uint32_t k = 0; k < 16u; k++) {
                if (w[k] == 2u) {
                    const uint64_t c = base + ((uint64_t)k * 1024u + tid) * p.gran;
                    if (ld_u32_any(p.stream + c + 20) == p.T && ld_u32_any(p.stream + c + 24 + p.T) == p.T)
                        atomicMax(&p.seg_found_inv[j], (unsigned long long)~c);
                }
            }
        }
        // the matches were published with device-scope atomics (performed at the memory side): once every wave has
        // seen its own acknowledged, the arrival may be counted -- no fence (an agent-scope fence from 1024 threads
        // of every workgroup cost more than the search)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) s_last = atomicAdd(&p.seg_arrive[j], 1u) == M - 1u ? 1u : 0u;
        __syncthreads();
        if (!s_last) return;
        if (tid == 0) {
            start = ~__hip_atomic_load(&p.seg_found_inv[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p.seg_found_inv[j] = 0;   // the workspace stays clean for the next call
            p.seg_arrive[j] = 0;
        }
    } else if (m != 0) {
        return;
    }
    if (tid != 0) return;
    uint64_t *pos = p.seg_pos + (size_t)j * p.seg_cap;
    uint32_t n = 0;
    uint64_t off = start, len = 0;
    uint32_t ended = 0;   // 1: the chain stops inside this segment (no further whole frame)
    if (start != ~0ull) {
        while (off < q_hi) {
            if (!frame_at(p.stream, p.stream_bytes, meta, p.T, off, &len)) { ended = 1; break; }
            if (n < p.seg_cap) pos[n] = off;
            n++;
            off += len;
        }
    }
    p.seg_start[j] = start;
    p.seg_end[j] = off;
    p.seg_count[j] = n <= p.seg_cap ? n : ~0u;   // overflow of the temporary list: treat the segment as unusable
    p.seg_ended[j] = ended;
}

__global__ __launch_bounds__(1024) void scan_stitch_kernel(ScanParams p, uint32_t n_seg, int max_frames, uint64_t *offsets, uint32_t *count) {
    __shared__ unsigned long long s_start[kMaxScanSegments], s_end[kMaxScanSegments];
    __shared__ uint32_t s_count[kMaxScanSegments], s_ended[kMaxScanSegments], s_base[kMaxScanSegments], s_use[kMaxScanSegments];
    __shared__ uint32_t s_total;
    const uint32_t tid = threadIdx.x;
    if (tid < n_seg) { s_start[tid] = p.seg_start[tid]; s_end[tid] = p.seg_end[tid]; s_count[tid] = p.seg_count[tid]; s_ended[tid] = p.seg_ended[tid]; s_use[tid] = 0; }
    __syncthreads();
    if (tid == 0) {
        const uint64_t meta = 32ull + 2ull * p.T;
        uint64_t cur = 0, len = 0;
        uint32_t total = 0, j = 0;
        bool done = false;
        while (!done && total < (uint32_t)max_frames) {
            // drop segments that can never be arrived at: already passed, found no start (~0), or overflowed their list
            while (j < n_seg && (s_start[j] < cur || s_start[j] == ~0ull || s_count[j] == ~0u)) j++;
            if (j < n_seg && s_start[j] == cur) {                        // the exact chain arrives at this segment's start
                s_use[j] = 1; s_base[j] = total;
                total += s_count[j];
                cur = s_end[j];
                done = s_ended[j] != 0u;
                j++;
            } else {                                                     // hop by hop until it does (or the stream ends)
                if (!frame_at(p.stream, p.stream_bytes, meta, p.T, cur, &len)) { done = true; break; }
                offsets[total++] = cur;
                cur += len;
            }
        }
        s_total = total < (uint32_t)max_frames ? total : (uint32_t)max_frames;
        *count = s_total;
    }
    __syncthreads();
    const uint32_t total = s_total;
    for (uint32_t j = tid >> 6; j < n_seg; j += 16u) {   // a wave per segment: the copies of different segments overlap
        if (!s_use[j]) continue;
        const uint64_t *pos = p.seg_pos + (size_t)j * p.seg_cap;
        for (uint32_t i = tid & 63u; i < s_count[j]; i += 64u)
            if (s_base[j] + i < total) offsets[s_base[j] + i] = pos[i];
    }
    for (uint32_t i = total + tid; i < (uint32_t)max_frames; i += 1024u) offsets[i] = ~0ull;   // as scan_stream_kernel
}

hipError_t launch_scan_spec(const ScanParams &p, uint32_t n_seg, int max_frames, uint64_t *d_offsets, uint32_t *d_count, hipStream_t s) {
    hipLaunchKernelGGL(scan_spec_kernel, dim3(n_seg * p.wg_per_seg), dim3(1024), 0, s, p);
    hipLaunchKernelGGL(scan_stitch_kernel, dim3(1), dim3(1024), 0, s, p, n_seg, max_frames, d_offsets, d_count);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// synthetic frames (same function as oracle/synth.c)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void synth_kernel(int mode, uint64_t seed, uint64_t first_frame, int n_frames,
                                                    int W, int H, uint8_t *images) {
    const uint32_t xb_per_row = (uint32_t)((W + 7) >> 3);
    const uint64_t per_frame = (uint64_t)xb_per_row * (uint64_t)H;
    const uint64_t total = per_frame * (uint64_t)n_frames;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t fi = g / per_frame;
        const uint64_t rem = g - fi * per_frame;
        const uint32_t y = (uint32_t)(rem / xb_per_row);
        const uint32_t xb = (uint32_t)(rem - (uint64_t)y * xb_per_row);
        const uint64_t frame = first_frame + fi;
        const uint64_t rk = mix64(seed ^ (frame << 42) ^ ((uint64_t)y << 21) ^ (uint64_t)xb);
        uint64_t out = 0;
        if (mode == 0) {
            out = rk;
        } else if (mode == 1) {
            const uint64_t tk = mix64(~seed ^ (frame << 42) ^ ((uint64_t)(y >> 3) << 21) ^ (uint64_t)xb);
            const uint32_t d = (uint32_t)(tk % 9u);
            const uint32_t m = (uint32_t)((tk >> 32) % (257u - (1u << d)));
            const uint32_t top = (1u << d) - 1u;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint32_t pix = m + ((uint32_t)(rk >> (8 * k)) & top);
                if ((y & 7u) == 0u && k == 0) pix = m;
                if ((y & 7u) == 0u && k == 1) pix = m + top;
                out |= (uint64_t)(pix & 0xFFu) << (8 * k);
            }
        } else if (mode == 2) {
            out = (seed & 0xFFull) * 0x0101010101010101ull;
        } else if (mode >= 4) {   // every tile of depth mode - 4 (0..8): the regular lane strides of uniform content
            const uint32_t d = (uint32_t)(mode - 4) > 8u ? 8u : (uint32_t)(mode - 4);
            const uint32_t top = (1u << d) - 1u, m = 100u > 255u - top ? 255u - top : 100u;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint32_t pix = m + ((uint32_t)(rk >> (8 * k)) & top);
                if ((y & 7u) == 0u && k == 0) pix = m;
                if ((y & 7u) == 0u && k == 1) pix = m + top;
                out |= (uint64_t)(pix & 0xFFu) << (8 * k);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t x = 8u * xb + (uint32_t)k;
                const uint32_t pix = ((x >> 4) + (y >> 5) + (uint32_t)(frame & 15u) + ((uint32_t)(rk >> (8 * k)) & 7u)) & 0xFFu;
                out |= (uint64_t)pix << (8 * k);
            }
        }
        uint8_t *dst = images + fi * (uint64_t)W * (uint64_t)H + (uint64_t)y * (uint64_t)W + 8ull * xb;
        const int valid = W - 8 * (int)xb < 8 ? W - 8 * (int)xb : 8;
        if (valid == 8 && (reinterpret_cast<uintptr_t>(dst) & 7u) == 0) {
            *reinterpret_cast<uint64_t *>(dst) = out;
        } else {
            for (int k = 0; k < valid; k++) dst[k] = (uint8_t)(out >> (8 * k));
        }
    }
}

hipError_t launch_synth(int mode, uint64_t seed, uint64_t first_frame, int n_frames, int W, int H,
                        uint8_t *d_images, hipStream_t s) {
    const uint64_t total = (uint64_t)((W + 7) >> 3) * (uint64_t)H * (uint64_t)n_frames;
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)blocks), dim3(256), 0, s, mode, seed, first_frame, n_frames, W, H,
                       d_images);
    return hipGetLastError();
}

}  // namespace dbde
