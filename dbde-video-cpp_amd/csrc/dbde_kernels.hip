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

namespace dbde {

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

// Inclusive scan of one value per lane across the 256-thread block.
// Returns this lane's inclusive prefix; block_total = sum over the block.
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
    uint32_t t0 = s_wave_tot[0], t1 = s_wave_tot[1], t2 = s_wave_tot[2], t3 = s_wave_tot[3];
    uint32_t base = (wave > 0 ? t0 : 0u) + (wave > 1 ? t1 : 0u) + (wave > 2 ? t2 : 0u);
    block_total = t0 + t1 + t2 + t3;
    return base + incl;
}

// LDS staging images are XOR-swizzled at 16-byte slot granularity so that the regular lane
// strides of uniform-depth content (lane stride = 2*depth qwords; 128 B for depth 8) do not
// pile onto one bank.  Both are involutions that only permute slots inside an aligned group.
__device__ __forceinline__ uint32_t swz8(uint32_t slot) { return slot ^ ((slot >> 3) & 7u); }      // 128-B groups
__device__ __forceinline__ uint32_t swz16(uint32_t slot) { return slot ^ ((slot >> 4) & 15u); }    // 256-B groups
// qword index -> swizzled qword index (the half inside the slot is kept)
__device__ __forceinline__ uint32_t swzq8(uint32_t q) { return (swz8(q >> 1) << 1) | (q & 1u); }
__device__ __forceinline__ uint32_t swzq16(uint32_t q) { return (swz16(q >> 1) << 1) | (q & 1u); }

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

// Subtract the minimum, pack each row to 8*d bits, concatenate rows into d U64 words in LDS.
__device__ __forceinline__ void pack_tile_to_lds(const uint32_t (&v)[16], uint32_t mn, uint32_t d,
                                                 uint64_t *s_out, uint32_t q) {
    const uint32_t m4 = mn * 0x01010101u;   // every byte >= mn: no borrow crosses a byte
    Funnel fn;
    fn.reset();
#pragma unroll
    for (int r = 0; r < 8; r++) {
        uint64_t row = pack_row(v[2 * r] - m4, v[2 * r + 1] - m4, d);
        uint64_t word;
        if (fn.push(row, 8u * d, word)) { s_out[swzq8(q)] = word; q++; }
    }
}

// Decoupled look-back executed by one full wave.  Walks 64 predecessors at a time, nearest
// first.  Chunks below lo_bound do not exist (prefix 0).  Returns false on POISON / time-out.
__device__ __forceinline__ bool lookback(const u64a *state, uint32_t c, uint32_t frame_first,
                                         uint32_t lo_bound, int lane, uint32_t &inframe_excl,
                                         uint32_t &global_excl) {
    uint32_t inf = 0, glob = 0;
    long long top = (long long)c - 1;
    const uint64_t t_start = wall_clock64();
    for (;;) {
        const long long idx = top - lane;
        const bool real = idx >= (long long)lo_bound;
        uint32_t spins = 0;
        for (;;) {
            u64a word = kStInc;   // virtual predecessor: inclusive prefix 0
            if (real) word = __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t st = (uint32_t)(word >> 62);
            const uint64_t incm = __ballot(st == 2u);
            const uint64_t rdy = __ballot(st != 0u);
            const uint64_t bad = __ballot(st == 3u);
            const uint64_t low = incm & (0ull - incm);               // nearest INC lane, as a bit
            const uint64_t need = incm ? ((low << 1) - 1ull) : ~0ull;  // lanes 0..first INC
            if (bad & need) return false;
            if ((rdy & need) == need) {
                const int first = incm ? (__ffsll((long long)incm) - 1) : 64;
                uint32_t g = 0, i = 0;
                const bool same_frame = real && idx >= (long long)frame_first;
                if (lane < first) {          // AGG records
                    g = (uint32_t)word;
                    i = same_frame ? g : 0u;
                } else if (lane == first) {  // the INC record that ends the walk
                    g = (uint32_t)word;
                    i = same_frame ? (uint32_t)((word >> 32) & 0x3FFFFFFFull) : 0u;
                }
                glob += wave_sum(g);
                inf += wave_sum(i);
                if (incm) {
                    inframe_excl = inf;
                    global_excl = glob;
                    return true;
                }
                top -= 64;
                break;
            }
            if (++spins > 64u && (wall_clock64() - t_start) > 200000000ull) return false;   // 2 s at 100 MHz
            __builtin_amdgcn_s_sleep(2);
        }
    }
}

template <bool FAST_IN, bool ALIGNED_OUT>
__global__ __launch_bounds__(kBlockThreads) void encode_kernel(EncParams p) {
    __shared__ __attribute__((aligned(16))) uint64_t s_out[kMaxChunkWords];   // 32 KiB payload staging
    __shared__ uint32_t s_wave_tot[4];
    __shared__ uint32_t s_bcast[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // Chunks are claimed in ticket order, so every predecessor of a claimed chunk has been
    // claimed by a workgroup that is already running: look-back cannot wait on unstarted work.
    uint32_t c;
    if (p.flags & 1u) {   // EXPERIMENT ONLY: dispatch order taken as chunk order (not contract-safe)
        c = blockIdx.x;
    } else {
        if (tid == 0) s_bcast[0] = atomicAdd(&p.ctrl[0], 1u);
        __syncthreads();
        c = s_bcast[0];
    }
    if (c >= p.n_chunks) return;
    const uint32_t f = c / p.chunks_per_frame;
    const uint32_t cf = c - f * p.chunks_per_frame;
    const uint32_t t0 = cf * kChunkTiles + 2u * (uint32_t)tid;
    const bool hasA = t0 < p.T, hasB = t0 + 1u < p.T;
    const uint8_t *img = p.images + (size_t)f * p.frame_pixels;

    // ---- load two tiles -----------------------------------------------------------------
    uint32_t va[16], vb[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { va[i] = 0; vb[i] = 0; }
    if (FAST_IN) {   // W % 16 == 0, base 16-aligned: both tiles in one strip, one 16-B load per row
        if (hasA) {
            const uint32_t ty = t0 / p.w, tx = t0 - ty * p.w;
            const uint8_t *base = img + (size_t)(8u * tx);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                int yy = 8 * (int)ty + r;
                yy = yy < p.H ? yy : p.H - 1;   // bottom padding = repeat the last row
                const uint4 q = *reinterpret_cast<const uint4 *>(base + (size_t)yy * (size_t)p.W);
                va[2 * r] = q.x; va[2 * r + 1] = q.y;
                vb[2 * r] = q.z; vb[2 * r + 1] = q.w;
            }
        }
    } else {
        if (hasA) load_tile_generic(img, p.W, p.H, p.w, t0, va);
        if (hasB) load_tile_generic(img, p.W, p.H, p.w, t0 + 1u, vb);
    }

    // ---- per-tile statistics (dbde_util.cpp:30-68) ---------------------------------------
    uint32_t mnA, mxA, mnB, mxB;
    tile_minmax(va, mnA, mxA);
    tile_minmax(vb, mnB, mxB);
    const uint32_t dA = hasA ? depth_of_range(mxA - mnA) : 0u;
    const uint32_t dB = hasB ? depth_of_range(mxB - mnB) : 0u;

    // ---- offsets inside the chunk ---------------------------------------------------------
    uint32_t chunk_total;
    const uint32_t incl = block_scan_incl(dA + dB, s_wave_tot, lane, wave, chunk_total);
    const uint32_t offA = incl - (dA + dB), offB = offA + dA;

    const bool slot_mode = p.slot_stride != 0;
    const uint32_t frame_first = f * p.chunks_per_frame;
    const bool is_head = slot_mode ? (cf == 0u) : (c == 0u);
    if (tid == 0) {
        const u64a rec = is_head ? (kStInc | ((u64a)chunk_total << 32) | (u64a)chunk_total)
                                 : (kStAgg | (u64a)chunk_total);
        __hip_atomic_store(&p.state[c], rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- pack into LDS (independent of the look-back) -------------------------------------
    pack_tile_to_lds(va, mnA, dA, s_out, offA);
    pack_tile_to_lds(vb, mnB, dB, s_out, offB);

    // ---- chunk offset inside the frame and the launch -------------------------------------
    if (wave == 0) {
        uint32_t inf = 0, glob = 0;
        bool ok = true;
        if (!is_head) ok = lookback(p.state, c, frame_first, slot_mode ? frame_first : 0u, lane, inf, glob);
        if (lane == 0) {
            if (!ok) {
                __hip_atomic_store(&p.state[c], kStPoison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicOr(p.sticky, 1u);
            } else if (!is_head) {
                const u64a rec = kStInc | ((u64a)((inf + chunk_total) & 0x3FFFFFFFu) << 32) |
                                 (u64a)(uint32_t)(glob + chunk_total);
                __hip_atomic_store(&p.state[c], rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s_bcast[1] = inf;
            s_bcast[2] = glob;
            s_bcast[3] = ok ? 1u : 0u;
        }
    }
    __syncthreads();
    if (!s_bcast[3]) return;
    const uint32_t inf = s_bcast[1], glob = s_bcast[2];

    // ---- addresses (dbde_util.cpp:137-146 layout) ------------------------------------------
    const uint64_t meta = 32ull + 2ull * p.T;   // frame header + three I32 + two byte arrays
    const uint64_t frame_base = slot_mode ? (uint64_t)f * p.slot_stride
                                          : (uint64_t)f * meta + 8ull * (uint64_t)(uint32_t)(glob - inf);
    uint8_t *fb = p.out + frame_base;
    uint8_t *depth_arr = fb + 24;
    uint8_t *min_arr = fb + 28 + p.T;

    if (ALIGNED_OUT) {   // fb % 8 == 0 and T % 4 == 0: t0 is even, so both arrays are 2-aligned here
        if (hasB) {
            *reinterpret_cast<uint16_t *>(depth_arr + t0) = (uint16_t)(dA | (dB << 8));
            *reinterpret_cast<uint16_t *>(min_arr + t0) = (uint16_t)(mnA | (mnB << 8));
        } else if (hasA) {
            depth_arr[t0] = (uint8_t)dA;
            min_arr[t0] = (uint8_t)mnA;
        }
    } else {
        if (hasA) { depth_arr[t0] = (uint8_t)dA; min_arr[t0] = (uint8_t)mnA; }
        if (hasB) { depth_arr[t0 + 1] = (uint8_t)dB; min_arr[t0 + 1] = (uint8_t)mnB; }
    }

    // ---- payload: LDS -> global, 16 B per lane ---------------------------------------------
    uint8_t *dst = fb + meta + 8ull * inf;
    if (ALIGNED_OUT) {
        const uint32_t q0 = (uint32_t)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1u);   // 1: dst is 8 mod 16
        const uint32_t lead = q0 < chunk_total ? q0 : chunk_total;
        if (lead && tid == 0) *reinterpret_cast<uint64_t *>(dst) = s_out[swzq8(0)];
        const uint32_t rest = chunk_total - lead;
        const uint32_t npairs = rest >> 1;
        for (uint32_t i = tid; i < npairs; i += kBlockThreads) {
            const uint32_t k = lead + 2u * i;
            ulonglong2 v2;
            v2.x = s_out[swzq8(k)];
            v2.y = s_out[swzq8(k + 1u)];
            *reinterpret_cast<ulonglong2 *>(dst + 8ull * k) = v2;
        }
        if ((rest & 1u) && tid == kBlockThreads - 1) {
            const uint32_t k = chunk_total - 1u;
            *reinterpret_cast<uint64_t *>(dst + 8ull * k) = s_out[swzq8(k)];
        }
    } else {
        for (uint32_t k = tid; k < chunk_total; k += kBlockThreads) store_u64_any(dst + 8ull * k, s_out[swzq8(k)]);
    }

    // ---- frame header and the I32 fields (dbde_util.cpp:140-146, 182-196) -------------------
    if (tid == 0) {
        if (cf == 0u) {
            const uint64_t index = p.indices ? p.indices[f] : p.first_index + f;
            const uint64_t el = p.elapsed_ns ? p.elapsed_ns[f] : 0ull;
            const uint64_t elbits = (uint64_t)__double_as_longlong(__ull2double_rn(el));   // trap T1: F64 on the wire
            if (ALIGNED_OUT) {
                uint32_t *h = reinterpret_cast<uint32_t *>(fb);
                h[0] = 2u;
                h[1] = (uint32_t)index; h[2] = (uint32_t)(index >> 32);
                h[3] = (uint32_t)elbits; h[4] = (uint32_t)(elbits >> 32);
                h[5] = p.T;
                *reinterpret_cast<uint32_t *>(fb + 24 + p.T) = p.T;
            } else {
                store_u32_bytes(fb, 2u);
                store_u32_bytes(fb + 4, (uint32_t)index); store_u32_bytes(fb + 8, (uint32_t)(index >> 32));
                store_u32_bytes(fb + 12, (uint32_t)elbits); store_u32_bytes(fb + 16, (uint32_t)(elbits >> 32));
                store_u32_bytes(fb + 20, p.T);
                store_u32_bytes(fb + 24 + p.T, p.T);
            }
            if (p.frame_offsets) p.frame_offsets[f] = frame_base;
        }
        if (cf == p.chunks_per_frame - 1u) {
            const uint32_t n64 = inf + chunk_total;
            if (ALIGNED_OUT) *reinterpret_cast<uint32_t *>(fb + 28 + 2ull * p.T) = n64;
            else store_u32_bytes(fb + 28 + 2ull * p.T, n64);
            if (p.frame_bytes) p.frame_bytes[f] = meta + 8ull * n64;
        }
    }
}

hipError_t launch_encode(const EncParams &p, bool fast_in, bool aligned_out, hipStream_t s) {
    dim3 grid(p.n_chunks), block(kBlockThreads);
    if (fast_in && aligned_out) hipLaunchKernelGGL((encode_kernel<true, true>), grid, block, 0, s, p);
    else if (fast_in) hipLaunchKernelGGL((encode_kernel<true, false>), grid, block, 0, s, p);
    else if (aligned_out) hipLaunchKernelGGL((encode_kernel<false, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((encode_kernel<false, false>), grid, block, 0, s, p);
    return hipGetLastError();
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
    const uint64_t need = 32ull + 2ull * T;   // header + metadata must lie inside the stream
    const bool in_range = off + need <= p.stream_bytes;
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
        for (uint32_t i = tid; i < npieces; i += blockDim.x) {
            const uint4 q = *reinterpret_cast<const uint4 *>(a_lo + 16ull * i);
            const uint32_t wv[4] = {q.x, q.y, q.z, q.w};
            // position of this piece's byte 0 relative to the array start
            const long long pos0 = 16ll * i - (long long)head;
            uint32_t sum_lo = 0, sum_hi = 0;
            const uint32_t k_lo = (uint32_t)((pos0 < 0 ? 0 : pos0) >> 9);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t x = wv[j];
                const long long pj = pos0 + 4 * j;
                // keep only bytes with 0 <= position < T
                uint32_t mask = 0xFFFFFFFFu;
                if (pj < 0) mask = (pj <= -4) ? 0u : (0xFFFFFFFFu << (8 * (int)(-pj)));
                if (pj + 4 > (long long)T) {
                    const long long keep = (long long)T - pj;   // bytes to keep
                    mask &= keep <= 0 ? 0u : (keep >= 4 ? 0xFFFFFFFFu : (0xFFFFFFFFu >> (8 * (int)(4 - keep))));
                }
                x &= mask;
                bad_depth |= (x & 0xF0F0F0F0u) | ((x + 0x77777777u) & 0x80808080u);   // any byte > 8
                // a dword belongs to one chunk unless it straddles a 512-byte boundary of the array
                const long long pc = pj < 0 ? 0 : pj;
                const uint32_t kj = (uint32_t)(pc >> 9);
                const uint32_t kend = (uint32_t)((pj + 3 < 0 ? 0 : pj + 3) >> 9);
                if (kj == kend) {
                    const uint32_t sb = __builtin_amdgcn_sad_u8(x, 0u, 0u);
                    if (kj == k_lo) sum_lo += sb; else sum_hi += sb;
                } else {   // split the dword at the boundary byte by byte
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const long long pb = pj + b;
                        const uint32_t v = (x >> (8 * b)) & 0xFFu;
                        if (pb >= 0) { if ((uint32_t)(pb >> 9) == k_lo) sum_lo += v; else sum_hi += v; }
                    }
                }
            }
            if (sum_lo) atomicAdd(&s_sum[k_lo], sum_lo);
            if (sum_hi && k_lo + 1u < cpf) atomicAdd(&s_sum[k_lo + 1u], sum_hi);
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
        if (off + 20 <= p.stream_bytes) {
            field = load_u32_bytes(fb);
            index = load_u64_bytes(fb + 4);
            elapsed = f64_to_u64_x86(__longlong_as_double((long long)load_u64_bytes(fb + 12)));
        }
        if (ok) {
            const int32_t nb = (int32_t)load_u32_bytes(fb + 20);
            const int32_t nm = (int32_t)load_u32_bytes(fb + 24 + T);
            const int32_t n64 = (int32_t)load_u32_bytes(fb + 28 + 2ull * T);
            ok = nb == (int32_t)T && nm == (int32_t)T && n64 == (int32_t)total && !(s_flags & 1u);
            // the payload itself must also be inside the stream
            if (ok && off + need + 8ull * total > p.stream_bytes) ok = false;
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
    hipLaunchKernelGGL(decode_index_kernel, dim3(n_frames), dim3(1024), lds, s, p);
    return hipGetLastError();
}

// Row r of a tile is the 8*d-bit integer at byte (r*d) of the tile payload: fetch the two
// qwords that hold it from the swizzled LDS image, funnel-shift, expand, add the minimum.
__device__ __forceinline__ void unpack_tile_from_lds(const uint64_t *s_in, uint32_t byte_base, uint32_t d,
                                                     uint32_t mn, uint32_t (&v)[16]) {
    const uint32_t m4 = mn * 0x01010101u;
    const uint64_t keep = d >= 8u ? ~0ull : ((1ull << (8u * d)) - 1ull);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint32_t a = byte_base + d * (uint32_t)r;
        const uint32_t q = a >> 3;
        const uint32_t sh = (a & 7u) * 8u;
        const uint64_t lo = s_in[swzq16(q)], hi = s_in[swzq16(q + 1u)];
        const uint64_t row = (sh ? ((lo >> sh) | (hi << (64u - sh))) : lo) & keep;
        uint32_t x, y;
        expand_row(row, d, x, y);
        v[2 * r] = add_bytes(x, m4);
        v[2 * r + 1] = add_bytes(y, m4);
    }
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

// LDS image of a chunk's payload: up to 15 bytes of alignment shift + 32 KiB + one qword of
// over-read, rounded up to whole 256-byte swizzle groups.
constexpr uint32_t kDecLdsSlots = ((15u + kMaxChunkWords * 8u + 8u + 15u) / 16u + 15u) / 16u * 16u;
constexpr int kDecMaxPieces = (kDecLdsSlots + kBlockThreads - 1) / kBlockThreads;   // 16-B pieces per thread

template <bool FAST_IMG>
__global__ __launch_bounds__(kBlockThreads) void decode_kernel(DecParams p) {
    __shared__ __attribute__((aligned(16))) uint64_t s_in[kDecLdsSlots * 2];
    __shared__ uint32_t s_wave_tot[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t c = blockIdx.x;
    const uint32_t f = c / p.chunks_per_frame;
    const uint32_t cf = c - f * p.chunks_per_frame;
    // everything the address arithmetic needs, requested together
    const uint32_t ok = p.frame_ok[f];
    const uint64_t foff = p.frame_offsets[f];
    const uint32_t *co = p.chunk_off + (size_t)f * (p.chunks_per_frame + 1u) + cf;
    const uint32_t w_begin = co[0], w_end = co[1];
    if (!ok) return;   // rejected frame: image untouched (dbde_util.cpp:296-303)

    const uint8_t *fb = p.stream + foff;
    const uint32_t t0 = cf * kChunkTiles + 2u * (uint32_t)tid;
    const bool hasA = t0 < p.T, hasB = t0 + 1u < p.T;
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
    const uint32_t n16r = (n16 + 15u) & ~15u;
#pragma unroll
    for (int j = 0; j < kDecMaxPieces; j++) {
        const uint32_t i = (uint32_t)tid + (uint32_t)j * kBlockThreads;
        const uint32_t src_slot = swz16(i);
        if (i < n16r && src_slot < n16) {
            const uint32_t wave_slot0 = (uint32_t)j * kBlockThreads + (uint32_t)wave * 64u;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(asrc + 16ull * src_slot),
                (__attribute__((address_space(3))) void *)(&s_in[2u * wave_slot0]), 16, 0, 0);
        }
    }
    uint32_t dA = 0, dB = 0, mA = 0, mB = 0;
    if (hasA) { dA = depth_arr[t0]; mA = min_arr[t0]; }
    if (hasB) { dB = depth_arr[t0 + 1]; mB = min_arr[t0 + 1]; }

    uint32_t chunk_total;   // equals chunk_words for a validated frame
    const uint32_t incl = block_scan_incl(dA + dB, s_wave_tot, lane, wave, chunk_total);   // barrier inside
    const uint32_t offA = incl - (dA + dB), offB = offA + dA;

    uint32_t va[16], vb[16];
    unpack_tile_from_lds(s_in, shift + 8u * offA, dA, mA, va);
    unpack_tile_from_lds(s_in, shift + 8u * offB, dB, mB, vb);

    uint8_t *img = p.images + (size_t)f * p.frame_pixels;
    if (FAST_IMG) {
        if (hasA) {
            const uint32_t ty = t0 / p.w, tx = t0 - ty * p.w;
            uint8_t *base = img + (size_t)(8u * tx);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int yy = 8 * (int)ty + r;
                if (yy < p.H) {
                    uint4 q;
                    q.x = va[2 * r]; q.y = va[2 * r + 1]; q.z = vb[2 * r]; q.w = vb[2 * r + 1];
                    *reinterpret_cast<uint4 *>(base + (size_t)yy * (size_t)p.W) = q;
                }
            }
        }
    } else {
        if (hasA) store_tile_generic(img, p.W, p.H, p.w, t0, va);
        if (hasB) store_tile_generic(img, p.W, p.H, p.w, t0 + 1u, vb);
    }
}

hipError_t launch_decode(const DecParams &p, bool fast_img, hipStream_t s) {
    dim3 grid(p.n_chunks), block(kBlockThreads);
    if (fast_img) hipLaunchKernelGGL((decode_kernel<true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((decode_kernel<false>), grid, block, 0, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// stream scanner: frame-to-frame hop (reference README.md:12-23: sizes are only in-band)
// ---------------------------------------------------------------------------------------
__global__ void scan_stream_kernel(const uint8_t *stream, uint64_t stream_bytes, uint32_t T, int max_frames,
                                   uint64_t *offsets, uint32_t *count) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint64_t off = 0;
    int n = 0;
    const uint64_t meta = 32ull + 2ull * T;
    while (n < max_frames && off + meta <= stream_bytes) {
        const uint32_t n64 = load_u32_bytes(stream + off + 28 + 2ull * T);
        const uint64_t len = meta + 8ull * n64;
        if ((int32_t)n64 < 0 || off + len > stream_bytes) break;
        offsets[n++] = off;
        off += len;
    }
    *count = (uint32_t)n;
}

hipError_t launch_scan_stream(const uint8_t *stream, uint64_t stream_bytes, uint32_t T, int max_frames,
                              uint64_t *d_offsets, uint32_t *d_count, hipStream_t s) {
    hipLaunchKernelGGL(scan_stream_kernel, dim3(1), dim3(64), 0, s, stream, stream_bytes, T, max_frames,
                       d_offsets, d_count);
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
