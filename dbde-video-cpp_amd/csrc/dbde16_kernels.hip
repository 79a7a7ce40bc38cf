// dbde16_kernels.hip -- DBDE16 on MI355X (gfx950): the higher-bit-depth extension reference README.md:65 points at,
// as specified in oracle/dbde16_oracle.c (U16 pixels, depth 0..16, U16 minima, nm = 2T).  PARITY UNPINNED: the
// reference defines no such format; the kernels are checked against the DBDE16 oracle, which in turn agrees with
// the pinned 8-bit oracle on 8-bit images.
//
// Correctness-first companion of dbde_kernels.hip, same decomposition, ONE tile per lane (a tile row is 8 pixels =
// 16 bytes, so every image access is still one 16-byte access per row and lane):
//   encode: large launches of 16-byte aligned rows go through the 8-bit path's persistent encoder
//           (dbde_kernels.hip: encode_kernel<IN, OUT, PIX = 2>); everything else:
//           enc16_kernel  -- one pass: a workgroup per 256-tile chunk reduces, publishes its word count, packs, and
//                            finds its prefix by summing the records in front of it (ticket order, two-level sums)
//   decode: the 8-bit path's index kernels (IdxParams::min_bytes = 2) + dec16_kernel.
// A tile row is the 8*d-bit integer at byte r*d of the tile payload, exactly as in the 8-bit format: rows are
// assembled / taken apart as two 4-pixel halves of 4*d <= 64 bits.
#include "dbde16_kernels.h"

namespace dbde16 {

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef u32x4_t __attribute__((aligned(1))) u32x4_unaligned;

__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x) {   // DPP inclusive scan over the wave
    uint32_t t = x;
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x111, 0xF, 0xF, false);
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x112, 0xF, 0xF, false);
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x114, 0xF, 0xF, false);
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x118, 0xF, 0xF, false);
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x142, 0xA, 0xF, false);
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x143, 0xC, 0xF, false);
    return t;
}
__device__ __forceinline__ void store_u32_bytes(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ void store_u64_any(uint8_t *p, uint64_t v) { __builtin_memcpy(p, &v, 8); }

// One tile (possibly partial: clamp-to-edge = the format's constant padding) into 8 rows x 4 dwords (2 pixels each).
__device__ __forceinline__ void load_tile16(const uint16_t *img, int W, int H, uint32_t ty, uint32_t tx, uint32_t (&v)[32]) {
    const int x0 = 8 * (int)tx;
    if (x0 + 8 <= W) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            int yy = 8 * (int)ty + r;
            yy = yy < H ? yy : H - 1;
            const u32x4_t q = *reinterpret_cast<const u32x4_unaligned *>(img + (size_t)yy * (size_t)W + x0);
            v[4 * r] = q[0]; v[4 * r + 1] = q[1]; v[4 * r + 2] = q[2]; v[4 * r + 3] = q[3];
        }
    } else {   // right-edge tile: pixel by pixel, the last valid one repeated
        for (int r = 0; r < 8; r++) {
            int yy = 8 * (int)ty + r;
            yy = yy < H ? yy : H - 1;
            for (int c = 0; c < 8; c += 2) {
                const int xa = x0 + c < W ? x0 + c : W - 1, xb = x0 + c + 1 < W ? x0 + c + 1 : W - 1;
                v[4 * r + c / 2] = (uint32_t)img[(size_t)yy * (size_t)W + xa] | ((uint32_t)img[(size_t)yy * (size_t)W + xb] << 16);
            }
        }
    }
}

__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ void tile_minmax16(const uint32_t (&v)[32], uint32_t &mn, uint32_t &mx) {
    uint32_t lo = v[0], hi = v[0];
#pragma unroll
    for (int i = 1; i < 32; i++) { lo = pk_min_u16(lo, v[i]); hi = pk_max_u16(hi, v[i]); }
    mn = (lo & 0xFFFFu) < (lo >> 16) ? (lo & 0xFFFFu) : (lo >> 16);
    mx = (hi & 0xFFFFu) > (hi >> 16) ? (hi & 0xFFFFu) : (hi >> 16);
}
__device__ __forceinline__ uint32_t depth_of(uint32_t range) { return range ? 32u - (uint32_t)__builtin_clz(range) : 0u; }

struct Tile16 { uint32_t f, cf, t, ty, tx; bool has; };
__device__ __forceinline__ Tile16 tile_of(const Params16 &p, uint32_t c, uint32_t tid) {
    Tile16 k;
    k.f = c / p.chunks_per_frame;
    k.cf = c - k.f * p.chunks_per_frame;
    k.t = k.cf * kChunkTiles16 + tid;
    k.has = k.t < p.T;
    const uint32_t t = k.has ? k.t : 0u;
    k.ty = t / p.w;
    k.tx = t - k.ty * p.w;
    return k;
}

// ---- encode: ONE pass ------------------------------------------------------------------------------------------
// Workgroup = one chunk of 256 tiles, chunk id = arrival ticket (every chunk in front of a workgroup belongs to a
// workgroup that is already running: no assumption about dispatch order).  Load, reduce, publish the chunk's word
// count as an 8-byte record, pack into LDS while the record travels, then sum the records in front of it (wave 0;
// two levels, groups of 64, so no record depends on a chain of earlier prefixes), store.
// Concatenated layout: the workgroup that holds a frame's last chunk publishes the frame's word count; a frame's base
// is the sum of the counts in front of it.  All records are zeroed before the launch (dbde16_hip_encode_frames).
typedef unsigned long long u64a;
constexpr u64a kReady = 1ull << 63;

// Four pixels (two dwords, 16 bits each, already minus the minimum) -> the 4*d-bit integer p0 | p1<<d | p2<<2d | p3<<3d.
__device__ __forceinline__ uint64_t pack4x16(uint32_t a, uint32_t b, uint32_t d) {
    const uint64_t lo = (uint64_t)(a & 0xFFFFu) | ((uint64_t)(a >> 16) << d);
    const uint64_t hi = (uint64_t)(b & 0xFFFFu) | ((uint64_t)(b >> 16) << d);
    return lo | (hi << (2u * d));
}

__device__ __forceinline__ uint64_t wave_sum64(uint64_t x) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) x += __shfl_xor(x, m, 64);
    return x;
}
// Wave-wide: sums of four record ranges (at most 64 records each: lane l reads record l of every range in one
// poll), waiting until every record is published.  False after 2 s (the caller raises the sticky failure word).
__device__ __forceinline__ uint32_t sum_records(const u64a *r0, uint32_t n0, const u64a *r1, uint32_t n1, const u64a *r2, uint32_t n2,
                                                const u64a *r3, uint32_t n3, uint32_t lane, uint64_t t_start,
                                                uint64_t &s0, uint64_t &s1, uint64_t &s2, uint64_t &s3) {
    u64a w0 = kReady, w1 = kReady, w2 = kReady, w3 = kReady;
    for (;;) {
        if (lane < n0 && w0 == kReady) w0 = __hip_atomic_load(&r0[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane < n1 && w1 == kReady) w1 = __hip_atomic_load(&r1[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane < n2 && w2 == kReady) w2 = __hip_atomic_load(&r2[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane < n3 && w3 == kReady) w3 = __hip_atomic_load(&r3[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all((int)(((w0 & w1 & w2 & w3) >> 63) & 1ull))) break;
        if (wall_clock64() - t_start > 200000000ull) return 0u;   // 2 s at 100 MHz: give up, loudly
        __builtin_amdgcn_s_sleep(1);
        // records not there yet read as 0: mark them "to be read again"
        w0 = (w0 & kReady) ? w0 : kReady; w1 = (w1 & kReady) ? w1 : kReady; w2 = (w2 & kReady) ? w2 : kReady; w3 = (w3 & kReady) ? w3 : kReady;
    }
    s0 = wave_sum64(lane < n0 ? (w0 & ~kReady) : 0ull); s1 = wave_sum64(lane < n1 ? (w1 & ~kReady) : 0ull);
    s2 = wave_sum64(lane < n2 ? (w2 & ~kReady) : 0ull); s3 = wave_sum64(lane < n3 ? (w3 & ~kReady) : 0ull);
    return 1u;
}

// LDS payload image of a chunk: kPayWords16 words + one shared trash word.  The worst case (256 tiles of depth 16) is
// what it holds today -- registers (the persistent loop needs ~120) limit the kernel to four workgroups per CU anyway;
// a fifth would need <= 32,768 bytes of LDS and 96 registers (tried: 27 spills, 1.51 ms against 1.20 ms).  A smaller
// image works too: a chunk with more words stores the first kPayWords16, then packs again for the rest (tested with
// kPayWords16 = 4088; pack_tile16 / the second pass below are written for any value).
constexpr uint32_t kPayWords16 = kChunkTiles16 * 16u;

// The tile's payload words into the LDS image: straight-line funnel over 16 half rows of 4*d <= 64 bits (every half
// row stores the word it is filling).  Words [q_lo, q_lo + kPayWords16) of the chunk go to image[0..), the others and
// everything of a tile without payload to the trash word.
__device__ __forceinline__ void pack_tile16(const uint32_t (&v)[32], uint32_t mn, uint32_t d, uint64_t *s_pay, uint32_t q0, uint32_t q_lo) {
    const uint32_t mn2 = mn * 0x00010001u;   // every 16-bit half >= mn: no borrow crosses a half
    const uint32_t nb = 4u * d;
    uint32_t q = d ? q0 - q_lo : kPayWords16;   // (unsigned: words in front of q_lo wrap to huge and land in the trash)
    uint64_t acc = 0;
    uint32_t fill = 0;
#pragma unroll
    for (int h = 0; h < 16; h++) {
        const uint64_t bits = pack4x16(v[2 * h] - mn2, v[2 * h + 1] - mn2, d);
        const uint64_t merged = acc | (bits << fill);
        s_pay[q < kPayWords16 ? q : kPayWords16] = merged;
        const uint32_t nf = fill + nb;
        const bool emit = nf >= 64u;
        acc = emit ? ((bits >> 1) >> (63u - fill)) : merged;
        fill = nf & 63u;
        q += emit ? 1u : 0u;
    }
}

// `n` words of the LDS image to dst (the chunk's payload is contiguous): 16-byte stores where dst is word-aligned.
__device__ __forceinline__ void copy_out16(const uint64_t *s_pay, uint8_t *dst, uint32_t n, uint32_t tid) {
    if ((reinterpret_cast<uintptr_t>(dst) & 7u) == 0u) {   // 16-byte stores between a possible odd first and last word
        const uint32_t head = (uint32_t)(reinterpret_cast<uintptr_t>(dst) >> 3) & 1u;
        const uint32_t h1 = head < n ? head : n;
        if (tid == 0 && h1) *reinterpret_cast<uint64_t *>(dst) = s_pay[0];
        const uint32_t pairs = (n - h1) >> 1;
        typedef uint64_t u64x2_t __attribute__((ext_vector_type(2)));
        for (uint32_t i = tid; i < pairs; i += kChunkTiles16) {
            u64x2_t q = {s_pay[h1 + 2u * i], s_pay[h1 + 2u * i + 1u]};
            *reinterpret_cast<u64x2_t *>(dst + 8ull * (h1 + 2u * i)) = q;
        }
        if (tid == 64u && ((n - h1) & 1u)) *reinterpret_cast<uint64_t *>(dst + 8ull * (n - 1u)) = s_pay[n - 1u];
    } else {
        for (uint32_t i = tid; i < n; i += kChunkTiles16) store_u64_any(dst + 8ull * i, s_pay[i]);
    }
}

#ifndef DBDE16_WAVES
#define DBDE16_WAVES 4
#endif
__global__ __launch_bounds__(kChunkTiles16, DBDE16_WAVES) void enc16_kernel(Params16 p) {
    __shared__ __attribute__((aligned(16))) uint64_t s_pay[kPayWords16 + 1];   // + the trash word
    __shared__ uint32_t s_tot[kChunkTiles16 / 64];
    __shared__ uint32_t s_chunk, s_ok, s_boot[2];
    __shared__ unsigned long long s_pre[2];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
#ifdef DBDE_DIAG
    uint64_t dt[6];
#define DIAG_MARK(i) dt[i] = wall_clock64()
#else
#define DIAG_MARK(i)
#endif
    // Chunk ids.  A workgroup's first chunk is its arrival rank.  STATIC (chunk = rank + k * G, no atomic per chunk: a
    // single-address ticket per 256-tile chunk caps the launch at about 73 chunks per microsecond and costs 2 us of
    // every chunk's life, measured) is only safe when all G workgroups run at the same time; that is proven, not
    // assumed: all G have arrived before anyone left.  If the arrivals do not complete within 20 us the launch
    // falls back to TICKETS (the arrival counter keeps counting, ids stay dense), which needs nothing but running
    // workgroups.  One CAS decides for the whole launch.
    const uint32_t n_chunks = p.n_frames * p.chunks_per_frame, G = gridDim.x;
    if (tid == 0) {
        const uint32_t rank = atomicAdd(&p.ticket[0], 1u);
        uint32_t mode = n_chunks <= G ? 1u : 0u;   // one chunk per workgroup at most: nothing to agree on
        const uint64_t t0 = wall_clock64();
        while (!mode) {
            mode = __hip_atomic_load(&p.ticket[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (mode) break;
            const uint32_t arrived = __hip_atomic_load(&p.ticket[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (arrived >= G && !p.force_tickets) atomicCAS(&p.ticket[1], 0u, 1u);
            else if (p.force_tickets || wall_clock64() - t0 > 2000ull) atomicCAS(&p.ticket[1], 0u, 2u);
            else __builtin_amdgcn_s_sleep(2);
        }
        s_boot[0] = rank; s_boot[1] = mode;
    }
    __syncthreads();
    const bool static_mode = __builtin_amdgcn_readfirstlane(s_boot[1]) == 1u;
    uint32_t c = __builtin_amdgcn_readfirstlane(s_boot[0]);
    while (c < n_chunks) {
#ifdef DBDE_DIAG
        dt[0] = wall_clock64();
#endif
        DIAG_MARK(1);
        const Tile16 k = tile_of(p, c, tid);
        uint32_t v[32];
        load_tile16(p.images + (size_t)k.f * p.frame_pixels, p.W, p.H, k.ty, k.tx, v);
        uint32_t mn, mx;
        tile_minmax16(v, mn, mx);
        const uint32_t d = k.has ? depth_of(mx - mn) : 0u;
        const uint32_t incl = wave_scan_incl(d);
        if (lane == 63u) s_tot[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0, total = 0;
        for (uint32_t q = 0; q < kChunkTiles16 / 64; q++) { wbase += q < wave ? s_tot[q] : 0u; total += s_tot[q]; }
        if (tid == 0) __hip_atomic_store(&p.state[c], kReady | (u64a)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        DIAG_MARK(2);

        // pack while the record travels
        pack_tile16(v, mn, d, s_pay, wbase + incl - d, 0u);

        // Prefix inside the frame and the frame's base WITHOUT a serial chain: two-level sums of records that each depend
        // on published counts only.  A = the chunks in front of this one in its group of 64, B = the frame's groups in
        // front (a group's count is published by its 64th chunk = A + own), C / D = the same two levels over frame word
        // counts (concatenated layout; a frame's count is published by its last chunk).  One poll of four loads per lane.
        const uint64_t meta = 32ull + 3ull * p.T;
        DIAG_MARK(3);
        if (wave == 0) {
            const uint64_t t_start = wall_clock64();
            const uint32_t g = k.cf >> 6, nA = k.cf & 63u, gpf = (p.chunks_per_frame + 63u) >> 6;
            const bool concat = p.slot_stride == 0ull;
            const uint32_t nC = concat ? (k.f & 63u) : 0u, fg = concat ? (k.f >> 6) : 0u;
            const u64a *rA = p.state + (size_t)k.f * p.chunks_per_frame + (size_t)g * 64u;
            const u64a *rB = p.gsum + (size_t)k.f * gpf;
            const u64a *rC = p.fsize + (size_t)(k.f & ~63u);
            uint64_t sA = 0, sB = 0, sC = 0, sD = 0, z0, z1, z2;
            uint32_t ok = 1u;
            const uint32_t nB = g < 64u ? g : 64u, nD = fg < 64u ? fg : 64u;
            const bool pub_group = nA == 63u, pub_frame = concat && k.cf == p.chunks_per_frame - 1u;
            if (!pub_group && !pub_frame) {
                ok = sum_records(rA, nA, rB, nB, rC, nC, p.fgsum, nD, lane, t_start, sA, sB, sC, sD);
            } else {   // a chunk that publishes a higher-level record does so BEFORE it waits for records of that level
                ok = sum_records(rA, nA, rA, 0u, rA, 0u, rA, 0u, lane, t_start, sA, z0, z1, z2);
                if (pub_group && lane == 0 && ok)
                    __hip_atomic_store(&p.gsum[(size_t)k.f * gpf + g], kReady | (u64a)(sA + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ok) ok = sum_records(rB, nB, rC, nC, rA, 0u, rA, 0u, lane, t_start, sB, sC, z0, z1);
            }
            for (uint32_t i = 64u; i < g && ok; i += 64u) {    // frames of more than 4096 chunks
                uint64_t x = 0;
                ok = sum_records(rB + i, g - i < 64u ? g - i : 64u, rA, 0u, rA, 0u, rA, 0u, lane, t_start, x, z0, z1, z2);
                sB += x;
            }
            const uint64_t inf = sA + sB, fwords = inf + total;
            if (pub_frame && lane == 0 && ok) {
                __hip_atomic_store(&p.fsize[k.f], kReady | (u64a)fwords, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (nC == 63u) __hip_atomic_store(&p.fgsum[fg], kReady | (u64a)(sC + fwords), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if ((pub_group || pub_frame) && ok) ok = sum_records(p.fgsum, nD, rA, 0u, rA, 0u, rA, 0u, lane, t_start, sD, z0, z1, z2);
            for (uint32_t i = 64u; i < fg && ok; i += 64u) {   // launches of more than 4096 frames
                uint64_t x = 0;
                ok = sum_records(p.fgsum + i, fg - i < 64u ? fg - i : 64u, rA, 0u, rA, 0u, rA, 0u, lane, t_start, x, z0, z1, z2);
                sD += x;
            }
            if (lane == 0) {
                s_pre[0] = inf;
                s_pre[1] = concat ? (uint64_t)k.f * meta + 8ull * (sC + sD) : (uint64_t)k.f * p.slot_stride;
                s_ok = ok;
                if (!ok) atomicOr(p.sticky, 1u);
            }
        }
        DIAG_MARK(4);
        __syncthreads();
        if (!s_ok) return;
        DIAG_MARK(5);
        const uint32_t inf = (uint32_t)s_pre[0];
        uint8_t *fb = p.out + s_pre[1];
        if (k.has) {   // metadata of this lane's tile
            fb[24 + k.t] = (uint8_t)d;
            uint8_t *m = fb + 28 + p.T + 2ull * k.t;
            m[0] = (uint8_t)mn; m[1] = (uint8_t)(mn >> 8);
        }
        uint8_t *dst = fb + meta + 8ull * inf;   // the chunk's contiguous payload
        copy_out16(s_pay, dst, total < kPayWords16 ? total : kPayWords16, tid);
        if (total > kPayWords16) {   // (nearly) every tile of depth 16: the words the image had no room for
            __syncthreads();
            load_tile16(p.images + (size_t)k.f * p.frame_pixels, p.W, p.H, k.ty, k.tx, v);
            pack_tile16(v, mn, d, s_pay, wbase + incl - d, kPayWords16);
            __syncthreads();
            copy_out16(s_pay, dst + 8ull * kPayWords16, total - kPayWords16, tid);
        }
        if (tid == 0) {
            if (k.cf == 0u) {   // frame header and the first I32 fields (trap T1: elapsed travels as an F64; 0 here)
                store_u32_bytes(fb, 2u);
                store_u64_any(fb + 4, p.first_index + k.f);
                store_u64_any(fb + 12, 0ull);
                store_u32_bytes(fb + 20, p.T);
                store_u32_bytes(fb + 24 + p.T, 2u * p.T);
                if (p.frame_offsets) p.frame_offsets[k.f] = s_pre[1];
            }
            if (k.cf == p.chunks_per_frame - 1u) {   // the frame's word count is known here
                const uint32_t words = inf + total;
                store_u32_bytes(fb + 28 + 3ull * p.T, words);
                if (p.frame_bytes) p.frame_bytes[k.f] = meta + 8ull * words;
            }
        }
#ifdef DBDE_DIAG
        if (tid == 0 && (c & 63u) == 5u) {   // a sample of workgroups, wave 0's view: ticket | load+reduce | pack | look-back | barrier wait | store
            __builtin_amdgcn_s_waitcnt(0);
            const uint64_t t6 = wall_clock64();
            for (int i = 0; i < 5; i++) atomicAdd(&p.diag[i], (unsigned long long)(dt[i + 1] - dt[i]));
            atomicAdd(&p.diag[5], (unsigned long long)(t6 - dt[5]));
            atomicAdd(&p.diag[6], 1ull);
        }
#endif
        // the next chunk; the barrier also hands the LDS image back
        if (!static_mode && tid == 0) s_chunk = atomicAdd(&p.ticket[0], 1u);
        __syncthreads();
        c = static_mode ? c + G : __builtin_amdgcn_readfirstlane(s_chunk);
    }
}

// ---- decode ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kChunkTiles16) void dec16_kernel(DecParams16 p) {
    __shared__ __attribute__((aligned(16))) uint8_t s_in[kChunkTiles16 * 128 + 256 + 48];   // + the shift, rounded up to a swizzle group
    __shared__ uint32_t s_tot[kChunkTiles16 / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
#ifdef DBDE_DIAG
    uint64_t dt[5]; dt[0] = wall_clock64();
#endif
    // workgroup -> chunk as in the 8-bit decoder: inside a group of 128 chunks each XCD (workgroup id mod 8) takes 16
    // CONSECUTIVE chunks, so that an XCD's L2 sees runs of the stream and of the image instead of every eighth 32 KB
    uint32_t c = blockIdx.x;
    {
        const uint32_t grp = c / 128u, rem = c % 128u;
        if ((grp + 1u) * 128u <= gridDim.x) c = grp * 128u + (rem & 7u) * 16u + (rem >> 3);
    }
    const uint32_t f = c / p.chunks_per_frame, cf = c - f * p.chunks_per_frame;
    // the three index words are requested together (one memory round trip, not three), then looked at
    const uint32_t *co = p.chunk_off + (size_t)f * (p.chunks_per_frame + 1u) + cf;
    const uint32_t ok = p.frame_ok[f], w_begin = co[0], w_end = co[1];
    const uint64_t f_off = p.frame_offsets[f];
    asm volatile("" :: "s"(w_begin), "s"(w_end), "s"(f_off));   // ... before the branch
    if (!ok) return;   // rejected frame: image untouched
    const uint8_t *fb = p.stream + f_off;
    const uint32_t words = w_end - w_begin;
    const uint32_t t = cf * kChunkTiles16 + tid;
    const bool has = t < p.T;
    // the chunk's payload: one contiguous byte range, fetched as aligned 16-byte pieces (source aligned down; the
    // first tile's bytes then start `shift` bytes into the image; frames are at least 8 bytes into any buffer)
    const uint8_t *src = fb + 32ull + 3ull * p.T + 8ull * w_begin;
    const uint32_t shift = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 15u);
    const uint8_t *asrc = src - shift;
    const uint32_t n16 = (shift + 8u * words + 15u) >> 4;
    // Every tile of depth 16 (the chunk's word count says so): lanes read the image at a 128-byte stride, which piles
    // onto two banks; the image is then held swizzled -- physical 16-byte slot i = logical slot i ^ ((i >> 4) & 15),
    // a permutation inside 256-byte groups, applied on the SOURCE side of the DMA -- as in the 8-bit decoder.
    const uint32_t n_tiles = p.T - cf * kChunkTiles16 < kChunkTiles16 ? p.T - cf * kChunkTiles16 : kChunkTiles16;
    const bool swz = words != 0u && words == 16u * n_tiles && (shift & 7u) == 0u;
    const uint32_t n16r = swz ? (n16 + 15u) & ~15u : n16;
    uint32_t d = 0, mn = 0;   // this lane's tile: depth and minimum, in flight with the payload
    if (has) {
        d = fb[24 + t];
        const uint8_t *m = fb + 28 + p.T + 2ull * t;
        mn = (uint32_t)m[0] | ((uint32_t)m[1] << 8);
    }
    // whole pieces inside the readable extent: all of a thread's loads (at most 8: 256 tiles x 128 bytes + the
    // shift) are in flight before the first is stored, one memory round trip for the chunk
    const uint64_t room = (uint64_t)((p.stream + p.stream_bytes) - asrc);
    const uint32_t n16_in = room / 16u < (uint64_t)n16 ? (uint32_t)(room / 16u) : n16;
    {   // LDS-DMA (global_load_lds_dwordx4): the wave's 64 pieces land at wave-uniform base + lane * 16, no staging
        // registers; all of a thread's pieces (at most 9: 256 tiles x 128 bytes + the shift) are in flight together
        constexpr int kMaxPieces = (kChunkTiles16 * 128 + 16 + 16 * kChunkTiles16 - 1) / (16 * kChunkTiles16);
#pragma unroll
        for (int j = 0; j < kMaxPieces; j++) {
            const uint32_t i = tid + (uint32_t)j * kChunkTiles16;
            const uint32_t src_slot = swz ? i ^ ((i >> 4) & 15u) : i;
            if (i < n16r && src_slot < n16_in)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(asrc + 16ull * src_slot),
                                                 (__attribute__((address_space(3))) void *)(s_in + 16u * ((uint32_t)j * kChunkTiles16 + wave * 64u)),
                                                 16, 0, 2);
        }
    }
    for (uint32_t i = n16_in + tid; i < n16; i += kChunkTiles16) {   // a piece that would cross the end of the extent: byte by byte
        uint32_t wq[4] = {0, 0, 0, 0};
        for (uint32_t b = 0; b < 16u; b++) {
            const uint8_t *s1 = asrc + 16ull * i + b;
            if (s1 < p.stream + p.stream_bytes) wq[b >> 2] |= (uint32_t)*s1 << (8u * (b & 3u));
        }
        u32x4_t q = {wq[0], wq[1], wq[2], wq[3]};
        *reinterpret_cast<u32x4_t *>(s_in + 16u * (swz ? i ^ ((i >> 4) & 15u) : i)) = q;
    }
    const uint32_t incl = wave_scan_incl(d);
    if (lane == 63u) s_tot[wave] = incl;
    DIAG_MARK(1);
    // a barrier does not drain vector memory: every wave sees its own DMA land (and its LDS writes retire) first
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    DIAG_MARK(2);
    uint32_t wbase = 0;
    for (uint32_t q = 0; q < kChunkTiles16 / 64; q++) wbase += q < wave ? s_tot[q] : 0u;
    if (!has) return;
    const uint32_t byte0 = shift + 8u * (wbase + incl - d);
    const uint32_t ty = t / p.w, tx = t - ty * p.w;
    const int x0t = 8 * (int)tx;
    const uint64_t fmask = d >= 16u ? 0xFFFFull : ((1ull << d) - 1ull);
    uint16_t *img = p.images + (size_t)f * p.frame_pixels;
    const bool rows16 = (p.W & 7) == 0 && (reinterpret_cast<uintptr_t>(p.images) & 15u) == 0u;   // every tile row a 16-byte aligned store
    // A row's 8*d bits start at byte r*d of the tile payload: two 64-bit windows, one per 4-pixel half (the second
    // starts 4*d bits = d/2 bytes, and 4 bits when d is odd, later).
#ifndef DBDE16_ALIGNED_READS
#define DBDE16_ALIGNED_READS 1   // A/B switch: 0 = two unaligned ds_read_b64 per row and 64-bit shifts per pixel
#endif
    if (swz) {   // depth 16 everywhere, rows 8-byte aligned: a row is its sixteen bytes plus the minimum
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
        const uint32_t mn2 = mn * 0x00010001u;
        uint2 lo8[8], hi8[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t A = byte0 + 16u * (uint32_t)r, B = A + 8u;
            lo8[r] = *reinterpret_cast<const uint2 *>(s_in + (A ^ ((A >> 4) & 0xF0u)));
            hi8[r] = *reinterpret_cast<const uint2 *>(s_in + (B ^ ((B >> 4) & 0xF0u)));
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            auto add = [&](uint32_t x) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, x) + __builtin_bit_cast(u16x2, mn2)); };
            const u32x4_t o = {add(lo8[r].x), add(lo8[r].y), add(hi8[r].x), add(hi8[r].y)};
            const int yy = 8 * (int)ty + r;
            if (yy < p.H) {
                uint16_t *row = img + (size_t)yy * (size_t)p.W + x0t;
                if (x0t + 8 <= p.W) {
                    if (rows16) __builtin_nontemporal_store(o, reinterpret_cast<u32x4_t *>(row));
                    else __builtin_nontemporal_store(o, reinterpret_cast<u32x4_unaligned *>(row));
                } else {
                    for (int i = 0; i < 8; i++) if (x0t + i < p.W) row[i] = (uint16_t)(o[i >> 1] >> (16 * (i & 1)));
                }
            }
        }
        return;
    }
#if DBDE16_ALIGNED_READS
    // Each window comes out of the three ALIGNED dwords that hold it (v_alignbyte, as in the 8-bit decoder: LDS reads
    // of 8 bytes at odd addresses are what made mixed depths the slow content here); a pixel is v_alignbit at i*d
    // (shift counts are taken modulo 32: from 32 on the window's high dword is shifted instead), two pixels per dword,
    // the minimum added with v_pk_add_u16 (modulo 2^16, as the spec says).
    uint32_t lw[8][3], hw[8][3];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint32_t a = byte0 + (uint32_t)r * d, ah = a + (d >> 1);
        const uint32_t *q = reinterpret_cast<const uint32_t *>(s_in + (a & ~3u));
        const uint32_t *qh = reinterpret_cast<const uint32_t *>(s_in + (ah & ~3u));
        lw[r][0] = q[0]; lw[r][1] = q[1]; lw[r][2] = q[2];
        hw[r][0] = qh[0]; hw[r][1] = qh[1]; hw[r][2] = qh[2];
    }
    DIAG_MARK(3);
    const uint32_t m32 = (uint32_t)fmask, mn2 = mn * 0x00010001u, sh_odd = (d & 1u) * 4u;
    const bool c2 = 2u * d >= 32u, c3 = 3u * d >= 32u;
    auto four = [&](uint32_t x0, uint32_t x1, uint32_t &o0, uint32_t &o1) __attribute__((always_inline)) {
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
        const uint32_t p0 = x0 & m32;
        const uint32_t p1 = __builtin_amdgcn_alignbit(x1, x0, d) & m32;
        const uint32_t p2 = __builtin_amdgcn_alignbit(c2 ? 0u : x1, c2 ? x1 : x0, 2u * d) & m32;
        const uint32_t p3 = __builtin_amdgcn_alignbit(c3 ? 0u : x1, c3 ? x1 : x0, 3u * d) & m32;
        o0 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, p0 | (p1 << 16)) + __builtin_bit_cast(u16x2, mn2));
        o1 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, p2 | (p3 << 16)) + __builtin_bit_cast(u16x2, mn2));
    };
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint32_t a = byte0 + (uint32_t)r * d, ah = a + (d >> 1);
        const uint32_t x0 = __builtin_amdgcn_alignbyte(lw[r][1], lw[r][0], a), x1 = __builtin_amdgcn_alignbyte(lw[r][2], lw[r][1], a);
        const uint32_t w0 = __builtin_amdgcn_alignbyte(hw[r][1], hw[r][0], ah), w1 = __builtin_amdgcn_alignbyte(hw[r][2], hw[r][1], ah);
        const uint32_t h0 = __builtin_amdgcn_alignbit(w1, w0, sh_odd), h1 = w1 >> sh_odd;
        uint32_t o0, o1, o2, o3;
        four(x0, x1, o0, o1);
        four(h0, h1, o2, o3);
        const u32x4_t o = {o0, o1, o2, o3};
        const int yy = 8 * (int)ty + r;
        if (yy < p.H) {
            uint16_t *row = img + (size_t)yy * (size_t)p.W + x0t;
            if (x0t + 8 <= p.W) {
                // written once, read by nobody here: non-temporal, as in the 8-bit decoder
                if (rows16) __builtin_nontemporal_store(o, reinterpret_cast<u32x4_t *>(row));
                else __builtin_nontemporal_store(o, reinterpret_cast<u32x4_unaligned *>(row));
            } else {
                for (int i = 0; i < 8; i++) if (x0t + i < p.W) row[i] = (uint16_t)(o[i >> 1] >> (16 * (i & 1)));
            }
        }
    }
#else
    uint64_t lo[8], hi[8];
    {
        const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t *)s_in + byte0;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            asm volatile("ds_read_b64 %0, %1" : "=v"(lo[r]) : "v"(la + (uint32_t)r * d) : "memory");
            asm volatile("ds_read_b64 %0, %1" : "=v"(hi[r]) : "v"(la + (uint32_t)r * d + (d >> 1)) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    DIAG_MARK(3);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint64_t h2 = hi[r] >> ((d & 1u) * 4u);
        uint32_t px[8];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            px[i] = ((uint32_t)((lo[r] >> (i * d)) & fmask) + mn) & 0xFFFFu;       // modulo 2^16, as the spec says
            px[4 + i] = ((uint32_t)((h2 >> (i * d)) & fmask) + mn) & 0xFFFFu;
        }
        const int yy = 8 * (int)ty + r;
        if (yy < p.H) {
            uint16_t *row = img + (size_t)yy * (size_t)p.W + x0t;
            if (x0t + 8 <= p.W) {
                u32x4_t o;
                o[0] = px[0] | (px[1] << 16); o[1] = px[2] | (px[3] << 16); o[2] = px[4] | (px[5] << 16); o[3] = px[6] | (px[7] << 16);
                if (rows16) __builtin_nontemporal_store(o, reinterpret_cast<u32x4_t *>(row));
                else __builtin_nontemporal_store(o, reinterpret_cast<u32x4_unaligned *>(row));
            } else {
                for (int i = 0; i < 8; i++) if (x0t + i < p.W) row[i] = (uint16_t)px[i];
            }
        }
    }
#endif
#ifdef DBDE_DIAG
    if (tid == 0 && (c & 63u) == 5u) {   // a sample of workgroups: fetch issue | barrier | LDS reads | unpack + stores (issue)
        DIAG_MARK(4);
        for (int i = 0; i < 4; i++) atomicAdd(&p.diag[8 + i], (unsigned long long)(dt[i + 1] - dt[i]));
        atomicAdd(&p.diag[12], 1ull);
    }
#endif
}

int encode16_blocks_per_cu() {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, enc16_kernel, kChunkTiles16, 0) != hipSuccess || n < 1) n = 1;
    return n;
}

hipError_t launch_encode16(const Params16 &p, int n_frames, uint32_t resident_blocks, hipStream_t s) {
    const uint32_t n_chunks = (uint32_t)n_frames * p.chunks_per_frame;
    hipLaunchKernelGGL(enc16_kernel, dim3(n_chunks < resident_blocks ? n_chunks : resident_blocks), dim3(kChunkTiles16), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_decode16(const DecParams16 &p, int n_frames, hipStream_t s) {
    hipLaunchKernelGGL(dec16_kernel, dim3((uint32_t)n_frames * p.chunks_per_frame), dim3(kChunkTiles16), 0, s, p);
    return hipGetLastError();
}

}  // namespace dbde16
