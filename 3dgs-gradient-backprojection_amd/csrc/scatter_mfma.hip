// scatter_mfma.hip -- the weighted scatter-accumulate as a BLOCK-SPARSE product on the matrix cores (GWBP_FLAG_GROUP_SCATTER):
//
//   F[g, c0:c0+128] += sum_p w_g(p) * feats[p, c0:c0+128]        (backproject.py:127-131 via colors.grad)
//
// Why: the vector kernels (scatter_full.hip, scatter_wide.hip) pay 2 v_readlane + 1 address + 1 LDS row read per (pair,
// 128..256 channels) on top of the FMAs and are bound by vector issue + LDS return (36 cycles per (pair,
// 256 ch) per SIMD = 7 FMA/clk/SIMD of 32).  A tile's records overlap: 16 records ordered by the bounding box of their
// pixel masks cover a union of ~125 pixels with ~45 each (C2), so a dense (16 records x union pixels) weight table is
// ~36 % non-zero -- and a dense table is what v_mfma_f32_16x16x4_f32 eats: one instruction = 16 records x 4 pixels x 16
// channels = 1024 fp32 FMAs in 32 cycles, no cross-lane traffic, no per-pair vector instruction at all.  The product is
// EXACT fp32: the instruction is a k-ordered fmaf chain, a zero weight contributes +-0, so every record's sum is the same
// ascending-pixel fmaf chain the vector kernels compute (bit for bit; only the order of the flush atomics differs).
//
// Three kernels:
//   k_group_sort  (front stage, one wave per tile)  counting sort of the tile's records by an 8-bit key = coarse bounding
//                 box of the record's pixel mask; consecutive 16 records = one group; per group the union of the masks ->
//                 K-steps (4 pixels each) -> blocks (4 K-steps); group table + one allocation of blocks per tile
//   k_pack        (front stage, one wave per group)  dense operand table: for K-step s and lane (k = lane / 16, i = lane % 16)
//                 A = weight of record i at the (4 s + k)-th pixel of the union, else 0 (mask bit test + prefix popcount +
//                 one gathered 4-B load); stored as float4 per lane per block, plus the four pixel bytes per (block, k)
//   k_scatter_mfma (the scatter)  persistent workgroups and (tile, 128-channel chunk) items exactly like k_scatter_full:
//                 the tile's 256 px x 128 ch slab is staged in LDS once; a wave claims a group, streams its blocks (one
//                 coalesced 16-B load per lane per 4 K-steps, three blocks ahead), reads B = slab[pixel k][16 n + j] with
//                 eight ds_read_b32 per K-step and issues eight MFMAs into 32 accumulator registers; the flush is 32 atomic
//                 wave-instructions (4 records x 64 contiguous bytes each), issued BEHIND the next group's first loads so
//                 that the in-order vmcnt never parks a group behind its predecessor's atomics.
//
// 0 x NaN: a dense table multiplies pixels a record does not touch by zero.  Slab staging checks every value; a (tile, chunk)
// whose slab holds a non-finite value is processed by the exact sparse loop at the end of this file instead (same result as
// the vector kernels: NaN reaches exactly the records that touch the pixel).
#include <stdlib.h>

#include "gwbp_dev.h"

namespace gwbp {

namespace {

constexpr int kMaxSeg = 4096; // records of one tile sorted / grouped together (longer lists: several segments)
constexpr int kChunk = 128;
constexpr int kThreads = 1024;
// Slab rows are PADDED by 16 floats (576 B instead of 512): a K-step reads four pixel rows at once (one per 16 lanes), and
// with a 512-B pitch all four start on the same LDS bank (4-way conflict on every B read: SQ_LDS_BANK_CONFLICT was 2.2e8 of
// 4.8e8 LDS cycles per C2 view); at 576 B consecutive pixels are 16 banks apart.  147 KB of the CU's 160 KB.
constexpr int kRowFloats = kChunk + 16;
constexpr int kSlabFloats = kTilePix * kRowFloats;         // 36864 floats = 144 KB
constexpr size_t kLdsBytes = (size_t)kSlabFloats * 4 + 32; // slab + work counter + two item slots + non-finite flags
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32 shfl_xor_u(u32 v, int m) { return (u32)__shfl_xor((int)v, m, 64); }
// OR over the 16 lanes of a row (lanes 16 r .. 16 r + 15), result in every lane of the row
__device__ __forceinline__ u64 row_or(u64 v)
{
    u32 lo = (u32)v, hi = (u32)(v >> 32);
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
        lo |= shfl_xor_u(lo, m);
        hi |= shfl_xor_u(hi, m);
    }
    return ((u64)hi << 32) | lo;
}

// 8-bit sort key of a record: its pixel mask's bounding box in units of 4 pixels, (y0, x0) major, (y1, x1) minor.
// tools/block_density.py-style census on a C2 view: 0.38 of the dense table non-zero (depth order: 0.23).
__device__ __forceinline__ u32 bbox_key(const u64 (&m)[4])
{
    int qf = 0, ql = 0;
#pragma unroll
    for (int q = 3; q >= 0; --q)
        if (m[q])
            qf = q;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (m[q])
            ql = q;
    const u64 mf = m[qf] ? m[qf] : 1ull, ml = m[ql] ? m[ql] : 1ull;
    const int y0 = 4 * qf + (__builtin_ctzll(mf) >> 4), y1 = 4 * ql + ((63 - __builtin_clzll(ml)) >> 4);
    u64 c = m[0] | m[1] | m[2] | m[3];
    c |= c >> 32;
    c |= c >> 16;
    const u32 cols = ((u32)c & 0xFFFFu) | 0x10000u; // sentinel keeps ctz/clz defined
    const int x0 = min(__builtin_ctz(cols), 15), x1 = max(31 - __builtin_clz(cols & 0xFFFFu ? cols & 0xFFFFu : 1u), 0);
    return (u32)((((y0 >> 2) * 4 + (x0 >> 2)) << 4) | ((y1 >> 2) * 4 + (x1 >> 2)));
}

__global__ __launch_bounds__(64) void k_group_sort(const u32 *__restrict__ tile_offsets, const u32 *__restrict__ hdr_count,
                                                   const Header *__restrict__ headers, uint2 *__restrict__ tile_grp,
                                                   GrpInfo *__restrict__ grp_info, u32 *__restrict__ grp_gid,
                                                   u32 *__restrict__ grp_rec, u32 *__restrict__ pack_ctr, u32 grp_cap,
                                                   u32 blk_cap, Counters *__restrict__ ctr, int prio)
{
    front_priority(prio);
    __shared__ unsigned char s_key[kMaxSeg];
    __shared__ unsigned short s_order[kMaxSeg];
    __shared__ u32 s_hist[256];
    __shared__ u32 s_gks[kMaxSeg / kGrp]; // K-steps per group of the segment, then its block offset
    const int tile = blockIdx.x, lane = threadIdx.x;
    const u32 R = hdr_count[tile];
    const u32 hb = tile_offsets[tile];
    const u32 n_seg_full = R / kMaxSeg, rem = R % kMaxSeg;
    const u32 n_groups = n_seg_full * (kMaxSeg / kGrp) + (rem + kGrp - 1) / kGrp;
    u32 gbase = 0;
    if (lane == 0 && n_groups)
        gbase = atomicAdd(&pack_ctr[kPackGroups], n_groups);
    gbase = uniform(gbase);
    const bool fits = (u64)gbase + n_groups <= (u64)grp_cap;
    if (lane == 0) {
        tile_grp[tile] = make_uint2(gbase, fits ? n_groups : 0u);
        if (!fits)
            atomicOr(&ctr->overflow, kOverflowGroups);
    }
    if (!fits || n_groups == 0)
        return;
    u32 g0 = gbase;
    for (u32 seg0 = 0; seg0 < R; seg0 += kMaxSeg) {
        const u32 n = min((u32)kMaxSeg, R - seg0);
        const Header *hs = headers + hb + seg0;
        // (one wave: its LDS operations complete in program order, no barriers)
        for (int i = lane; i < 256; i += 64)
            s_hist[i] = 0;
        for (u32 i = lane; i < n; i += 64) {
            const u64 m[4] = {hs[i].mask[0], hs[i].mask[1], hs[i].mask[2], hs[i].mask[3]};
            const u32 key = bbox_key(m);
            s_key[i] = (unsigned char)key;
            atomicAdd(&s_hist[key], 1u);
        }
        { // exclusive scan of the 256 bins: lane l owns bins 4 l .. 4 l + 3
            const u32 c0 = s_hist[4 * lane], c1 = s_hist[4 * lane + 1], c2 = s_hist[4 * lane + 2], c3 = s_hist[4 * lane + 3];
            u32 incl = c0 + c1 + c2 + c3;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const u32 up = (u32)__shfl_up((int)incl, o, 64);
                if (lane >= o)
                    incl += up;
            }
            const u32 excl = incl - (c0 + c1 + c2 + c3);
            s_hist[4 * lane] = excl, s_hist[4 * lane + 1] = excl + c0, s_hist[4 * lane + 2] = excl + c0 + c1;
            s_hist[4 * lane + 3] = excl + c0 + c1 + c2;
        }
        for (u32 i = lane; i < n; i += 64) {
            const u32 pos = atomicAdd(&s_hist[s_key[i]], 1u);
            s_order[pos] = (unsigned short)i;
        }
        // groups of the segment, four per round: row r of the wave = group 4 it + r, lane i of the row = its i-th record
        const u32 ngs = (n + kGrp - 1) / kGrp;
        for (u32 it = 0; 4 * it < ngs; ++it) {
            const u32 gi = 4 * it + (u32)(lane >> 4);
            const u32 slot = kGrp * gi + (u32)(lane & 15);
            const bool valid = gi < ngs && slot < n;
            const u32 r = valid ? (u32)s_order[slot] : 0u;
            u64 m[4] = {0ull, 0ull, 0ull, 0ull};
            u32 gid = 0xFFFFFFFFu;
            if (valid) {
                const Header &h = hs[r];
                m[0] = h.mask[0], m[1] = h.mask[1], m[2] = h.mask[2], m[3] = h.mask[3];
                gid = h.gid;
            }
            u32 cnt = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                cnt += (u32)__popcll(row_or(m[q]));
            // empty slots of the last group: the Gaussian of the group's first record (their operand rows are all zero, so the
            // scatter's unconditional flush adds +0 to a row that exists)
            const u32 gid_first = (u32)__shfl((int)gid, lane & 48, 64);
            if (!valid)
                gid = gid_first;
            if (gi < ngs) {
                grp_gid[(size_t)(g0 + gi) * kGrp + (lane & 15)] = gid;
                grp_rec[(size_t)(g0 + gi) * kGrp + (lane & 15)] = valid ? hb + seg0 + r : 0xFFFFFFFFu;
                if ((lane & 15) == 0)
                    s_gks[gi] = (cnt + 3) >> 2;
            }
        }
        // blocks (4 K-steps): exclusive scan over the segment's groups (<= 256: lane l owns groups 4 l .. 4 l + 3), then ONE
        // allocation for the segment
        {
            u32 ks[4], nb[4], sum = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32 gi = 4 * lane + j;
                ks[j] = gi < ngs ? s_gks[gi] : 0u;
                nb[j] = (ks[j] + 3) >> 2;
                sum += nb[j];
            }
            u32 incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const u32 up = (u32)__shfl_up((int)incl, o, 64);
                if (lane >= o)
                    incl += up;
            }
            const u32 total = (u32)__shfl((int)incl, 63, 64);
            u32 base = 0;
            if (lane == 0 && total)
                base = atomicAdd(&pack_ctr[kPackBlocks], total);
            base = uniform(base);
            const bool ok = (u64)base + total <= (u64)blk_cap;
            if (!ok && lane == 0)
                atomicOr(&ctr->overflow, kOverflowGroups);
            u32 off = base + incl - sum;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32 gi = 4 * lane + j;
                if (gi < ngs) {
                    GrpInfo info;
                    info.blk_off = ok ? off : 0u, info.n_ks = ok ? ks[j] : 0u; // dropped on overflow: the view is invalid anyway
                    grp_info[g0 + gi] = info;
                }
                off += nb[j];
            }
        }
        g0 += ngs;
    }
}

constexpr int kPackWaves = 4;

__global__ __launch_bounds__(kPackWaves * 64) void k_pack(const Header *__restrict__ headers, const WPair *__restrict__ wpool,
                                                         const GrpInfo *__restrict__ grp_info, const u32 *__restrict__ grp_rec,
                                                         const u32 *__restrict__ pack_ctr, float4 *__restrict__ apool,
                                                         u32 *__restrict__ kpix, u32 grp_cap, int prio)
{
    front_priority(prio);
    __shared__ unsigned char s_pix_all[kPackWaves][kTilePix + 16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i16 = lane & 15, k4 = lane >> 4;
    unsigned char *s_pix = s_pix_all[wave];
    const u32 n_groups = min(pack_ctr[kPackGroups], grp_cap);
    for (u32 g = blockIdx.x * kPackWaves + wave; g < n_groups; g += gridDim.x * kPackWaves) {
        const u32 blk_off = uniform(grp_info[g].blk_off), n_ks = uniform(grp_info[g].n_ks);
        if (n_ks == 0)
            continue;
        const u32 rec = grp_rec[(size_t)g * kGrp + i16];
        u64 m[4] = {0ull, 0ull, 0ull, 0ull};
        u32 wo[4] = {0u, 0u, 0u, 0u};
        if (rec != 0xFFFFFFFFu) {
            const Header &h = headers[rec];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                m[q] = h.mask[q], wo[q] = h.woff[q];
        }
        // the union's pixels in ascending order -> s_pix[0 .. cnt); padded with the last pixel up to whole blocks
        u32 cnt = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u64 U = uniform64(row_or(m[q]));
            if ((U >> lane) & 1ull)
                s_pix[cnt + mbcnt(U)] = (unsigned char)(64 * q + lane);
            cnt += (u32)__popcll(U);
        }
        const u32 n_blk = (n_ks + 3) >> 2;
        {
            const unsigned char last = s_pix[cnt - 1];
            for (u32 p = cnt + lane; p < n_blk * 16; p += 64)
                s_pix[p] = last;
        }
        // (two blocks per round -- eight gathered weights in flight per lane -- on twice as many waves was measured SLOWER:
        // 0.83 against 0.60 ms per C2 view; the kernel moves ~2 GB, half of it the dense table it writes)
        for (u32 blk = 0; blk < n_blk; ++blk) {
            float av[4];
            u32 pk = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const u32 pos = 16 * blk + 4 * t + k4; // K-step 4 blk + t, slot k4
                const u32 p = s_pix[pos];
                const u32 q = p >> 6, b = p & 63u;
                const u64 mm = q == 0 ? m[0] : q == 1 ? m[1] : q == 2 ? m[2] : m[3];
                const u32 w0 = q == 0 ? wo[0] : q == 1 ? wo[1] : q == 2 ? wo[2] : wo[3];
                const bool has = pos < cnt && ((mm >> b) & 1ull);
                const u32 idx = (u32)__popcll(mm & ((1ull << b) - 1ull));
                av[t] = has ? wpool[w0 + idx].w : 0.f;
                pk |= p << (8 * t);
            }
            apool[(size_t)(blk_off + blk) * 64 + lane] = make_float4(av[0], av[1], av[2], av[3]);
            if (i16 == 0)
                kpix[(size_t)(blk_off + blk) * 4 + k4] = pk;
        }
    }
}

__device__ __forceinline__ float lds_read_b32(u32 a)
{
#if __HIP_DEVICE_COMPILE__
    return *reinterpret_cast<const __attribute__((address_space(3))) float *>((size_t)a);
#else
    (void)a;
    return 0.f;
#endif
}

constexpr int kPF = 3; // blocks (of 4 K-steps) in flight ahead of the MFMAs

// compiler-only fence: memory operations are neither moved across it nor merged over it (no instruction is emitted)
__device__ __forceinline__ void order_fence() { asm volatile("" ::: "memory"); }

struct Slot { // one block of a group's operand stream: A of 4 K-steps (this lane's float4) and the 4 pixel bytes of its k slot
    float4 a;
    u32 p;
};

// (grp_gid, apool, kpix and F are deliberately NOT __restrict__: hipcc may then not move their loads across the
// order_fence()s or the atomics, which is what keeps the counted waits of the operand stream exact)
// (waves_per_eu(5, 5): 96 VGPRs instead of 118 -- the spilled values are per-thread constants reloaded at item and group
// boundaries; alone the kernel loses 0.1 ms, in the two-stream pipeline the front stage gets its second wave per SIMD back:
// 5.5 -> 4.9 ms per C2 view)
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_scatter_mfma(ViewDev V, int n_chunks, const uint2 *__restrict__ tile_grp,
                                                          const GrpInfo *__restrict__ grp_info, const u32 *grp_gid,
                                                          const float4 *apool, const u32 *kpix,
                                                          const u32 *__restrict__ tile_offsets, const u32 *__restrict__ hdr_count,
                                                          const Header *__restrict__ headers, const WPair *__restrict__ wpool,
                                                          FeatMap M, int D, float scale_f, float *F,
                                                          u32 *__restrict__ queues, Counters *__restrict__ ctr, int dbg_arg)
{
#ifdef GWBP_PROFILE
    const int dbg = dbg_arg; // ablation bits (PROFILE build only): 1 plain stores for the atomics, 2 no MFMA / LDS work, 4 no staging
#else
    constexpr int dbg = 0;   // the product kernel must not even contain the branches: they would break the counted waits
    (void)dbg_arg;
#endif
    // The group tables exist only if THIS view was blended with GWBP_FLAG_GROUP_SCATTER: refuse otherwise (F untouched,
    // overflow bit 2 raised), like k_scatter_wide
    if (uniform(ctr->blend_kind) != kBlendGroups) {
        if (blockIdx.x == 0 && threadIdx.x == 0)
            atomicOr(&ctr->overflow, kOverflowMismatch);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) float lds[];
    u32 *s_next = reinterpret_cast<u32 *>(lds + kSlabFloats);
    u32 *s_item = s_next + 1; // two slots: iteration k reads [k & 1], thread 0 fills [(k + 1) & 1] meanwhile
    u32 *s_bad = s_next + 3;  // [k & 1]: item k's slab holds a non-finite value (two flags: thread 0 clears the NEXT item's
                              // flag while the current item runs -- clearing the current one would race with the staging
                              // waves that are already raising it)
    const float *__restrict__ feats = M.p;

    // persistent workgroups, per-XCD-class queues: as k_scatter_full
    const u32 xcls = blockIdx.x & 7u;
    const int n_tiles = V.tile_w * V.tile_h;
    const u32 n_items = (u32)((n_tiles - (int)xcls + 7) / 8) * (u32)n_chunks;
    u32 *queue = queues + xcls * 16;
    const int lane = threadIdx.x & 63;
    const int j16 = lane & 15, k4 = lane >> 4;
    const u32 lane_col = (u32)j16 * 4u;
    if (threadIdx.x == 0)
        s_item[0] = atomicAdd(queue, 1u), s_bad[0] = 0u, s_bad[1] = 0u;
    __syncthreads();
    for (u32 k = 0;; ++k) {
    const u32 item = uniform(s_item[k & 1u]);
    if (item >= n_items)
        break;
    const int chunk = (int)(item % (u32)n_chunks);
    const int tile = (int)((item / (u32)n_chunks) * 8u + xcls);
    const uint2 tg = tile_grp[tile];
    const u32 g_base = uniform(tg.x), n_grp = uniform(tg.y);
    const int tx = tile % V.tile_w, ty = tile / V.tile_w;
    const int c0 = chunk * kChunk;
    if (threadIdx.x == 0)
        *s_next = 0, s_bad[(k + 1u) & 1u] = 0u;
    u32 nxt = 0;
    if (threadIdx.x == 0)
        nxt = atomicAdd(queue, 1u); // claim the next item; the value is only needed after the slab is staged
    if (n_grp != 0 && !(dbg & 4)) {
        // stage the 256 px x 128 ch slab: 32 float4 per pixel row, eight 16-B loads per thread in flight
        constexpr int vpr = kChunk >> 2;
        constexpr int kIt = kTilePix * vpr / kThreads; // 8
        float4 vals[kIt];
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int idx = it * kThreads + threadIdx.x;
            const int p = idx / vpr, v = idx - p * vpr;
            const int ix = tx * kTile + (p & 15), iy = ty * kTile + (p >> 4);
            // pixels past the image edge are never referenced by an entry: load a clamped (valid) address
            vals[it] = *reinterpret_cast<const float4 *>(feats + (int64_t)min(iy, V.H - 1) * M.fs_y +
                                                         (int64_t)min(ix, V.W - 1) * M.fs_x + c0 + 4 * v);
        }
        bool bad = false;
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int idx = it * kThreads + threadIdx.x;
            const int p = idx / vpr, v = idx - p * vpr;
            *reinterpret_cast<float4 *>(lds + p * kRowFloats + 4 * v) = vals[it];
            // x - x is 0 for every finite x and NaN for +-inf / NaN
            const float z = (vals[it].x - vals[it].x) + (vals[it].y - vals[it].y) + (vals[it].z - vals[it].z) +
                            (vals[it].w - vals[it].w);
            bad |= !(z == 0.f);
        }
        if (__ballot(bad) != 0ull && lane == 0)
            atomicOr(&s_bad[k & 1u], 1u);
    }
    if (threadIdx.x == 0)
        s_item[(k + 1u) & 1u] = nxt;
    __syncthreads();

    auto claim = [&]() __attribute__((always_inline)) -> u32 { // one lane, one LDS atomic (see k_scatter_full)
        u32 h = 0;
        if (lane == 0) {
            const u32 addr = (u32)(kSlabFloats * sizeof(float)), one = 1u;
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(h) : "v"(addr), "v"(one) : "memory");
        }
        return uniform(h);
    };

    if (n_grp != 0 && uniform(s_bad[k & 1u]) == 0u) {
        // ---- block-sparse path ---------------------------------------------------------------------------------------
        // The operand stream of a wave is SEAMLESS across its groups.  Slot u of sl[] holds block (u mod kPF) of the current
        // group and is refilled, right after its block has been consumed, with the block kPF further on -- which in a group's
        // LAST round is block u of the NEXT group (whose header was fetched two groups ago): a group boundary issues no
        // stream load, nothing waits for a cold first block, and no slot is written while an older load to it is in flight.
        // Slots are never rotated (a register move would wait for every load in flight).
        // vmcnt retires in order and an atomic stays counted for ~3000 cycles under load (scatter_full.hip), so the waits
        // for the stream loads must be COUNTED; the code is shaped so that hipcc's own wait insertion counts exactly:
        // every sub-step issues exactly two loads (arithmetic of the blocks past the group's last one is skipped, the
        // loads are not), a boundary issues [gid of the next group] [32 atomics] (+ a scalar header load), always; a
        // group's first round is a separate copy of the code (its loads have a boundary behind them), and so is the first
        // group of an item (they do not).
        Slot sl[kPF];
        // group headers are fetched TWO groups ahead with SCALAR loads (wave-uniform address): they are consumed at a
        // boundary, where the K loop's LDS reads have long drained and lgkmcnt(0) costs nothing -- a vector load would have
        // to be awaited with vmcnt, and hipcc's count for it collapses across the rounds loop (it emitted vmcnt(2), i.e. a
        // wait for the next group's first blocks, at every boundary)
        auto fetch_info = [&](u32 gg) __attribute__((always_inline)) -> uint2 {
            const GrpInfo gi = grp_info[gg];
            return make_uint2(gi.blk_off, gi.n_ks);
        };
        auto claim_group = [&](bool &ok) __attribute__((always_inline)) -> u32 {
            const u32 hh = claim();
            ok = hh < n_grp;
            return g_base + min(hh, n_grp - 1); // past the end: the last group again (loaded, never computed or flushed)
        };
        bool have, have2, have3;
        const u32 g1 = claim_group(have);
        const uint2 i1 = fetch_info(g1);
        const u32 g2f = claim_group(have2);
        const uint2 i2 = fetch_info(g2f);
        u32 g3 = claim_group(have3);
        uint2 info3 = fetch_info(g3);
        u32 n_blk = max((uniform(i1.y) + 3u) >> 2, 1u), n_blk2 = max((uniform(i2.y) + 3u) >> 2, 1u);
        const float4 *ap = apool + (size_t)uniform(i1.x) * 64 + lane, *ap2 = apool + (size_t)uniform(i2.x) * 64 + lane;
        const u32 *pp = kpix + (size_t)uniform(i1.x) * 4 + k4, *pp2 = kpix + (size_t)uniform(i2.x) * 4 + k4;
        uint4 gid4 = reinterpret_cast<const uint4 *>(grp_gid + (size_t)g1 * kGrp)[k4];
        u32 g2 = g2f;
#pragma unroll
        for (int u = 0; u < kPF; ++u) {
            const u32 bb = min((u32)u, n_blk - 1);
            sl[u].a = ap[(size_t)bb * 64], sl[u].p = pp[(size_t)bb * 4];
        }
        order_fence();
        u32 exp_counter = (u32)(threadIdx.x >> 6); // (experiment builds only)
        (void)exp_counter;
        auto group = [&]() __attribute__((always_inline)) {
            const u32 cur_blk = n_blk;
            f32x4_t acc[8];
#pragma unroll
            for (int n = 0; n < 8; ++n)
                acc[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            auto round = [&](u32 blk0) __attribute__((always_inline)) {
                const bool last_round = blk0 + kPF >= cur_blk; // wave-uniform
#pragma unroll
                for (int u = 0; u < kPF; ++u) {
                    const float4 a4 = sl[u].a;
                    const u32 p4 = sl[u].p;
#ifdef GWBP_NO_COMPUTE // experiment build only
                    if (false) {
#else
                    if (blk0 + u < cur_blk && !(dbg & 2)) { // wave-uniform; no memory operation inside
#endif
                        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
                        // all four K-steps of a block run: k_pack zero-fills the K-steps past the group's last one.
                        // (Issuing the eight B reads of K-step t + 1 ahead of the MFMAs of K-step t -- two register sets,
                        // scheduling barriers -- was measured SLOWER: 3.91 against 3.78 ms per C2 view, 128 VGPRs + spills.)
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const u32 row = ((p4 >> (8 * t)) & 255u) * (u32)(kRowFloats * 4) + lane_col; // padded pixel row
                            float b[8];
#pragma unroll
                            for (int n = 0; n < 8; ++n)
                                b[n] = lds_read_b32(row + 64u * n);
#pragma unroll
                            for (int n = 0; n < 8; ++n)
                                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], b[n], acc[n], 0, 0, 0);
                        }
                    }
                    // refill: the block kPF further on -- of this group, or block u of the next one
                    const float4 *rp = last_round ? ap2 + (size_t)min((u32)u, n_blk2 - 1) * 64
                                                  : ap + (size_t)min(blk0 + u + kPF, cur_blk - 1) * 64;
                    const u32 *rq = last_round ? pp2 + (size_t)min((u32)u, n_blk2 - 1) * 4
                                               : pp + (size_t)min(blk0 + u + kPF, cur_blk - 1) * 4;
                    sl[u].a = *rp, sl[u].p = *rq;
                    order_fence(); // the refill is issued HERE (hipcc would otherwise sink or hoist it across blocks)
                }
            };
            round(0);
            for (u32 blk0 = kPF; blk0 < cur_blk; blk0 += kPF)
                round(blk0);
            // boundary: [32 atomics of this group] [gid of the next group] (+ a scalar header load), always.  Exactly 32
            // atomics, unconditional: empty slots of a tile's last group carry a real Gaussian id and an all-zero operand
            // row (they add +0).  D register v of block n = record 4 k4 + v, channel 16 n + j16 -> 4 records x 64
            // contiguous bytes per instruction.
            order_fence();
            const u32 gv[4] = {gid4.x, gid4.y, gid4.z, gid4.w};
#ifdef GWBP_FEWER_FLUSHES // experiment builds only (results invalid): 3 of 8 groups (1, 2) or all of them (3) "flush" with
                          // plain stores -- same VMEM count; 2 and 3: into one L2-resident row per workgroup
            const bool plain = GWBP_FEWER_FLUSHES == 3 || (exp_counter++ & 7u) >= 5u;
#else
            constexpr bool plain = false;
#endif
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float *Fg = F + (int64_t)gv[v] * D + c0 + j16;
#ifdef GWBP_FEWER_FLUSHES
#if GWBP_FEWER_FLUSHES >= 2 // ... and those stores go to one L2-resident line set per workgroup: no memory-side cost at all
                if (plain)
                    Fg = F + (int64_t)(blockIdx.x & 255u) * D + c0 + j16;
#endif
#endif
                if (!(dbg & 1) && !plain) {
#pragma unroll
                    for (int n = 0; n < 8; ++n)
                        atomicAdd(Fg + 16 * n, acc[n][v] * scale_f);
                } else { // ablation (PROFILE build): same VMEM count, no atomics
#pragma unroll
                    for (int n = 0; n < 8; ++n)
                        __builtin_nontemporal_store(acc[n][v] * scale_f, Fg + 16 * n);
                }
            }
            order_fence();
            // the next group becomes the current one, the group after it the next one
            have = have2, have2 = have3;
            n_blk = n_blk2, ap = ap2, pp = pp2;
            gid4 = reinterpret_cast<const uint4 *>(grp_gid + (size_t)g2 * kGrp)[k4];
            g2 = g3;
            {
                const u32 off3 = uniform(info3.x);
                n_blk2 = max((uniform(info3.y) + 3u) >> 2, 1u);
                ap2 = apool + (size_t)off3 * 64 + lane, pp2 = kpix + (size_t)off3 * 4 + k4;
            }
            g3 = claim_group(have3);
            info3 = fetch_info(g3);
            order_fence();
        };
        if (have) {
            group(); // peeled: no boundary between its stream loads and its first round
            while (have)
                group();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the last stream loads still target sl[]'s registers
    } else if (n_grp != 0) {
        // ---- exact sparse path for a slab with non-finite values: one record per wave at a time, lane = 2 channels ------
        const u32 nh = hdr_count[tile];
        const Header *hbase = headers + tile_offsets[tile];
        for (u32 h = claim(); h < nh; h = claim()) {
            const Header *hp = hbase + h;
            const u32 gid = uniform(hp->gid), woff = uniform(hp->woff[0]), c = uniform(hp->counts);
            const u32 T = (c & 0xFFu) + ((c >> 8) & 0xFFu) + ((c >> 16) & 0xFFu) + (c >> 24);
            float2 acc = make_float2(0.f, 0.f);
            for (u32 e = 0; e < T; ++e) { // wave-uniform entry: a broadcast load
                const WPair wp = wpool[woff + e];
                const float2 f = *reinterpret_cast<const float2 *>(lds + wp.pix * kRowFloats + 2 * lane);
                acc.x = __builtin_fmaf(wp.w, f.x, acc.x);
                acc.y = __builtin_fmaf(wp.w, f.y, acc.y);
            }
            float *Fg = F + (int64_t)gid * D + c0 + 2 * lane;
            atomicAdd(Fg, acc.x * scale_f);
            atomicAdd(Fg + 1, acc.y * scale_f);
        }
    }
    __syncthreads(); // every wave is done with this slab and work counter
    } // item loop
    // the last workgroup of the class to leave re-arms the queue (see k_scatter_full)
    if (threadIdx.x == 0) {
        const u32 left = atomicAdd(queue + 1, 1u);
        if (left == gridDim.x / 8u - 1u) {
            atomicExch(queue + 1, 0u);
            atomicExch(queue, 0u);
        }
    }
}

} // namespace

int launch_pack_groups(const Layout &L, const Ws &W, const ViewDev &V, hipStream_t s)
{
    const int prio = (L.flags & GWBP_FLAG_FRONT_PRIORITY) ? 1 : 0;
    const int n_tiles = V.tile_w * V.tile_h;
    int n_cu = 0;
    const int rc = device_cus(&n_cu);
    if (rc)
        return rc;
    hipLaunchKernelGGL(k_group_sort, dim3(n_tiles), dim3(64), 0, s, W.tile_offsets, W.hdr_count, W.headers, W.tile_grp,
                       W.grp_info, W.grp_gid, W.grp_rec, W.pack_ctr, (u32)L.grp_cap, (u32)L.blk_cap, W.counters, prio);
    hipLaunchKernelGGL(k_pack, dim3(n_cu * 4), dim3(kPackWaves * 64), 0, s, W.headers, W.wpool, W.grp_info, W.grp_rec,
                       W.pack_ctr, W.apool, W.kpix, (u32)L.grp_cap, prio);
    return check_hip(hipGetLastError(), "group packing launch");
}

bool scatter_mfma_takes(const FeatMap &M, int D)
{
    return D % kChunk == 0 && M.fs_c == 1 && (M.fs_x % 4 == 0) && (M.fs_y % 4 == 0) &&
           ((reinterpret_cast<uintptr_t>(M.p) & 15) == 0) && !M.ymap && !M.xmap && !M.bilinear() && !M.enc;
}

int launch_scatter_mfma(const Layout &L, const Ws &W, const ViewDev &V, const FeatMap &M, int D, float scale_f, float *F,
                        hipStream_t s)
{
    int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(k_scatter_mfma), (int)kLdsBytes, 9);
    if (rc)
        return rc;
    int n_cu = 0;
    if ((rc = device_cus(&n_cu)))
        return rc;
    int grid = L.scatter_wgs > 0 ? L.scatter_wgs : n_cu;
    grid = (grid + 7) & ~7;
    u32 *queues = W.shards + kShards * 16;
    hipLaunchKernelGGL(k_scatter_mfma, dim3(grid), dim3(kThreads), kLdsBytes, s, V, D / kChunk, W.tile_grp, W.grp_info,
                       W.grp_gid, W.apool, W.kpix, W.tile_offsets, W.hdr_count, W.headers, W.wpool, M, D, scale_f, F, queues,
                       W.counters, profile_knob("GWBP_ABLATE"));
    return check_hip(hipGetLastError(), "scatter_mfma launch");
}

} // namespace gwbp
