// sort.hip -- two-level stable LSD radix sort producing gsplat's (tile, depth, index) order, and the per-tile
// offset table.
//
// Replaces cub::DeviceRadixSort::SortPairs on 64-bit (tile << 32 | depth) keys + isect_offset_encode inside gsplat
// 1.4.0's rasterization():
//   level 1  the N Gaussians are sorted ONCE by depth bits (4 byte-passes over N 32-bit keys; culled ones last)
//   emit     intersections are emitted in that order (project.hip: k_emit), key = tile id
//   level 2  stable sort of the intersections by tile id only (ceil(tile_bits / 8) = 2 byte-passes at 1600x1060)
// -> (tile, depth, index) order with 2 passes over the 4.4 M intersections instead of 6 passes over 64-bit keys.
// The number of intersections lives in device memory (Counters::n_isect), so the level-2 kernels are launched for
// the caller's capacity and idle blocks exit at once: no host read-back between projection and blend.
//
// Per 8-bit pass: (1) k_hist   - per-block digit histogram, hist[digit][block]
//                 (2) k_scan   - one workgroup per digit: exclusive scan over blocks + digit total
//                 (3) k_radix_scatter - wave-striped stable ranking (ballot "match" per digit, per-wave LDS
//                                counters); only the ranks survive the barriers, keys and values are re-read for the scatter.
//
// Round 6 built the "front diet" the round-5 review asked for and MEASURED it against this form on one box
// (profiles/r6_front_diet.txt): per LEVEL one k_hist_all that reads the keys once and produces the digit totals of every pass,
// then ONE kernel per pass, k_onesweep -- the same ranking, but a block's base inside each digit comes from a DECOUPLED
// LOOK-BACK over the blocks in front of it (status words {flag, count} per (block, digit), block index from a ticket) instead
// of from a histogram kernel and a scan kernel: 18 sort launches per view -> 8, a third of the key reads.  Same order bit for bit
// (the whole GPU suite passes on it), and SLOWER: every pass's 245 (depth sort) / ~880 (tile sort) blocks are resident at the
// same time on this chip, so the look-back degenerates into a serial wavefront over the blocks -- k_onesweep 165 us per pass
// beside the scatter kernel against 94 + 21 + 8 us for the three kernels, the C2 step 3.49-3.50 against 3.45-3.48 ms/view, C5
// 1.33-1.36 against 1.33-1.34, alone +0.05 ms per view.  Halving the front's launches does not move the step; what the front
// stage costs is the work of its kernels (k_blend above all), not their number.  The code stays behind -DGWBP_SORT_ONESWEEP
// (make VARIANT=onesweep EXTRA=-DGWBP_SORT_ONESWEEP) as the record of that measurement.
#include "gwbp_dev.h"

namespace gwbp {

constexpr int kSortThreads = 256;
constexpr int kItemsPerThread = kSortItems / kSortThreads; // 16
constexpr int kWaves = kSortThreads / 64;
constexpr int kWaveItems = kSortItems / kWaves; // 1024 keys per wave segment

__global__ __launch_bounds__(kSortThreads) void k_hist(const u32 *__restrict__ keys, const u32 *__restrict__ n_dev,
                                                       u32 n_host, int shift, int nblk, u32 *__restrict__ hist, int prio)
{
    front_priority(prio);
    const u32 n = n_dev ? *n_dev : n_host;
    const u32 base = blockIdx.x * (u32)kSortItems;
    // The level-2 passes are launched for the caller's CAPACITY (the intersection count lives on the device): a block beyond
    // the data leaves at once, without touching the histogram -- k_scan stops at the last block that has data (round 5: the
    // capacity is 16 N, 4-7 x the data; idle blocks used to write 256 zeros each and k_scan to scan them, 78 us per pass at C4)
    if (base >= n)
        return;
    __shared__ u32 s_h[256];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    {
#pragma unroll 4
        for (int it = 0; it < kItemsPerThread; ++it) {
            const u32 idx = base + it * kSortThreads + threadIdx.x;
            if (idx < n)
                atomicAdd(&s_h[(u32)(keys[idx] >> shift) & 0xFFu], 1u);
        }
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblk + blockIdx.x] = s_h[threadIdx.x];
}

// block d: exclusive scan of hist[d][0..nact) in place, digit_total[d] = sum; nact = the blocks that hold data
__global__ __launch_bounds__(256) void k_scan(int nblk_all, const u32 *__restrict__ n_dev, u32 n_host, u32 *__restrict__ hist,
                                              u32 *__restrict__ digit_total, int prio)
{
    front_priority(prio);
    u32 *row = hist + (size_t)blockIdx.x * nblk_all; // (the row pitch stays the capacity's)
    const u32 n = n_dev ? *n_dev : n_host;
    const int nblk = (int)min((u32)nblk_all, (n + (u32)kSortItems - 1u) / (u32)kSortItems);
    __shared__ u32 s_wave[4];
    __shared__ u32 s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0)
        s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += 256) {
        const int i = base + threadIdx.x;
        const u32 v = (i < nblk) ? row[i] : 0u;
        u32 incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 t = __shfl_up(incl, o, 64);
            if (lane >= o)
                incl += t;
        }
        if (lane == 63)
            s_wave[wave] = incl;
        __syncthreads();
        u32 woff = 0;
        for (int w = 0; w < wave; ++w)
            woff += s_wave[w];
        const u32 carry = s_carry;
        if (i < nblk)
            row[i] = carry + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 255)
            s_carry = carry + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0)
        digit_total[blockIdx.x] = s_carry;
}

__global__ __launch_bounds__(kSortThreads) void k_radix_scatter(const u32 *__restrict__ keys_in,
                                                                const u32 *__restrict__ vals_in,
                                                                u32 *__restrict__ keys_out,
                                                                u32 *__restrict__ vals_out,
                                                                const u32 *__restrict__ n_dev, u32 n_host, int shift,
                                                                int nblk, const u32 *__restrict__ hist,
                                                                const u32 *__restrict__ digit_total, int prio)
{
    front_priority(prio);
    const u32 n = n_dev ? *n_dev : n_host;
    const u32 base = blockIdx.x * (u32)kSortItems;
    if (base >= n)
        return;
    __shared__ u32 s_cnt[kWaves][256]; // per-wave running digit counters, then per-wave bases
    __shared__ u32 s_start[256];       // global start of each digit (exclusive scan of digit_total)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

#pragma unroll
    for (int w = 0; w < kWaves; ++w)
        s_cnt[w][threadIdx.x] = 0;
    { // exclusive scan of the 256 digit totals
        const u32 v = digit_total[threadIdx.x];
        u32 incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 t = __shfl_up(incl, o, 64);
            if (lane >= o)
                incl += t;
        }
        __shared__ u32 s_w[kWaves];
        if (lane == 63)
            s_w[wave] = incl;
        __syncthreads();
        u32 woff = 0;
        for (int w = 0; w < wave; ++w)
            woff += s_w[w];
        s_start[threadIdx.x] = woff + incl - v;
    }
    __syncthreads();

    // Only the rank of an item survives the two barriers; keys (hence digits) and values are re-read for the scatter (the
    // block's 16 KB are L2-resident): 60 VGPRs.  Holding key[16] + val[16] + rank[16] put the kernel at 106: one block per
    // CU beside the persistent scatter workgroup, where each pass then took 200-260 us instead of 30.  (Ranks parked in
    // LDS as u16 instead -- 19 VGPRs, 13 KB -- were measured too: 4 % slower alone, and no faster step beside the fused
    // blend+scatter kernel, whose four 120-VGPR waves per SIMD leave this kernel no register room.)
    u32 r_rank[kItemsPerThread];
    const u32 seg = base + wave * (u32)kWaveItems;
    const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int it = 0; it < kItemsPerThread; ++it) {
        const u32 idx = seg + it * 64 + lane;
        const bool valid = idx < n;
        const u32 key = valid ? keys_in[idx] : ~0u;
        const u32 dg = (u32)(key >> shift) & 0xFFu;
        u64 peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (dg >> b) & 1u;
            const u64 m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        u32 old = 0;
        const int leader = valid ? (__ffsll((long long)peers) - 1) : 0;
        if (valid && lane == leader) {
            old = s_cnt[wave][dg];
            s_cnt[wave][dg] = old + (u32)__popcll(peers);
        }
        old = __shfl(old, leader, 64);
        r_rank[it] = old + (u32)__popcll(peers & lt);
    }
    __syncthreads();
    { // thread = digit: turn per-wave counts into global bases
        const u32 d = threadIdx.x;
        u32 run = s_start[d] + hist[(size_t)d * nblk + blockIdx.x];
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const u32 c = s_cnt[w][d];
            s_cnt[w][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kItemsPerThread; ++it) {
        const u32 idx = seg + it * 64 + lane;
        if (idx < n) {
            const u32 key = keys_in[idx];
            const u32 pos = s_cnt[wave][(key >> shift) & 0xFFu] + r_rank[it];
            keys_out[pos] = key;
            vals_out[pos] = vals_in[idx];
        }
    }
}


// ---- round 6 experiment (-DGWBP_SORT_ONESWEEP): one histogram kernel per level + one look-back kernel per pass --------------
#ifdef GWBP_SORT_ONESWEEP
constexpr u64 kFlagAggregate = 1ull << 32, kFlagInclusive = 2ull << 32; // status word = flag << 32 | count (counts reach 2^32 - 4096)

// Digit totals of ALL passes of a level from one read of the keys: per-block LDS histograms (passes x 256 bins), non-zero bins
// added to the level's tables with global atomics (the tables live in the region gwbp_project's memset clears, and the last
// kernel of gwbp_bin_sort clears them again).  Each active block also zeroes its row of status buffer 0 for the first pass.
__global__ __launch_bounds__(kSortThreads) void k_hist_all(const u32 *__restrict__ keys, const u32 *__restrict__ n_dev, u32 n_host,
                                                           int passes, u32 *__restrict__ totals, u64 *__restrict__ status0, int prio)
{
    front_priority(prio);
    const u32 n = n_dev ? *n_dev : n_host;
    const u32 base = blockIdx.x * (u32)kSortItems;
    if (base >= n)
        return; // launched for the capacity: blocks beyond the data leave at once
    __shared__ u32 s_h[kMaxPasses][256];
#pragma unroll
    for (int p = 0; p < kMaxPasses; ++p)
        s_h[p][threadIdx.x] = 0;
    status0[(size_t)blockIdx.x * 256 + threadIdx.x] = 0ull;
    __syncthreads();
#pragma unroll 4
    for (int it = 0; it < kItemsPerThread; ++it) {
        const u32 idx = base + it * kSortThreads + threadIdx.x;
        if (idx < n) {
            const u32 key = keys[idx];
            for (int p = 0; p < passes; ++p)
                atomicAdd(&s_h[p][(key >> (8 * p)) & 0xFFu], 1u);
        }
    }
    __syncthreads();
    for (int p = 0; p < passes; ++p) {
        const u32 c = s_h[p][threadIdx.x];
        if (c)
            atomicAdd(&totals[p * 256 + threadIdx.x], c);
    }
}

// One 8-bit pass.  totals = this pass's 256 digit totals; status = this pass's look-back words [block][digit] (zero on entry),
// status_next = the other buffer, whose row this block zeroes for the next pass; ticket = this pass's block counter.
__global__ __launch_bounds__(kSortThreads) void k_onesweep(const u32 *__restrict__ keys_in, const u32 *__restrict__ vals_in,
                                                           u32 *__restrict__ keys_out, u32 *__restrict__ vals_out,
                                                           const u32 *__restrict__ n_dev, u32 n_host, int shift,
                                                           const u32 *__restrict__ totals, u64 *__restrict__ status,
                                                           u64 *__restrict__ status_next, u32 *__restrict__ ticket, int prio)
{
    front_priority(prio);
    const u32 n = n_dev ? *n_dev : n_host;
    if (blockIdx.x * (u32)kSortItems >= n)
        return; // exactly ceil(n / kSortItems) blocks stay and take tickets 0 .. that - 1
    __shared__ u32 s_cnt[kWaves][256]; // per-wave running digit counters, then per-wave bases
    __shared__ u32 s_start[256];       // global start of each digit (exclusive scan of the digit totals)
    __shared__ u32 s_tile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0)
        s_tile = atomicAdd(ticket, 1u); // the block's index = its place in the START order: everything in front of it runs or is done
#pragma unroll
    for (int w = 0; w < kWaves; ++w)
        s_cnt[w][threadIdx.x] = 0;
    { // exclusive scan of the 256 digit totals
        const u32 v = totals[threadIdx.x];
        u32 incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 t = __shfl_up(incl, o, 64);
            if (lane >= o)
                incl += t;
        }
        __shared__ u32 s_w[kWaves];
        if (lane == 63)
            s_w[wave] = incl;
        __syncthreads();
        u32 woff = 0;
        for (int w = 0; w < wave; ++w)
            woff += s_w[w];
        s_start[threadIdx.x] = woff + incl - v;
    }
    __syncthreads();
    const u32 tile = s_tile;
    const u32 base = tile * (u32)kSortItems;
    status_next[(size_t)tile * 256 + threadIdx.x] = 0ull;

    // (ranking exactly as in k_radix_scatter: only the ranks survive the barriers, keys and values are re-read for the scatter)
    u32 r_rank[kItemsPerThread];
    const u32 seg = base + wave * (u32)kWaveItems;
    const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int it = 0; it < kItemsPerThread; ++it) {
        const u32 idx = seg + it * 64 + lane;
        const bool valid = idx < n;
        const u32 key = valid ? keys_in[idx] : ~0u;
        const u32 dg = (u32)(key >> shift) & 0xFFu;
        u64 peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (dg >> b) & 1u;
            const u64 m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        u32 old = 0;
        const int leader = valid ? (__ffsll((long long)peers) - 1) : 0;
        if (valid && lane == leader) {
            old = s_cnt[wave][dg];
            s_cnt[wave][dg] = old + (u32)__popcll(peers);
        }
        old = __shfl(old, leader, 64);
        r_rank[it] = old + (u32)__popcll(peers & lt);
    }
    __syncthreads();
    { // thread = digit: publish this block's count, look back for the blocks in front, turn per-wave counts into global bases
        const u32 d = threadIdx.x;
        u32 c[kWaves], mine = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            c[w] = s_cnt[w][d];
            mine += c[w];
        }
        u64 *row = status + d;
        __hip_atomic_store(row + (size_t)tile * 256, (tile == 0 ? kFlagInclusive : kFlagAggregate) | (u64)mine, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        u32 excl = 0;
        if (tile != 0) {
            // four predecessors' words in flight at a time; an empty word (its block has not ranked yet) is polled again
            long long j = (long long)tile - 1;
            bool done = false;
            while (!done) {
                u64 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    v[u] = (j - u >= 0) ? __hip_atomic_load(row + (size_t)(j - u) * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                        : kFlagInclusive;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (done)
                        break;
                    while ((v[u] >> 32) == 0ull) {
                        __builtin_amdgcn_s_sleep(1);
                        v[u] = __hip_atomic_load(row + (size_t)(j - u) * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    excl += (u32)v[u];
                    done = (v[u] >> 32) == 2ull;
                }
                j -= 4;
            }
            __hip_atomic_store(row + (size_t)tile * 256, kFlagInclusive | (u64)(excl + mine), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        u32 run = s_start[d] + excl;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            s_cnt[w][d] = run;
            run += c[w];
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kItemsPerThread; ++it) {
        const u32 idx = seg + it * 64 + lane;
        if (idx < n) {
            const u32 key = keys_in[idx];
            const u32 pos = s_cnt[wave][(key >> shift) & 0xFFu] + r_rank[it];
            keys_out[pos] = key;
            vals_out[pos] = vals_in[idx];
        }
    }
}
#endif // GWBP_SORT_ONESWEEP

// Small scenes (N <= 12 K Gaussians): the whole level-1 sort -- four 8-bit passes over (depth key, index) -- in ONE
// workgroup with the pairs resident in LDS, instead of 4 x (k_hist, k_scan, k_radix_scatter) = 12 launches of 5-15 us each.
// A view of such a scene is a chain of dependent launches that the HOST can barely enqueue fast enough (bench.py C1:
// host_enqueue_ms_per_view = 0.17 of a 0.18 ms step), so every launch removed is time.  Same ranking scheme as
// k_radix_scatter (wave-striped ballot match, per-wave digit counters): the result is the same stable order bit for bit.
constexpr int kSmallThreads = 1024;
constexpr int kSmallWaves = kSmallThreads / 64;
constexpr int kSmallMaxItems = 12 * kSmallThreads; // 12 items per thread: 103 VGPRs (16: 128 + spills)

// The tail also does the scan in front of the emit (k_sorted_blocksums + k_scan_blocksums of project.hip: two more launches):
// after the last pass a thread's items are positions of the depth order, the tiles they touch are prefix-summed in that
// order (wave scan per item slot, carried over the slots, then over the waves), the exclusive sums at every 256th position
// are what k_emit expects as block offsets, and the total becomes Counters::n_isect (or the overflow flag).  (Emitting
// from here as well was measured: one workgroup writes the 92 K pairs of a C1 view in 90 us, k_emit's 40 blocks in 6.)
struct SmallEmit {
    const u32 *touched; // tiles per Gaussian
    u32 *blocksums;     // exclusive prefix of `touched` in depth order at positions 0, 256, 512, ...
    Counters *ctr;
    u32 isect_cap;
};

template <int IPT> // items per thread: N <= IPT * 1024
__global__ __launch_bounds__(kSmallThreads) void k_sort_small(u32 *__restrict__ keys, u32 *__restrict__ vals, u32 n, int prio,
                                                              SmallEmit em)
{
    front_priority(prio);
    extern __shared__ u32 s_small[];
    u32 *kbuf = s_small, *vbuf = s_small + IPT * kSmallThreads;
    u32(*s_cnt)[256] = reinterpret_cast<u32(*)[256]>(vbuf + IPT * kSmallThreads); // [wave][digit]
    u32 *s_w = reinterpret_cast<u32 *>(s_cnt + kSmallWaves);                        // 4 wave totals of the digit scan
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 seg = (u32)wave * (u32)(IPT * 64);
    const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int it = 0; it < IPT; ++it) {
        const u32 idx = seg + it * 64 + lane;
        kbuf[idx] = idx < n ? keys[idx] : ~0u;
        vbuf[idx] = idx < n ? vals[idx] : 0u;
    }
    for (int shift = 0; shift < 32; shift += 8) {
#pragma unroll
        for (int i = 0; i < kSmallWaves * 256 / kSmallThreads; ++i)
            (&s_cnt[0][0])[i * kSmallThreads + threadIdx.x] = 0;
        __syncthreads(); // counters zeroed, the pairs are in LDS
        u32 k[IPT], v[IPT], r[IPT];
#pragma unroll
        for (int it = 0; it < IPT; ++it) {
            const u32 idx = seg + it * 64 + lane;
            const bool valid = idx < n;
            k[it] = kbuf[idx], v[it] = vbuf[idx];
            const u32 dg = (k[it] >> shift) & 0xFFu;
            u64 peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (dg >> b) & 1u;
                const u64 m = __ballot(bit);
                peers &= bit ? m : ~m;
            }
            u32 old = 0;
            const int leader = valid ? (__ffsll((long long)peers) - 1) : 0;
            if (valid && lane == leader) {
                old = s_cnt[wave][dg];
                s_cnt[wave][dg] = old + (u32)__popcll(peers);
            }
            old = __shfl(old, leader, 64);
            r[it] = old + (u32)__popcll(peers & lt);
        }
        __syncthreads(); // every item is in registers, every count is in
        u32 total = 0, incl = 0;
        if (threadIdx.x < 256) { // thread = digit: per-wave bases inside the digit, the digit's total, scan over digits
#pragma unroll
            for (int w = 0; w < kSmallWaves; ++w) {
                const u32 c = s_cnt[w][threadIdx.x];
                s_cnt[w][threadIdx.x] = total;
                total += c;
            }
            incl = total;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const u32 t = __shfl_up(incl, o, 64);
                if (lane >= o)
                    incl += t;
            }
            if (lane == 63)
                s_w[wave] = incl;
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            u32 start = incl - total;
            for (int w = 0; w < wave; ++w)
                start += s_w[w];
#pragma unroll
            for (int w = 0; w < kSmallWaves; ++w)
                s_cnt[w][threadIdx.x] += start;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < IPT; ++it) {
            const u32 idx = seg + it * 64 + lane;
            if (idx < n) {
                const u32 pos = s_cnt[wave][(k[it] >> shift) & 0xFFu] + r[it];
                kbuf[pos] = k[it], vbuf[pos] = v[it];
            }
        }
        __syncthreads(); // the bases in s_cnt are still being read until here: the next pass zeroes them
    }
    u32 pos[IPT];
    u32 running = 0; // tiles touched by this wave's earlier item slots
#pragma unroll
    for (int it = 0; it < IPT; ++it) {
        const u32 idx = seg + it * 64 + lane;
        const bool valid = idx < n;
        const u32 gid = valid ? vbuf[idx] : 0u;
        if (valid)
            keys[idx] = kbuf[idx], vals[idx] = gid;
        const u32 cnt = valid ? em.touched[gid] : 0u;
        u32 incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 t = __shfl_up(incl, o, 64);
            if (lane >= o)
                incl += t;
        }
        pos[it] = running + incl - cnt;
        running += (u32)__builtin_amdgcn_readlane((int)incl, 63);
    }
    __syncthreads(); // s_cnt is free again: its first row carries the wave totals
    u32 *s_tot = &s_cnt[0][0];
    if (lane == 0)
        s_tot[wave] = running;
    __syncthreads();
    u32 woff = 0;
    u64 total = 0; // 64 bits: a wrapped 32-bit sum must not pass the capacity check
#pragma unroll
    for (int w = 0; w < kSmallWaves; ++w) {
        const u32 t = s_tot[w];
        woff += w < wave ? t : 0u;
        total += t;
    }
    if (threadIdx.x == 0) {
        if (total > (u64)em.isect_cap) {
            atomicOr(&em.ctr->overflow, 1u);
            em.ctr->n_isect = 0; // downstream stages see an empty view; caller must retry with larger caps
        } else {
            em.ctr->n_isect = (u32)total;
        }
    }
    if ((lane & 63) == 0) { // idx = seg + 64 it is a multiple of 256 for every fourth item slot of lane 0
#pragma unroll
        for (int it = 0; it < IPT; ++it) {
            const u32 idx = seg + it * 64;
            if ((idx & (u32)(kScanBlock - 1)) == 0u && idx < n)
                em.blocksums[idx / (u32)kScanBlock] = woff + pos[it];
        }
    }
}

template <int IPT>
static int launch_sort_small(u32 *keys, u32 *vals, u32 n, int prio, int slot, const SmallEmit &em, hipStream_t s)
{
    const size_t lds = (size_t)(2 * IPT * kSmallThreads + kSmallWaves * 256 + 4) * sizeof(u32);
    if (lds > 64 * 1024) {
        const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(k_sort_small<IPT>), (int)lds, slot);
        if (rc)
            return rc;
    }
    hipLaunchKernelGGL(k_sort_small<IPT>, dim3(1), dim3(kSmallThreads), lds, s, keys, vals, n, prio, em);
    return GWBP_OK;
}

// isect_offset_encode: offsets[t] = first index whose tile >= t; offsets[n_tiles] = n.
__global__ __launch_bounds__(256) void k_tile_offsets(const u32 *__restrict__ keys,
                                                      const Counters *__restrict__ ctr, int n_tiles,
                                                      u32 *__restrict__ offsets, int prio)
{
    front_priority(prio);
    const u32 n = ctr->n_isect;
    const u32 i0 = blockIdx.x * 256u + threadIdx.x;
    if (n == 0) {
        for (u32 t = i0; t <= (u32)n_tiles; t += gridDim.x * 256u)
            offsets[t] = 0;
        return;
    }
    // grid-stride over the DATA (the count lives on the device; a grid sized for the capacity was 312 K mostly idle blocks at C4)
    for (u32 i = i0; i < n; i += gridDim.x * 256u) {
        const int t = (int)keys[i];
        const int tp = (i == 0) ? -1 : (int)keys[i - 1];
        for (int tt = tp + 1; tt <= t; ++tt)
            offsets[tt] = i;
        if (i == n - 1)
            for (int tt = t + 1; tt <= n_tiles; ++tt)
                offsets[tt] = n;
    }
}

// gsplat's meta: isect_ids = tile << 32 | depth bits (rebuilt from the tile key and the Gaussian's depth), flatten_ids
__global__ void k_export_sorted(const u32 *__restrict__ keys, const u32 *__restrict__ vals,
                                const G2D *__restrict__ g2d, const Counters *__restrict__ ctr, int64_t cap,
                                int64_t *__restrict__ o_keys, int32_t *__restrict__ o_vals)
{
    const u32 n = ctr->n_isect;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (int64_t)gridDim.x * blockDim.x) {
        const bool in = i < (int64_t)n;
        if (o_keys)
            o_keys[i] = in ? (int64_t)(((u64)keys[i] << 32) | (u32)__float_as_int(g2d[vals[i]].depth)) : -1;
        if (o_vals)
            o_vals[i] = in ? (int32_t)vals[i] : -1;
    }
}

// Tiles ordered by descending list length (counting sort into 1024 buckets of 4 entries, one workgroup).  k_blend runs
// one wave per tile, so its critical path is the longest list: starting those waves first keeps them off the tail,
// which matters most when the kernel shares the CUs with the persistent scatter workgroups (ViewPipeline).
// 256 threads x 4 buckets each (a 1024-thread workgroup is hard to place beside the scatter workgroups, see
// k_scan_blocksums).
constexpr int kOrderThreads = 256;
__global__ __launch_bounds__(kOrderThreads) void k_tile_order(const u32 *__restrict__ tile_offsets, int n_tiles,
                                                              u32 *__restrict__ order, u32 *__restrict__ sweep, int prio)
{
    front_priority(prio);
#ifdef GWBP_SORT_ONESWEEP
    // the last kernel of gwbp_bin_sort leaves the look-back sort's digit tables and tickets zero for the next call on this
    // workspace (gwbp_project's memset clears them as well: a view normally starts there)
    for (int i = threadIdx.x; i < kSweepWords; i += kOrderThreads)
        sweep[i] = 0u;
#else
    (void)sweep;
#endif
    __shared__ u32 s_cnt[1024];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        s_cnt[4 * threadIdx.x + j] = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < n_tiles; t += kOrderThreads) {
        const u32 len = tile_offsets[t + 1] - tile_offsets[t];
        atomicAdd(&s_cnt[1023u - min(len >> 2, 1023u)], 1u); // bucket 0 = longest lists
    }
    __syncthreads();
    // exclusive scan of the 1024 bucket counts: thread t owns buckets 4t .. 4t+3
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ u32 s_w[kOrderThreads / 64];
    u32 c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        c[j] = s_cnt[4 * threadIdx.x + j];
    const u32 v = c[0] + c[1] + c[2] + c[3];
    u32 incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 tt = __shfl_up(incl, o, 64);
        if (lane >= o)
            incl += tt;
    }
    if (lane == 63)
        s_w[wave] = incl;
    __syncthreads();
    u32 run = incl - v;
    for (int w = 0; w < wave; ++w)
        run += s_w[w];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s_cnt[4 * threadIdx.x + j] = run;
        run += c[j];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < n_tiles; t += kOrderThreads) {
        const u32 len = tile_offsets[t + 1] - tile_offsets[t];
        order[atomicAdd(&s_cnt[1023u - min(len >> 2, 1023u)], 1u)] = (u32)t;
    }
}

__global__ void k_copy_u32(const u32 *__restrict__ src, int32_t *__restrict__ dst, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        dst[i] = (int32_t)src[i];
}

// `level` 0 = the depth sort of the Gaussians (tables 0..3, tickets 0..3), 1 = the tile sort of the intersections (4..7)
static void radix_passes(const Layout &L, const Ws &W, u32 *const keys[2], u32 *const vals[2], const u32 *n_dev, u32 n_host,
                         int nblk, int passes, int prio, int level, hipStream_t s)
{
#ifdef GWBP_SORT_ONESWEEP
    u64 *status[2] = {reinterpret_cast<u64 *>(W.hist), reinterpret_cast<u64 *>(W.hist) + (size_t)L.n_sort_blocks * 256};
    u32 *totals = W.sweep + level * kMaxPasses * 256;
    u32 *tickets = W.sweep + kSweepTickets + level * kMaxPasses;
    hipLaunchKernelGGL(k_hist_all, dim3(nblk), dim3(kSortThreads), 0, s, keys[0], n_dev, n_host, passes, totals, status[0], prio);
    for (int p = 0; p < passes; ++p) {
        const int in = p & 1, out = in ^ 1;
        hipLaunchKernelGGL(k_onesweep, dim3(nblk), dim3(kSortThreads), 0, s, keys[in], vals[in], keys[out], vals[out], n_dev, n_host,
                           p * 8, totals + p * 256, status[p & 1], status[(p + 1) & 1], tickets + p, prio);
    }
    return;
#endif
    (void)level, (void)L;
    for (int p = 0; p < passes; ++p) {
        const int in = p & 1, out = in ^ 1;
        hipLaunchKernelGGL(k_hist, dim3(nblk), dim3(kSortThreads), 0, s, keys[in], n_dev, n_host, p * 8, nblk, W.hist, prio);
        hipLaunchKernelGGL(k_scan, dim3(256), dim3(256), 0, s, nblk, n_dev, n_host, W.hist, W.digit_total, prio);
        hipLaunchKernelGGL(k_radix_scatter, dim3(nblk), dim3(kSortThreads), 0, s, keys[in], vals[in], keys[out],
                           vals[out], n_dev, n_host, p * 8, nblk, W.hist, W.digit_total, prio);
    }
}

int launch_bin_sort(const Layout &L, const Ws &W, const ViewDev &V, int64_t *isect_ids, int32_t *flatten_ids,
                    int32_t *tile_offsets, hipStream_t s)
{
    const int prio = (L.flags & GWBP_FLAG_FRONT_PRIORITY) ? 1 : 0;
    const int n_tiles = V.tile_w * V.tile_h;
    if (L.n > 0) {
        // level 1: Gaussians by depth (4 passes -> the result is back in buffer 0)
        const int nblk1 = (int)((L.n + kSortItems - 1) / kSortItems);
        if (L.n <= kSmallMaxItems) { // small scene: depth sort and the emit's scan in one single-workgroup launch
            const SmallEmit em = {W.touched, W.blocksums, W.counters, (u32)L.isect_cap};
            int rc1;
            if (L.n <= 4 * kSmallThreads)
                rc1 = launch_sort_small<4>(W.dkeys[0], W.dvals[0], (u32)L.n, prio, 11, em, s);
            else if (L.n <= 8 * kSmallThreads)
                rc1 = launch_sort_small<8>(W.dkeys[0], W.dvals[0], (u32)L.n, prio, 12, em, s);
            else
                rc1 = launch_sort_small<12>(W.dkeys[0], W.dvals[0], (u32)L.n, prio, 13, em, s);
            if (rc1 || (rc1 = launch_emit_scanned(L, W, V, W.dvals[0], s)))
                return rc1;
        } else {
            radix_passes(L, W, W.dkeys, W.dvals, nullptr, (u32)L.n, nblk1, 4, prio, 0, s);
            // emit intersections front to back
            const int rc = launch_emit(L, W, V, W.dvals[0], s);
            if (rc)
                return rc;
        }
    }
    // level 2: intersections by tile id
    const int passes = sort_passes(n_tiles);
    const int nblk2 = (int)((L.isect_cap + kSortItems - 1) / kSortItems);
    radix_passes(L, W, W.keys, W.vals, &W.counters->n_isect, 0u, nblk2, passes, prio, 1, s);
    const int fin = passes & 1;
    const int ob = (int)((L.isect_cap + 255) / 256);
    hipLaunchKernelGGL(k_tile_offsets, dim3(ob > 0 ? (ob < 4096 ? ob : 4096) : 1), dim3(256), 0, s, W.keys[fin], W.counters, n_tiles,
                       W.tile_offsets, prio);
    hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(kOrderThreads), 0, s, W.tile_offsets, n_tiles, W.tile_order, W.sweep, prio);
    if (isect_ids || flatten_ids)
        hipLaunchKernelGGL(k_export_sorted, dim3(1024), dim3(256), 0, s, W.keys[fin], W.vals[fin], W.g2d, W.counters,
                           L.isect_cap, isect_ids, flatten_ids);
    if (tile_offsets)
        hipLaunchKernelGGL(k_copy_u32, dim3((n_tiles + 256) / 256), dim3(256), 0, s, W.tile_offsets, tile_offsets,
                           n_tiles + 1);
    return check_hip(hipGetLastError(), "bin_sort launch");
}

} // namespace gwbp
