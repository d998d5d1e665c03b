// blend.hip -- per-tile front-to-back alpha blend producing the SPARSE WEIGHT STORE of one view.
//
// Replaces gsplat 1.4.0's rasterize_to_pixels forward (and makes its backward replay unnecessary): the reference
// runs that blend 2 x (16 fwd + 16 bwd) times per view at D = 512 (backproject.py:115-147 through
// channel_chunk = 32); here it runs ONCE and its only product is w = alpha * T for every contributing
// (Gaussian, pixel) pair:
//
//   workgroup = one 16x16 tile, 256 threads; wave q owns tile rows 4q..4q+3, lane l -> pixel q*64 + l
//   per batch of <= kBatch list entries: Gaussian records staged in LDS (2 x 16-B broadcast reads per evaluation)
//   per (Gaussian, wave): mask = ballot(contributes); the popc(mask) {w, pixel} entries are written compacted
//     (lane rank = mbcnt, 8-B stores) into a per-wave stream carved from a global pool in pages of kPage entries,
//     each list zero-padded to a multiple of 8 entries (= one s_load_dwordx16 in the scatter kernel)
//   per (Gaussian, tile) with any contribution: one 64-B Header {gid, 4 x woff, 4 x mask}, compacted in list order
//
// Store size: 8 B per pair (+ padding) + 64 B per header (C2: ~0.8 GB + ~116 MB per view), written once, then read by the
// scatter kernel once per 128-channel chunk through L2.
#include <stdlib.h>

#include "gwbp_dev.h"

namespace gwbp {

constexpr int kBatch = 128;

__global__ __launch_bounds__(256) void k_blend(ViewDev V, const u32 *__restrict__ tile_offsets,
                                               const u32 *__restrict__ vals, const G2D *__restrict__ g2d,
                                               Counters *__restrict__ ctr, Header *__restrict__ headers,
                                               u32 *__restrict__ hdr_count, WPair *__restrict__ wpool,
                                               u32 pair_cap, u32 *__restrict__ shards, float *__restrict__ alphas, int dbg)
{
    // kBatch list entries are staged per round; 128 keeps the workgroup at 10 KB of LDS so that two of them fit beside
    // the 139 KB scatter workgroup when the two kernels overlap (ViewPipeline)
    __shared__ float4 s_a[kBatch]; // mx, my, opac, gid bits
    __shared__ float4 s_b[kBatch]; // ca, cb, cc, strip mask
    __shared__ u64 s_mask[kBatch][4];
    __shared__ u32 s_woff[kBatch][4];
    __shared__ u32 s_wsum[4];
    __shared__ u32 s_hdrn;

    const int tile = blockIdx.x;
    const int tx = tile % V.tile_w, ty = tile / V.tile_w;
    const int lane = threadIdx.x & 63;
    const int wave = (int)uniform(threadIdx.x >> 6); // scalar: everything derived from it stays wave-uniform
    const int ix = tx * kTile + (lane & 15), iy = ty * kTile + wave * 4 + (lane >> 4);
    const bool inside = ix < V.W && iy < V.H;
    const float px = (float)ix + 0.5f, py = (float)iy + 0.5f;
    const u32 beg = tile_offsets[tile], end = tile_offsets[tile + 1];

    // The weight pool is carved into kShards regions with their own head words: a single head saturates at
    // ~88 returning same-address atomics per microsecond, which cost 1.0 ms/view with ~96 K page grabs per view.
    const u32 shard = (u32)tile % (u32)kShards;
    const u32 shard_cap = (pair_cap / (u32)kShards) & ~(u32)(kPage - 1);
    const u32 shard_base = shard * shard_cap;
    u32 *shard_head = shards + shard * 16;

    float T = 1.0f;
    bool done = !inside;
    u32 page_pos = 0, page_left = 0, npairs = 0; // wave-uniform
    bool dead = false;                           // wave-uniform: pool exhausted
    if (threadIdx.x == 0)
        s_hdrn = 0;

    for (u32 batch = beg; batch < end; batch += kBatch) {
        // barrier + early exit when every pixel of the tile has terminated (gsplat: __syncthreads_count(done))
        if (__syncthreads_count(done) == 256)
            break;
        const u32 bn = min((u32)kBatch, end - batch);
        if (threadIdx.x < bn) {
            const u32 gid = vals[batch + threadIdx.x];
            const float4 *gp = reinterpret_cast<const float4 *>(g2d + gid);
            const float4 a = gp[0], b = gp[1];
            // Conservative strip mask: bit q set <=> the Gaussian MAY reach alpha >= 1/255 somewhere in tile rows
            // 4q..4q+3.  For a fixed dy the minimum of sigma over dx is dy^2 / (2 Syy) (Syy = ca / det(conic)), so a
            // row with dy^2 > 2 Syy ln(255 o) cannot contribute; same in x for the whole tile.  Distances carry a
            // 5 % + 1 px margin (>> fp32 error of sigma: conic entries are bounded by 1/eps2d), so a rejected
            // (Gaussian, strip) provably has no contributing pixel and results are unchanged bit for bit.
            u32 smask = 0;
            const float L = __logf(255.0f * a.z); // ln(255 o); o <= 1/255 can never reach alpha >= 1/255
            if (L > 0.f) {
                const float idet = 1.0f / (b.x * b.z - b.y * b.y);
                const float ex = 1.05f * __builtin_sqrtf(2.0f * L * b.z * idet) + 1.0f;
                const float ey = 1.05f * __builtin_sqrtf(2.0f * L * b.x * idet) + 1.0f;
                const float x0 = (float)(tx * kTile) + 0.5f, y0 = (float)(ty * kTile) + 0.5f;
                const bool xhit = !(a.x + ex < x0 || a.x - ex > x0 + 15.0f);
                if (xhit || !(ex == ex)) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float ya = y0 + 4.0f * q;
                        const bool miss = (a.y + ey < ya) || (a.y - ey > ya + 3.0f);
                        smask |= (miss ? 0u : 1u) << q;
                    }
                }
                if (!(ex == ex) || !(ey == ey))
                    smask = 0xFu; // degenerate conic: never reject
            }
            s_a[threadIdx.x] = make_float4(a.x, a.y, a.z, __int_as_float((int)gid));
            s_b[threadIdx.x] = make_float4(b.x, b.y, b.z, __int_as_float((int)smask));
            s_mask[threadIdx.x][0] = 0ull, s_mask[threadIdx.x][1] = 0ull;
            s_mask[threadIdx.x][2] = 0ull, s_mask[threadIdx.x][3] = 0ull;
        }
        __syncthreads();

        // Per-lane control flow is written branch-free (selects); the only branches in the loop are wave-uniform.
        // Two list entries are evaluated per iteration: their sigma / exp / alpha chains are independent (only the
        // T update is sequential), which doubles the instruction-level parallelism of this latency-bound loop.
        auto emit = [&](u32 j, bool valid, float w) {
            const u64 mask = __ballot(valid);
            if (mask == 0ull)
                return;
            const u32 cnt = (u32)__popcll(mask);
            const u32 padded = (cnt + (kListPad - 1)) & ~(u32)(kListPad - 1);
            if (padded > page_left) {
                u32 old = 0;
                if (lane == 0)
                    old = atomicAdd(shard_head, (u32)kPage);
                old = uniform(old);
                page_pos = shard_base + old;
                page_left = kPage;
                if (old + (u32)kPage > shard_cap) {
                    dead = true;
                    if (lane == 0)
                        atomicOr(&ctr->overflow, 2u);
                }
            }
            if (!dead && !(dbg & 1)) {
                if (valid) {
                    WPair e;
                    e.w = w, e.pix = (u32)(wave * 64 + lane);
                    wpool[page_pos + mbcnt(mask)] = e;
                }
                if ((u32)lane < padded - cnt) { // {0, 0} tail so the scatter loop needs no remainder handling
                    WPair z;
                    z.w = 0.f, z.pix = 0u;
                    wpool[page_pos + cnt + lane] = z;
                }
                if (lane == 0) {
                    s_mask[j][wave] = mask;
                    s_woff[j][wave] = page_pos;
                }
            }
            page_pos += padded;
            page_left -= padded;
            npairs += cnt;
        };
        auto alpha_of = [&](const float4 &a, const float4 &b, float &sigma) -> float {
            const float dx = a.x - px, dy = a.y - py;
            sigma = __builtin_fmaf(b.y * dx, dy, 0.5f * __builtin_fmaf(b.x * dx, dx, (b.z * dy) * dy));
            if (dbg & 8)
                return __builtin_fminf(kAlphaMax, a.z * (1.0f / (1.0f + sigma)));
            return __builtin_fminf(kAlphaMax, a.z * exp_neg(-__builtin_fmaxf(sigma, 0.f)));
        };
        for (u32 j = 0; j < bn && !(dbg & 2); j += 2) {
            if (__ballot(!done) == 0ull)
                break; // every pixel of this wave's strip has terminated
            const u32 j1 = min(j + 1, bn - 1);
            const float4 b0 = s_b[j], b1 = s_b[j1];
            const float4 a0 = s_a[j], a1 = s_a[j1];
            const bool hit0 = (uniform((u32)__float_as_int(b0.w)) >> wave) & 1u;
            const bool hit1 = ((uniform((u32)__float_as_int(b1.w)) >> wave) & 1u) && (j + 1 < bn);
            if (!hit0 && !hit1 && !(dbg & 16))
                continue; // neither Gaussian can reach this wave's strip
            float sg0, sg1;
            const float al0 = alpha_of(a0, b0, sg0), al1 = alpha_of(a1, b1, sg1);
            // sequential part (front to back): entry j, then entry j+1
            const bool ok0 = hit0 && !done && (sg0 >= 0.f) && (al0 >= kAlphaMin);
            const float nT0 = T * (1.0f - al0);
            const bool term0 = ok0 && (nT0 <= kTMin); // the terminating Gaussian is NOT counted
            const bool v0 = ok0 && !term0;
            const float w0 = al0 * T;
            T = v0 ? nT0 : T;
            done = done || term0;
            const bool ok1 = hit1 && !done && (sg1 >= 0.f) && (al1 >= kAlphaMin);
            const float nT1 = T * (1.0f - al1);
            const bool term1 = ok1 && (nT1 <= kTMin);
            const bool v1 = ok1 && !term1;
            const float w1 = al1 * T;
            T = v1 ? nT1 : T;
            done = done || term1;
            if (!(dbg & 4)) {
                emit(j, v0, w0);
                emit(j1, v1, w1);
            }
        }
        __syncthreads();

        // compact this batch's non-empty (Gaussian, tile) records into the tile's header run, in list order
        bool has = false;
        u64 m0 = 0, m1 = 0, m2 = 0, m3 = 0;
        if (threadIdx.x < bn) {
            m0 = s_mask[threadIdx.x][0], m1 = s_mask[threadIdx.x][1];
            m2 = s_mask[threadIdx.x][2], m3 = s_mask[threadIdx.x][3];
            has = (m0 | m1 | m2 | m3) != 0ull;
        }
        const u64 hb = __ballot(has);
        if (lane == 0)
            s_wsum[wave] = (u32)__popcll(hb);
        __syncthreads();
        u32 off = s_hdrn + mbcnt(hb);
        for (int w = 0; w < wave; ++w)
            off += s_wsum[w];
        if (has) {
            Header h;
            h.gid = (u32)__float_as_int(s_a[threadIdx.x].w);
            h.woff[0] = s_woff[threadIdx.x][0], h.woff[1] = s_woff[threadIdx.x][1];
            h.woff[2] = s_woff[threadIdx.x][2], h.woff[3] = s_woff[threadIdx.x][3];
            h.counts = (u32)__popcll(m0) | ((u32)__popcll(m1) << 8) | ((u32)__popcll(m2) << 16) |
                       ((u32)__popcll(m3) << 24);
            h.pad[0] = h.pad[1] = 0;
            h.mask[0] = m0, h.mask[1] = m1, h.mask[2] = m2, h.mask[3] = m3;
            headers[beg + off] = h;
        }
        __syncthreads();
        if (threadIdx.x == 0)
            s_hdrn += s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        hdr_count[tile] = s_hdrn;
        if (s_hdrn)
            atomicAdd(&ctr->n_headers, s_hdrn);
    }
    if (lane == 0 && npairs)
        atomicAdd(&ctr->n_pairs, (u64)npairs);
    if (alphas && inside)
        alphas[(size_t)iy * V.W + ix] = 1.0f - T;
}

// Test/debug: expand the weight store into (gid, pix, w) triples.
__global__ __launch_bounds__(256) void k_dump_pairs(ViewDev V, const u32 *__restrict__ tile_offsets,
                                                    const u32 *__restrict__ hdr_count,
                                                    const Header *__restrict__ headers,
                                                    const WPair *__restrict__ wpool, int64_t cap,
                                                    int32_t *__restrict__ gid, int32_t *__restrict__ pix,
                                                    float *__restrict__ w, u64 *__restrict__ n_out)
{
    const int tile = blockIdx.x;
    const int tx = tile % V.tile_w, ty = tile / V.tile_w;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 nh = hdr_count[tile];
    const Header *hb = headers + tile_offsets[tile];
    for (u32 h = wave; h < nh; h += 4) {
        const Header hd = hb[h];
        for (int q = 0; q < 4; ++q) {
            const u64 m = hd.mask[q];
            if (!m)
                continue;
            const u32 cnt = (u32)__popcll(m);
            u64 base = 0;
            if (lane == 0)
                base = atomicAdd(n_out, (u64)cnt);
            base = uniform64(base);
            if ((m >> lane) & 1ull) {
                const u32 r = mbcnt(m);
                const u64 o = base + r;
                if ((int64_t)o < cap) {
                    const int ix = tx * kTile + (lane & 15), iy = ty * kTile + q * 4 + (lane >> 4);
                    gid[o] = (int32_t)hd.gid;
                    pix[o] = iy * V.W + ix;
                    w[o] = wpool[hd.woff[q] + r].w;
                }
            }
        }
    }
}

__global__ void k_pool_stats(const u32 *__restrict__ shards, Counters *__restrict__ ctr)
{
    u32 mx = 0;
    for (int i = 0; i < kShards; ++i)
        mx = max(mx, shards[i * 16]);
    ctr->pool_head = mx * (u32)kShards;
}

int launch_blend(const Layout &L, const Ws &W, const ViewDev &V, float *alphas, hipStream_t s)
{
    const int n_tiles = V.tile_w * V.tile_h;
    const int fin = sort_passes(n_tiles) & 1;
    static int extra_lds = -1; // experiment knob: pad the workgroup's LDS footprint to cap co-residency
    if (extra_lds < 0)
        extra_lds = getenv("GWBP_BLEND_LDS") ? atoi(getenv("GWBP_BLEND_LDS")) : 0;
    hipLaunchKernelGGL(k_blend, dim3(n_tiles), dim3(256), (size_t)extra_lds, s, V, W.tile_offsets, W.vals[fin], W.g2d, W.counters,
                       W.headers, W.hdr_count, W.wpool, (u32)L.pair_cap, W.shards, alphas,
                       getenv("GWBP_ABLATE_BLEND") ? atoi(getenv("GWBP_ABLATE_BLEND")) : 0);
    hipLaunchKernelGGL(k_pool_stats, dim3(1), dim3(1), 0, s, W.shards, W.counters);
    return check_hip(hipGetLastError(), "blend launch");
}

int launch_dump_pairs(const Layout &L, const Ws &W, const ViewDev &V, int64_t cap, int32_t *gid, int32_t *pix,
                      float *w, u64 *n_dev, hipStream_t s)
{
    (void)L;
    const int n_tiles = V.tile_w * V.tile_h;
    hipLaunchKernelGGL(k_dump_pairs, dim3(n_tiles), dim3(256), 0, s, V, W.tile_offsets, W.hdr_count, W.headers,
                       W.wpool, cap, gid, pix, w, n_dev);
    return check_hip(hipGetLastError(), "dump_pairs launch");
}

} // namespace gwbp
