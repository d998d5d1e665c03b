// blend.hip -- per-tile front-to-back alpha blend producing the SPARSE WEIGHT STORE of one view.
//
// Replaces gsplat 1.4.0's rasterize_to_pixels forward (and makes its backward replay unnecessary): the reference
// runs that blend 2 x (16 fwd + 16 bwd) times per view at D = 512 (backproject.py:115-147 through
// channel_chunk = 32); here it runs ONCE and its only product is w = alpha * T for every contributing
// (Gaussian, pixel) pair:
//
//   workgroup = one 16x16 tile, 256 threads; wave q owns tile rows 4q..4q+3, lane l -> pixel q*64 + l
//   per batch of <= 256 list entries: Gaussian records staged in LDS (2 x 16-B broadcast reads per evaluation)
//   per (Gaussian, wave): mask = ballot(contributes); the popc(mask) {w, pixel} entries are written compacted
//     (lane rank = mbcnt, 8-B stores) into a per-wave stream carved from a global pool in pages of kPage entries,
//     each list zero-padded to a multiple of 8 entries (= one s_load_dwordx16 in the scatter kernel)
//   per (Gaussian, tile) with any contribution: one 64-B Header {gid, 4 x woff, 4 x mask}, compacted in list order
//
// Store size: 8 B per pair (+ padding) + 64 B per header (C2: ~0.8 GB + ~116 MB per view), written once, then read by the
// scatter kernel once per 128-channel chunk through L2.
#include "gwbp_dev.h"

namespace gwbp {

__global__ __launch_bounds__(256) void k_blend(ViewDev V, const u32 *__restrict__ tile_offsets,
                                               const u32 *__restrict__ vals, const G2D *__restrict__ g2d,
                                               Counters *__restrict__ ctr, Header *__restrict__ headers,
                                               u32 *__restrict__ hdr_count, WPair *__restrict__ wpool,
                                               u32 pair_cap, float *__restrict__ alphas)
{
    __shared__ float4 s_a[256]; // mx, my, opac, gid bits
    __shared__ float4 s_b[256]; // ca, cb, cc, -
    __shared__ u64 s_mask[256][4];
    __shared__ u32 s_woff[256][4];
    __shared__ u32 s_wsum[4];
    __shared__ u32 s_hdrn;

    const int tile = blockIdx.x;
    const int tx = tile % V.tile_w, ty = tile / V.tile_w;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ix = tx * kTile + (lane & 15), iy = ty * kTile + wave * 4 + (lane >> 4);
    const bool inside = ix < V.W && iy < V.H;
    const float px = (float)ix + 0.5f, py = (float)iy + 0.5f;
    const u32 beg = tile_offsets[tile], end = tile_offsets[tile + 1];

    float T = 1.0f;
    bool done = !inside;
    u32 page_pos = 0, page_left = 0, npairs = 0; // wave-uniform
    bool dead = false;                           // wave-uniform: pool exhausted
    if (threadIdx.x == 0)
        s_hdrn = 0;

    for (u32 batch = beg; batch < end; batch += 256) {
        // barrier + early exit when every pixel of the tile has terminated (gsplat: __syncthreads_count(done))
        if (__syncthreads_count(done) == 256)
            break;
        const u32 bn = min(256u, end - batch);
        if (threadIdx.x < bn) {
            const u32 gid = vals[batch + threadIdx.x];
            const float4 *gp = reinterpret_cast<const float4 *>(g2d + gid);
            const float4 a = gp[0], b = gp[1];
            s_a[threadIdx.x] = make_float4(a.x, a.y, a.z, __int_as_float((int)gid));
            s_b[threadIdx.x] = b;
        }
        __syncthreads();

        bool wave_active = __ballot(!done) != 0ull;
        for (u32 j = 0; j < bn; ++j) {
            u64 mask = 0ull;
            u32 woff = 0;
            if (wave_active) {
                const float4 a = s_a[j], b = s_b[j];
                const float dx = a.x - px, dy = a.y - py;
                const float sigma =
                    __builtin_fmaf(b.y * dx, dy, 0.5f * __builtin_fmaf(b.x * dx, dx, (b.z * dy) * dy));
                bool valid = false;
                float w = 0.f;
                if (!done && sigma >= 0.f) {
                    const float alpha = __builtin_fminf(kAlphaMax, a.z * exp_neg(-sigma));
                    if (alpha >= kAlphaMin) {
                        const float next_T = T * (1.0f - alpha);
                        if (next_T <= kTMin) {
                            done = true; // the terminating Gaussian is NOT counted
                        } else {
                            w = alpha * T;
                            T = next_T;
                            valid = true;
                        }
                    }
                }
                mask = __ballot(valid);
                if (mask != 0ull) {
                    const u32 cnt = (u32)__popcll(mask);
                    const u32 padded = (cnt + (kListPad - 1)) & ~(u32)(kListPad - 1);
                    if (padded > page_left) {
                        u32 old = 0;
                        if (lane == 0)
                            old = atomicAdd(&ctr->pool_head, (u32)kPage);
                        page_pos = uniform(old);
                        page_left = kPage;
                        if ((u64)page_pos + kPage > (u64)pair_cap) {
                            dead = true;
                            if (lane == 0)
                                atomicOr(&ctr->overflow, 2u);
                        }
                    }
                    if (!dead) {
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
                    }
                    woff = page_pos;
                    page_pos += padded;
                    page_left -= padded;
                    npairs += cnt;
                }
                wave_active = __ballot(!done) != 0ull;
            }
            if (lane == 0) {
                s_mask[j][wave] = dead ? 0ull : mask;
                s_woff[j][wave] = woff;
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

int launch_blend(const Layout &L, const Ws &W, const ViewDev &V, float *alphas, hipStream_t s)
{
    const int n_tiles = V.tile_w * V.tile_h;
    const int fin = sort_passes(n_tiles) & 1;
    hipLaunchKernelGGL(k_blend, dim3(n_tiles), dim3(256), 0, s, V, W.tile_offsets, W.vals[fin], W.g2d, W.counters,
                       W.headers, W.hdr_count, W.wpool, (u32)L.pair_cap, alphas);
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
