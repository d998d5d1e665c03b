// scatter.hip -- the weighted scatter-accumulate  F[g,:] += sum_p w_g(p) * feats[p,:],  d[g] += sum_p w_g(p)
// (what the reference harvests from gsplat's rasterize_to_pixels backward as colors.grad, backproject.py:127-150),
// its transpose (forward render) and the final normalisation (backproject.py:166-169).
//
// k_scatter: workgroup = (tile, 128-channel chunk).
//   - the tile's 256 px x 128 ch feature slab is staged ONCE in LDS (128 KB, 512-B rows, 16-B loads; each feature
//     byte is read from HBM exactly once per view)
//   - waves pull (Gaussian, tile) headers from an LDS work counter; the Gaussian is wave-uniform, lanes = channel
//     pairs (ds_read_b64: 256 B/clk, conflict-free), weights arrive as one coalesced vector load per quarter and are
//     broadcast with v_readlane, pixel indices come from scalar bit-scans of the 64-bit masks
//   - one flush per (Gaussian, tile, chunk): two 256-B contiguous fp32 atomic wave-instructions (channels are
//     transposed across lanes first so each instruction covers 64 consecutive dwords)
//   - blockIdx -> (tile, chunk) keeps the chunks of one tile on one XCD (blockIdx % 8) so the weight store is fetched
//     from HBM once and re-read from that XCD's L2
// Roofline accounting: HBM bytes/view = 4HWD + 8 N_vis (D+1) + weight store; the binding ceiling of this
// first version is the fp32 atomic rate (~1.3 TB/s of added bytes), see DESIGN.md section 5.
#include <stdlib.h>

#include "gwbp_dev.h"

namespace gwbp {

constexpr int kChunk = 128;         // channels per workgroup
constexpr int kScatterThreads = 1024;

__device__ __forceinline__ float readlane_f(float v, u32 l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)l));
}


// Pixel list of a quarter mask without scalar bit-scans: lane l with bit l set sends l to lane rank(l) (= number of
// set bits below l); lanes whose bit is clear fill the tail [cnt, 64) in order.  One full permutation, all lanes
// active: afterwards lane k < cnt holds the pixel index of the k-th contributing pixel of the quarter.
__device__ __forceinline__ int pixel_list(u64 m, int lane, u32 cnt)
{
    const u32 rank = mbcnt(m);
    const bool bit = (m >> lane) & 1ull;
    const u32 dest = bit ? rank : cnt + ((u32)lane - rank);
    return __builtin_amdgcn_ds_permute((int)(dest << 2), lane);
}
__device__ __forceinline__ int readlane_i(int v, u32 l) { return __builtin_amdgcn_readlane(v, (int)l); }

__global__ __launch_bounds__(kScatterThreads) void k_scatter(
    ViewDev V, int n_tiles_pad, int n_chunks, int pitch, const u32 *__restrict__ tile_offsets,
    const u32 *__restrict__ hdr_count, const Header *__restrict__ headers, const WPair *__restrict__ wpool,
    FeatMap M, int D, float scale_f, float scale_d, float *__restrict__ F, float *__restrict__ dsum_out)
{
    const float *__restrict__ feats = M.p;
    const int64_t fs_x = M.fs_x, fs_y = M.fs_y, fs_c = M.fs_c;
    // dynamic LDS only (no static __shared__ in front of it: keeps the carve base 16-B aligned);
    // layout: [256][pitch] floats, then the work counter
    extern __shared__ __attribute__((aligned(16))) float lds[];
    u32 *s_next = reinterpret_cast<u32 *>(lds + kTilePix * pitch);

    // XCD-aware decode: blocks b and b+8 share an XCD; give one tile's chunks the same b % 8.
    const u32 b = blockIdx.x;
    const u32 x = b & 7u, sidx = b >> 3;
    const int chunk = (int)(sidx % (u32)n_chunks);
    const int tile = (int)((sidx / (u32)n_chunks) * 8u + x);
    (void)n_tiles_pad;
    if (tile >= V.tile_w * V.tile_h)
        return;
    const u32 nh = hdr_count[tile];
    if (nh == 0)
        return;
    const int tx = tile % V.tile_w, ty = tile / V.tile_w;
    const int c0 = chunk * kChunk;
    const int cw = min(pitch, D - c0); // valid channels of this chunk
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0)
        *s_next = 0;

    // ---- stage the feature slab -------------------------------------------------------------------------
    const bool vec_ok = !M.bilinear() && (fs_c == 1) && ((pitch & 3) == 0) && ((cw & 3) == 0) && ((fs_x & 3) == 0) &&
                        ((fs_y & 3) == 0) && ((c0 & 3) == 0) && ((reinterpret_cast<uintptr_t>(feats) & 15) == 0);
    if (vec_ok) {
        const int vpr = pitch >> 2; // float4 per pixel row
        const int total = kTilePix * vpr;
        for (int idx = threadIdx.x; idx < total; idx += kScatterThreads) {
            const int p = idx / vpr, v = idx - p * vpr;
            const int ix = tx * kTile + (p & 15), iy = ty * kTile + (p >> 4);
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ix < V.W && iy < V.H && 4 * v < cw)
                val = *reinterpret_cast<const float4 *>(feats + M.pixel(iy, ix) + c0 + 4 * v);
            *reinterpret_cast<float4 *>(lds + p * pitch + 4 * v) = val;
        }
    } else {
        const int total = kTilePix * pitch;
        for (int idx = threadIdx.x; idx < total; idx += kScatterThreads) {
            const int p = idx / pitch, c = idx - p * pitch;
            const int ix = tx * kTile + (p & 15), iy = ty * kTile + (p >> 4);
            float val = 0.f;
            if (ix < V.W && iy < V.H && c < cw)
                val = M.sample(feats + (int64_t)(c0 + c) * fs_c, iy, ix);
            lds[p * pitch + c] = val;
        }
    }
    __syncthreads();

    // ---- accumulate ---------------------------------------------------------------------------------------
    // Software pipeline of depth 1 over records: the next record's header (scalar loads) and its four weight
    // vectors (one coalesced load per quarter) are requested before the current record is processed.
    const Header *hbase = headers + tile_offsets[tile];
    // a "pixel" whose row lies at or beyond the largest LDS allocation gfx950 has (160 KB), for any pitch: reads as 0
    const int no_pix = (160 * 1024 / 4 + pitch - 1) / pitch;
    const bool lane_on = 2 * lane < pitch;               // this lane owns channels c0 + 2*lane, +1 (pitch is even)
    const float *lrow = lds + (lane_on ? 2 * lane : 0);  // idle lanes read lane 0's pair and are masked at the flush

    struct Rec {
        u32 gid;
        u64 m[4];
        float wv[4];
        bool valid;
    };
    auto fetch = [&]() -> Rec {
        Rec r;
        u32 h = 0;
        if (lane == 0)
            h = atomicAdd(s_next, 1u);
        h = uniform(h);
        r.valid = h < nh;
        r.gid = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            r.m[q] = 0ull, r.wv[q] = 0.f;
        if (r.valid) {
            const Header *hp = hbase + h;
            r.gid = uniform(hp->gid);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                r.m[q] = uniform64(hp->mask[q]);
                const u32 cnt = (u32)__popcll(r.m[q]);
                const u32 woff = uniform(hp->woff[q]);
                if ((u32)lane < cnt)
                    r.wv[q] = wpool[woff + lane].w;
            }
        }
        return r;
    };

    Rec cur = fetch();
    while (cur.valid) {
        const Rec nxt = fetch();
        float acc0 = 0.f, acc1 = 0.f;
        float wacc = 0.f; // per-lane partial of sum_p w (reduced once per record, chunk 0 only)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u64 m = cur.m[q];
            if (m == 0ull)
                continue;
            const u32 cnt = (u32)__popcll(m);
            const float wv = cur.wv[q]; // lanes >= cnt hold 0
            wacc += wv;
            // (tail lanes [cnt, 64) carry w = 0; they must not read a real pixel -- 0 x NaN / 0 x inf -- so they point
            // beyond any LDS allocation, where a read returns 0)
            const int pl = pixel_list(m, lane, cnt); // (a full permutation: every lane takes part)
            const int pv = (u32)lane < cnt ? pl : no_pix;
            const float *qrow = lrow + q * 64 * pitch;
            // four pairs per step: independent LDS reads in flight; the tail reads unset pixels with w = 0
            for (u32 k = 0; k < cnt; k += 4) {
                const float w0 = readlane_f(wv, k), w1 = readlane_f(wv, k + 1);
                const float w2 = readlane_f(wv, k + 2), w3 = readlane_f(wv, k + 3);
                const int p0 = readlane_i(pv, k), p1 = readlane_i(pv, k + 1);
                const int p2 = readlane_i(pv, k + 2), p3 = readlane_i(pv, k + 3);
                const float2 f0 = *reinterpret_cast<const float2 *>(qrow + p0 * pitch);
                const float2 f1 = *reinterpret_cast<const float2 *>(qrow + p1 * pitch);
                const float2 f2 = *reinterpret_cast<const float2 *>(qrow + p2 * pitch);
                const float2 f3 = *reinterpret_cast<const float2 *>(qrow + p3 * pitch);
                acc0 = __builtin_fmaf(w0, f0.x, acc0), acc1 = __builtin_fmaf(w0, f0.y, acc1);
                acc0 = __builtin_fmaf(w1, f1.x, acc0), acc1 = __builtin_fmaf(w1, f1.y, acc1);
                acc0 = __builtin_fmaf(w2, f2.x, acc0), acc1 = __builtin_fmaf(w2, f2.y, acc1);
                acc0 = __builtin_fmaf(w3, f3.x, acc0), acc1 = __builtin_fmaf(w3, f3.y, acc1);
            }
        }
        // transpose channel pairs across lanes so each atomic instruction covers 64 consecutive dwords
        acc0 *= scale_f, acc1 *= scale_f;
        float *Fg = F + (int64_t)cur.gid * D + c0;
        {
            const int src = lane >> 1;
            const float a = __shfl(acc0, src, 64), bb = __shfl(acc1, src, 64);
            const float v = (lane & 1) ? bb : a;
            if (lane < cw)
                atomicAdd(Fg + lane, v);
        }
        if (pitch > 64) {
            const int src = 32 + (lane >> 1);
            const float a = __shfl(acc0, src, 64), bb = __shfl(acc1, src, 64);
            const float v = (lane & 1) ? bb : a;
            if (64 + lane < cw)
                atomicAdd(Fg + 64 + lane, v);
        }
        if (chunk == 0 && dsum_out) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
                wacc += __shfl_xor(wacc, o, 64);
            if (lane == 0)
                atomicAdd(dsum_out + cur.gid, wacc * scale_d);
        }
        cur = nxt;
    }
}

// backproject.py:63,166-169 -- one wave per Gaussian row.
__global__ __launch_bounds__(256) void k_finalize(int64_t N, int D, const float *__restrict__ F,
                                                  const float *__restrict__ d, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= N)
        return;
    const float den = 1e-12f + d[g];
    const float *Fg = F + g * D;
    float *og = out + g * D;
    float ss = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float xv = Fg[c] / den;
        ss = __builtin_fmaf(xv, xv, ss);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        ss += __shfl_xor(ss, o, 64);
    const float nrm = __builtin_sqrtf(ss);
    for (int c = lane; c < D; c += 64) {
        float xv = (Fg[c] / den) / nrm;
        if (xv != xv)
            xv = 0.f;
        og[c] = xv;
    }
}

__global__ void k_accum_stats(const Counters *__restrict__ c, gwbp_stats *__restrict__ acc)
{
    acc->n_pairs += c->n_pairs;
    acc->n_isect += c->n_isect;
    acc->n_visible += c->n_visible;
    acc->n_headers += c->n_headers;
    acc->pool_used = max(acc->pool_used, c->pool_head);
    acc->overflow |= c->overflow;
}

static int chunk_pitch(int D)
{
    int p = D < kChunk ? D : kChunk;
    return (p + 3) & ~3; // multiple of 4 floats: 16-B LDS rows, even for the b64 channel pairs
}

// d[g] += scale_d * (sum of the record's weights), one thread per (Gaussian, tile) record; the sums were taken by
// k_blend (Header::wsum).  Used with k_scatter_wide, whose counted-wait visit loop has no room for a conditional
// denominator operation.
__global__ __launch_bounds__(256) void k_accum_d(const u32 *__restrict__ tile_offsets, const u32 *__restrict__ hdr_count,
                                                 const Header *__restrict__ headers, float scale_d,
                                                 float *__restrict__ dsum_out, Counters *__restrict__ ctr)
{
    if (ctr->blend_kind != kBlendHalves) { // no weight sums in this view's headers: refuse, flag (see k_scatter_wide)
        if (blockIdx.x == 0 && threadIdx.x == 0)
            atomicOr(&ctr->overflow, kOverflowMismatch);
        return;
    }
    const int tile = blockIdx.x;
    const u32 nh = hdr_count[tile];
    const Header *hb = headers + tile_offsets[tile];
    for (u32 h = threadIdx.x; h < nh; h += blockDim.x)
        atomicAdd(dsum_out + hb[h].gid, __int_as_float((int)hb[h].wsum) * scale_d);
}

int launch_accum_d(const Layout &L, const Ws &W, const ViewDev &V, float scale_d, float *d, hipStream_t s)
{
    if (L.flags & GWBP_FLAG_NARROW_SCATTER)
        return set_error(GWBP_EINVAL, "gwbp_accumulate_d needs a blend without GWBP_FLAG_NARROW_SCATTER (no weight sums)");
    const int n_tiles = V.tile_w * V.tile_h;
    if (d && n_tiles > 0)
        hipLaunchKernelGGL(k_accum_d, dim3(n_tiles), dim3(256), 0, s, W.tile_offsets, W.hdr_count, W.headers, scale_d, d,
                           W.counters);
    return check_hip(hipGetLastError(), "accum_d launch");
}

int launch_scatter(const Layout &L, const Ws &W, const ViewDev &V, const FeatMap &M, int D, float scale_f,
                   float scale_d, float *F, float *d, hipStream_t s)
{
    const int n_tiles = V.tile_w * V.tile_h;
    const int n_tiles_pad = (n_tiles + 7) & ~7;
    const int n_chunks = (D + kChunk - 1) / kChunk;
    const int pitch = chunk_pitch(D);
    const size_t lds_bytes = (size_t)kTilePix * pitch * sizeof(float) + 16;
    {
        const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(k_scatter), 160 * 1024 - 64, 5);
        if (rc)
            return rc;
    }
    // fast paths: D % 256 == 0 channel-contiguous (scatter_wide.hip); D % 128 == 0 or D <= 64, any strides (scatter_full.hip)
    if (D % 256 == 0 && M.fs_c == 1 && !(L.flags & GWBP_FLAG_NARROW_SCATTER)) {
        if (d) {
            const int rc = launch_accum_d(L, W, V, scale_d, d, s);
            if (rc)
                return rc;
        }
        return launch_scatter_wide(L, W, V, M, D, scale_f, F, s);
    }
    if (D % kChunk == 0 || D <= 64)
        return launch_scatter_full(L, W, V, M, D, scale_f, scale_d, F, d, s);
    else
        hipLaunchKernelGGL(k_scatter, dim3(n_tiles_pad * n_chunks), dim3(kScatterThreads), lds_bytes, s, V,
                           n_tiles_pad, n_chunks, pitch, W.tile_offsets, W.hdr_count, W.headers, W.wpool, M, D, scale_f,
                           scale_d, F, d);
    return check_hip(hipGetLastError(), "scatter launch");
}

int launch_finalize(int64_t N, int D, const float *F, const float *d, float *out, hipStream_t s)
{
    if (N == 0)
        return GWBP_OK;
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, N, D, F, d, out);
    return check_hip(hipGetLastError(), "finalize launch");
}

int launch_accum_stats(const Ws &W, gwbp_stats *accum, hipStream_t s)
{
    hipLaunchKernelGGL(k_accum_stats, dim3(1), dim3(1), 0, s, W.counters, accum);
    return check_hip(hipGetLastError(), "accum_stats launch");
}

} // namespace gwbp
