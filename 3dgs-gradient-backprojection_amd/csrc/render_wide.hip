// render_wide.hip -- k_render_rows / k_render_rows4: forward render of a D-wide colour table from the weight store,
//   out[p, :] = sum_g w_g(p) * colors[g, :]          (render_colors of rasterization(); segment.py:209-220 renders the
//   512-d feature field this way, and the drop-in operator's forward needs it for the reference's own loop).
//
// k_render_rows (D < 128 or D % 4 != 0): a wave owns ONE ROW of a tile (16 pixels) for one 128-channel chunk; lanes = channel pairs.
// k_render_rows4 (everything else, further down): the same walk with four or eight channels per lane.  The 16 accumulators are
// registers (float2 acc[16]), so there is no LDS image and no read-modify-write chain: a pair costs one v_readlane (w)
// and one v_pk_fma_f32 -- the pixel is a compile-time index because the row's 16 mask bits are tested one by one with
// scalar branches.  Every pixel is summed front to back by its one owner: deterministic, gsplat's order.
//
// Per block of 64 records the lanes look at one header each (mask word of the wave's quarter, Gaussian id, entry
// offset), vote which records touch the row, and the wave then walks those "visits" through a four-slot software
// pipeline: the visit's entry weights (<= 16 dwords) and its colour chunk (512 B) are requested four visits ahead.  All
// loads of a step are unconditional (exhausted slots re-read a valid address), so hipcc's own vmcnt bookkeeping gives
// each step an exact counted wait.
#include "gwbp_dev.h"

namespace gwbp {

namespace {

constexpr int kChunk = 128;
constexpr int kSlots = 4;

struct Desc { // wave-uniform
    u32 gid, off, bits;
    bool have;
};
struct Data {
    float w;
    float2 col;
};

__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

__global__ __launch_bounds__(256) void k_render_rows(ViewDev V, int n_chunks, const u32 *__restrict__ tile_offsets,
                                                     const u32 *__restrict__ hdr_count,
                                                     const Header *__restrict__ headers,
                                                     const WPair *__restrict__ wpool, const float *__restrict__ colors,
                                                     int D, float *__restrict__ out)
{
    // blockIdx -> (tile, quarter, chunk); the 4 * n_chunks blocks of a tile share an XCD (b % 8): headers, entries and
    // colour rows of the tile are pulled from HBM once and re-read from that XCD's L2
    const u32 b = blockIdx.x;
    const u32 x = b & 7u, sidx = b >> 3;
    const u32 per_tile = 4u * (u32)n_chunks;
    const u32 inner = sidx % per_tile;
    const int tile = (int)((sidx / per_tile) * 8u + x);
    if (tile >= V.tile_w * V.tile_h)
        return;
    const int q = (int)(inner & 3u), chunk = (int)(inner >> 2);
    const int lane = threadIdx.x & 63;
    const int wave = (int)uniform(threadIdx.x >> 6);
    const int tx = tile % V.tile_w, ty = tile / V.tile_w;
    const int iy = ty * kTile + 4 * q + wave;
    if (iy >= V.H)
        return; // whole wave; no barriers in this kernel
    const u32 sh = 16u * (u32)wave; // the row's 16 bits inside mask[q]
    const int c0 = chunk * kChunk;
    const int cw = min(kChunk, D - c0);
    const bool on0 = 2 * lane < cw, on1 = 2 * lane + 1 < cw;
    const int cl = on0 ? 2 * lane : 0; // lanes past the chunk re-read channel c0 (valid) and keep zeros
    const u32 nh = uniform(hdr_count[tile]);
    const Header *hbase = headers + uniform(tile_offsets[tile]);

    float2 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
        acc[p] = make_float2(0.f, 0.f);

    for (u32 h0 = 0; h0 < nh; h0 += 64) {
        // ---- lanes = headers: which of these 64 records touch the row, and where their entries are ------------
        const u32 hh = min(h0 + (u32)lane, nh - 1);
        const Header *hp = hbase + hh;
        const u64 m = (h0 + (u32)lane < nh) ? hp->mask[q] : 0ull;
        const u32 my_bits = (u32)(m >> sh) & 0xFFFFu;
        const u32 my_gid = hp->gid;
        const u32 my_off = hp->woff[q] + (u32)__popcll(m & ((1ull << sh) - 1ull));
        u64 rem = __ballot(my_bits != 0u);
        if (rem == 0ull)
            continue;

        Desc ds[kSlots];
        Data dt[kSlots];
        auto refill = [&](Desc &d_, Data &x_) __attribute__((always_inline)) {
            u32 gid_s = d_.gid, off_s = d_.off, bits_s = 0u; // exhausted: re-read the slot's last (valid) addresses
            d_.have = rem != 0ull;
            if (d_.have) {
                const int l = __ffsll((long long)rem) - 1;
                rem &= rem - 1ull;
                gid_s = (u32)__builtin_amdgcn_readlane((int)my_gid, l);
                off_s = (u32)__builtin_amdgcn_readlane((int)my_off, l);
                bits_s = (u32)__builtin_amdgcn_readlane((int)my_bits, l);
            }
            d_.gid = gid_s, d_.off = off_s, d_.bits = bits_s;
            const u32 cnt = (u32)__popc(bits_s);
            x_.w = wpool[off_s + min((u32)lane, cnt ? cnt - 1u : 0u)].w;
            const float *cg = colors + (int64_t)gid_s * D + c0 + cl;
            x_.col.x = cg[0];
            x_.col.y = cg[on1 ? 1 : 0];
        };
        auto process = [&](const Desc &d_, const Data &x_) __attribute__((always_inline)) {
            if (!d_.have)
                return;
            const u32 cnt = (u32)__popc(d_.bits);
            const float wv = ((u32)lane < cnt) ? x_.w : 0.f;
            const float cx_ = on0 ? x_.col.x : 0.f, cy_ = on1 ? x_.col.y : 0.f;
            int k = 0;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                if ((d_.bits >> p) & 1u) { // wave-uniform: s_bitcmp1 + s_cbranch
                    const float w = readlane_f(wv, k);
                    ++k;
                    acc[p].x = __builtin_fmaf(w, cx_, acc[p].x);
                    acc[p].y = __builtin_fmaf(w, cy_, acc[p].y);
                }
            }
        };
        // the first touched record gives every slot a valid address before any refill may run dry
        {
            const int l0 = __ffsll((long long)rem) - 1;
            const u32 g0 = (u32)__builtin_amdgcn_readlane((int)my_gid, l0);
            const u32 o0 = (u32)__builtin_amdgcn_readlane((int)my_off, l0);
#pragma unroll
            for (int s = 0; s < kSlots; ++s)
                ds[s].gid = g0, ds[s].off = o0, ds[s].bits = 0u, ds[s].have = false;
        }
#pragma unroll
        for (int s = 0; s < kSlots; ++s)
            refill(ds[s], dt[s]);
        while (ds[0].have) { // slots fill in order, so slot 0 runs dry first only when everything has
#pragma unroll
            for (int s = 0; s < kSlots; ++s) {
                process(ds[s], dt[s]);
                refill(ds[s], dt[s]);
            }
        }
    }

    // ---- write the row: 16 pixels x (this lane's two channels); a pixel's chunk is 512 contiguous bytes per wave ----
    float *orow = out + ((int64_t)iy * V.W + (int64_t)tx * kTile) * D + c0 + 2 * lane;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        if (tx * kTile + p < V.W) {
            if (on0)
                orow[(int64_t)p * D] = acc[p].x;
            if (on1)
                orow[(int64_t)p * D + 1] = acc[p].y;
        }
    }
}


// D >= 128, D % 4 == 0 (the 512-d field of segment.py:209-220; D = 192: 1.56 ms instead of two 128-channel chunks = 2.44, D = 132:
// 1.47 where k_render_rows takes 1.53 for 128): a wave owns one tile row for a chunk of 256 * Q channels, lanes =
// channel quads (lane l: channels 4 l .. 4 l + 3 of each of the chunk's Q blocks of 256; lanes past the end of a partial last
// block re-read a valid address and store nothing).
// Round 5: k_render_rows issues ~70 instructions per (record, row) visit -- 29 vector, 40 scalar -- of which ~6 are FMAs, and it
// is bound by exactly that: a SIMD issues one vector and one scalar instruction per four cycles, from different waves
// (SQ_ACTIVE_INST_VALU 61 %, SQ_ACTIVE_INST_SCA 65 % of the SIMD time; C2, D = 512: 5.37 ms for 7.1 GB).  The visit's
// bookkeeping (three readlanes, addresses, the walk over the 16 mask bits) does not depend on the channel count, so it is paid
// once per 256 (Q = 1: 3.0 ms) or 512 channels (Q = 2) instead of once per 128, and a pair costs one v_readlane + 2 Q
// v_pk_fma_f32.  Same order of additions per pixel as k_render_rows: bit-identical output.
// (Measured and dropped, profiles/r5_render.txt: the walk over the mask bits as one inline-asm chain entered at the row's first
// pixel and left behind its last entry -- two or three taken branches per visit instead of one per set bit and one back -- is 4 %
// SLOWER: it needs five scalar instructions per pair, and scalar issue counts like vector issue; testing the mask nibble by
// nibble first changes nothing; more visits in flight per wave lose: 2 / 4 / 6 / 8 slots -> 3.20 / 3.07 / 3.38 / 3.47 ms at
// Q = 1, the registers cost occupancy; ONE stream of visits over all the tile's records -- the next 64 headers prefetched, the
// slots never drained between header blocks -- 2.88 against 2.75: the extra test in every refill costs more than the two round
// trips per block it hides.)
#ifndef GWBP_RENDER_SLOTS
#define GWBP_RENDER_SLOTS 4 // visits in flight per wave at 256 channels per wave (95 VGPRs: five waves per SIMD) ...
#endif
#ifndef GWBP_RENDER_SLOTS2
#define GWBP_RENDER_SLOTS2 3 // ... and at 512 (166 VGPRs: three waves per SIMD)
#endif

template <int Q, int SLOTS>
__global__ __launch_bounds__(256) void k_render_rows4(ViewDev V, int n_chunks, const u32 *__restrict__ tile_offsets,
                                                      const u32 *__restrict__ hdr_count,
                                                      const Header *__restrict__ headers,
                                                      const WPair *__restrict__ wpool, const float *__restrict__ colors,
                                                      int D, float *__restrict__ out, int cbase)
{
    struct Data4 {
        float w;
        float4 col[Q];
    };
    const u32 b = blockIdx.x;
    const u32 x = b & 7u, sidx = b >> 3;
    const u32 per_tile = 4u * (u32)n_chunks;
    const u32 inner = sidx % per_tile;
    const int tile = (int)((sidx / per_tile) * 8u + x);
    if (tile >= V.tile_w * V.tile_h)
        return;
    const int q = (int)(inner & 3u), chunk = (int)(inner >> 2);
    const int lane = threadIdx.x & 63;
    const int wave = (int)uniform(threadIdx.x >> 6);
    const int tx = tile % V.tile_w, ty = tile / V.tile_w;
    const int iy = ty * kTile + 4 * q + wave;
    if (iy >= V.H)
        return; // whole wave; no barriers in this kernel
    const u32 sh = 16u * (u32)wave;
    bool on[Q];
    int c0[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const int c = cbase + chunk * 256 * Q + 256 * j + 4 * lane; // (cbase: the first channel of this launch)
        on[j] = c < D;
        c0[j] = on[j] ? c : cbase + chunk * 256 * Q; // (the chunk's first channel exists)
    }
    const u32 nh = uniform(hdr_count[tile]);
    const Header *hbase = headers + uniform(tile_offsets[tile]);

    float4 acc[16][Q];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int j = 0; j < Q; ++j)
            acc[p][j] = make_float4(0.f, 0.f, 0.f, 0.f);

    for (u32 h0 = 0; h0 < nh; h0 += 64) {
        const u32 hh = min(h0 + (u32)lane, nh - 1);
        const Header *hp = hbase + hh;
        const u64 m = (h0 + (u32)lane < nh) ? hp->mask[q] : 0ull;
        const u32 my_bits = (u32)(m >> sh) & 0xFFFFu;
        const u32 my_gid = hp->gid;
        const u32 my_off = hp->woff[q] + (u32)__popcll(m & ((1ull << sh) - 1ull));
        u64 rem = __ballot(my_bits != 0u);
        if (rem == 0ull)
            continue;

        Desc ds[SLOTS];
        Data4 dt[SLOTS];
        auto refill = [&](Desc &d_, Data4 &x_) __attribute__((always_inline)) {
            u32 gid_s = d_.gid, off_s = d_.off, bits_s = 0u;
            d_.have = rem != 0ull;
            if (d_.have) {
                const int l = __ffsll((long long)rem) - 1;
                rem &= rem - 1ull;
                gid_s = (u32)__builtin_amdgcn_readlane((int)my_gid, l);
                off_s = (u32)__builtin_amdgcn_readlane((int)my_off, l);
                bits_s = (u32)__builtin_amdgcn_readlane((int)my_bits, l);
            }
            d_.gid = gid_s, d_.off = off_s, d_.bits = bits_s;
            // lane k < 16 reads entry k of the visit: lanes past its last entry read the next records' entries or the slack behind
            // the pool (make_layout: 1 KB, 15 entries = 120 B are needed) and are never looked at (v_readlane k < count)
            x_.w = wpool[off_s + (u32)(lane & 15)].w;
#pragma unroll
            for (int j = 0; j < Q; ++j)
                x_.col[j] = *reinterpret_cast<const float4 *>(colors + (int64_t)gid_s * D + c0[j]);
        };
        auto process = [&](const Desc &d_, const Data4 &x_) __attribute__((always_inline)) {
            if (!d_.have)
                return;
            int k = 0;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                if ((d_.bits >> p) & 1u) { // wave-uniform
                    const float w = readlane_f(x_.w, k);
                    ++k;
#pragma unroll
                    for (int j = 0; j < Q; ++j) {
                        acc[p][j].x = __builtin_fmaf(w, x_.col[j].x, acc[p][j].x);
                        acc[p][j].y = __builtin_fmaf(w, x_.col[j].y, acc[p][j].y);
                        acc[p][j].z = __builtin_fmaf(w, x_.col[j].z, acc[p][j].z);
                        acc[p][j].w = __builtin_fmaf(w, x_.col[j].w, acc[p][j].w);
                    }
                }
            }
        };
        {
            const int l0 = __ffsll((long long)rem) - 1;
            const u32 g0 = (u32)__builtin_amdgcn_readlane((int)my_gid, l0);
            const u32 o0 = (u32)__builtin_amdgcn_readlane((int)my_off, l0);
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
                ds[s].gid = g0, ds[s].off = o0, ds[s].bits = 0u, ds[s].have = false;
        }
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            refill(ds[s], dt[s]);
        while (ds[0].have) {
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                process(ds[s], dt[s]);
                refill(ds[s], dt[s]);
            }
        }
    }

#pragma unroll
    for (int j = 0; j < Q; ++j) {
        float *orow = out + ((int64_t)iy * V.W + (int64_t)tx * kTile) * D + c0[j];
#pragma unroll
        for (int p = 0; p < 16; ++p)
            if (on[j] && tx * kTile + p < V.W)
                *reinterpret_cast<float4 *>(orow + (int64_t)p * D) = acc[p][j];
    }
}

} // namespace

int launch_render(const Layout &L, const Ws &W, const ViewDev &V, const float *colors, int D, float *out, hipStream_t s)
{
    (void)L;
    const int n_tiles = V.tile_w * V.tile_h;
    const int n_tiles_pad = (n_tiles + 7) & ~7;
    if (D >= 128 && D % 4 == 0 && ((reinterpret_cast<uintptr_t>(colors) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 &&
        !profile_knob("GWBP_RENDER_NARROW")) {
        // whole blocks of 512 channels: 512 per wave (C2, D = 512: 2.75 ms against 3.05 with 256 per wave; 166 registers, three
        // waves per SIMD); what is left (<= 511 channels) in blocks of 256, the last one possibly partial
#ifdef GWBP_RENDER_NO_Q2
        const int n8 = 0;
#else
        const int n8 = D / 512;
#endif
        if (n8 > 0)
            hipLaunchKernelGGL((k_render_rows4<2, GWBP_RENDER_SLOTS2>), dim3((unsigned)n_tiles_pad * 4u * (unsigned)n8), dim3(256), 0, s, V, n8,
                               W.tile_offsets, W.hdr_count, W.headers, W.wpool, colors, D, out, 0);
        const int rest = D - 512 * n8, n4 = (rest + 255) / 256;
        if (n4 > 0)
            hipLaunchKernelGGL((k_render_rows4<1, GWBP_RENDER_SLOTS>), dim3((unsigned)n_tiles_pad * 4u * (unsigned)n4), dim3(256), 0, s,
                               V, n4, W.tile_offsets, W.hdr_count, W.headers, W.wpool, colors, D, out, 512 * n8);
        return check_hip(hipGetLastError(), "render launch");
    }
    const int n_chunks = (D + kChunk - 1) / kChunk;
    hipLaunchKernelGGL(k_render_rows, dim3((unsigned)n_tiles_pad * 4u * (unsigned)n_chunks), dim3(256), 0, s, V, n_chunks,
                       W.tile_offsets, W.hdr_count, W.headers, W.wpool, colors, D, out);
    return check_hip(hipGetLastError(), "render launch");
}

} // namespace gwbp
