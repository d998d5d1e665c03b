// render_wide.hip -- k_render_rows: forward render of a D-wide colour table from the weight store,
//   out[p, :] = sum_g w_g(p) * colors[g, :]          (render_colors of rasterization(); segment.py:209-220 renders the
//   512-d feature field this way, and the drop-in operator's forward needs it for the reference's own loop).
//
// A wave owns ONE ROW of a tile (16 pixels) for one 128-channel chunk; lanes = channel pairs.  The 16 accumulators are
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

} // namespace

int launch_render(const Layout &L, const Ws &W, const ViewDev &V, const float *colors, int D, float *out, hipStream_t s)
{
    (void)L;
    const int n_tiles = V.tile_w * V.tile_h;
    const int n_tiles_pad = (n_tiles + 7) & ~7;
    const int n_chunks = (D + kChunk - 1) / kChunk;
    hipLaunchKernelGGL(k_render_rows, dim3((unsigned)n_tiles_pad * 4u * (unsigned)n_chunks), dim3(256), 0, s, V, n_chunks,
                       W.tile_offsets, W.hdr_count, W.headers, W.wpool, colors, D, out);
    return check_hip(hipGetLastError(), "render launch");
}

} // namespace gwbp
