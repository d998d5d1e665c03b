// gwbp_dev.h -- internal declarations shared by the HIP translation units of libgwbp.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gwbp.h"

namespace gwbp {

typedef unsigned long long u64;
typedef unsigned int u32;

// ---- constants of the arithmetic contract (gsplat 1.4.0 semantics) -------------
constexpr float kAlphaMin = 0x1.010102p-8f;    // 1/255
constexpr float kAlphaMax = 0x1.ff7ceep-1f;    // 0.999
constexpr float kTMin = 0x1.a36e2ep-14f;       // 1e-4
constexpr float kRadiusFloor = 0x1.47ae14p-7f; // 0.01
constexpr float kClampMargin = 0x1.333334p-2f; // 0.3 (x/z clamp margin in tan_fov units)

constexpr int kTile = GWBP_TILE;     // 16 x 16 pixels
constexpr int kTilePix = 256;
constexpr int kPage = 1024;          // weight-pool page (pairs) grabbed per (tile, wave) stream
constexpr int kQueues = 8;           // scatter work queues, one per XCD class (blockIdx % 8), 64 B apart, after the shard heads
constexpr int kShards = 32;          // independently counted regions of the weight pool (one head word per 64-B line)
constexpr int kListPad = 8;          // every record's entry list is padded to a multiple of kListPad entries (kHalves: each half's)
constexpr unsigned kPadPix = 640;    // "pixel" of a padding entry {0, kPadPix}: its slab row lies beyond the 160 KB an LDS allocation
                                     // can have in the 256-channel kernel (1 KB rows), where an out-of-range read returns 0; every
                                     // other consumer masks the padding by the quarters' counts
constexpr int kSortItems = 4096;     // keys per sort block (256 threads x 16)
constexpr int kScanBlock = 256;      // Gaussians per project/emit block
constexpr int kMaxPasses = 4;        // 8-bit passes per sort level (32-bit keys)
// the sort's small tables, cleared with the counters by gwbp_project's memset: digit totals [2 levels][kMaxPasses][256] (one
// k_hist_all per level fills a level's tables from one read of the keys), then one block ticket per (level, pass)
constexpr int kSweepTickets = 2 * kMaxPasses * 256;
constexpr int kSweepWords = kSweepTickets + 2 * kMaxPasses;

// ---- device-resident tables -------------------------------------------------------------------------------
// Projected Gaussian, 32 B, read by the blend kernel with two 16-B loads.
struct __attribute__((aligned(16))) G2D {
    float mx, my, opac, depth;
    float ca, cb, cc;
    int radius; // 0 = culled
};

// One entry of the sparse weight store: w = alpha * T of one (Gaussian, pixel) pair and the pixel's index inside its
// 16x16 tile (row-major, 0..255).  8 B so that eight entries are one s_load_dwordx16 and {w, pix} is an aligned SGPR pair.
struct __attribute__((aligned(8))) WPair {
    float w;
    u32 pix;
};

// One per (Gaussian, tile) pair that contributes at least one weight.  mask[q] bit l <=> pixel q*64+l of the tile
// (row-major 16x16) has a weight; the popc(mask[q]) entries of quarter q are contiguous at wpool[woff[q]..] in
// ascending pixel order.  INVARIANTS the scatter kernels rely on: quarters 0 | 1 are back to back (woff[1] = woff[0] + cnt0) and
// so are quarters 2 | 3; woff[0] % kListPad == 0; the record ends with {0, kPadPix} entries up to the next multiple of kListPad.
// A store blended for the 256-channel kernel (k_blend<kHalves>) additionally pads the FIRST half: woff[2] % kListPad == 0 with
// {0, kPadPix} entries between the halves (k_scatter_wide fetches a half's entries eight at a time); otherwise woff[2] =
// woff[1] + cnt1.  A reader of the whole record as one run (k_scatter_full) takes woff[3] + cnt3 - woff[0] entries and drops
// those with pix >= 256.
// counts = cnt0 | cnt1 << 8 | cnt2 << 16 | cnt3 << 24 (cnt <= 64).
struct __attribute__((aligned(64))) Header {
    u32 gid;
    u32 woff[4];
    u32 counts;
    u32 wsum; // float bits: sum of the record's weights = the record's share of d[gid] (k_accum_d)
    u32 carry_row; // k_blend<kHalves>: the record's rank among its tile's records that have entries in both halves (its carry row
                   // in k_scatter_wide), 0xFFFFFFFF for a record that lies in one half; unused otherwise
    u64 mask[4];
};
static_assert(sizeof(Header) == 64, "header is one 64-B line");

constexpr int kCarryRows = 1024; // carry rows per scatter workgroup; a tile's records beyond it are flushed per half
constexpr int kCarryWgs = 256;   // scatter workgroups that own a carry slice (persistent grid: one per CU)

// Mirrors gwbp_stats (include/gwbp.h) field for field.
struct Counters {
    u64 n_pairs;
    u32 n_isect;
    u32 n_visible;
    u32 n_headers;
    u32 pool_head;
    u32 overflow;
    u32 blend_kind; // gwbp_stats::reserved: kBlendHalves once k_blend<kHalves> has written THIS view's weight sums
};
constexpr u32 kBlendHalves = 1u;
constexpr u32 kBlendFused = 2u; // gwbp_blend_scatter: the view was blended AND scattered, its weight store is empty
constexpr u32 kBlendToken = 3u; // gwbp_blend_tokens: the workspace holds per-(Gaussian, tile) token-quadrant weight sums, no store
constexpr u32 kOverflowMismatch = 4u; // gwbp_stats::overflow bit 2, see include/gwbp.h
constexpr u32 kOverflowRingStall = 16u; // bit 4: a wave of the producer / consumer kernel gave up waiting on its LDS ring (internal error)
constexpr u32 kOverflowTokenGeometry = 8u; // bit 3: gwbp_blend_tokens met a tile that spans more than 2 x 2 tokens
static_assert(sizeof(Counters) == sizeof(gwbp_stats), "Counters must mirror gwbp_stats");

struct Layout {
    size_t total;
    size_t counters, shards, sweep, g2d, rect, touched, blocksums, dkeys[2], dvals[2], keys[2], vals[2], hist, digit_total, tile_offsets, tile_order, hdr_count,
        headers, carry, wpool;
    int64_t n, isect_cap, pair_cap;
    int max_tiles, n_scan_blocks, n_sort_blocks, scatter_wgs, flags;
};

struct Ws {
    Counters *counters;
    u32 *shards; // kShards head words, 16 u32 apart
    u32 *sweep;  // kSweepWords: digit totals of every sort pass + block tickets (zero between gwbp_bin_sort calls)
    G2D *g2d;
    uint2 *rect; // x = xmin | xmax<<16, y = ymin | ymax<<16
    u32 *touched;
    u32 *blocksums;
    u32 *dkeys[2], *dvals[2]; // depth sort of the Gaussians: key = depth bits (0xFFFFFFFF if culled), value = index
    u32 *keys[2];             // intersection sort: key = tile id
    u32 *vals[2];             //                    value = Gaussian index
    u32 *hist;
    u32 *digit_total;
    u32 *tile_offsets;
    u32 *tile_order; // tiles by descending list length (heavy tiles start first in k_blend)
    u32 *hdr_count;
    Header *headers;
    float *carry;        // kCarryWgs x kCarryRows x 256 floats
    WPair *wpool;
};

int make_layout(const gwbp_caps *caps, Layout *L);
int bind_workspace(const gwbp_caps *caps, void *ws, size_t bytes, Layout *L, Ws *W);
int set_error(int code, const char *fmt, ...);
int check_hip(hipError_t e, const char *what);
// Per-device facts, cached per device ORDINAL (a process may drive several GPUs): CU count of the current device, and
// "this kernel's dynamic-LDS limit has been raised on the current device" (slot = one bit per kernel, < 32).
int device_cus(int *n_cu);
int ensure_dynamic_lds(const void *func, int bytes, int slot);
// Profiling / ablation knobs (GWBP_ABLATE, GWBP_ABLATE_BLEND, GWBP_BLEND_LDS): an environment variable is read only by a
// library built with -DGWBP_PROFILE (make PROFILE=1 -> libgwbp_profile.so); the product library always gets 0.
int profile_knob(const char *name);

struct ViewDev { // per-launch copy of gwbp_view (kernel argument, 128 B)
    float R[9];
    float t[3];
    float fx, fy, cx, cy;
    int W, H, tile_w, tile_h;
    float near_plane, far_plane, eps2d, radius_clip;
};
int make_view(const gwbp_view *v, const gwbp_caps *caps, ViewDev *out);

// 8-bit passes of the tile-id sort of the intersections (the depth order comes from the Gaussian pre-sort)
inline int sort_passes(int n_tiles)
{
    int tile_bits = 1;
    while ((1 << tile_bits) < n_tiles)
        ++tile_bits;
    return (tile_bits + 7) / 8;
}

// stage launchers (one per .hip file)
int launch_project(const Layout &L, const Ws &W, const ViewDev &V, const float *means, const float *quats,
                   const float *scales, const float *opac, int32_t *radii, float *means2d, float *depths,
                   float *conics, hipStream_t s);
int launch_emit(const Layout &L, const Ws &W, const ViewDev &V, const u32 *order, hipStream_t s);
int launch_emit_scanned(const Layout &L, const Ws &W, const ViewDev &V, const u32 *order, hipStream_t s);
int launch_bin_sort(const Layout &L, const Ws &W, const ViewDev &V, int64_t *isect_ids, int32_t *flatten_ids,
                    int32_t *tile_offsets, hipStream_t s);
struct FeatMap;
// M != nullptr: the fused small-D form (gwbp_blend_scatter): F[gid, :D] and d are accumulated by the blend itself
int launch_blend(const Layout &L, const Ws &W, const ViewDev &V, float *alphas, float *d, float scale_d, hipStream_t s,
                 const FeatMap *M = nullptr, int D = 0, float scale_f = 1.0f, float *F = nullptr);
// token-space path of a nearest-upsampled low-resolution map (blend.hip: k_blend<kToken>; token.hip)
int launch_blend_tokens(const Layout &L, const Ws &W, const ViewDev &V, float *alphas, const int32_t *ymap, const int32_t *xmap,
                        hipStream_t s);
int launch_zero_omega(const Layout &L, const Ws &W, hipStream_t s);
int launch_token_apply(const Layout &L, const Ws &W, const ViewDev &V, const float *tokens, int64_t ts_y, int64_t ts_x, int D,
                       const int32_t *ymap, const int32_t *xmap, float scale_f, float scale_d, float *F, float *d, hipStream_t s);
// A 2-D feature map as the scatter kernels address it: feats[row(y)*fs_y + col(x)*fs_x + c*fs_c] (strides in floats).
// ymap/xmap (device, optional) send an output pixel to the row/column of a lower-resolution map: the
// F.interpolate(mode="nearest") of backproject.py:244-248 without materialising the upsampled map.  With ly/lx as well
// the map is sampled BILINEARLY (backproject.py:110-112, align_corners=False): ymap/xmap hold the lower texel, ly/lx
// the weight of the upper one, lr_h/lr_w clamp the upper texel; ATen's operation order is kept.
struct FeatMap {
    const float *p;
    int64_t fs_y, fs_x, fs_c;
    const int32_t *ymap, *xmap;
    const float *ly, *lx;
    int32_t lr_h, lr_w;
    // gwbp_scatter_encoded (small-D kernel only): the map has enc_k channels per pixel and is multiplied by
    // enc[enc_k, D] while the tile's slab is staged (backproject_compressed.py:127 without the [H,W,D] intermediate)
    const float *enc = nullptr;
    int32_t enc_k = 0;
    __host__ __device__ __forceinline__ int64_t pixel(int iy, int ix) const
    {
        const int64_t yy = ymap ? ymap[iy] : iy, xx = xmap ? xmap[ix] : ix;
        return yy * fs_y + xx * fs_x;
    }
    __host__ __device__ __forceinline__ bool bilinear() const { return ly != nullptr; }
    // one channel of one pixel; chan = p + c * fs_c
    __device__ __forceinline__ float sample(const float *chan, int iy, int ix) const
    {
        if (!bilinear())
            return chan[pixel(iy, ix)];
        const int y0 = ymap[iy], x0 = xmap[ix];
        const int y1 = min(y0 + 1, lr_h - 1), x1 = min(x0 + 1, lr_w - 1);
        const float h1 = ly[iy], w1 = lx[ix], h0 = 1.0f - h1, w0 = 1.0f - w1;
        const float a = chan[y0 * fs_y + x0 * fs_x], b = chan[y0 * fs_y + x1 * fs_x];
        const float c = chan[y1 * fs_y + x0 * fs_x], d = chan[y1 * fs_y + x1 * fs_x];
        return h0 * (w0 * a + w1 * b) + h1 * (w0 * c + w1 * d); // UpSampleBilinear2d: same association
    }
};
int launch_scatter(const Layout &L, const Ws &W, const ViewDev &V, const FeatMap &M, int D, float scale_f,
                   float scale_d, float *F, float *d, hipStream_t s);
int launch_scatter_full(const Layout &L, const Ws &W, const ViewDev &V, const FeatMap &M, int D, float scale_f,
                        float scale_d, float *F, float *d, hipStream_t s);
int launch_accum_d(const Layout &L, const Ws &W, const ViewDev &V, float scale_d, float *d, hipStream_t s);
int launch_scatter_wide(const Layout &L, const Ws &W, const ViewDev &V, const FeatMap &M, int D, float scale_f,
                        float *F, hipStream_t s);
int launch_render(const Layout &L, const Ws &W, const ViewDev &V, const float *colors, int D, float *out,
                  hipStream_t s);
int launch_render_px(const Ws &W, const ViewDev &V, const float *colors, int D, float *out, float *alphas,
                     hipStream_t s);
int launch_sh_colors(int64_t N, int degree, int K, const float *means, const float *coeffs, const float *campos,
                     float *out, hipStream_t s);
int launch_encode_map(const float *feats, int64_t fs_y, int64_t fs_x, int H, int W, int K, const float *enc, int n_out,
                      float *out, int workgroups, hipStream_t s);
int launch_finalize(int64_t N, int D, const float *F, const float *d, float *out, hipStream_t s);
int launch_dump_pairs(const Layout &L, const Ws &W, const ViewDev &V, int64_t cap, int32_t *gid, int32_t *pix,
                      float *w, u64 *n_dev, hipStream_t s);
int launch_accum_stats(const Ws &W, gwbp_stats *accum, hipStream_t s);

// ---- device helpers -----------------------------------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ float dot3f(float a0, float a1, float a2, float b0, float b1, float b2)
{
    return __builtin_fmaf(a2, b2, __builtin_fmaf(a1, b1, a0 * b0));
}

// exp(x), x <= 0: ln2 hi/lo range reduction, degree-7 Taylor (Horner, fma), exponent added to the bit pattern.
// Deterministic (no v_exp_f32), so weights do not depend on a transcendental unit; ~12 VALU ops.
__device__ __forceinline__ float exp_neg_core(float x) // -80 <= x <= 0
{
    const float t = x * 0x1.715476p+0f;
    const float n = __builtin_rintf(t);
    float r = __builtin_fmaf(n, -0x1.62e4p-1f, x);
    r = __builtin_fmaf(n, -0x1.7f7d1cp-20f, r);
    float p = 0x1.a01a02p-13f;
    p = __builtin_fmaf(p, r, 0x1.6c16c2p-10f);
    p = __builtin_fmaf(p, r, 0x1.111112p-7f);
    p = __builtin_fmaf(p, r, 0x1.555556p-5f);
    p = __builtin_fmaf(p, r, 0x1.555556p-3f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    return __int_as_float(__float_as_int(p) + (((int)n) << 23));
}
__device__ __forceinline__ float exp_neg(float x)
{
    return exp_neg_core(__builtin_fmaxf(x, -80.0f));
}
// exp_neg(-fmaxf(sigma, 0)) bit for bit with ONE clamp instruction: med3(-sigma, 0, -80) is -sigma inside [0, 80], 0 below
// (exp of either zero is 1.0) and -80 above; fmaxf twice costs three (each with a canonicalising v_max in front).
__device__ __forceinline__ float exp_neg_sigma(float sigma)
{
    return exp_neg_core(__builtin_amdgcn_fmed3f(-sigma, 0.0f, -80.0f));
}

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ u32 mbcnt(u64 mask)
{
    return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}
// GWBP_FLAG_FRONT_PRIORITY: the front-stage kernels (project / sort / blend of view v+1) raise their waves' issue
// priority.  Beside the persistent scatter kernel of view v the SIMD arbiter otherwise serves the four older scatter waves
// first and the front's dependent instruction chains stretch 3x (k_blend 1.0 -> 2.9 ms at equal occupancy).
#ifndef GWBP_FRONT_PRIO
#define GWBP_FRONT_PRIO 3 // (levels 1, 2 and 3 measured equal at C2, round 5: 3.43-3.45 ms/view each; 0 = off costs 4 %)
#endif
__device__ __forceinline__ void front_priority(int on)
{
    if (on)
        __builtin_amdgcn_s_setprio(GWBP_FRONT_PRIO);
}
// Sum over the 64 lanes, returned wave-uniform: four DPP adds inside each row of 16 lanes, then the four row totals.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_f<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v); // row_half_mirror
    v += dpp_f<0x140>(v); // row_mirror
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ u32 uniform(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 uniform64(u64 v)
{
    return ((u64)uniform((u32)(v >> 32)) << 32) | uniform((u32)v);
}
#endif

} // namespace gwbp
