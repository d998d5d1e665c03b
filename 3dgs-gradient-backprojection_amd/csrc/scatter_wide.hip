// scatter_wide.hip -- k_scatter_wide: the D % 256 == 0, channel-contiguous path of the weighted scatter-accumulate
// (C2 D = 512, C4 D = 768, dino-sized 1024); semantics identical to k_scatter / k_scatter_full.
//
//   F[g, c0:c0+256] += sum_p w_g(p) * feats[p, c0:c0+256]        (backproject.py:127-131 via colors.grad)
//
// Why a second fast path: k_scatter_full spends 4 vector instructions per (pair, 128 channels) -- 2 v_readlane,
// 1 address add, 1 v_pk_fma_f32 -- and is VALU-issue bound.  With 4 channels per lane (ds_read_b128 + 2 v_pk_fma_f32)
// the two v_readlane and the address add are shared by 256 channels: 2.5 instructions per (pair, 128 channels)
// (tools/ubench_scatter.hip mode 5: 1.53x the per-channel rate of mode 0).  The price is LDS capacity: 256 px x 256 ch
// x 4 B does not fit, so a work item = (tile, 256-channel chunk) runs in TWO passes over half-tile slabs
// (tile rows 0..7, then 8..15; 128 px x 256 ch = 128 KB each).  The flush traffic must not grow with it -- fp32
// atomics run memory-side at ~1.3 TB/s chip-wide and the per-(Gaussian, tile) flush already sits at ~75 % of that --
// so a record with entries in both halves is NOT flushed twice: the top pass parks its partial sums in a carry row
// (plain stores into a per-workgroup slice that stays in L2), the bottom pass starts from them and issues the one
// atomic flush.  k_blend hands over the records as two per-tile lists (HalfHdr), so no wave ever claims an empty visit.
//
// Slab layout: row = pixel (1 KB), position 4*l + k of a row holds channel c0 + 64*k + l.  Lane l reads its 16 B with
// one conflict-free ds_read_b128 and owns channels {l, l+64, l+128, l+192}: the flush is four 256-B contiguous
// atomic wave-instructions with no cross-lane transpose.  Staging loads are therefore dword loads (256 B per
// wave-instruction, coalesced) feeding one ds_write_b128 per (pixel, lane).
//
// Every VMEM instruction of a visit is unconditional WITHIN a pass, so the counted s_waitcnt in front of a visit's entries is
// exact (scatter_full.hip explains why it must be): the top pass issues 2 entry loads + 4 flush operations per visit
// (vmcnt(6)), the bottom pass 2 entry + 4 carry loads + 4 flush operations (vmcnt(10)); the visit loop is instantiated once
// per pass.  The denominator d is not accumulated here: the blend adds every record's weight sum to d itself
// (gwbp_blend_weights_d: one 4-B atomic per record on the front's stream), or k_accum_d (scatter.hip) does from the headers
// when gwbp_scatter is handed d -- a fifth, conditional operation per visit would break the count, and a stand-in for it
// cost more than the kernel saved (in-kernel d with a stand-in store: 4.51 ms/view in the pipeline; separate: 4.25).
// experiments/r1_scatter_wide/README.md has the measurements and the pitfalls of the first attempt.

#include <stdlib.h>

#include <type_traits>

#include "gwbp_dev.h"

namespace gwbp {

#ifdef GWBP_STAMPS
// In-kernel stamps (PROFILE build only, tools/stamp_scatter.py): shader cycles summed over the waves of all workgroups,
// [0] slab staging incl. its barrier, [1] visit loop, [2] drain (s_waitcnt vmcnt(0)), [3] end-of-phase barrier wait,
// [4] phases, [5] visits.
__device__ unsigned long long g_wide_prof[8];
#define GWBP_STAMP(x) const unsigned long long x = __builtin_amdgcn_s_memtime()
#else
#define GWBP_STAMP(x)
#endif

namespace {

constexpr int kWide = 256;            // channels per chunk
constexpr int kHalfPix = kTilePix / 2; // pixels per slab
constexpr int kThreads = 1024;
constexpr int kSlabFloats = kHalfPix * kWide; // 32768 floats = 128 KB
constexpr size_t kLdsBytes = (size_t)kSlabFloats * 4 + 16; // slab + work counter + two item slots

struct Visit { // wave-uniform description of one (record, half) visit
    u32 gid;
    u32 off;  // first entry
    u32 n;    // entries (1..128)
    u32 span; // nonzero: the record has entries in both halves and owns carry row `row`
    u32 row;
};

__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ u32 readlane_u(u32 v, int l) { return (u32)__builtin_amdgcn_readlane((int)v, l); }
// 16-B LDS read at a BYTE ADDRESS (the slab is the start of the kernel's only LDS allocation): lets the compiler fold
// the whole address into one v_lshl_add_u32
__device__ __forceinline__ float4 lds_read_b128(u32 a)
{
#if __HIP_DEVICE_COMPILE__
    return *reinterpret_cast<const __attribute__((address_space(3))) float4 *>((size_t)a);
#else
    (void)a;
    return make_float4(0.f, 0.f, 0.f, 0.f);
#endif
}

struct EV { // entries 64j .. 64j+63 of a visit, one per lane
    float w;
    u32 pix;
};
struct Pre { // everything a visit prefetches: two entry vectors and the four carry dwords of this lane
    EV e[2];
    float c[4];
};
// tied operands: see scatter_full.hip (the load must land in the registers the struct lives in)
__device__ __forceinline__ void issue_e(EV &dst, const WPair *p)
{
    asm volatile("global_load_dwordx2 %0, %1, off" : "+v"(*reinterpret_cast<float2 *>(&dst)) : "v"(p) : "memory");
}
// sc1: served by L2, never by this CU's L1 (the row was written by another wave of this workgroup one pass earlier)
template <int OFF> // byte offset as an instruction immediate: the four carry loads of a visit share ONE 64-bit address
__device__ __forceinline__ void issue_c(float &dst, const float *p)
{
    asm volatile("global_load_dword %0, %1, off offset:%2 sc1" : "+v"(dst) : "v"(p), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_pre(Pre &x)
{
    asm volatile("s_waitcnt vmcnt(%6)"
                 : "+v"(*reinterpret_cast<float2 *>(&x.e[0])), "+v"(*reinterpret_cast<float2 *>(&x.e[1])),
                   "+v"(x.c[0]), "+v"(x.c[1]), "+v"(x.c[2]), "+v"(x.c[3])
                 : "n"(N)
                 : "memory");
}

// Structure-preserving ablations (make PROFILE=1 ABL=<bits>; results INVALID by design, never in the product library):
// compile-time, so every build keeps the visit's VMEM count and hence its counted waits.
//   1  every flush = plain stores into ONE L2-resident row of the workgroup's carry slice (no memory-side atomic cost)
//   2  no LDS reads / FMAs
//   4  no slab staging
//   8  with 1: only 3 of 8 visits flush that way (what merging 2 x 2 tile blocks would save)
//  16  parks and resumes all use carry row 0 (no carry traffic beyond L2)
#if defined(GWBP_PROFILE) && defined(GWBP_ABL)
constexpr int kAbl = GWBP_ABL;
#else
constexpr int kAbl = 0;
#endif

constexpr int kLoads = 6;  // VMEM loads per visit (prefetch)
constexpr int kFlush = 4;  // VMEM flush operations per visit

__global__ __launch_bounds__(kThreads) void k_scatter_wide(
    ViewDev V, int n_chunks, const u32 *__restrict__ tile_offsets, const u32 *__restrict__ cnt_a,
    const u32 *__restrict__ cnt_b, const HalfHdr *__restrict__ half_a, const HalfHdr *__restrict__ half_b,
    const WPair *__restrict__ wpool, FeatMap M, int D, float scale_f, float *__restrict__ F,
    u32 *__restrict__ queues, float *__restrict__ carry_all, Counters *__restrict__ ctr)
{
    // The half-tile lists exist only if THIS view was blended without GWBP_FLAG_NARROW_SCATTER; otherwise they are
    // uninitialised or a previous view's.  Refuse (F untouched, overflow bit 2 raised) instead of scattering garbage.
    if (uniform(ctr->blend_kind) != kBlendHalves) {
        if (blockIdx.x == 0 && threadIdx.x == 0)
            atomicOr(&ctr->overflow, kOverflowMismatch);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) float lds[];
    u32 *s_next = reinterpret_cast<u32 *>(lds + kSlabFloats);
    u32 *s_item = s_next + 1; // two slots: iteration k reads [k & 1], thread 0 fills [(k + 1) & 1] meanwhile

    // persistent workgroups, per-XCD-class queues: as k_scatter_full
    const u32 xcls = blockIdx.x & 7u;
    const int n_tiles = V.tile_w * V.tile_h;
    const u32 n_items = (u32)((n_tiles - (int)xcls + 7) / 8) * (u32)n_chunks;
    u32 *queue = queues + xcls * 16;
    const int lane = threadIdx.x & 63;
    float *carry = carry_all + (size_t)blockIdx.x * kCarryRows * kWide; // this workgroup's slice
    const float *feats = M.p;
    if (threadIdx.x == 0)
        s_item[0] = atomicAdd(queue, 1u);
    __syncthreads();
#ifdef GWBP_STAMPS
    unsigned long long prof_acc[6] = {0, 0, 0, 0, 0, 0};
#endif
    for (u32 k = 0;; ++k) {
    const u32 item = uniform(s_item[k & 1u]); // wave-uniform by construction: keep every derived address scalar
    if (item >= n_items)
        break;
    const int chunk = (int)(item % (u32)n_chunks);
    const int tile = (int)((item / (u32)n_chunks) * 8u + xcls);
    const int tx = tile % V.tile_w, ty = tile / V.tile_w;
    const int c0 = chunk * kWide;

#pragma unroll 1
    for (int phase = 0; phase < 2; ++phase) {
    GWBP_STAMP(ts0);
    const u32 nh = uniform(phase ? cnt_b[tile] : cnt_a[tile]);
    const HalfHdr *hbase = (phase ? half_b : half_a) + tile_offsets[tile];
    if (threadIdx.x == 0)
        *s_next = 0;
    u32 nxt = 0;
    if (phase == 0 && threadIdx.x == 0)
        nxt = atomicAdd(queue, 1u); // claim the next item under the slab loads
    if (!(kAbl & 4) && nh != 0 && M.bilinear()) {
        // Bilinear low-resolution map (backproject.py:110-112 folded in): every slab value is the blend of four texels
        // (L2 / Infinity-Cache resident: the 480 x 480 x 512 map of the lseg script is 472 MB), in ATen's association.
        // One (pixel, lane) unit per round: 16 dword loads in flight per thread.
        constexpr int kAll = kHalfPix * 64;
        constexpr int kUnits = (kAll + kThreads - 1) / kThreads;
#pragma unroll 1
        for (int u = 0; u < kUnits; ++u) {
            const int idx = min(u * kThreads + (int)threadIdx.x, kAll - 1);
            const int pix = phase * kHalfPix + (idx >> 6);
            const int ix = min(tx * kTile + (pix & 15), V.W - 1), iy = min(ty * kTile + (pix >> 4), V.H - 1);
            const int y0 = M.ymap[iy], x0 = M.xmap[ix];
            const int y1 = min(y0 + 1, M.lr_h - 1), x1 = min(x0 + 1, M.lr_w - 1);
            const float h1 = M.ly[iy], w1 = M.lx[ix], h0 = 1.0f - h1, w0 = 1.0f - w1;
            const float *b0 = feats + c0 + lane;
            const float *pa = b0 + y0 * M.fs_y + x0 * M.fs_x, *pb = b0 + y0 * M.fs_y + x1 * M.fs_x;
            const float *pc = b0 + y1 * M.fs_y + x0 * M.fs_x, *pd = b0 + y1 * M.fs_y + x1 * M.fs_x;
            float r[4];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
                r[k4] = h0 * (w0 * pa[64 * k4] + w1 * pb[64 * k4]) + h1 * (w0 * pc[64 * k4] + w1 * pd[64 * k4]);
            if (kAll % kThreads == 0 || u * kThreads + (int)threadIdx.x < kAll)
                *reinterpret_cast<float4 *>(lds + (idx >> 6) * kWide + 4 * lane) = make_float4(r[0], r[1], r[2], r[3]);
        }
    } else if (!(kAbl & 4) && nh != 0) {
        // stage 128 px x 256 ch: unit = (pixel, lane) -> 4 coalesced dword loads + one ds_write_b128; 8 units per thread
        constexpr int kAll = kHalfPix * 64;                          // 8192 (pixel, lane) units
        constexpr int kUnits = (kAll + kThreads - 1) / kThreads;     // 8 at 1024 threads
        float4 vals[kUnits];
        int64_t offs[kUnits]; // pixel offsets first: with index maps they are loads themselves, and must not sit
                              // between the feature loads (their wait would serialise the slab fetch)
#pragma unroll
        for (int u = 0; u < kUnits; ++u) {
            const int idx = min(u * kThreads + (int)threadIdx.x, kAll - 1);
            const int p = idx >> 6; // 0..127 inside the half; l = idx & 63 = lane
            const int pix = phase * kHalfPix + p;
            const int ix = tx * kTile + (pix & 15), iy = ty * kTile + (pix >> 4);
            // pixels past the image edge are never referenced by an entry: load a clamped (valid) address
            offs[u] = M.pixel(min(iy, V.H - 1), min(ix, V.W - 1));
        }
#pragma unroll
        for (int u = 0; u < kUnits; ++u) {
            const float *src = feats + offs[u] + c0 + lane;
            vals[u] = make_float4(src[0], src[64], src[128], src[192]);
        }
#pragma unroll
        for (int u = 0; u < kUnits; ++u) {
            const int idx = u * kThreads + threadIdx.x;
            if (kAll % kThreads == 0 || idx < kAll)
                *reinterpret_cast<float4 *>(lds + (idx >> 6) * kWide + 4 * lane) = vals[u];
        }
    }
    if (phase == 0 && threadIdx.x == 0)
        s_item[(k + 1u) & 1u] = nxt;
    __syncthreads();
    GWBP_STAMP(ts1);
#ifdef GWBP_STAMPS
    u32 n_vis_prof = 0;
#endif

    // dynamic LDS starts at address 0 (no static __shared__ in this kernel): slab row r lives at byte r * 1024
    const u32 row_base = (u32)(lane * 16) - (phase ? (u32)(kHalfPix << 10) : 0u);

    auto claim = [&]() __attribute__((always_inline)) -> u32 {
        // one lane, one LDS atomic -- written as asm so that hipcc's atomic optimiser does not wrap the already
        // single-lane add in its wave-aggregation sequence (v_mbcnt x2, s_bcnt1, compare, second exec mask, add: ~8
        // instructions per visit).  The counter sits right behind the slab; dynamic LDS starts at address 0.
        u32 h = 0;
        if (lane == 0) {
            const u32 addr = (u32)(kSlabFloats * sizeof(float)), one = 1u;
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(h) : "v"(addr), "v"(one) : "memory");
        }
        return uniform(h);
    };
    auto load_visit = [&](u32 h) __attribute__((always_inline)) -> Visit { // scalar loads; an invalid claim re-reads the last header (never processed)
        const HalfHdr *hp = hbase + min(h, nh - 1);
        Visit r;
        r.gid = uniform(hp->gid);
        r.off = uniform(hp->off);
        const u32 ns = uniform(hp->n_span);
        r.n = ns & 0xFFu;
        r.span = ns & 0x100u;
        r.row = uniform(hp->row);
        return r;
    };
    // exactly 2 (top pass) / kLoads (bottom pass) VMEM loads: the top pass never resumes a record, so it issues no carry
    // loads at all (they were ~1.8 M x 4 L2 requests per view that fetched nothing)
    auto prefetch = [&](const Visit &R, Pre &x, auto bottom) __attribute__((always_inline)) {
        const u32 last = R.n - 1;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            issue_e(x.e[j], wpool + (R.off + min((u32)(64 * j + lane), last)));
        if constexpr (decltype(bottom)::value) {
            // carry dwords of this lane (non-spanning records: row 0, value ignored -- the count must stay exact)
            const float *cr = carry + (size_t)((R.span && !(kAbl & 16)) ? R.row : 0u) * kWide + lane;
            issue_c<0>(x.c[0], cr);
            issue_c<256>(x.c[1], cr);
            issue_c<512>(x.c[2], cr);
            issue_c<768>(x.c[3], cr);
        }
    };

    float4 acc;
    // n in 1..64 entries held by lanes 0..n-1 of ev (lanes >= n: w = 0, pix = any pixel)
    auto run_vec = [&](const EV &ev, u32 n) __attribute__((always_inline)) {
        // ONE batch of 8 float4 in flight (32 VGPRs): the other waves of the SIMD cover the LDS latency between the
        // eight reads and the first FMA.  Two batches (the k_scatter_full scheme) put this kernel at 117 VGPRs: 16 waves
        // then own the CU's register file, the overlapped front-stage kernels cannot co-reside and the two-stream
        // pipeline degenerates to the serial schedule.  At 83 VGPRs the front runs beside it as with k_scatter_full.
        constexpr int kB = 8;
        float4 f[kB];
        // LDS address = slab row of the pixel + this lane's 16 B: ONE v_lshl_add_u32 per pair straight from the
        // v_readlane'd pixel index (no scalar mask/shift: row_base already carries -128 rows in the bottom pass, where
        // every real entry has pix >= 128; the {0, 0} padding entries then point below the slab -- an out-of-range LDS
        // read returns 0 and their weight is 0 anyway).
#define GWBP_ISSUE(B, J0, J1) /* (a half batch for short tails was measured slower: 4.00 vs 3.93 ms/view) */                                                                                         \
    _Pragma("unroll") for (int j = J0; j < J1; ++j)                                                                   \
    {                                                                                                                 \
        const u32 px_ = readlane_u(ev.pix, kB * (B) + j);                                                             \
        f[j] = lds_read_b128((px_ << 10) + row_base);                                                                 \
    }
#define GWBP_FMA(B, J0, J1)                                                                                           \
    _Pragma("unroll") for (int j = J0; j < J1; ++j)                                                                   \
    {                                                                                                                 \
        const float w = readlane_f(ev.w, kB * (B) + j);                                                               \
        acc.x = __builtin_fmaf(w, f[j].x, acc.x);                                                                     \
        acc.y = __builtin_fmaf(w, f[j].y, acc.y);                                                                     \
        acc.z = __builtin_fmaf(w, f[j].z, acc.z);                                                                     \
        acc.w = __builtin_fmaf(w, f[j].w, acc.w);                                                                     \
    }
#pragma unroll
        for (int B = 0; B < 64 / kB; ++B) {
            if ((u32)kB * B >= n)
                break;
            GWBP_ISSUE(B, 0, kB)
            GWBP_FMA(B, 0, kB)
        }
#undef GWBP_ISSUE
#undef GWBP_FMA
    };
    auto process = [&](const Visit &R, const Pre &x) __attribute__((always_inline)) { // exactly kFlush VMEM operations
#ifdef GWBP_STAMPS
        ++n_vis_prof;
#endif
        const bool resume = phase && R.span;
        acc = resume ? make_float4(x.c[0], x.c[1], x.c[2], x.c[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (!(kAbl & 2)) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if ((u32)(64 * j) >= R.n)
                    break;
                const u32 n = min(64u, R.n - 64u * j);
                EV ev;
                ev.w = ((u32)lane < n) ? x.e[j].w : 0.f; // clamped loads / the other half's entries: zero them ...
                // ... and send them past the slab (out-of-range LDS reads return 0): 0 x inf at the record's last pixel is NaN
                ev.pix = ((u32)lane < n) ? x.e[j].pix : 640u;
                run_vec(ev, n);
            }
        }
        if (!phase && R.span) { // park the partial sums: plain stores, same shape as the atomics
            float *cr = carry + (size_t)((kAbl & 16) ? 0u : R.row) * kWide + lane;
            cr[0] = acc.x, cr[64] = acc.y, cr[128] = acc.z, cr[192] = acc.w;
        } else {
            float *Fg = F + (int64_t)R.gid * D + c0 + lane;
            if (scale_f != 1.0f) // wave-uniform; the .sum() reduction of backproject.py:127 needs no scaling
                acc.x *= scale_f, acc.y *= scale_f, acc.z *= scale_f, acc.w *= scale_f;
            if (!(kAbl & 1) || ((kAbl & 8) && (R.gid & 7u) >= 3u)) {
                atomicAdd(Fg, acc.x);
                atomicAdd(Fg + 64, acc.y);
                atomicAdd(Fg + 128, acc.z);
                atomicAdd(Fg + 192, acc.w);
            } else { // ablation: same VMEM count, no memory-side cost
                float *dump = carry + (size_t)(kCarryRows - 1) * kWide + lane;
                dump[0] = acc.x, dump[64] = acc.y, dump[128] = acc.z, dump[192] = acc.w;
            }
        }
    };

    // visit pipeline: header scalar loads two visits ahead, entry (+ carry) loads one visit ahead (A/B buffers); one copy
    // of the loop per pass, because the number of loads per visit -- hence the counted wait -- differs
    auto visits = [&](auto bottom) __attribute__((always_inline)) {
        constexpr int kL = decltype(bottom)::value ? kLoads : 2;
        Pre pA = {{{0.f, 0u}, {0.f, 0u}}, {0.f, 0.f, 0.f, 0.f}}, pB = pA;
        u32 h = claim();
        if (h < nh) {
            Visit Rcur = load_visit(h);
            prefetch(Rcur, pA, bottom);
            h = claim();
            bool vnxt = h < nh;
            Visit Rnxt = load_visit(h);

            // peeled first visit: only loads(1) are guaranteed younger than loads(0)
            prefetch(Rnxt, pB, bottom);
            h = claim();
            bool vnn = h < nh;
            Visit Rnn = load_visit(h);
            wait_pre<kL>(pA);
            process(Rcur, pA);
            while (vnxt) {
                // odd: current visit's data in pB; next loads into pA
                Rcur = Rnxt, Rnxt = Rnn, vnxt = vnn;
                prefetch(Rnxt, pA, bottom);
                h = claim();
                vnn = h < nh;
                Rnn = load_visit(h);
                wait_pre<kL + kFlush>(pB);
                process(Rcur, pB);
                if (!vnxt)
                    break;
                // even: current in pA; next into pB
                Rcur = Rnxt, Rnxt = Rnn, vnxt = vnn;
                prefetch(Rnxt, pB, bottom);
                h = claim();
                vnn = h < nh;
                Rnn = load_visit(h);
                wait_pre<kL + kFlush>(pA);
                process(Rcur, pA);
            }
        }
    };
    if (phase)
        visits(std::true_type{});
    else
        visits(std::false_type{});
    // Drain: (1) the last prefetch still targets pA/pB's registers, (2) the carry rows parked in the top pass must be
    // in L2 before any wave of the bottom pass loads them, (3) the slab is about to be overwritten.
    GWBP_STAMP(ts2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GWBP_STAMP(ts3);
    __syncthreads();
#ifdef GWBP_STAMPS
    {
        GWBP_STAMP(ts4);
        prof_acc[0] += ts1 - ts0, prof_acc[1] += ts2 - ts1, prof_acc[2] += ts3 - ts2, prof_acc[3] += ts4 - ts3;
        prof_acc[4] += 1ull, prof_acc[5] += (unsigned long long)n_vis_prof;
    }
#endif
    } // phase
    } // item loop
#ifdef GWBP_STAMPS
    if (lane == 0)
        for (int i = 0; i < 6; ++i)
            atomicAdd(&g_wide_prof[i], prof_acc[i]);
#endif
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

#ifdef GWBP_STAMPS
extern "C" int gwbp_profile_read_wide(unsigned long long *out8_host)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out8_host, HIP_SYMBOL(g_wide_prof), sizeof(z)) != hipSuccess)
        return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_wide_prof), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

int launch_scatter_wide(const Layout &L, const Ws &W, const ViewDev &V, const FeatMap &M, int D, float scale_f,
                        float *F, hipStream_t s)
{
    int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(k_scatter_wide), (int)kLdsBytes, 0);
    if (rc)
        return rc;
    int n_cu = 0;
    if ((rc = device_cus(&n_cu)))
        return rc;
    // persistent workgroups: one per CU, at most kCarryWgs (each owns a carry slice), a multiple of the 8 XCD classes
    int grid = L.scatter_wgs > 0 ? L.scatter_wgs : n_cu;
    grid = (grid + 7) & ~7;
    if (grid > kCarryWgs)
        grid = kCarryWgs;
    u32 *queues = W.shards + kShards * 16;
    hipLaunchKernelGGL(k_scatter_wide, dim3(grid), dim3(kThreads), kLdsBytes, s, V, D / kWide, W.tile_offsets,
                       W.half_count[0], W.half_count[1], W.half[0], W.half[1], W.wpool, M, D, scale_f, F, queues, W.carry,
                       W.counters);
    return check_hip(hipGetLastError(), "scatter_wide launch");
}

} // namespace gwbp
