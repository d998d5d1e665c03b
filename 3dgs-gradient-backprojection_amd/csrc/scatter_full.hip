// scatter_full.hip -- k_scatter_full: the D % 128 == 0 (and D <= 64) fast path of the weighted scatter-accumulate
// (C2/C4/C5-input: D = 512 / 768 / 1024; dino 384); semantics identical to k_scatter (scatter.hip).  Channel-contiguous
// maps are staged with 16-B loads, any other strides element-wise; optional row/column maps address a low-resolution
// feature map (nearest upsampling, backproject.py:244-248).
//
//   F[g, c0:c0+128] += sum_p w_g(p) * feats[p, c0:c0+128]        (backproject.py:127-131 via colors.grad)
//
// work item = (16x16 tile, 128-channel chunk); 1024 threads = 16 waves, ONE PERSISTENT workgroup per CU (128 KB LDS)
// pulling items from a per-XCD-class queue:
//   * the tile's 256 px x 128 ch slab is staged once (eight 16-B loads per thread in flight, then eight LDS writes):
//     every feature byte is read from HBM exactly once per view
//   * a wave owns one (Gaussian, tile) record at a time (LDS work counter); the Gaussian is wave-uniform and
//     lanes = channel pairs (ds_read_b64: conflict-free 512-B rows)
//   * the record's {w, pixel} entries are contiguous in the weight store (k_blend writes a record's four quarter
//     lists back to back): lane s of load j fetches entry 64j + s with one 8-B load, so a typical record
//     (45 entries) costs ONE coalesced 512-B load.  The accumulate loop walks the entries in batches of 8 with
//     compile-time lane selects: per pair 2 v_readlane (w, pixel) + 1 v_lshl_add (LDS address) + 1 ds_read_b64 +
//     1 v_pk_fma_f32; the next batch's eight LDS reads are issued before the current batch's FMAs.
//     Measured ceiling of exactly this instruction mix (tools/ubench_scatter.hip, 16 waves/CU): 453 pairs/us/CU
//     = 2.98 ms per C2 view; ds_read_b64 + v_pk_fma alone: 1.86 ms.
//   * records are software-pipelined: claim + header scalar loads two records ahead, entry loads one record ahead.
//     The entry loads are inline asm and awaited with a COUNTED s_waitcnt: vmcnt retires in order and a float atomic
//     stays counted for ~3000 cycles under load, so the vmcnt(0) hipcc would insert in front of every record's
//     entries serialised each record behind its predecessor's atomics (first profile: 7.2 ms/view).
//         [loads(i): 2] [atomics(i-1): 2 F (+1 d)] [loads(i+1): 2]   ->   s_waitcnt vmcnt(4)
//     Every VMEM instruction in the steady-state loop is unconditional, so 4 is a guaranteed lower bound of the
//     younger operations (an over-wait is always safe).
//   * flush: channel pairs are transposed across lanes (ds_bpermute) so each of the two atomic wave-instructions
//     covers 64 consecutive dwords (256 contiguous bytes: the shape that runs at the full fp32 atomic rate)
#include <stdlib.h>

#include "gwbp_dev.h"

namespace gwbp {

namespace {

constexpr int kChunk = 128;
constexpr int kThreads = 1024;
constexpr u32 kLdsMax = 160u * 1024u; // no LDS allocation on gfx950 is larger: a read at or beyond this byte offset returns 0
constexpr int kSlabFloats = kTilePix * kChunk; // 32768 floats = 128 KB
constexpr size_t kLdsBytes = (size_t)kSlabFloats * 4 + 16; // slab + work counter + two item slots

struct Rec { // wave-uniform (SGPR) description of one (Gaussian, tile) record
    u32 gid;
    u32 woff; // first entry; the record's entries (all four quarters) are contiguous in the weight store
    u32 T;    // entries in the record (1..256)
};

__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ u32 readlane_u(u32 v, int l) { return (u32)__builtin_amdgcn_readlane((int)v, l); }

struct EV { // entries 64j .. 64j+63 of a record, one per lane
    float w;
    u32 pix;
};
// "+v" (tied operand): the load lands in the SAME physical registers that currently hold dst.  With a plain "=v"
// output hipcc may rename the destination per iteration and reconcile the names with a v_mov on a loop back-edge --
// a copy of registers whose data has not arrived yet (observed in a single-body variant of the record loop).
__device__ __forceinline__ void issue_e(EV &dst, const WPair *p)
{
    asm volatile("global_load_dwordx2 %0, %1, off" : "+v"(*reinterpret_cast<float2 *>(&dst)) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_e(EV (&e)[2])
{
    asm volatile("s_waitcnt vmcnt(%2)"
                 : "+v"(*reinterpret_cast<float2 *>(&e[0])), "+v"(*reinterpret_cast<float2 *>(&e[1]))
                 : "n"(N)
                 : "memory");
}

// Small-D inner loop with SEVERAL pairs per wave-instruction: with D <= 32 channels a wave has room for P = 64 / Dp pairs
// at once (Dp = D rounded up to 4, 8 or 16), lane = (pair slot, channel).  The slot's {w, pixel} comes from the
// lane that holds the entry (ds_bpermute), so a step of P pairs costs 2 ds_bpermute + ds_read_b32 + v_fmac instead of
// P x (2 v_readlane + address + ds_read_b32 + v_fmac).  The caller sums the P partial results of a channel afterwards.
template <int LGD>
__device__ __forceinline__ float packed_small_vec(const char *slab, u32 row_bytes, u32 cbase, int lane, float ev_w,
                                                  u32 ev_pix, u32 n, float acc)
{
    constexpr int P = 64 >> LGD;                 // pairs per step
    constexpr int kSteps = 64 / P;               // steps that cover one 64-entry vector
    constexpr int SB = 4;                        // steps per batch (LDS operations in flight; 8 would cost the 64-VGPR budget of two workgroups per CU)
    const int slot4 = (lane >> LGD) * 4;         // byte address of this lane's entry inside a step
#pragma unroll 1
    for (int t0 = 0; t0 < kSteps; t0 += SB) {
        if ((u32)(P * t0) >= n) // wave-uniform; entries past n carry w = 0, so whole batches can go
            break;
        float wv[SB], fv[SB];
        u32 pv[SB];
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            const int addr = slot4 + 4 * P * (t0 + i); // entry P*(t0+i) + slot, always < 64
            wv[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(ev_w)));
            pv[i] = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)ev_pix);
        }
#pragma unroll
        for (int i = 0; i < SB; ++i)
            fv[i] = *reinterpret_cast<const float *>(slab + (pv[i] * row_bytes + cbase));
#pragma unroll
        for (int i = 0; i < SB; ++i)
            acc = __builtin_fmaf(wv[i], fv[i], acc);
    }
    return acc;
}

// SMALL = false: D % 128 == 0, 128-channel chunks, lanes = channel pairs (ds_read_b64 + v_pk_fma_f32).
// SMALL = true : D <= 64 (C1 D = 32, C5 D = 16, the drop-in's 3-channel denominator pass), one chunk, lane l = channel l
//                (ds_read_b32 + v_fmac), any feature-map strides, slab pitch = D rounded up to 4 floats.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int kEncN = 16; // output channels of one MFMA tile of the fused encoder (D <= 16)

template <bool SMALL, int VEC> // VEC (slab staging): 1 = 16-B loads, 2 = 16-B loads + bilinear blend, 0 = element-wise,
                               // 3 (SMALL only) = the 512 -> 16 encoder fused in: slab = pixels @ encoder on the MFMA units
__global__ __launch_bounds__(kThreads) void k_scatter_full(
    ViewDev V, int n_chunks, const u32 *__restrict__ tile_offsets, const u32 *__restrict__ hdr_count,
    const Header *__restrict__ headers, const WPair *__restrict__ wpool, FeatMap M, int pitch_rt, int D,
    float scale_f, float scale_d, float *__restrict__ F, float *__restrict__ dsum_out, u32 *__restrict__ queues,
    int dbg_arg)
{
#ifdef GWBP_PROFILE
    const int dbg = dbg_arg; // ablation bits (make PROFILE=1 only; results invalid)
#else
    constexpr int dbg = 0;   // the product kernel does not even contain the ablation branches
    (void)dbg_arg;
#endif

    const float *__restrict__ feats = M.p;
    const int64_t fs_c = M.fs_c;
    const int pitch = SMALL ? pitch_rt : kChunk;
    // dynamic LDS only (no static __shared__ in front of it: the carve base stays 16-B aligned)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    u32 *s_next = reinterpret_cast<u32 *>(lds + kTilePix * pitch);

    // PERSISTENT workgroups: the grid is one workgroup per CU; each pulls (tile, chunk) items from the work queue of
    // its XCD class.  Blocks b and b+8 share an XCD, so class x = b % 8 owns the tiles t with t % 8 == x and a tile's
    // chunks are consecutive items of one queue: the weight store is pulled from HBM once and re-read from that XCD's
    // L2.  The next item is claimed (one returning atomic by thread 0) while the current slab loads are in flight --
    // the vmcnt(0) the slab staging needs anyway covers it, so dynamic scheduling costs no extra wait.  Compared with
    // one workgroup per item this removes ~105 workgroup launches per CU per view and keeps the CU's LDS claimed, so
    // the overlapped front-stage kernels (ViewPipeline) can never take over a CU between two scatter workgroups.
    const u32 xcls = blockIdx.x & 7u;
    const int n_tiles = V.tile_w * V.tile_h;
    const u32 n_items = (u32)((n_tiles - (int)xcls + 7) / 8) * (u32)n_chunks; // tiles of this class x chunks
    u32 *queue = queues + xcls * 16;
    u32 *s_item = s_next + 1; // two slots: iteration k reads [k & 1], thread 0 fills [(k + 1) & 1] meanwhile
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0)
        s_item[0] = atomicAdd(queue, 1u);
    // VEC == 3: the encoder lives in LDS behind the slab for the whole (persistent) kernel, re-ordered so that MFMA step
    // (j, i) reads one conflict-free 256-B row: s_enc[((j * 4 + i) * 4 + q) * 16 + n] = enc[16 j + 4 q + i][n]
    float *s_enc = lds + kTilePix * pitch + 4;
    if constexpr (SMALL && VEC == 3) {
        for (int idx = threadIdx.x; idx < M.enc_k * kEncN; idx += kThreads) {
            const int n = idx & 15, q = (idx >> 4) & 3, i = (idx >> 6) & 3, j = idx >> 8;
            s_enc[idx] = n < D ? M.enc[(int64_t)(16 * j + 4 * q + i) * D + n] : 0.f;
        }
    }
    __syncthreads();
    for (u32 k = 0;; ++k) {
    const u32 item = uniform(s_item[k & 1u]); // wave-uniform by construction: keep every derived address scalar
    if (item >= n_items)
        break;
    const int chunk = (int)(item % (u32)n_chunks);
    const int tile = (int)((item / (u32)n_chunks) * 8u + xcls);
    const u32 nh = hdr_count[tile];
    const int tx = tile % V.tile_w, ty = tile / V.tile_w;
    const int c0 = chunk * kChunk;
    if (threadIdx.x == 0)
        *s_next = 0;

    u32 nxt = 0;
    if (threadIdx.x == 0)
        nxt = atomicAdd(queue, 1u); // claim the next item; the value is only needed after the slab is staged
    if (!(dbg & 4) && nh != 0) {
        if constexpr (SMALL && VEC == 3) {
            // Fused encoder (backproject_compressed.py:127): wave w stages tile row w -- 16 pixels x enc_k channels read
            // straight from the full-width map (the only HBM stream of this kernel: 512 KB per tile at enc_k = 512),
            // times the encoder on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, a k-ordered fmaf chain), 16 x 16
            // results into the slab.  Lane (m = lane % 16, q = lane / 16) loads the float4 of pixel m at channels
            // 16 j + 4 q .. + 3; its component i feeds MFMA step (j, i), whose k slot q is channel 16 j + 4 q + i.
            const int wv = threadIdx.x >> 6, m = lane & 15, q = lane >> 4;
            const int ix = min(tx * kTile + m, V.W - 1), iy = min(ty * kTile + wv, V.H - 1); // edge pixels: never read back
            const float4 *src = reinterpret_cast<const float4 *>(feats + M.pixel(iy, ix)) + q;
            const int nb = M.enc_k >> 4;
            constexpr int kPre = 4; // float4 per lane in flight: 4 KB per wave, 128 KB per CU at two workgroups (64-VGPR budget)
            float4 a[kPre];
#pragma unroll
            for (int u = 0; u < kPre; ++u)
                a[u] = src[4 * min(u, nb - 1)];
            f32x4_t acc4 = {0.f, 0.f, 0.f, 0.f};
            for (int j0 = 0; j0 < nb; j0 += kPre) {
#pragma unroll
                for (int u = 0; u < kPre; ++u) {
                    const int j = j0 + u;
                    if (j >= nb)
                        break;
                    const float4 av = a[u];
                    if (j + kPre < nb)
                        a[u] = src[4 * (j + kPre)];
                    const float *b = s_enc + j * 256 + lane;
                    acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b[0], acc4, 0, 0, 0);
                    acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b[64], acc4, 0, 0, 0);
                    acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b[128], acc4, 0, 0, 0);
                    acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b[192], acc4, 0, 0, 0);
                }
            }
            // C layout: lane holds output channel n = lane % 16 of pixels 4 * (lane / 16) + r of the wave's row
            if (m < pitch) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    lds[(wv * kTile + 4 * q + r) * pitch + m] = acc4[r];
            }
        } else if constexpr (SMALL) { // 256 px x pitch floats, element-wise (any strides, zero past D or past the image)
            const int total = kTilePix * pitch;
            for (int idx = threadIdx.x; idx < total; idx += kThreads) {
                const int p = idx / pitch, c = idx - p * pitch;
                const int ix = tx * kTile + (p & 15), iy = ty * kTile + (p >> 4);
                float val = 0.f;
                if (ix < V.W && iy < V.H && c < D)
                    val = M.sample(feats + (int64_t)c * fs_c, iy, ix);
                lds[idx] = val;
            }
        } else if constexpr (VEC == 2) {
            // bilinear low-resolution map, channel-contiguous: every float4 of the slab is the blend of four float4
            // (L2/MALL-resident texels); one unit per round: four 16-B loads in flight per thread, 16 registers
            constexpr int vpr = kChunk >> 2;
            constexpr int kIt = kTilePix * vpr / kThreads; // 8
#pragma unroll 1
            for (int it = 0; it < kIt; ++it) {
                const int idx = it * kThreads + threadIdx.x;
                const int p = idx / vpr, v = idx - p * vpr;
                const int ix = min(tx * kTile + (p & 15), V.W - 1), iy = min(ty * kTile + (p >> 4), V.H - 1);
                const int y0 = M.ymap[iy], x0 = M.xmap[ix];
                const int y1 = min(y0 + 1, M.lr_h - 1), x1 = min(x0 + 1, M.lr_w - 1);
                const float h1 = M.ly[iy], w1 = M.lx[ix], h0 = 1.0f - h1, w0 = 1.0f - w1;
                const float *b0 = feats + c0 + 4 * v;
                const float4 qa = *reinterpret_cast<const float4 *>(b0 + y0 * M.fs_y + x0 * M.fs_x);
                const float4 qb = *reinterpret_cast<const float4 *>(b0 + y0 * M.fs_y + x1 * M.fs_x);
                const float4 qc = *reinterpret_cast<const float4 *>(b0 + y1 * M.fs_y + x0 * M.fs_x);
                const float4 qd = *reinterpret_cast<const float4 *>(b0 + y1 * M.fs_y + x1 * M.fs_x);
                float4 r;
                r.x = h0 * (w0 * qa.x + w1 * qb.x) + h1 * (w0 * qc.x + w1 * qd.x);
                r.y = h0 * (w0 * qa.y + w1 * qb.y) + h1 * (w0 * qc.y + w1 * qd.y);
                r.z = h0 * (w0 * qa.z + w1 * qb.z) + h1 * (w0 * qc.z + w1 * qd.z);
                r.w = h0 * (w0 * qa.w + w1 * qb.w) + h1 * (w0 * qc.w + w1 * qd.w);
                *reinterpret_cast<float4 *>(lds + p * kChunk + 4 * v) = r;
            }
        } else if constexpr (VEC == 0) {
            // any strides (e.g. the channel-major [D,H,W] map that permute(1,2,0) of backproject.py:249 hands over):
            // lane = 8 pixels of a tile row x 8 channels -> 32-B runs of a channel plane on the load side, a 4-way
            // bank conflict (2x a ds_write_b32) on the LDS side; 32 dwords per thread, eight in flight
            constexpr int kIt = kSlabFloats / kThreads; // 32
            const int pl = threadIdx.x & 7, cl = (threadIdx.x >> 3) & 7, rest = threadIdx.x >> 6; // 16 waves
#pragma unroll 1
            for (int it0 = 0; it0 < kIt; it0 += 8) {
                float vals[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int u = (it0 + j) * 16 + rest; // 512 units of (8 px, 8 ch): 32 pixel groups x 16 channel groups
                    const int p = (u & 31) * 8 + pl, c = (u >> 5) * 8 + cl;
                    const int ix = tx * kTile + (p & 15), iy = ty * kTile + (p >> 4);
                    const int cx_ = min(ix, V.W - 1), cy_ = min(iy, V.H - 1);
                    vals[j] = M.sample(feats + (int64_t)(c0 + c) * fs_c, cy_, cx_);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int u = (it0 + j) * 16 + rest;
                    const int p = (u & 31) * 8 + pl, c = (u >> 5) * 8 + cl;
                    lds[p * kChunk + c] = vals[j];
                }
            }
        } else { // vec_ok == 1: stage the 256 px x 128 ch slab, 32 float4 per pixel row
            constexpr int vpr = kChunk >> 2;
            constexpr int kIt = kTilePix * vpr / kThreads; // 8
            float4 vals[kIt];
            int64_t offs[kIt]; // pixel offsets first (index maps make them loads; keep them out of the feature loads)
#pragma unroll
            for (int it = 0; it < kIt; ++it) {
                const int idx = it * kThreads + threadIdx.x;
                const int p = idx / vpr;
                const int ix = tx * kTile + (p & 15), iy = ty * kTile + (p >> 4);
                // pixels past the image edge are never referenced by an entry: load a clamped (valid) address
                // instead of branching, so the eight loads of a thread are all in flight before the first LDS write
                offs[it] = M.pixel(min(iy, V.H - 1), min(ix, V.W - 1));
            }
#pragma unroll
            for (int it = 0; it < kIt; ++it) {
                const int idx = it * kThreads + threadIdx.x;
                const int p = idx / vpr, v = idx - p * vpr;
                vals[it] = *reinterpret_cast<const float4 *>(feats + offs[it] + c0 + 4 * v);
            }
#pragma unroll
            for (int it = 0; it < kIt; ++it) {
                const int idx = it * kThreads + threadIdx.x;
                const int p = idx / vpr, v = idx - p * vpr;
                *reinterpret_cast<float4 *>(lds + p * kChunk + 4 * v) = vals[it];
            }
        }
    }
    if (threadIdx.x == 0)
        s_item[(k + 1u) & 1u] = nxt;
    __syncthreads();

    const Header *hbase = headers + tile_offsets[tile];
    // byte offset of this lane's channel (pair) inside a pixel row; idle lanes of the small path re-read the last channel
    // small path: Dp = D rounded up to 4 / 8 / 16 / 32 / 64 channels per pair slot, lane = (slot, channel)
    // (measured: D = 16 scatter 1.40 -> 1.24 ms at C5; two pairs per step at D = 32 were SLOWER than one -- the step is
    // bound by its three LDS operations -- so 17..64 channels keep one pair per instruction)
    const int lgd = !SMALL ? 6 : (D <= 4 ? 2 : D <= 8 ? 3 : D <= 16 ? 4 : 6);
    const int chan = lane & ((1 << lgd) - 1);
    const u32 lane_base = SMALL ? (u32)(min(chan, pitch - 1) * sizeof(float)) : (u32)(2 * lane * sizeof(float));
    const u32 row_bytes = (u32)pitch * (u32)sizeof(float);
    // a "pixel" whose slab row lies at or beyond the largest LDS allocation the hardware has, whatever this launch's pitch
    // and whatever sits behind the slab (work counter, encoder): zero-weight padding lanes read it (0 x inf at a real pixel is NaN)
    const u32 kNoPix = (kLdsMax + row_bytes - 1u) / row_bytes;
    const char *slab = reinterpret_cast<const char *>(lds);
    const bool want_d = (chunk == 0) && (dsum_out != nullptr);

    auto claim = [&]() __attribute__((always_inline)) -> u32 {
        // one lane, one LDS atomic, as asm: hipcc's atomic optimiser otherwise wraps the already single-lane add in its
        // wave-aggregation sequence (~8 more instructions per record); the counter sits right behind the slab and the
        // kernel's dynamic LDS starts at address 0
        u32 h = 0;
        if (lane == 0) {
            const u32 addr = (u32)(kTilePix * pitch * (int)sizeof(float)), one = 1u;
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(h) : "v"(addr), "v"(one) : "memory");
        }
        return uniform(h);
    };
    auto load_rec = [&](u32 h) __attribute__((always_inline)) -> Rec { // scalar loads; an invalid claim re-reads the last header (never processed)
        const Header *hp = hbase + min(h, nh - 1);
        Rec r;
        r.gid = uniform(hp->gid);
        r.woff = uniform(hp->woff[0]);
        const u32 c = uniform(hp->counts);
        // the record's entries as ONE run: all four quarters, plus the padding a store blended for the 256-channel kernel has
        // between its halves (Header, gwbp_dev.h) -- those entries carry pix = kPadPix and are dropped below
        r.T = uniform(hp->woff[3]) + (c >> 24) - r.woff;
        return r;
    };
    // slot s of the record's entry stream -> index into the weight pool
    auto wslot = [&](const Rec &R, u32 s) __attribute__((always_inline)) -> u32 { return R.woff + s; };
    auto prefetch = [&](const Rec &R, EV (&e)[2]) __attribute__((always_inline)) {
        // exactly 2 VMEM loads (slots 0..127, clamped to the last one); one coalesced 512-B read per 64 entries
        const u32 last = R.T ? R.T - 1 : 0u;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            issue_e(e[j], wpool + wslot(R, min((u32)(64 * j + lane), last)));
    };

    float2 acc = make_float2(0.f, 0.f);
    // n in 1..64 entries held by lanes 0..n-1 of ev (lanes >= n: w = 0, pix = any valid pixel)
    auto run_vec = [&](const EV &ev, u32 n) __attribute__((always_inline)) {
        if constexpr (SMALL) {
            if (lgd < 6) { // wave-uniform: several pairs per instruction
                switch (lgd) {
                case 2: acc.x = packed_small_vec<2>(slab, row_bytes, lane_base, lane, ev.w, ev.pix, n, acc.x); break;
                case 3: acc.x = packed_small_vec<3>(slab, row_bytes, lane_base, lane, ev.w, ev.pix, n, acc.x); break;
                default: acc.x = packed_small_vec<4>(slab, row_bytes, lane_base, lane, ev.w, ev.pix, n, acc.x); break;
                }
                return;
            }
        }
        // Batches of kB entries, two in flight: the next batch's LDS reads are issued before the current batch's FMAs.
        // Lane selects are compile-time constants; exits are wave-uniform.  Lists are padded to kB entries with {0, 0}.
        constexpr int kB = kListPad;
        float2 fa[kB], fb[kB];
#define GWBP_ISSUE8(B, f)                                                                                             \
    _Pragma("unroll") for (int j = 0; j < kB; ++j)                                                                    \
    {                                                                                                                 \
        const u32 px_ = readlane_u(ev.pix, kB * (B) + j);                                                             \
        if constexpr (SMALL)                                                                                          \
            f[j].x = *reinterpret_cast<const float *>(slab + (px_ * row_bytes + lane_base));                          \
        else                                                                                                          \
            f[j] = *reinterpret_cast<const float2 *>(slab + ((px_ << 9) + lane_base));                                \
    }
#define GWBP_FMA8(B, f)                                                                                               \
    _Pragma("unroll") for (int j = 0; j < kB; ++j)                                                                    \
    {                                                                                                                 \
        const float w = readlane_f(ev.w, kB * (B) + j);                                                               \
        acc.x = __builtin_fmaf(w, f[j].x, acc.x);                                                                     \
        if constexpr (!SMALL)                                                                                         \
            acc.y = __builtin_fmaf(w, f[j].y, acc.y);                                                                 \
    }
        constexpr int kNB = 64 / kB;
        GWBP_ISSUE8(0, fa)
#pragma unroll
        for (int B = 0; B < kNB; B += 2) {
            const bool m1 = (u32)kB * (B + 1) < n;
            if (m1) {
                GWBP_ISSUE8(B + 1, fb)
            }
            GWBP_FMA8(B, fa)
            if (!m1)
                break;
            const bool m2 = (u32)kB * (B + 2) < n;
            if (m2 && B + 2 < kNB) {
                GWBP_ISSUE8((B + 2) & (kNB - 1), fa)
            }
            GWBP_FMA8(B + 1, fb)
            if (!m2)
                break;
        }
#undef GWBP_ISSUE8
#undef GWBP_FMA8
    };
    auto process = [&](const Rec &R, const EV (&e)[2]) __attribute__((always_inline)) { // exactly 2 (+1 if want_d) VMEM atomics, always
        acc = make_float2(0.f, 0.f);
        float wacc = 0.f;
        if (!(dbg & 2)) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if ((u32)(64 * j) >= R.T)
                    break;
                const u32 n = min(64u, R.T - 64u * j);
                EV ev;
                const bool real = (u32)lane < n && e[j].pix < (u32)kTilePix; // (not a padding entry between the halves)
                ev.w = real ? e[j].w : 0.f; // clamped loads: zero the lanes past the list ...
                // ... and point them past the slab (an out-of-range LDS read returns 0): w = 0 times the record's last pixel
                // would turn an inf feature there into NaN (0 x inf)
                ev.pix = real ? e[j].pix : kNoPix;
                wacc += ev.w;
                run_vec(ev, n);
            }
            for (u32 j = 2; 64 * j < R.T; ++j) { // rare: more than 128 entries in one (Gaussian, tile) record
                const u32 n = min(64u, R.T - 64u * j);
                const WPair wp = wpool[wslot(R, min(64 * j + lane, R.T - 1))];
                EV ev;
                const bool real = (u32)lane < n && wp.pix < (u32)kTilePix;
                ev.w = real ? wp.w : 0.f;
                ev.pix = real ? wp.pix : kNoPix;
                wacc += ev.w;
                run_vec(ev, n);
            }
        }
        float *Fg = F + (int64_t)R.gid * D + c0;
        if constexpr (SMALL) { // lane l = channel l: one atomic instruction (always issued, lanes >= D masked off)
            for (int o = 1 << lgd; o < 64; o <<= 1) // packed pairs: add the slots' partial sums (wave-uniform trip count)
                acc.x += __shfl_xor(acc.x, o, 64);
            if (lane < D) {
                if (!(dbg & 1))
                    atomicAdd(Fg + lane, acc.x * scale_f);
                else
                    __builtin_nontemporal_store(acc.x * scale_f, Fg + lane);
            }
        } else {
            const float a0 = acc.x * scale_f, a1 = acc.y * scale_f;
            { // channels c0 + [0, 64)
                const int src = lane >> 1;
                const float a = __shfl(a0, src, 64), bb = __shfl(a1, src, 64);
                if (!(dbg & 1))
                    atomicAdd(Fg + lane, (lane & 1) ? bb : a);
                else
                    __builtin_nontemporal_store((lane & 1) ? bb : a, Fg + lane); // ablation: same VMEM count
            }
            { // channels c0 + [64, 128)
                const int src = 32 + (lane >> 1);
                const float a = __shfl(a0, src, 64), bb = __shfl(a1, src, 64);
                if (!(dbg & 1))
                    atomicAdd(Fg + 64 + lane, (lane & 1) ? bb : a);
                else
                    __builtin_nontemporal_store((lane & 1) ? bb : a, Fg + 64 + lane);
            }
        }
        if (want_d) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
                wacc += __shfl_xor(wacc, o, 64);
            if (lane == 0)
                atomicAdd(dsum_out + R.gid, wacc * scale_d);
        }
    };

    EV eA[2] = {{0.f, 0u}, {0.f, 0u}}, eB[2] = {{0.f, 0u}, {0.f, 0u}};
    u32 h = claim();
    if (h < nh) {
        Rec Rcur = load_rec(h);
        prefetch(Rcur, eA);
        h = claim();
        bool vnxt = h < nh;
        Rec Rnxt = load_rec(h);

        // peeled first phase: only loads(1) are guaranteed younger than loads(0) (atomics of a previous item are older)
        prefetch(Rnxt, eB);
        h = claim();
        bool vnn = h < nh;
        Rec Rnn = load_rec(h);
        wait_e<2>(eA);
        process(Rcur, eA);
        while (vnxt) {
            // odd phase: current record's entries in eB; next record loads into eA
            Rcur = Rnxt, Rnxt = Rnn, vnxt = vnn;
            prefetch(Rnxt, eA);
            h = claim();
            vnn = h < nh;
            Rnn = load_rec(h);
            wait_e<SMALL ? 3 : 4>(eB);
            process(Rcur, eB);
            if (!vnxt)
                break;
            // even phase: current in eA; next into eB
            Rcur = Rnxt, Rnxt = Rnn, vnxt = vnn;
            prefetch(Rnxt, eB);
            h = claim();
            vnn = h < nh;
            Rnn = load_rec(h);
            wait_e<SMALL ? 3 : 4>(eA);
            process(Rcur, eA);
        }
        // The last prefetch (a clamped re-read for a record that does not exist) is still in flight and will write
        // eA/eB's physical registers when it lands; hipcc considers those registers dead here and would reuse them for
        // the next item's address arithmetic.  Drain before leaving the record loop (the slab staging of the next
        // item needs vmcnt(0) anyway).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads(); // every wave is done with this slab and work counter
    } // item loop
    // A launch consumes its queue.  The last workgroup of the class to leave re-arms it, so the same weight store can be
    // scattered again (second feature map, drop-in backward after a forward) without a host-side memset between the
    // launches (a hipMemsetAsync here cost 0.35 ms per view in the two-stream pipeline).  Word 1 of the queue's line
    // counts the leavers; every leaver made its last claim before it counts itself.
    if (threadIdx.x == 0) {
        const u32 left = atomicAdd(queue + 1, 1u);
        if (left == gridDim.x / 8u - 1u) {
            atomicExch(queue + 1, 0u);
            atomicExch(queue, 0u);
        }
    }
}

} // namespace

int launch_scatter_full(const Layout &L, const Ws &W, const ViewDev &V, const FeatMap &M, int D, float scale_f,
                        float scale_d, float *F, float *d, hipStream_t s)
{
    const bool small = D <= 64;
    // 16-B vector staging needs channel-contiguous, 16-B aligned pixel rows
    const bool aligned = M.fs_c == 1 && (M.fs_x % 4 == 0) && (M.fs_y % 4 == 0) &&
                         ((reinterpret_cast<uintptr_t>(M.p) & 15) == 0);
    const int vec_ok = aligned ? (M.bilinear() ? 2 : 1) : 0; // 1: 16-B staging, 2: 16-B bilinear staging, 0: element-wise
    const int n_chunks = small ? 1 : D / kChunk;
    const int pitch = small ? ((D + 3) & ~3) : kChunk;
    const bool enc = M.enc != nullptr;
    if (enc && (!small || D > kEncN || M.enc_k < 16 || (M.enc_k & 15) || M.enc_k > 1024 || !aligned || M.bilinear() || M.ymap))
        return set_error(GWBP_EINVAL, "fused encoder needs D <= 16, K %% 16 == 0, K <= 1024 and a plain channel-contiguous map");
    // slab + work counter + two item slots (+ the encoder table for the fused-encoder staging)
    const size_t lds_bytes = (size_t)kTilePix * pitch * sizeof(float) + 16 + (enc ? (size_t)M.enc_k * kEncN * sizeof(float) : 0);
    const void *fns[4] = {reinterpret_cast<const void *>(k_scatter_full<false, 0>),
                          reinterpret_cast<const void *>(k_scatter_full<false, 1>),
                          reinterpret_cast<const void *>(k_scatter_full<false, 2>),
                          reinterpret_cast<const void *>(k_scatter_full<true, 0>)};
    for (int i = 0; i < 4; ++i) {
        const int rc = ensure_dynamic_lds(fns[i], i < 3 ? (int)kLdsBytes : 65536 + 16, 1 + i);
        if (rc)
            return rc;
    }
    if (enc) {
        const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(k_scatter_full<true, 3>), 16384 + 16 + 1024 * kEncN * 4, 8);
        if (rc)
            return rc;
    }
    int n_cu = 0;
    {
        const int rc = device_cus(&n_cu);
        if (rc)
            return rc;
    }
    // persistent workgroups: one per CU (two for the small-D path whose slab is <= 64 KB); caps.scatter_workgroups
    // overrides; always a multiple of the 8 XCD classes
    int grid = L.scatter_wgs > 0 ? L.scatter_wgs : (small ? 2 * n_cu : n_cu);
    grid = (grid + 7) & ~7;
    u32 *queues = W.shards + kShards * 16;
    const int dbg = profile_knob("GWBP_ABLATE");
#define GWBP_LAUNCH(S, Vc)                                                                                            \
    hipLaunchKernelGGL((k_scatter_full<S, Vc>), dim3(grid), dim3(kThreads), lds_bytes, s, V, n_chunks, W.tile_offsets,  \
                       W.hdr_count, W.headers, W.wpool, M, pitch, D, scale_f, scale_d, F, d, queues, dbg)
    if (enc)
        GWBP_LAUNCH(true, 3);
    else if (small)
        GWBP_LAUNCH(true, 0);
    else if (vec_ok == 1)
        GWBP_LAUNCH(false, 1);
    else if (vec_ok == 2)
        GWBP_LAUNCH(false, 2);
    else
        GWBP_LAUNCH(false, 0);
#undef GWBP_LAUNCH
    return check_hip(hipGetLastError(), "scatter_full launch");
}

} // namespace gwbp
