// token.hip -- the scatter of a NEAREST-UPSAMPLED low-resolution map in TOKEN space (dino variant, backproject.py:242-289).
//
// The reference upsamples the network's 64 x 64 x 1024 patch-token map to the view's 1600 x 1060 pixels with
// F.interpolate(mode="nearest") (backproject.py:244-248) and back-projects the 6.9 GB result.  Every pixel of a token carries the
// same vector, so
//     F_v[g, :] = sum_p w_g(p) feats[p, :] = sum_t omega_{g,t} tok[t, :],     omega_{g,t} = sum_{p in t} w_g(p),
// and when a token is at least a tile wide and high a 16 x 16 tile sees at most 2 x 2 of them: k_blend<kToken> (blend.hip) reduces
// every contributing (Gaussian, tile) record to FOUR weight sums and files them at the record's EMIT position.  k_emit wrote the
// intersections of one Gaussian contiguously (row-major over its tile rectangle, Gaussians in depth order), so here one wave
// walks a Gaussian's sums back to back, multiplies them with token rows and adds the result to F[g, :] with ONE plain
// read-modify-write per view -- the formulation "whose partial sums are combined on chip across tiles" that the full-resolution
// path cannot have, available because the D-wide operand is a 16 MB table instead of a 6.9 GB map:
// no atomics, no weight store, no slabs, no carry rows, no sort by Gaussian.
//
// HBM traffic per view: 8 B x D per Gaussian that receives weight (C2 geometry, D = 1024: 0.53 M rows -> 4.4 GB) + the 16-B sums
// (0.06 GB); the token rows (about five 4-KB rows per touched Gaussian) come from LDS, the few outside the window from L2.
//
// Work map (k_token_apply<NC, true, FULL>, the product): a workgroup of four waves takes 256 consecutive entries of the
// (tile, depth)-sorted intersection list and works on the Gaussians whose HOME tile -- first tile of their rectangle, emit slot 0 --
// is the entry's tile: every Gaussian exactly once, neighbours on the screen back to back.  The 3 x 3 token rows under the first
// entry's tile are staged in LDS (per pass over the channels: 9 x 256 NC floats); a Gaussian's rectangle of up to 2 x 2 tiles reaches at most three token columns
// and rows, so almost every token read is an LDS read.  A wave first finds, lane-parallel, which of its Gaussians carry weight at
// all, then walks those with two register sets: the NEXT Gaussian's F row, first 16 sums and d are requested (unconditionally, so
// that the waits are counted ones) before the current one is multiplied out and stored -- a row's read, its write and the next
// row's read are all in flight together.
//
// Measured per C2-geometry view at D = 1024, alone / in the pipeline, same box each step (profiles/r6_token_*.txt, DESIGN.md
// section 9): first version -- one (Gaussian, 256-channel chunk) per wave iteration, chunk tied to the XCD -- 2.17 ms; one wave
// per Gaussian over all channels in depth order 1.27 / 1.85; tile order + LDS window 1.07 / 1.73; + the request one Gaussian
// ahead 0.87 / 1.42 (4.45 GB of 4-KB rows read and written at 5.1 TB/s).  Depth order (k_token_apply<NC, false, FULL>) stays as
// the -DGWBP_TOKEN_DEPTH_ORDER A/B build.  Channels: 256-channel chunks, up to four side by side per pass, as few passes as that
// allows (D = 1024: 1 x 4; 1536: 2 x 3; 384: 1 x 2 with half a chunk masked off) -- any D % 4 == 0.  Tried and dropped: XCD-local channel groups (1.32-1.66 alone), tile order without the window (no gain:
// the L2 gathers did not bind, the per-Gaussian latency chain did), a 2 x 2 register window, channel-group waves.
#include "gwbp_dev.h"

namespace gwbp {

constexpr int kTokCh = 256;       // channels per wave pass: one float4 per lane
constexpr int kTokPerWave = 16;   // Gaussians (consecutive in depth order) per wave
constexpr int kTokWaves = 4;
#ifndef GWBP_TOK_FLIGHT
#define GWBP_TOK_FLIGHT 1
#endif
// entries (token rows x NC chunks) a wave has in flight per inner-loop iteration.  1: 104 registers at NC = 4, i.e. four waves per
// SIMD leave room for a front-stage wave beside them (2: 118 registers; alone the same 0.87 ms, the DINO64 step 1.3 % slower)
constexpr int kTokFlight = GWBP_TOK_FLIGHT;
constexpr int kTokGroup = kTokPerWave * kTokWaves; // Gaussians per workgroup
constexpr int kTokMaxTiles = 256;                  // tile columns / rows of the largest view the token path takes (4096 px): 2 KB of
                                                   // LDS tables, so that four workgroups with their 36 KB token windows share a CU

typedef float f4 __attribute__((ext_vector_type(4)));


__global__ __launch_bounds__(256) void k_zero_omega(float4 *__restrict__ omega, const Counters *__restrict__ ctr, int prio)
{
    front_priority(prio);
    const u32 n = ctr->n_isect;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
        omega[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

struct TokenApplyArgs {
    int64_t N;
    const u32 *order;   // Gaussians in depth order (the emit order)
    const u32 *touched; // emit slots per Gaussian (0 = culled)
    const u32 *estart;  // first emit slot
    const uint2 *rect;  // tile rectangle of the emit
    const float *omega; // [n_isect][4]
    const int32_t *ymap, *xmap;
    const float *tokens; // tokens[row * ts_y + col * ts_x + c]
    int64_t ts_y, ts_x;
    const u32 *sorted_tiles, *sorted_gids; // the intersections in (tile, depth) order: tile id, Gaussian
    int D, W, H;
    int n_pass; // passes over the channels, 256 NC each
    float scale_f, scale_d;
    float *F, *d;
    Counters *ctr;
};

// One wave = one Gaussian at a time, all channels of a PASS: NC chunks of 256 channels side by side (NC x float4 per lane; D = 1024:
// NC = 4, one pass; D = 1536: two passes of NC = 3), so the Gaussian's weight sums are read and decoded once per pass and every
// token row read / F row read-modify-write of the Gaussian is in flight together.  The operands of the NEXT Gaussian are requested
// before this one is worked on.  FULL: every (pass, chunk, lane) maps to a channel (D a multiple of 256 NC); otherwise the lanes
// beyond D are predicated off (any D % 4 == 0: the 384 / 768 / 1536 channels of the other DINOv2 backbones).
template <int NC, bool TILE_ORDER, bool FULL>
__global__ __launch_bounds__(64 * kTokWaves) void k_token_apply(TokenApplyArgs A)
{
    if (A.ctr->blend_kind != kBlendToken) { // the view in this workspace was not blended by gwbp_blend_tokens
        if (blockIdx.x == 0 && threadIdx.x == 0)
            atomicOr(&A.ctr->overflow, kOverflowMismatch);
        return;
    }
    // An intersection-capacity overflow leaves NO emit positions behind (k_emit returns at once, estart[] holds whatever the depth
    // sort left there): the view is invalid anyway (the host grows the workspace and runs it again) and nothing may be read
    // through estart.
    if (A.ctr->overflow & 1u)
        return;
    const u32 n_isect = A.ctr->n_isect;
    if (TILE_ORDER && blockIdx.x * (u32)(64 * kTokWaves) >= n_isect)
        return; // launched for the capacity (the intersection count lives on the device): blocks beyond the data leave at once
    // (all returns above are workgroup-uniform; from here on every wave reaches every barrier)
    // first token column / row of every tile column / row (the index maps at the tiles' first pixels): a few hundred ints that
    // every entry's token lookup reads -- from LDS, not through a dependent global load in front of the token row reads
    __shared__ int s_tc0[kTokMaxTiles], s_tr0[kTokMaxTiles];
    const int tile_w = (A.W + kTile - 1) / kTile, tile_h = (A.H + kTile - 1) / kTile;
    for (int i = threadIdx.x; i < tile_w; i += 64 * kTokWaves)
        s_tc0[i] = A.xmap[min(i * kTile, A.W - 1)];
    for (int i = threadIdx.x; i < tile_h; i += 64 * kTokWaves)
        s_tr0[i] = A.ymap[min(i * kTile, A.H - 1)];
    __syncthreads();
    const int lane = (int)(threadIdx.x & 63u), wave = (int)uniform(threadIdx.x >> 6);
    // TILE_ORDER: the 3 x 3 token rows below / right of the first token of the workgroup's first tile are staged in LDS (per pass:
    // its 256 NC channels): the home Gaussians of that tile (rectangles of up to 2 x 2 tiles reach at most three token columns
    // and rows) read their token rows from there instead of from L2
    int win_c = 0, win_r = 0;
    if constexpr (TILE_ORDER) {
        const u32 t0 = A.sorted_tiles[blockIdx.x * (u32)(64 * kTokWaves)];
        win_c = s_tc0[min((int)(t0 % (u32)tile_w), tile_w - 1)], win_r = s_tr0[min((int)(t0 / (u32)tile_w), tile_h - 1)];
    }
    // WHICH Gaussians a wave takes.  TILE_ORDER: 64 consecutive entries of the (tile, depth)-sorted intersection list, of which the
    // wave works on the Gaussians whose HOME tile (first tile of the rectangle, emit slot 0) is the entry's tile -- every Gaussian
    // exactly once, consecutive Gaussians read the SAME few token rows (the LDS window).  Otherwise: 16 consecutive Gaussians of
    // the depth order (= the emit order; screen positions at random, token rows gathered from all over the map through L2).
    u32 m_gid = 0, m_cnt = 0, m_es = 0, m_rx = 0, m_ry = 0;
    if constexpr (TILE_ORDER) {
        const u32 i0 = (blockIdx.x * (u32)kTokWaves + (u32)wave) * 64u;
        if (i0 + (u32)lane < n_isect) {
            const u32 gid = A.sorted_gids[i0 + lane], tile = A.sorted_tiles[i0 + lane];
            const uint2 rc = A.rect[gid];
            if (tile == (rc.y & 0xFFFFu) * (u32)tile_w + (rc.x & 0xFFFFu)) {
                m_gid = gid, m_cnt = A.touched[gid], m_es = A.estart[gid];
                m_rx = rc.x, m_ry = rc.y;
            }
        }
    } else {
        const int64_t i0 = ((int64_t)blockIdx.x * kTokWaves + wave) * kTokPerWave;
        if (lane < kTokPerWave && i0 + lane < A.N) {
            m_gid = A.order[i0 + lane];
            m_cnt = A.touched[m_gid];
            if (m_cnt) {
                m_es = A.estart[m_gid];
                const uint2 rc = A.rect[m_gid];
                m_rx = rc.x, m_ry = rc.y;
            }
        }
    }
    const int quad = lane & 3, sl = lane >> 2; // lane = (slot within a batch of 16, token quadrant qx | qy << 1)
    // Which of the wave's Gaussians have weight at all, found lane-parallel: lane j looks through the (up to 16) first sums of ITS
    // Gaussian, 16 independent loads in flight.  The walk below then only visits Gaussians whose row it will write, which is what
    // lets it request the NEXT Gaussian's row and sums unconditionally, a Gaussian ahead.
    u64 mine = __ballot(m_cnt != 0u);
    if (mine != 0ull) {
        const u32 last = m_cnt ? min(m_cnt, 16u) - 1u : 0u;
        const float4 *om4 = reinterpret_cast<const float4 *>(A.omega) + m_es;
        float4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            v[i] = om4[min((u32)i, last)];
        u32 bits = 0u; // (OR of the bit patterns without the sign: != 0 exactly when some sum is not +-0; no short circuit, which
                       // would put a wait and a branch behind every load)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            bits |= (__float_as_uint(v[i].x) | __float_as_uint(v[i].y) | __float_as_uint(v[i].z) | __float_as_uint(v[i].w)) &
                    0x7FFFFFFFu;
        mine = __ballot(m_cnt != 0u && (m_cnt > 16u || bits != 0u));
    }
    constexpr int kCW = kTokCh * NC; // channels per pass
    const float *dsrc = A.d ? A.d : A.omega;
    for (int pass = 0; pass < A.n_pass; ++pass) {
        const int cbase = pass * kCW; // first channel of the pass
        if constexpr (TILE_ORDER) {
            extern __shared__ __attribute__((aligned(16))) float s_win[];
            if (pass > 0)
                __syncthreads(); // every wave is done with the previous pass's window
            const int tc_max = A.xmap[A.W - 1], tr_max = A.ymap[A.H - 1];
            constexpr int per_row = kCW / 4; // float4 per window row
            for (int i = threadIdx.x; i < 9 * per_row; i += 64 * kTokWaves) {
                const int w9 = i / per_row, c4 = i - w9 * per_row;
                if (FULL || cbase + c4 * 4 < A.D) {
                    const long long o = (long long)min(win_r + w9 / 3, tr_max) * A.ts_y + (long long)min(win_c + w9 % 3, tc_max) * A.ts_x;
                    reinterpret_cast<f4 *>(s_win)[i] = *reinterpret_cast<const f4 *>(A.tokens + o + cbase + c4 * 4);
                }
            }
            __syncthreads();
        }
        u64 rest = mine;
        if (rest == 0ull)
            continue; // (wave-uniform; the wave still meets the others at the barriers)
        bool valid[NC]; // does this lane's float4 of chunk c hold channels?  (the last chunks of a D that is no multiple of 256 NC)
#pragma unroll
        for (int c = 0; c < NC; ++c)
            valid[c] = FULL || cbase + c * kTokCh + lane * 4 < A.D;
        // the token rows of one Gaussian's weight sums, times the sums, into acc.  Returns whether any sum is non-zero
        auto accumulate = [&](int k, float om_first, f4 (&acc)[NC], float &dsum) -> bool {
            const u32 cnt = (u32)__builtin_amdgcn_readlane((int)m_cnt, k), es = (u32)__builtin_amdgcn_readlane((int)m_es, k);
            const u32 rx = (u32)__builtin_amdgcn_readlane((int)m_rx, k), ry = (u32)__builtin_amdgcn_readlane((int)m_ry, k);
            const u32 x0 = rx & 0xFFFFu, rw = (rx >> 16) - x0, y0 = ry & 0xFFFFu;
            const float *tbase = A.tokens + (size_t)cbase + (size_t)lane * 4;
            // one batch of 16 emit slots x 4 quadrants, one sum per lane
            auto batch = [&](float om, u32 slot, u64 nz) {
                dsum += om;
                // the token under this lane's (tile, quadrant): first token of the tile + (qx, qy); only dereferenced where
                // om != 0, i.e. where the blend found a pixel of that token
                const u32 ty = y0 + slot / rw, tx = x0 + slot % rw;
                const int tc = s_tc0[min(tx, (u32)tile_w - 1u)] + (quad & 1);
                const int tr = s_tr0[min(ty, (u32)tile_h - 1u)] + (quad >> 1);
                u64 miss = nz;
                if constexpr (TILE_ORDER) {
                    // the entries whose token lies in the workgroup's LDS window first, in a loop of their own without a global
                    // load (one loop with both sources has to wait for EVERYTHING in flight at the join, the next Gaussian's row
                    // included)
                    extern __shared__ __attribute__((aligned(16))) float s_win[];
                    const int er = tr - win_r, ec = tc - win_c;
                    const bool inside = (u32)er < 3u && (u32)ec < 3u;
                    u64 hits = nz & __ballot(inside);
                    miss = nz & ~hits;
                    const int woff = (er * 3 + ec) * kCW; // (floats; only read from lanes inside)
                    while (hits != 0ull) {                // kTokFlight entries' rows in flight
                        f4 t[kTokFlight][NC];
                        float w[kTokFlight];
#pragma unroll
                        for (int u = 0; u < kTokFlight; ++u) {
                            w[u] = 0.f;
                            if (hits != 0ull) { // wave-uniform
                                const int l = __ffsll((long long)hits) - 1;
                                hits &= hits - 1;
                                w[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(om), l));
                                const float *wrow = s_win + __builtin_amdgcn_readlane(woff, l) + lane * 4;
#pragma unroll
                                for (int c = 0; c < NC; ++c) // (lanes beyond D read LDS nobody wrote: never stored)
                                    t[u][c] = *reinterpret_cast<const f4 *>(wrow + c * kTokCh);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < kTokFlight; ++u)
                            if (w[u] != 0.f) { // (wave-uniform; a skipped slot must not turn 0 x NaN into NaN)
#pragma unroll
                                for (int c = 0; c < NC; ++c) {
                                    acc[c].x = __builtin_fmaf(w[u], t[u][c].x, acc[c].x);
                                    acc[c].y = __builtin_fmaf(w[u], t[u][c].y, acc[c].y);
                                    acc[c].z = __builtin_fmaf(w[u], t[u][c].z, acc[c].z);
                                    acc[c].w = __builtin_fmaf(w[u], t[u][c].w, acc[c].w);
                                }
                            }
                    }
                }
                if (miss == 0ull)
                    return;
                const long long toff = (long long)tr * A.ts_y + (long long)tc * A.ts_x;
                const int tlo = (int)(u32)toff, thi = (int)(toff >> 32);
                while (miss != 0ull) { // kTokFlight entries' rows (NC loads each) in flight
                    f4 t[kTokFlight][NC];
                    float w[kTokFlight];
#pragma unroll
                    for (int u = 0; u < kTokFlight; ++u) {
                        w[u] = 0.f;
                        if (miss != 0ull) { // wave-uniform
                            const int l = __ffsll((long long)miss) - 1;
                            miss &= miss - 1;
                            w[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(om), l));
                            const long long o = ((long long)__builtin_amdgcn_readlane(thi, l) << 32) |
                                                (long long)(u32)__builtin_amdgcn_readlane(tlo, l);
#pragma unroll
                            for (int c = 0; c < NC; ++c) {
                                t[u][c] = f4{0.f, 0.f, 0.f, 0.f};
                                if (valid[c])
                                    t[u][c] = *reinterpret_cast<const f4 *>(tbase + o + c * kTokCh);
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kTokFlight; ++u)
                        if (w[u] != 0.f) {
#pragma unroll
                            for (int c = 0; c < NC; ++c) {
                                acc[c].x = __builtin_fmaf(w[u], t[u][c].x, acc[c].x);
                                acc[c].y = __builtin_fmaf(w[u], t[u][c].y, acc[c].y);
                                acc[c].z = __builtin_fmaf(w[u], t[u][c].z, acc[c].z);
                                acc[c].w = __builtin_fmaf(w[u], t[u][c].w, acc[c].w);
                            }
                        }
                }
            };
            // the first batch's sums are in registers already; the loop over further batches (rectangles of more than 16 tiles)
            // is a loop of its own, so that the common case has no load of sums in front of its token reads
            bool any = false;
            {
                const u64 nz = __ballot(om_first != 0.f);
                if (nz != 0ull) {
                    any = true;
                    batch(om_first, (u32)sl, nz);
                }
            }
            for (u32 s0 = 16u; s0 < cnt; s0 += 16u) {
                const u32 slot = s0 + (u32)sl;
                const float om = slot < cnt ? A.omega[(size_t)(es + slot) * 4 + quad] : 0.f;
                const u64 nz = __ballot(om != 0.f);
                if (nz == 0ull)
                    continue;
                any = true;
                batch(om, slot, nz);
            }
            return any;
        };
        auto pop = [&]() -> int { // the wave's next Gaussian (a lane index), -1 behind the last
            if (rest == 0ull)
                return -1;
            const int k = __ffsll((long long)rest) - 1;
            rest &= rest - 1;
            return k;
        };
        // (sums, row, d) of a Gaussian: requests without conditions -- the waits in the walk are counted ones (vmcnt retires in
        // order and counts stores; a wait in front of a path-dependent number of younger loads would have to drain them all)
        auto request = [&](int k, float &om, f4 (&fold)[NC], float &dv) {
            const u32 cnt = (u32)__builtin_amdgcn_readlane((int)m_cnt, k), es = (u32)__builtin_amdgcn_readlane((int)m_es, k);
            const u32 gid = (u32)__builtin_amdgcn_readlane((int)m_gid, k);
            om = A.omega[(size_t)(es + min((u32)sl, cnt - 1u)) * 4 + quad];
            const float *row = A.F + (size_t)gid * (size_t)A.D + (size_t)cbase + (size_t)lane * 4;
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (valid[c])
                    fold[c] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(row + c * kTokCh));
            dv = dsrc[A.d ? gid : 0u];
        };
        // one Gaussian: request the next one's operands into the OTHER register set, work on this one, store; returns the next
        auto step = [&](int kc, float om_c, f4 (&fold_c)[NC], float d_c, float &om_n, f4 (&fold_n)[NC], float &d_n) -> int {
            const int kn = pop();
            request(kn >= 0 ? kn : kc, om_n, fold_n, d_n);
            const u32 gid = (u32)__builtin_amdgcn_readlane((int)m_gid, kc), cnt = (u32)__builtin_amdgcn_readlane((int)m_cnt, kc);
            f4 acc[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c)
                acc[c] = f4{0.f, 0.f, 0.f, 0.f};
            float dsum = 0.f;
            const bool any = accumulate(kc, (u32)sl < cnt ? om_c : 0.f, acc, dsum);
            if (any) { // (a Gaussian of more than 16 slots may still be without weight: nothing is written then)
                float *frow = A.F + (size_t)gid * (size_t)A.D + (size_t)cbase + (size_t)lane * 4;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    f4 r = fold_c[c];
                    r.x = __builtin_fmaf(A.scale_f, acc[c].x, r.x);
                    r.y = __builtin_fmaf(A.scale_f, acc[c].y, r.y);
                    r.z = __builtin_fmaf(A.scale_f, acc[c].z, r.z);
                    r.w = __builtin_fmaf(A.scale_f, acc[c].w, r.w);
                    if (valid[c])
                        __builtin_nontemporal_store(r, reinterpret_cast<f4 *>(frow + c * kTokCh));
                }
                if (pass == 0 && A.d) { // the wave owns d[gid] as well: plain read-modify-write
                    const float tot = wave_sum(dsum);
                    if (lane == 0)
                        A.d[gid] = __builtin_fmaf(A.scale_d, tot, d_c);
                }
            }
            return kn;
        };
        int kc = pop();
        float om_a, om_b, d_a, d_b;
        f4 fold_a[NC], fold_b[NC];
        request(kc, om_a, fold_a, d_a);
        for (;;) { // the two register sets swap roles: no copy (a copy would wait for the request it has just made)
            kc = step(kc, om_a, fold_a, d_a, om_b, fold_b, d_b);
            if (kc < 0)
                break;
            kc = step(kc, om_b, fold_b, d_b, om_a, fold_a, d_a);
            if (kc < 0)
                break;
        }
    }
}

int launch_zero_omega(const Layout &L, const Ws &W, hipStream_t s)
{
    const int prio = (L.flags & GWBP_FLAG_FRONT_PRIORITY) ? 1 : 0;
    hipLaunchKernelGGL(k_zero_omega, dim3(1024), dim3(256), 0, s, reinterpret_cast<float4 *>(W.headers), W.counters, prio);
    return check_hip(hipGetLastError(), "zero_omega launch");
}

int launch_token_apply(const Layout &L, const Ws &W, const ViewDev &V, const float *tokens, int64_t ts_y, int64_t ts_x, int D,
                       const int32_t *ymap, const int32_t *xmap, float scale_f, float scale_d, float *F, float *d, hipStream_t s)
{
    if (D < 4 || D % 4 != 0)
        return set_error(GWBP_EUNSUPPORTED, "gwbp_scatter_tokens: D must be a multiple of 4 (got %d); use gwbp_scatter_upsampled", D);
    if (!tokens || !ymap || !xmap || (L.n > 0 && !F))
        return set_error(GWBP_EINVAL, "gwbp_scatter_tokens: null tokens / index maps / F");
    if (ts_y < 0 || ts_x < D || (ts_y & 3) || (ts_x & 3) || (reinterpret_cast<uintptr_t>(tokens) & 15) ||
        (reinterpret_cast<uintptr_t>(F) & 15))
        return set_error(GWBP_EINVAL, "gwbp_scatter_tokens: token rows must be 16-B aligned runs of D contiguous channels "
                                      "(strides %lld %lld), F 16-B aligned", (long long)ts_y, (long long)ts_x);
    if (V.tile_w > kTokMaxTiles || V.tile_h > kTokMaxTiles)
        return set_error(GWBP_EUNSUPPORTED, "gwbp_scatter_tokens: views of more than %d tile columns / rows are not supported",
                         kTokMaxTiles);
    if (L.n == 0)
        return GWBP_OK;
    TokenApplyArgs A;
    A.N = L.n, A.order = W.dvals[0], A.touched = W.touched, A.estart = W.dkeys[1], A.rect = W.rect;
    A.omega = reinterpret_cast<const float *>(W.headers);
    A.ymap = ymap, A.xmap = xmap, A.tokens = tokens, A.ts_y = ts_y, A.ts_x = ts_x;
    A.D = D, A.W = V.W, A.H = V.H, A.scale_f = scale_f, A.scale_d = scale_d, A.F = F, A.d = d;
    A.ctr = W.counters;
    // 256-channel chunks: as few passes as four chunks side by side allow, the chunks spread evenly over them (D = 1024: one pass
    // of 4; 1536: two passes of 3; 384: one pass of 2 whose second chunk is half empty)
    const int n_chunk = (D + kTokCh - 1) / kTokCh;
    A.n_pass = (n_chunk + 3) / 4;
    const int nc = (n_chunk + A.n_pass - 1) / A.n_pass;
    const bool full = D % kTokCh == 0 && n_chunk == A.n_pass * nc;
    const int fin = sort_passes(V.tile_w * V.tile_h) & 1; // where the tile sort left its result
    A.sorted_tiles = W.keys[fin], A.sorted_gids = W.vals[fin];
    // the tile-order walk with the workgroup's 3 x 3 token window in LDS (9 x 256 NC floats per pass: at most 36 KB, four workgroups
    // per CU); -DGWBP_TOKEN_DEPTH_ORDER builds (same-box A/B) walk the Gaussians in depth order and read every token row from L2
#ifndef GWBP_TOKEN_DEPTH_ORDER
    constexpr bool tile_order = true;
#else
    constexpr bool tile_order = false;
#endif
    const size_t lds = tile_order ? (size_t)9 * nc * kTokCh * sizeof(float) : 0;
    const int64_t blocks = tile_order ? (L.isect_cap + 64 * kTokWaves - 1) / (64 * kTokWaves) : (L.n + kTokGroup - 1) / kTokGroup;
    if (blocks > 0x7FFFFFFFll)
        return set_error(GWBP_EINVAL, "gwbp_scatter_tokens: grid too large");
    const dim3 grid((unsigned)blocks), block(64 * kTokWaves);
#define GWBP_TOK_LAUNCH(NCV)                                                                                                   \
    do {                                                                                                                       \
        if (full)                                                                                                              \
            hipLaunchKernelGGL((k_token_apply<NCV, tile_order, true>), grid, block, lds, s, A);                                \
        else                                                                                                                   \
            hipLaunchKernelGGL((k_token_apply<NCV, tile_order, false>), grid, block, lds, s, A);                               \
    } while (0)
    if (nc == 4)
        GWBP_TOK_LAUNCH(4);
    else if (nc == 3)
        GWBP_TOK_LAUNCH(3);
    else if (nc == 2)
        GWBP_TOK_LAUNCH(2);
    else
        GWBP_TOK_LAUNCH(1);
#undef GWBP_TOK_LAUNCH
    return check_hip(hipGetLastError(), "token_apply launch");
}

} // namespace gwbp
