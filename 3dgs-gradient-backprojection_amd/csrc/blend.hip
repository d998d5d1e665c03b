// blend.hip -- per-tile front-to-back alpha blend producing the SPARSE WEIGHT STORE of one view.
//
// Replaces gsplat 1.4.0's rasterize_to_pixels forward (and makes its backward replay unnecessary): the reference
// runs that blend 2 x (16 fwd + 16 bwd) times per view at D = 512 (backproject.py:115-147 through
// channel_chunk = 32); here it runs ONCE and its only product is w = alpha * T for every contributing
// (Gaussian, pixel) pair:
//
//   workgroup = ONE WAVE = one 16x16 tile; lane l owns the four pixels (l % 16, 4q + l / 16), q = 0..3 ("quarters":
//     tile rows 4q..4q+3), i.e. pixel index q*64 + l.  Four independent T chains per lane give the latency-bound
//     loop 4-way instruction-level parallelism; a single wave needs no workgroup barrier and no header compaction.
//   per batch of 64 list entries: projected Gaussians staged in LDS (2 x 16-B broadcast reads per entry) together
//     with a conservative 4-bit strip mask (which quarters the Gaussian can reach at all)
//   per contributing (Gaussian, tile) RECORD: mask[q] = ballot(contributes in quarter q); the entries {w, pixel} of
//     the four quarters are written back to back (ascending pixel order, lane rank = mbcnt, 8-B stores) into the
//     wave's stream carved from a sharded global pool in pages of kPage entries, padded with {0, kPadPix} to a multiple of 8
//     per record (kHalves: per half-tile list, see below), followed by one 64-B Header {gid, woff[q], counts, mask[q]} in
//     list order.
//
// Store size: 8 B per pair (+ padding) + 64 B per header (C2: ~0.75 GB + ~115 MB per view), written once, then read
// by the scatter kernel once per 128-channel chunk through L2.
#include <stdlib.h>

#include "gwbp_dev.h"

namespace gwbp {

constexpr int kBatch = 64;

// What a contributing record turns into.
//   kStore:  entries {w, pixel} + Header in the weight store (read by k_scatter_full / k_scatter / k_render_*)
//   kHalves: the same plus the record's weight sum in its header (+ d[gid]): what gwbp_scatter's 256-channel path asks for
//            (k_scatter_wide builds its half-tile visit lists itself, from these headers)
//   kFused:  NO store: the record's sums  F[gid, :D] += sum_p w f[p, :],  d[gid] += sum_p w  are formed right here from
//            the tile's pixels held in registers (D <= 16: 4 pixels x 16 channels = 64 VGPRs per lane) and added to F / d
//            with one atomic instruction -- the small-D variants (backproject_compressed.py:127-165: D = 16) then need
//            neither the 0.8 GB store nor a scatter kernel.
//   kFusedEnc: kFused whose 16-channel pixels are COMPUTED by the kernel from a K-channel map and a [K, n <= 16] encoder
//            (backproject_compressed.py:127 inside the tile's prologue): the wave streams its tile's 256 pixels x K channels once
//            through the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, the same k-ordered chain as k_encode_map) and the
//            results land -- after a 4 x 4 exchange between the rows of 16 lanes -- in the registers kFused loads them into.
//            No [H, W, 16] map, no encoder kernel, and the HBM stream of one wave's prologue overlaps with the other waves'
//            blend loops on the same SIMD instead of two kernels competing for wave slots.
//   kToken:  NO store either, for a LOW-RESOLUTION map that the reference upsamples with F.interpolate(mode="nearest") and whose
//            texels ("tokens": the dino variant's 64 x 64 patch tokens, backproject.py:242-249) are at least a tile wide and high:
//            a tile then sees at most 2 x 2 tokens, and F_v[g, :] = sum_p w_g(p) feats[p, :] = sum_t omega_{g,t} tok[t, :] with
//            omega_{g,t} = sum_{p in t} w_g(p).  A contributing record turns into its FOUR token-quadrant weight sums, written
//            as one 16-B line at the record's EMIT position (k_emit wrote the intersections of a Gaussian contiguously, in the
//            order of its tile rectangle; estart[gid] is where they start) -- so the weight sums of one Gaussian lie back to back
//            whatever tiles they came from, and k_token_apply (token.hip) adds them to F with ONE plain read-modify-write of the
//            row per view: no pair entries, no headers, no atomics, no slabs, no carry rows.
//   kFusedPC: kFusedEnc split into PRODUCER and CONSUMER waves inside one persistent launch (round 6; the compressed variant,
//            backproject_compressed.py:127-165).  kFusedEnc runs every wave's HBM-bound encoder prologue (0.58 ms worth at C5) and
//            then its issue-bound blend loop (0.70 ms) back to back, and since all resident waves start together the chip
//            alternates between an HBM phase with idle issue slots and an issue phase with an idle memory system.  Here one
//            workgroup per CU holds kPcProd encoder waves that do nothing but stream tiles through the matrix cores into a ring of
//            encoded tiles in LDS (16 KB each), and WAVES - kPcProd blend waves that drain it: the HBM stream and the blend loops
//            run CONCURRENTLY for the whole launch.  Same encoded pixels, same weights, bit for bit.  Measured at C5: the kernel
//            alone 1.35 -> 1.19 ms, the step 1.34 -> 1.26 ms/view on one box (the north_star's 1.05 would need the blend waves
//            to run as fast with three waves per SIMD as with four; they are latency-bound, see the wave-mix table below).
enum BlendMode { kStore = 0, kHalves = 1, kFused = 2, kFusedEnc = 3, kToken = 4, kFusedPC = 5 };
constexpr int kFusedCh = 16;
constexpr int kEncWaves = 8;    // kFusedEnc: tiles (waves) per workgroup sharing one LDS copy of the encoder
constexpr int kEncMaxK = 512;   // ... whose K x 16 floats take at most 32 KB
// kFusedPC wave mix (tuning knobs; measured at C5 on one box, kernel alone / four views in flight, ms per view --
// profiles/r6_c5_split.txt: 4 + 8: 1.58 / 1.44, 2 + 10: 1.36 / 1.30, 4 + 12: 1.36 / 1.27, 3 + 12: 1.34 / 1.25, 3 + 13: 1.29 / 1.25,
// 2 + 14: 1.19 / 1.26, 1 + 12: 1.62 / 1.75 (starved); one wave per tile, kFusedEnc: 1.35 / 1.34; on the shipped mix, ring 3 / 6,
// non-temporal map loads, a shorter poll interval and 2 + 13 were all within the noise of a second box, ring 6 worse).  The blend waves are bound by
// the LATENCY of their own dependent chains, so their throughput grows with their number, and two encoder waves per CU
// (2 x 16 KB in flight x 256 CUs) keep up with them.
#ifndef GWBP_PC_WAVES
#define GWBP_PC_WAVES 16
#endif
#ifndef GWBP_PC_PROD
#define GWBP_PC_PROD 2
#endif
#ifndef GWBP_PC_RING
#define GWBP_PC_RING 5
#endif
constexpr int kPcWaves = GWBP_PC_WAVES; // kFusedPC: waves per workgroup (ONE workgroup per CU: four 128-register waves per SIMD)
constexpr int kPcProd = GWBP_PC_PROD;   // ... of which the first ones are encoder (producer) waves
constexpr int kPcRing = GWBP_PC_RING;   // ... filling a ring of this many encoded tiles (256 pixels x 16 outputs x 4 B = 16 KB each) in LDS
constexpr int kPcTileFloats = kTilePix * kFusedCh;
// ring slot states: 0 = empty, 1 = claimed (being filled or being read), tile + 2 = holds that tile's encoded pixels
constexpr u32 kPcEmpty = 0u, kPcBusy = 1u;
constexpr u32 kPcStalled = 0xFFu;       // "no slot": a wait on the ring gave up
constexpr u32 kPcMaxSpins = 1u << 22;   // s_sleep(8) polls (~0.25 us each) before a ring wait gives up: about a second

struct FusedArgs { // kFused only
    const float *feats; // feats[y * fs_y + x * fs_x + c], c < D
    int64_t fs_y, fs_x;
    int D;       // <= kFusedCh
    int vec4;    // rows may be read as float4 (D % 4 == 0, 16-B aligned base and strides)
    float scale_f;
    float *F;
    const float *enc; // kFusedEnc: [enc_k, D] row-major; feats then has enc_k channels per pixel
    int enc_k;
    // kToken only
    const int32_t *ymap, *xmap; // nearest-upsampling index maps (view.height / view.width entries): pixel -> token row / column
    float *omega;               // [isect_cap][4]: weight sums per (emit position, token quadrant qx | qy << 1), zeroed per view
    const u32 *estart;          // first emit position of every Gaussian (k_emit)
    const uint2 *rect;          // the tile rectangle the emit walked (k_project)
    // kFusedPC only
    u32 *pc_queue;              // next tile (index into tile_order) a producer wave takes; zero when the kernel starts
};


// Sum of 16 per-lane values over the 64 lanes, transposed: lane l returns the wave total of value c(l),
//   c(l) = bit2(l) | bit3(l) << 1 | bit0(l) << 2 | bit1(l) << 3        (the same in all four rows of 16 lanes).
// Each stage pairs lanes (row_half_mirror, row_ror:8, xor 1, xor 2 -- partners always agree on the earlier stages' sides,
// together they generate all 16 lanes of a row), keeps one half of the registers on each side and adds the partner's
// copy of the kept half: 8 + 4 + 2 + 1 outputs instead of 16 full reductions; two lane swaps add the four rows.
template <int CTRL>
__device__ __forceinline__ float dpp_get(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float rows_sum(float v) // + the other three rows of 16 lanes (lane-wise)
{
#if __HIP_DEVICE_COMPILE__
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
    v = __int_as_float(a[0]) + __int_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
    v = __int_as_float(b[0]) + __int_as_float(b[1]);
#endif
    return v;
}
// Stages 1 and 2 take their side from lane bits 2 and 3, i.e. from the 4-lane "bank" inside the row: the add of the second
// half is written under a DPP bank mask, no select at all (2 instructions per output).  Stages 3 and 4 (lane bits 0 and 1)
// select with v_cndmask under constant lane masks (3 per output).  33 vector instructions for the 16 -> 1 transposed sum;
// written as one block because the DPP read-after-write wait states (2) are placed by hand.  p[] is clobbered.
__device__ __forceinline__ float transposed_sum16(float (&p)[16])
{
#if __HIP_DEVICE_COMPILE__
    const u64 odd = 0xAAAAAAAAAAAAAAAAull, upper2 = 0xCCCCCCCCCCCCCCCCull; // lanes with bit 0 / bit 1 set
    asm("s_nop 1\n\t"
        // stage 1: i <-> 7 - i (row_half_mirror); lanes 0-3, 8-11 keep the even registers, lanes 4-7, 12-15 the odd ones
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %6, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %8, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %10, %10, %10 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %12, %12, %12 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %14, %14, %14 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %2, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %4, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %6, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %8, %9, %9 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %10, %11, %11 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %12, %13, %13 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %14, %15, %15 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        // stage 2: i <-> i ^ 8 (row_ror:8); lanes 0-7 keep registers 0, 4, 8, 12, lanes 8-15 registers 2, 6, 10, 14
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %8, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %12, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %8, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %12, %14, %14 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        // stage 3: i <-> i ^ 1; even lanes keep registers 0, 8, odd lanes 4, 12
        "v_cndmask_b32_e64 %1, %0, %4, %16\n\t"
        "v_cndmask_b32_e64 %2, %4, %0, %16\n\t"
        "v_cndmask_b32_e64 %3, %8, %12, %16\n\t"
        "v_cndmask_b32_e64 %5, %12, %8, %16\n\t"
        "v_add_f32_dpp %0, %2, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %8, %5, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        // stage 4: i <-> i ^ 2; lanes with bit 1 clear keep register 0, the others 8
        "v_cndmask_b32_e64 %1, %0, %8, %17\n\t"
        "v_cndmask_b32_e64 %2, %8, %0, %17\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %2, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "+v"(p[8]),
          "+v"(p[9]), "+v"(p[10]), "+v"(p[11]), "+v"(p[12]), "+v"(p[13]), "+v"(p[14]), "+v"(p[15])
        : "s"(odd), "s"(upper2));
#endif
    return rows_sum(p[0]);
}
__device__ __forceinline__ int transposed_channel(int lane)
{
    return ((lane >> 2) & 1) | (((lane >> 3) & 1) << 1) | ((lane & 1) << 2) | (((lane >> 1) & 1) << 3);
}

// lane-wise  mask bit ? a : b  with a lane mask that lives in an SGPR pair (one v_cndmask; spelled in C the compiler
// rebuilds a per-lane bool from the mask with two v_and and a 64-bit compare)
__device__ __forceinline__ float mask_select(u64 mask, float a, float b)
{
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
    return r;
}

// Sums of FOUR per-lane values over the 64 lanes, transposed: every lane l returns the wave total of value (l & 3).
// xor-1 and xor-2 exchanges halve the register count (each side keeps the values whose index bit matches its lane bit and adds
// the partner's copy), two row rotations by multiples of four add the quads of a row, two lane swaps add the four rows: 15
// vector instructions where four full reductions take about 50.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float transposed_sum4(float v0, float v1, float v2, float v3)
{
    const u64 odd = 0xAAAAAAAAAAAAAAAAull, upper2 = 0xCCCCCCCCCCCCCCCCull; // lanes with bit 0 / bit 1 set
    // stage 1 (i <-> i ^ 1): even lanes keep v0, v2, odd lanes v1, v3
    const float k0 = mask_select(odd, v1, v0) + dpp_mov<0xB1>(mask_select(odd, v0, v1));
    const float k1 = mask_select(odd, v3, v2) + dpp_mov<0xB1>(mask_select(odd, v2, v3));
    // stage 2 (i <-> i ^ 2): lanes with bit 1 clear keep k0 (values 0, 1), the others k1 (values 2, 3)
    float r = mask_select(upper2, k1, k0) + dpp_mov<0x4E>(mask_select(upper2, k0, k1));
    r += dpp_mov<0x124>(r); // row_ror:4
    r += dpp_mov<0x128>(r); // row_ror:8: all four quads of the row
    return rows_sum(r);
}

// One 8-B store per lane whose bit is set in `mask`: exec IS the ballot mask.  (`if ((mask >> lane) & 1)` makes the compiler
// rebuild the predicate per lane: two v_and, a 64-bit compare and a saveexec per quarter.)
__device__ __forceinline__ void store_pair_masked(u64 mask, WPair *dst, const WPair &e)
{
    u64 saved;
    const u64 bits = ((u64)e.pix << 32) | (u64)(u32)__float_as_int(e.w);
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, %1\n\t"
                 "global_store_dwordx2 %2, %3, off\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(saved)
                 : "s"(mask), "v"(dst), "v"(bits)
                 : "memory");
}

typedef float f32x4m __attribute__((ext_vector_type(4)));

// Exchange between the four rows of 16 lanes: on entry register x[rho] of lane row g holds block (rho, g); on exit register x[g]
// of lane row rho holds it (a 4 x 4 transpose of register index against lane row: two v_permlane16_swap, two v_permlane32_swap).
__device__ __forceinline__ void rows_transpose4(float (&x)[4])
{
#if __HIP_DEVICE_COMPILE__
    auto sw16 = [](float &a, float &b) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(a), __float_as_int(b), false, false);
        a = __int_as_float(r[0]), b = __int_as_float(r[1]);
    };
    auto sw32 = [](float &a, float &b) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(a), __float_as_int(b), false, false);
        a = __int_as_float(r[0]), b = __int_as_float(r[1]);
    };
    sw16(x[0], x[1]); // rows 1 <-> 0 and 3 <-> 2 of the pair
    sw16(x[2], x[3]);
    sw32(x[0], x[2]); // rows {2, 3} <-> {0, 1} of the pair
    sw32(x[1], x[3]);
#endif
}


// One quarter (4 tile rows x 16 pixels) of a tile through the encoder on the matrix cores: R[rho][r] of lane (column, g) becomes
// output 4 g + r of the pixel (column, row 4 q + rho).  One MFMA tile = ONE TILE ROW: M = the 16 encoder outputs (A operand: lane
// (n = lane % 16, kq = lane / 16) reads its encoder value from LDS), N = the row's 16 pixels (B operand: lane (column, kq) holds the
// float4 feats[pixel][16 j + 4 kq ..] of k-block j, component i feeds step i); the four rows are accumulated side by side, so
// the four encoder reads of a k-block serve 16 MFMAs.  PF k-blocks (PF x 4 rows x 16 B per lane) are in flight.
// address = wave-uniform row base (scalar registers) + ONE per-lane byte offset: four 64-bit pointers per lane would cost eight
// registers of the 128 the kernel may use.
template <int PF>
__device__ __forceinline__ void encode_quarter(const float *feats, int64_t fs_y, int H, int ty, int q, u32 lane_off, int nb,
                                               const float *ebase, f32x4m (&R)[4])
{
    const char *rowp[4];
#pragma unroll
    for (int rho = 0; rho < 4; ++rho)
        rowp[rho] = reinterpret_cast<const char *>(feats + (int64_t)min(ty * kTile + 4 * q + rho, H - 1) * fs_y);
    auto ld = [&](int rho, int j) __attribute__((always_inline)) -> float4 {
        return *reinterpret_cast<const float4 *>(rowp[rho] + ((size_t)lane_off + (size_t)(64 * j)));
    };
#pragma unroll
    for (int rho = 0; rho < 4; ++rho)
        R[rho] = f32x4m{0.f, 0.f, 0.f, 0.f};
    float4 b[PF][4];
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int rho = 0; rho < 4; ++rho)
            b[u][rho] = ld(rho, min(u, nb - 1));
    for (int j0 = 0; j0 < nb; j0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int j = j0 + u;
            if (j >= nb)
                break;
            float4 bv[4];
#pragma unroll
            for (int rho = 0; rho < 4; ++rho)
                bv[rho] = b[u][rho];
            if (j + PF < nb) {
#pragma unroll
                for (int rho = 0; rho < 4; ++rho)
                    b[u][rho] = ld(rho, j + PF);
            }
            const float *e = ebase + j * 256; // (kq, n) = lane; steps i are 64 floats apart
            const float a0 = e[0], a1 = e[64], a2 = e[128], a3 = e[192];
#pragma unroll
            for (int rho = 0; rho < 4; ++rho)
                R[rho] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv[rho].x, R[rho], 0, 0, 0);
#pragma unroll
            for (int rho = 0; rho < 4; ++rho)
                R[rho] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv[rho].y, R[rho], 0, 0, 0);
#pragma unroll
            for (int rho = 0; rho < 4; ++rho)
                R[rho] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, bv[rho].z, R[rho], 0, 0, 0);
#pragma unroll
            for (int rho = 0; rho < 4; ++rho)
                R[rho] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, bv[rho].w, R[rho], 0, 0, 0);
        }
    }
}

template <int MODE, int WAVES = 1>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_blend(ViewDev V, const u32 *__restrict__ tile_offsets,
                                              const u32 *__restrict__ vals, const G2D *__restrict__ g2d,
                                              Counters *__restrict__ ctr, Header *__restrict__ headers,
                                              u32 *__restrict__ hdr_count, WPair *__restrict__ wpool, u32 pair_cap,
                                              u32 *__restrict__ shards, const u32 *__restrict__ tile_order, float *__restrict__ alphas,
                                              int dbg_arg, int prio,
                                              float *__restrict__ d_out, float scale_d, FusedArgs fu)
{
#ifdef GWBP_PROFILE
    const int dbg = dbg_arg; // ablation bits (make PROFILE=1 only; results invalid)
#else
    constexpr int dbg = 0;   // the product kernel does not even contain the ablation branches
    (void)dbg_arg;
#endif
    constexpr bool WSUM = MODE == kHalves; // the record's weight sum in its header (+ d[gid] right here)
    constexpr bool PC = MODE == kFusedPC;
    constexpr bool FUSED = MODE == kFused || MODE == kFusedEnc || PC;
    constexpr bool TOKEN = MODE == kToken;
    constexpr int BW = PC ? WAVES - kPcProd : WAVES; // waves that blend (and own a staging area)
    front_priority(prio);
    __shared__ float4 s_a_[BW][kBatch]; // mx, my, opac, gid bits
    __shared__ float4 s_b_[BW][kBatch]; // ca, cb, cc, strip mask
    __shared__ float s_thr_[BW][kBatch]; // ln(255 o) + margin: sigma above this cannot reach alpha >= 1/255
    __shared__ u32 s_pos_[TOKEN ? WAVES : 1][TOKEN ? kBatch : 1]; // kToken: emit position of the (Gaussian, tile) pair
    const int wave = WAVES > 1 ? (int)uniform(threadIdx.x >> 6) : 0;
    const int bw = PC ? max(wave - kPcProd, 0) : wave;
    float4 *const s_a = s_a_[bw], *const s_b = s_b_[bw];
    float *const s_thr = s_thr_[bw];
    u32 *const s_pos = s_pos_[TOKEN ? wave : 0];
    const int lane = (int)(threadIdx.x & 63u);

    const int n_tiles_all = V.tile_w * V.tile_h;
    const int slot = (int)blockIdx.x * WAVES + wave;
    // kFusedPC: dynamic LDS = encoder | ring of kPcRing encoded tiles | ring states + number of producers that have finished
    float *pc_ring = nullptr;
    u32 *pc_state = nullptr;
    if constexpr (MODE == kFusedEnc || PC) {
        // the encoder, re-ordered per MFMA step like k_encode_map's: [K/16][i][kq][n] <- enc[16 j + 4 kq + i][n]
        extern __shared__ __attribute__((aligned(16))) float s_enc[];
        for (int idx = threadIdx.x; idx < fu.enc_k * kFusedCh; idx += 64 * WAVES) {
            const int n = idx & 15, kq = (idx >> 4) & 3, i = (idx >> 6) & 3, j = idx >> 8;
            s_enc[idx] = n < fu.D ? fu.enc[(int64_t)(16 * j + 4 * kq + i) * fu.D + n] : 0.f;
        }
        if constexpr (PC) {
            pc_ring = s_enc + fu.enc_k * kFusedCh;
            pc_state = reinterpret_cast<u32 *>(pc_ring + kPcRing * kPcTileFloats);
            if (threadIdx.x <= kPcRing)
                pc_state[threadIdx.x] = 0u; // kPcRing slot states + the finished-producers count
            if (blockIdx.x == 0 && threadIdx.x == 0)
                ctr->blend_kind = kBlendFused; // (no k_pool_stats launch behind the fused kernels: the pool is untouched)
        }
        __syncthreads();
    }
    if constexpr (PC) {
        if (wave < kPcProd) {
            // ---- producer wave: claim a tile (heaviest first, one global counter), claim an empty ring slot, stream the tile's
            // 256 pixels x K channels through the matrix cores into it, publish it; until the tiles run out
            extern __shared__ __attribute__((aligned(16))) float s_enc[];
            const int nb = fu.enc_k >> 4, kq = lane >> 4, col = lane & 15;
            const float *ebase = s_enc + lane;
            // (Control flow of this loop is kept to plain if / else with wave-uniform conditions, every LDS atomic alone under
            // `lane == 0`, the publication stored by all lanes -- no `break` out of the tile loop, no one-lane branch behind the
            // encode: with those, hipcc's control-flow structuriser built a loop that lane 0 left and lanes 1..63 went round
            // again, and the first version of this kernel hung the device.)
            bool more = true;
            while (more) {
                u32 t = 0;
                if (lane == 0)
                    t = atomicAdd(fu.pc_queue, 1u);
                t = uniform(t);
                if (t < (u32)n_tiles_all) {
                    const int ptile = (int)tile_order[t];
                    const int ptx = ptile % V.tile_w, pty = ptile / V.tile_w;
                    u32 rs = kPcStalled;
                    for (u32 spins = 0; spins < kPcMaxSpins && rs == kPcStalled; ++spins) {
                        for (u32 k = 0; k < (u32)kPcRing && rs == kPcStalled; ++k) {
                            u32 old = kPcBusy;
                            if (lane == 0)
                                old = atomicCAS(&pc_state[k], kPcEmpty, kPcBusy);
                            if (uniform(old) == kPcEmpty)
                                rs = k;
                        }
                        if (rs == kPcStalled)
                            __builtin_amdgcn_s_sleep(8);
                    }
                    if (rs != kPcStalled) {
                        float *dst = pc_ring + rs * kPcTileFloats;
                        const int cx = min(ptx * kTile + col, V.W - 1);
                        const u32 lane_off = (u32)(((int64_t)cx * fu.fs_x + 4 * kq) * (int64_t)sizeof(float));
#pragma unroll 1
                        for (int q = 0; q < 4; ++q) {
                            f32x4m R[4];
                            encode_quarter<4>(fu.feats, fu.fs_y, V.H, pty, q, lane_off, nb, ebase, R); // 4 k-blocks in flight: 16 KB per wave
                            // R[rho] of lane (col, g = kq) = outputs 4 g .. 4 g + 3 of pixel (col, row 4 q + rho): one 16-B store each;
                            // the 16-B chunk index is xor-ed with the pixel's low bits so that neither these stores nor the consumers'
                            // reads of four consecutive chunks per pixel pile onto the same LDS banks
#pragma unroll
                            for (int rho = 0; rho < 4; ++rho) {
                                const int pix = (4 * q + rho) * 16 + col;
                                *reinterpret_cast<f32x4m *>(dst + pix * kFusedCh + ((kq ^ (pix & 3)) << 2)) = R[rho];
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // the tile's stores before its publication
                        __hip_atomic_store(&pc_state[rs], (u32)ptile + 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
                        // (never seen: the blend waves free a slot as soon as they have copied it.  A protocol error must end the
                        // launch with a flag, not hang the device)
                        if (lane == 0)
                            atomicOr(&ctr->overflow, kOverflowRingStall);
                        more = false;
                    }
                } else {
                    more = false;
                }
            }
            if (lane == 0)
                atomicAdd(&pc_state[kPcRing], 1u); // (behind this wave's last publication: LDS operations of a wave complete in order)
            return;
        }
    }
    u32 pc_slot = 0;
    for (;;) { // (kFusedPC: a blend wave takes tile after tile from the ring; every other mode runs this body once)
        int tile_sel;
        if constexpr (PC) {
            // ---- consumer wave: take any published ring slot; leave when every producer has finished and nothing is published
            int got = -1;
            {
                // one look at the ring (wave-uniform loop, the atomics under `lane == 0`: see the producers' note)
                auto scan = [&]() __attribute__((always_inline)) {
                    for (u32 k = 0; k < (u32)kPcRing && got < 0; ++k) {
                        u32 v = 0, old = ~0u;
                        if (lane == 0) {
                            v = __hip_atomic_load(&pc_state[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (v >= 2u)
                                old = atomicCAS(&pc_state[k], v, kPcBusy);
                        }
                        v = uniform(v), old = uniform(old);
                        if (v >= 2u && old == v)
                            got = (int)((v - 2u) << 8 | k);
                    }
                };
                for (u32 spins = 0; got < 0; ++spins) {
                    if (spins >= kPcMaxSpins) { // (see the producers' guard)
                        if (lane == 0)
                            atomicOr(&ctr->overflow, kOverflowRingStall);
                        break;
                    }
                    scan();
                    if (got >= 0)
                        break;
                    u32 fin = 0;
                    if (lane == 0)
                        fin = __hip_atomic_load(&pc_state[kPcRing], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (uniform(fin) == (u32)kPcProd) {
                        scan(); // every producer has published its last tile: one more look at the ring, then leave
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            if (got < 0)
                return;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            pc_slot = (u32)got & 0xFFu;
            tile_sel = got >> 8;
        } else {
            if (WAVES > 1 && slot >= n_tiles_all)
                return; // (behind the only barrier of the kernel: the waves of a workgroup are independent from here on)
            tile_sel = (int)tile_order[slot]; // longest lists first
        }
        const int tile = tile_sel;
        const int tx = tile % V.tile_w, ty = tile / V.tile_w;
        const int ix = tx * kTile + (lane & 15), iy0 = ty * kTile + (lane >> 4);
        const float px = (float)ix + 0.5f;
        const u32 beg = tile_offsets[tile], end = tile_offsets[tile + 1];

        // The weight pool is carved into kShards regions with their own head words: a single head saturates at
        // ~88 returning same-address atomics per microsecond, which cost 1.0 ms/view with ~96 K page grabs per view.
        const u32 shard = (u32)tile % (u32)kShards;
        const u32 shard_cap = (pair_cap / (u32)kShards) & ~(u32)(kPage - 1);
        const u32 shard_base = shard * shard_cap;
        u32 *shard_head = shards + shard * 16;

        // Per-pixel state.  T = 0 encodes "terminated or outside the image": a pixel with T = 0 can never produce a valid
        // pair again (T (1 - alpha) = 0 <= 1e-4), so no separate done flag -- and none of the scalar mask bookkeeping a
        // bool per lane costs (the first version of this loop issued more SALU than VALU instructions).  Tout keeps the
        // transmittance to report (1 - Tout = alpha map), which the terminating Gaussian must not change.
        float T[4], Tout[4], py[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int iy = iy0 + 4 * q;
            py[q] = (float)iy + 0.5f;
            T[q] = (ix < V.W && iy < V.H) ? 1.0f : 0.0f;
            Tout[q] = 1.0f;
        }
        u32 page_pos = 0, page_left = 0, npairs = 0, hdr_n = 0, span_n = 0; // wave-uniform
        bool dead = false;                                     // wave-uniform: pool exhausted

        // kToken: which of the tile's (at most) 2 x 2 tokens a pixel falls into, as lane masks: cmask = lanes whose column lies in the
        // tile's SECOND token column, rmask[q] = lanes whose row 4 q + lane / 16 lies in its second token row.  The nearest index maps
        // are non-decreasing; a tile that spans more than two token columns or rows (texels narrower than a tile) violates the entry
        // point's precondition and raises overflow bit 3 -- the host never takes this path for such maps.
        u64 cmask = 0ull, rmask[4] = {0ull, 0ull, 0ull, 0ull};
        if constexpr (TOKEN) {
            const int c_rel = fu.xmap[min(ix, V.W - 1)] - fu.xmap[min(tx * kTile, V.W - 1)];
            cmask = __ballot(c_rel >= 1);
            bool wide = c_rel > 1 || c_rel < 0;
            const int r_base = fu.ymap[min(ty * kTile, V.H - 1)];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r_rel = fu.ymap[min(iy0 + 4 * q, V.H - 1)] - r_base;
                rmask[q] = __ballot(r_rel >= 1);
                wide |= r_rel > 1 || r_rel < 0;
            }
            if (__ballot(wide) != 0ull && lane == 0)
                atomicOr(&ctr->overflow, kOverflowTokenGeometry);
        }

        // kFused: the lane's four pixels, kFusedCh channels each, stay in registers for the whole tile (pixels outside the
        // image never get a weight: T = 0; they read a clamped address)
        float f[FUSED ? 4 : 1][FUSED ? kFusedCh : 1];
        if constexpr (MODE == kFusedEnc) {
            // f[q][c] = sum_k feats[pixel (ix, iy0 + 4 q)][k] enc[k][c].  One MFMA tile = ONE TILE ROW: M = the 16 encoder outputs
            // (A operand: lane (n = lane % 16, kq = lane / 16) reads its encoder value from LDS), N = the row's 16 pixels (B operand:
            // lane (column = lane % 16, kq) holds the float4 feats[pixel][16 j + 4 kq ..] of k-block j, component i feeds step i),
            // so the result registers of lane (column, g) are outputs 4 g .. 4 g + 3 of that row's pixel in ITS column.  Four
            // rows (one quarter) are accumulated side by side -- the four encoder reads of a k-block serve 16 MFMAs -- and a
            // 4 x 4 exchange between the lane rows hands every lane all 16 outputs of its own pixel (row 4 q + lane / 16).
            extern __shared__ __attribute__((aligned(16))) float s_enc[];
            const int kq = lane >> 4;
            const int cx = min(ix, V.W - 1);
            const int nb = fu.enc_k >> 4;
            const float *ebase = s_enc + lane;
            const u32 lane_off = (u32)(((int64_t)cx * fu.fs_x + 4 * kq) * (int64_t)sizeof(float));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4m R[4];
                encode_quarter<2>(fu.feats, fu.fs_y, V.H, ty, q, lane_off, nb, ebase, R); // 2 k-blocks in flight: 8 KB per wave
                // R[rho][r] of lane row g = output 4 g + r of the pixel (column, row 4 q + rho)  ->  f[q][4 g + r] of lane row rho
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x[4] = {R[0][r], R[1][r], R[2][r], R[3][r]};
                    rows_transpose4(x);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        f[q][4 * g + r] = x[g];
                }
            }
        }
        if constexpr (PC) {
            // the tile's encoded pixels from the ring slot into the registers the blend works from, then the slot is free again
            const float *src = pc_ring + pc_slot * kPcTileFloats;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int pix = (4 * q + (lane >> 4)) * 16 + (lane & 15);
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const f32x4m v = *reinterpret_cast<const f32x4m *>(src + pix * kFusedCh + ((c4 ^ (pix & 3)) << 2));
                    f[q][4 * c4] = v[0], f[q][4 * c4 + 1] = v[1], f[q][4 * c4 + 2] = v[2], f[q][4 * c4 + 3] = v[3];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // (waits for the reads: the slot may be refilled behind it)
            __hip_atomic_store(&pc_state[pc_slot], kPcEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // (all lanes, same word)
        }
        if constexpr (MODE == kFused) {
            const int cx = min(ix, V.W - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float *src = fu.feats + (int64_t)min(iy0 + 4 * q, V.H - 1) * fu.fs_y + (int64_t)cx * fu.fs_x;
                if (fu.vec4) {
#pragma unroll
                    for (int c4 = 0; c4 < kFusedCh / 4; ++c4) {
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (4 * c4 < fu.D)
                            v = reinterpret_cast<const float4 *>(src)[c4];
                        f[q][4 * c4] = v.x, f[q][4 * c4 + 1] = v.y, f[q][4 * c4 + 2] = v.z, f[q][4 * c4 + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < kFusedCh; ++c)
                        f[q][c] = c < fu.D ? src[c] : 0.f;
                }
            }
        }
        // A non-finite feature value (backproject.py:109: feats / feats.norm() of an all-zero pixel is NaN) must reach exactly the
        // Gaussians that have a weight AT that pixel -- the reference's backward adds fac * v_render for contributing pairs only.
        // The record sums below multiply a lane's pixels by a weight that is 0 where the record has no entry, and 0 x NaN is NaN,
        // which poisoned every record of the tile (found in round 5; the vector scatter kernels had been fixed in round 3).
        // Once per tile: bad[q] = lanes whose pixel of quarter q holds a non-finite value (scalar masks), those pixels are zeroed
        // in the registers, and a record whose entry mask meets bad[] gets NaN sums instead -- a few scalar instructions per
        // record, no vector work in tiles without such pixels.  (All channels of the record become NaN, also where the reference
        // would keep a finite or infinite one: the row is non-finite either way, and backproject.py:166-169 maps it to zeros.)
        u64 bad[4] = {0ull, 0ull, 0ull, 0ull};
        if constexpr (FUSED) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float chk = 0.f;
#pragma unroll
                for (int c = 0; c < kFusedCh; ++c)
                    chk = __builtin_fmaf(f[q][c], 0.f, chk); // NaN iff some value is NaN or +-inf
                bad[q] = __ballot(chk != chk);
                if (bad[q] != 0ull) { // (wave-uniform, rare)
#pragma unroll
                    for (int c = 0; c < kFusedCh; ++c)
                        f[q][c] = (chk != chk) ? 0.f : f[q][c];
                }
            }
        }
        // per-lane constants of the record flush: lanes 0..15 add channel transposed_channel(lane) of F[gid], lane 16 adds d[gid]
        const int my_ch = transposed_channel(lane);
        float *out_base = nullptr;
        size_t out_mul = 0;
        float out_scale = 0.f;
        bool out_on = false;
        if constexpr (FUSED) {
            out_base = lane < 16 ? fu.F + my_ch : d_out;
            out_mul = lane < 16 ? (size_t)fu.D : (size_t)1;
            out_scale = lane < 16 ? fu.scale_f : scale_d;
            out_on = lane < 16 ? my_ch < fu.D : (lane == 16 && d_out != nullptr);
        }

        for (u32 batch = beg; batch < end; batch += kBatch) {
            // quarters that still have a live pixel; stop when the whole tile has terminated (gsplat: all threads done)
            u32 alive = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                alive |= (__ballot(T[q] > 0.f) != 0ull ? 1u : 0u) << q;
            if (alive == 0)
                break;
            const u32 bn = min((u32)kBatch, end - batch);
            if ((u32)lane < bn) {
                const u32 gid = vals[batch + lane];
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
                    if (xhit) {
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
                s_a[lane] = make_float4(a.x, a.y, a.z, __int_as_float((int)gid));
                s_b[lane] = make_float4(b.x, b.y, b.z, __int_as_float((int)smask));
                // alpha = o exp(-sigma) >= 1/255  <=>  sigma <= ln(255 o); 1e-3 absorbs the error of __logf and exp_neg
                s_thr[lane] = L + 1e-3f;
                if constexpr (TOKEN) { // this (Gaussian, tile) pair's emit position: k_emit walked the rectangle row-major from estart
                    const uint2 rc = fu.rect[gid];
                    const u32 rx0 = rc.x & 0xFFFFu, rx1 = rc.x >> 16, ry0 = rc.y & 0xFFFFu;
                    s_pos[lane] = fu.estart[gid] + ((u32)ty - ry0) * (rx1 - rx0) + ((u32)tx - rx0);
                }
            }
            // single wave: LDS operations of one wave complete in program order, no barrier needed

            for (u32 j = 0; j < bn && !(dbg & 2); ++j) {
                const float4 b = s_b[j];
                const u32 sm = uniform((u32)__float_as_int(b.w)) & alive;
                if (sm == 0)
                    continue; // this Gaussian cannot reach any live quarter of the tile
                const float4 a = s_a[j];
                const float thr = s_thr[j];
                const float dx = a.x - px;
                const float adx = b.x * dx, bdx = b.y * dx; // shared by the four quarters (same column)
                u64 m[4];
                float w[4];
                // Per-lane control flow is branch-free and flag-free (float selects); the only branches are wave-uniform.
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    m[q] = 0ull, w[q] = 0.f;
                    if ((sm >> q) & 1u) {
                        const float dy = a.y - py[q];
                        const float sigma = __builtin_fmaf(bdx, dy, 0.5f * __builtin_fmaf(adx, dx, (b.z * dy) * dy));
                        // no live pixel of this quarter lies inside the alpha >= 1/255 ellipse: skip exp / T / ballot
                        if (__ballot(T[q] > 0.f && sigma <= thr) == 0ull)
                            continue;
                        const float alpha = __builtin_fminf(kAlphaMax, a.z * exp_neg_sigma(sigma));
                        // The three tests as lane masks (a ballot of a COMPARE is that compare's own SGPR result; a ballot of
                        // their conjunction costs v_cndmask + v_cmp_ne), combined on the scalar unit.
                        const float next_T = T[q] * (1.0f - alpha);
                        const u64 m_ok = __builtin_amdgcn_ballot_w64(sigma >= 0.f) &       // sigma < 0: skipped
                                         __builtin_amdgcn_ballot_w64(alpha >= kAlphaMin);  // alpha < 1/255: skipped
                        const u64 m_valid = m_ok & __builtin_amdgcn_ballot_w64(next_T > kTMin); // T' <= 1e-4: terminates, NOT counted
                        const float T_else = mask_select(m_ok, 0.f, T[q]);       // ok but not valid -> terminated
                        w[q] = alpha * T[q];
                        T[q] = mask_select(m_valid, next_T, T_else);
                        Tout[q] = mask_select(m_valid, next_T, Tout[q]);
                        m[q] = m_valid;
                    }
                }
                if ((m[0] | m[1] | m[2] | m[3]) == 0ull) {
                    // a quarter may have died without contributing: refresh the live set lazily
                    if ((j & 7u) == 7u) {
                        alive = 0;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            alive |= (__ballot(T[q] > 0.f) != 0ull ? 1u : 0u) << q;
                        if (alive == 0)
                            break;
                    }
                    continue;
                }
                // Entry layout of a record: quarters 0 | 1 back to back, then quarters 2 | 3 back to back.  kHalves: the second half
                // starts on a multiple of kListPad entries (64 B), i.e. BOTH half-tile lists are padded -- k_scatter_wide fetches a
                // half's entries eight at a time with s_load_dwordx16 and must find {0, kPadPix} behind the last real one.
                u32 cnt[4], base[4], total = 0, mid_pad = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    cnt[q] = (u32)__popcll(m[q]);
                    if (WSUM && q == 2) {
                        mid_pad = (0u - total) & (u32)(kListPad - 1);
                        total += mid_pad;
                    }
                    base[q] = total;
                    total += cnt[q];
                }
                if constexpr (TOKEN) {
                    // the record's four token-quadrant weight sums (index qx | qy << 1), one 16-B line at its emit position
                    float a0 = 0.f, a1 = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (m[q] != 0ull) { // wave-uniform
                            const float wq = mask_select(m[q], w[q], 0.f);
                            a1 += mask_select(rmask[q], wq, 0.f);
                            a0 += mask_select(rmask[q], 0.f, wq);
                        }
                    const float r = transposed_sum4(mask_select(cmask, 0.f, a0), mask_select(cmask, a0, 0.f),
                                                    mask_select(cmask, 0.f, a1), mask_select(cmask, a1, 0.f));
                    if (!(dbg & 1) && lane < 4)
                        fu.omega[(size_t)s_pos[j] * 4 + lane] = r;
                    ++hdr_n;
                    npairs += total;
                    continue;
                }
                if constexpr (FUSED) {
                    float p[kFusedCh], wl = 0.f;
#pragma unroll
                    for (int c = 0; c < kFusedCh; ++c)
                        p[c] = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (m[q] != 0ull) { // wave-uniform
                            const float wq = mask_select(m[q], w[q], 0.f); // the ballot register is the select mask
                            wl += wq;
#pragma unroll
                            for (int c = 0; c < kFusedCh; ++c)
                                p[c] = __builtin_fmaf(wq, f[q][c], p[c]);
                        }
                    float tot = transposed_sum16(p); // every row: channel my_ch
                    if (((m[0] & bad[0]) | (m[1] & bad[1]) | (m[2] & bad[2]) | (m[3] & bad[3])) != 0ull) // (scalar; see bad[] above)
                        tot = __builtin_nanf("");
                    wl += dpp_get<0xB1>(wl);
                    wl += dpp_get<0x4E>(wl);
                    wl += dpp_get<0x141>(wl);
                    wl += dpp_get<0x140>(wl);
                    const float ws = rows_sum(wl);
                    if (!(dbg & 1) && out_on) {
                        // one atomic instruction: lanes 0..15 the record's channel sums, lane 16 its share of d
                        const u32 gid = (u32)__float_as_int(a.w);
                        atomicAdd(out_base + (size_t)gid * out_mul, (lane < 16 ? tot : ws) * out_scale);
                    }
                    ++hdr_n;
                    npairs += total;
                    continue;
                }
                const u32 padded = (total + (kListPad - 1)) & ~(u32)(kListPad - 1);
                const u32 end_pad = padded - total;
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
#pragma unroll
                    for (int q = 0; q < 4; ++q) { // exec = the quarter's ballot mask
                        WPair e;
                        e.w = w[q], e.pix = (u32)(q * 64 + lane);
                        store_pair_masked(m[q], wpool + (page_pos + base[q] + mbcnt(m[q])), e);
                    }
                    if ((u32)lane < mid_pad + end_pad) { // {0, kPadPix} behind each half (kStore: behind the record): no remainder handling
                        WPair z;
                        z.w = 0.f, z.pix = kPadPix;
                        wpool[page_pos + ((u32)lane < mid_pad ? base[2] - mid_pad + lane : total + lane - mid_pad)] = z;
                    }
                    // only the 256-channel scatter path wants the record's weight sum (its visit loop has no room for d), hence the
                    // template parameter
                    float wsum = 0.f; // the record's share of d[gid] (k_accum_d)
                    if constexpr (WSUM) {
                        float wl = 0.f;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            wl += mask_select(m[q], w[q], 0.f);
                        wsum = wave_sum(wl);
                    }
                    if (lane == 0) {
                        Header h;
                        h.gid = (u32)__float_as_int(a.w);
                        h.woff[0] = page_pos, h.woff[1] = page_pos + base[1];
                        h.woff[2] = page_pos + base[2], h.woff[3] = page_pos + base[3];
                        h.counts = cnt[0] | (cnt[1] << 8) | (cnt[2] << 16) | (cnt[3] << 24);
                        // kHalves: a record with entries in BOTH halves of the tile gets the next carry row of its tile (k_scatter_wide
                        // parks its partial sums there between the two half-tile passes): compact indices keep a workgroup's carry
                        // rows few and hot in L2
                        const bool spans = WSUM && (cnt[0] + cnt[1]) != 0 && (cnt[2] + cnt[3]) != 0;
                        h.wsum = (u32)__float_as_int(wsum), h.carry_row = spans ? span_n : 0xFFFFFFFFu;
                        h.mask[0] = m[0], h.mask[1] = m[1], h.mask[2] = m[2], h.mask[3] = m[3];
                        headers[beg + hdr_n] = h;
                        if constexpr (WSUM) { // gwbp_blend_weights_d: the record's share of d[gid] right here (no k_accum_d)
                            if (d_out) // (spelled as the instruction: hipcc wraps a single-lane atomicAdd in its wave-aggregation code)
                                asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(d_out + h.gid), "v"(wsum * scale_d) : "memory");
                        }
                    }
                    ++hdr_n;
                    if (WSUM && (cnt[0] + cnt[1]) != 0 && (cnt[2] + cnt[3]) != 0)
                        ++span_n;
                }
                page_pos += padded;
                page_left -= padded;
                npairs += total - mid_pad;
            }
        }
        if (lane == 0) {
            if ((FUSED || TOKEN) && !PC && slot == 0)
                ctr->blend_kind = TOKEN ? kBlendToken : kBlendFused; // (no k_pool_stats launch behind these: the pool is untouched)
            hdr_count[tile] = (FUSED || TOKEN) ? 0u : hdr_n; // kFused / kToken: the store stays empty
            if (hdr_n)
                atomicAdd(&ctr->n_headers, hdr_n);
            if (npairs)
                atomicAdd(&ctr->n_pairs, (u64)npairs);
        }
        if (alphas) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int iy = iy0 + 4 * q;
                if (ix < V.W && iy < V.H)
                    alphas[(size_t)iy * V.W + ix] = 1.0f - Tout[q];
            }
        }
        if constexpr (!PC)
            break;
    } // for (;;): the next tile of a kFusedPC blend wave
}

// Small images (<= kQuarterMaxTiles tiles): the fused blend + scatter with ONE QUARTER of a tile per wave, lane = one pixel.
// A tile's blend is a dependent chain over its Gaussians; with a few hundred tiles the chip holds one wave per SIMD at
// most and k_blend is bound by that chain's latency (C1: 0.27 ms for 475 waves, then 0.13 ms of scatter kernel).  Four waves
// per tile run four shorter chains side by side, a pixel's CH channels fit in CH registers (so D <= 32 works), and a
// contributing (Gaussian, quarter) is flushed by the wave that found it: no store, no scatter kernel.  Same arithmetic
// per pixel as k_blend, statement for statement (the alpha map is compared bit for bit with the oracle's).
constexpr int kQuarterMaxTiles = 4096;
constexpr int kQuarterMaxCh = 32;

template <int CH> // 16 or 32 channels held per pixel (D <= CH)
__global__ __launch_bounds__(64) void k_blend_scatter_quarter(ViewDev V, const u32 *__restrict__ tile_offsets,
                                                              const u32 *__restrict__ vals, const G2D *__restrict__ g2d,
                                                              Counters *__restrict__ ctr, u32 *__restrict__ hdr_count,
                                                              const u32 *__restrict__ tile_order, float *__restrict__ alphas,
                                                              int dbg_arg, int prio, float *__restrict__ d_out, float scale_d,
                                                              FusedArgs fu)
{
#ifdef GWBP_PROFILE
    const int dbg = dbg_arg; // ablation bits (make PROFILE=1 only; results invalid)
#else
    constexpr int dbg = 0;   // the product kernel does not even contain the ablation branches
    (void)dbg_arg;
#endif
    front_priority(prio);
    __shared__ float4 s_a[kBatch]; // mx, my, opac, gid bits
    __shared__ float4 s_b[kBatch]; // ca, cb, cc, "may reach this quarter"
    __shared__ float s_thr[kBatch];
    const int tile = (int)tile_order[blockIdx.x >> 2]; // longest lists first
    const int q = (int)(blockIdx.x & 3u);
    const int tx = tile % V.tile_w, ty = tile / V.tile_w;
    const int lane = threadIdx.x;
    const int ix = tx * kTile + (lane & 15), iy = ty * kTile + 4 * q + (lane >> 4);
    const float px = (float)ix + 0.5f, py = (float)iy + 0.5f;
    const u32 beg = tile_offsets[tile], end = tile_offsets[tile + 1];
    float T = (ix < V.W && iy < V.H) ? 1.0f : 0.0f, Tout = 1.0f;

    float f[CH];
    {
        const float *src = fu.feats + (int64_t)min(iy, V.H - 1) * fu.fs_y + (int64_t)min(ix, V.W - 1) * fu.fs_x;
        if (fu.vec4) {
#pragma unroll
            for (int c4 = 0; c4 < CH / 4; ++c4) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (4 * c4 < fu.D)
                    v = reinterpret_cast<const float4 *>(src)[c4];
                f[4 * c4] = v.x, f[4 * c4 + 1] = v.y, f[4 * c4 + 2] = v.z, f[4 * c4 + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int c = 0; c < CH; ++c)
                f[c] = c < fu.D ? src[c] : 0.f;
        }
    }
    u64 bad = 0ull; // lanes whose pixel holds a non-finite value: zeroed here, NaN for the records that touch them (see k_blend)
    {
        float chk = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c)
            chk = __builtin_fmaf(f[c], 0.f, chk);
        bad = __ballot(chk != chk);
        if (bad != 0ull) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
                f[c] = (chk != chk) ? 0.f : f[c];
        }
    }
    // flush: lanes 16 g .. 16 g + 15 add channels 16 g + transposed_channel(lane) of F[gid], lane CH adds d[gid]
    const int my_ch = 16 * (lane >> 4) + transposed_channel(lane);
    float *const out_base = lane < CH ? fu.F + my_ch : d_out;
    const size_t out_mul = lane < CH ? (size_t)fu.D : (size_t)1;
    const float out_scale = lane < CH ? fu.scale_f : scale_d;
    const bool out_on = lane < CH ? my_ch < fu.D : (lane == CH && d_out != nullptr);
    u32 npairs = 0, nrec = 0;

    for (u32 batch = beg; batch < end; batch += kBatch) {
        if (__ballot(T > 0.f) == 0ull)
            break; // the quarter has terminated
        const u32 bn = min((u32)kBatch, end - batch);
        if ((u32)lane < bn) {
            const u32 gid = vals[batch + lane];
            const float4 *gp = reinterpret_cast<const float4 *>(g2d + gid);
            const float4 a = gp[0], b = gp[1];
            u32 reach = 0; // k_blend's conservative strip mask, this quarter's bit
            const float L = __logf(255.0f * a.z);
            if (L > 0.f) {
                const float idet = 1.0f / (b.x * b.z - b.y * b.y);
                const float ex = 1.05f * __builtin_sqrtf(2.0f * L * b.z * idet) + 1.0f;
                const float ey = 1.05f * __builtin_sqrtf(2.0f * L * b.x * idet) + 1.0f;
                const float x0 = (float)(tx * kTile) + 0.5f, ya = (float)(ty * kTile) + 0.5f + 4.0f * q;
                const bool xhit = !(a.x + ex < x0 || a.x - ex > x0 + 15.0f);
                const bool ymiss = (a.y + ey < ya) || (a.y - ey > ya + 3.0f);
                reach = (xhit && !ymiss) ? 1u : 0u;
                if (!(ex == ex) || !(ey == ey))
                    reach = 1u; // degenerate conic: never reject
            }
            s_a[lane] = make_float4(a.x, a.y, a.z, __int_as_float((int)gid));
            s_b[lane] = make_float4(b.x, b.y, b.z, __int_as_float((int)reach));
            s_thr[lane] = L + 1e-3f;
        }
        for (u32 j = 0; j < bn; ++j) {
            const float4 b = s_b[j];
            if (uniform((u32)__float_as_int(b.w)) == 0u)
                continue;
            const float4 a = s_a[j];
            const float thr = s_thr[j];
            const float dx = a.x - px;
            const float adx = b.x * dx, bdx = b.y * dx;
            const float dy = a.y - py;
            const float sigma = __builtin_fmaf(bdx, dy, 0.5f * __builtin_fmaf(adx, dx, (b.z * dy) * dy));
            if (__ballot(T > 0.f && sigma <= thr) == 0ull)
                continue;
            const float alpha = __builtin_fminf(kAlphaMax, a.z * exp_neg_sigma(sigma));
            const float next_T = T * (1.0f - alpha);
            const u64 m_ok = __builtin_amdgcn_ballot_w64(sigma >= 0.f) & __builtin_amdgcn_ballot_w64(alpha >= kAlphaMin);
            const u64 m_valid = m_ok & __builtin_amdgcn_ballot_w64(next_T > kTMin);
            const float T_else = mask_select(m_ok, 0.f, T);
            const float w = alpha * T;
            T = mask_select(m_valid, next_T, T_else);
            Tout = mask_select(m_valid, next_T, Tout);
            if (m_valid == 0ull)
                continue;
            const float wq = mask_select(m_valid, w, 0.f);
            float p0[16], tot1 = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c)
                p0[c] = wq * f[c];
            float tot0 = transposed_sum16(p0);
            if constexpr (CH == 32) {
                float p1[16];
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    p1[c] = wq * f[16 + c];
                tot1 = transposed_sum16(p1);
            }
            if ((m_valid & bad) != 0ull)
                tot0 = tot1 = __builtin_nanf("");
            float wl = wq;
            wl += dpp_get<0xB1>(wl);
            wl += dpp_get<0x4E>(wl);
            wl += dpp_get<0x141>(wl);
            wl += dpp_get<0x140>(wl);
            const float ws = rows_sum(wl);
            if (!(dbg & 1) && out_on) {
                const u32 gid = (u32)__float_as_int(a.w);
                const float val = lane < 16 ? tot0 : (lane < CH ? tot1 : ws);
                atomicAdd(out_base + (size_t)gid * out_mul, val * out_scale);
            }
            npairs += (u32)__popcll(m_valid);
            ++nrec;
        }
    }
    if (lane == 0) {
        if (blockIdx.x == 0)
            ctr->blend_kind = kBlendFused; // (no k_pool_stats launch behind the fused kernels: the pool is untouched)
        if (q == 0)
            hdr_count[tile] = 0u; // the store stays empty
        if (nrec)
            atomicAdd(&ctr->n_headers, nrec); // (Gaussian, quarter-tile) flushes here, not (Gaussian, tile) records
        if (npairs)
            atomicAdd(&ctr->n_pairs, (u64)npairs);
    }
    if (alphas && ix < V.W && iy < V.H)
        alphas[(size_t)iy * V.W + ix] = 1.0f - Tout;
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

__global__ void k_pool_stats(const u32 *__restrict__ shards, Counters *__restrict__ ctr, u32 blend_kind)
{
    ctr->blend_kind = blend_kind; // which scatter kernels may read this view's store (half-tile lists, weight sums)
    u32 mx = 0;
    for (int i = 0; i < kShards; ++i)
        mx = max(mx, shards[i * 16]);
    ctr->pool_head = mx * (u32)kShards;
}

int launch_blend(const Layout &L, const Ws &W, const ViewDev &V, float *alphas, float *d, float scale_d, hipStream_t s,
                 const FeatMap *M, int D, float scale_f, float *F)
{
    const bool fused = M != nullptr;
    FusedArgs fu = {};
    const bool fused_enc = fused && M->enc != nullptr;
    if (fused) {
        const int cap = (!fused_enc && V.tile_w * V.tile_h <= kQuarterMaxTiles) ? kQuarterMaxCh : kFusedCh;
        if (D < 1 || D > cap)
            return set_error(GWBP_EINVAL, "gwbp_blend_scatter: D must be 1..%d for this image size (got %d)", cap, D);
        if (M->fs_c != 1 || M->ymap || M->xmap)
            return set_error(GWBP_EINVAL, "gwbp_blend_scatter: a full-resolution map with unit channel stride is required");
        if (!M->p || !F)
            return set_error(GWBP_EINVAL, "null feats / F");
        if (M->fs_y < 0 || M->fs_x < 0)
            return set_error(GWBP_EINVAL, "gwbp_blend_scatter: negative strides (%lld %lld)", (long long)M->fs_y, (long long)M->fs_x);
        fu.feats = M->p, fu.fs_y = M->fs_y, fu.fs_x = M->fs_x, fu.D = D, fu.scale_f = scale_f, fu.F = F;
        fu.vec4 = (D % 4 == 0 && M->fs_y % 4 == 0 && M->fs_x % 4 == 0 && (reinterpret_cast<uintptr_t>(M->p) & 15) == 0) ? 1 : 0;
        if (fused_enc) {
            // the tile prologue reads every pixel as float4 k-blocks of 16 channels: same domain as gwbp_encode_map, K <= 512
            if (M->enc_k < 16 || M->enc_k % 16 != 0 || M->enc_k > kEncMaxK)
                return set_error(GWBP_EINVAL, "gwbp_blend_scatter_encoded: K must be a multiple of 16 in 16..%d (got %d)", kEncMaxK,
                                 M->enc_k);
            if (M->fs_y % 4 != 0 || M->fs_x % 4 != 0 || (reinterpret_cast<uintptr_t>(M->p) & 15) != 0 || M->fs_x < M->enc_k)
                return set_error(GWBP_EINVAL, "gwbp_blend_scatter_encoded: pixels must be 16-B aligned runs of K contiguous channels");
            // the prologue addresses a pixel as wave-uniform row base + ONE 32-bit per-lane byte offset (column * fs_x + k-block)
            if (((int64_t)(V.W - 1) * M->fs_x + M->enc_k) * (int64_t)sizeof(float) >= (int64_t)1 << 32)
                return set_error(GWBP_EINVAL, "gwbp_blend_scatter_encoded: a row spans %lld bytes, the kernel's per-lane offsets are "
                                 "32-bit (width %d, pixel stride %lld floats): encode with gwbp_encode_map instead",
                                 (long long)(((int64_t)(V.W - 1) * M->fs_x + M->enc_k) * 4), V.W, (long long)M->fs_x);
            fu.enc = M->enc, fu.enc_k = M->enc_k;
        }
    }
    if (!fused && d && (L.flags & GWBP_FLAG_NARROW_SCATTER))
        return set_error(GWBP_EINVAL, "gwbp_blend_weights_d needs a blend without GWBP_FLAG_NARROW_SCATTER (no weight sums)");
    const int prio = (L.flags & GWBP_FLAG_FRONT_PRIORITY) ? 1 : 0;
    const int n_tiles = V.tile_w * V.tile_h;
    const int fin = sort_passes(n_tiles) & 1;
    // profiling knobs, read once per process (results are invalid when GWBP_ABLATE_BLEND is set)
    const int ablate = profile_knob("GWBP_ABLATE_BLEND");
    const int extra_lds = profile_knob("GWBP_BLEND_LDS"); // experiment knob: pad the workgroup's LDS footprint
#define GWBP_BLEND(H)                                                                                                 \
    hipLaunchKernelGGL(k_blend<H>, dim3(n_tiles), dim3(64), (size_t)extra_lds, s, V, W.tile_offsets, W.vals[fin], W.g2d,  \
                       W.counters, W.headers, W.hdr_count, W.wpool, (u32)L.pair_cap, W.shards, W.tile_order, alphas,  \
                       ablate, prio, d, scale_d, fu)
    if (fused_enc && (L.flags & GWBP_FLAG_SPLIT_ENCODER)) {
        // producer / consumer form: ONE persistent workgroup per CU (kPcProd encoder waves + blend waves around a ring of encoded
        // tiles in LDS); the tile counter is the first scatter queue word, which gwbp_project's memset left zero
        const size_t lds = (size_t)fu.enc_k * kFusedCh * sizeof(float) + (size_t)kPcRing * kPcTileFloats * sizeof(float) + 64;
        int n_cu = 0;
        int rc = device_cus(&n_cu);
        if (rc || (rc = ensure_dynamic_lds(reinterpret_cast<const void *>(k_blend<kFusedPC, kPcWaves>),
                                           kEncMaxK * kFusedCh * 4 + kPcRing * kPcTileFloats * 4 + 64, 14)))
            return rc;
        fu.pc_queue = W.shards + kShards * 16;
        hipLaunchKernelGGL((k_blend<kFusedPC, kPcWaves>), dim3(n_cu), dim3(64 * kPcWaves), lds, s, V, W.tile_offsets, W.vals[fin],
                           W.g2d, W.counters, W.headers, W.hdr_count, W.wpool, (u32)L.pair_cap, W.shards, W.tile_order, alphas,
                           ablate, prio, d, scale_d, fu);
    } else if (fused_enc) {
        const size_t lds = (size_t)fu.enc_k * kFusedCh * sizeof(float);
        hipLaunchKernelGGL((k_blend<kFusedEnc, kEncWaves>), dim3((n_tiles + kEncWaves - 1) / kEncWaves), dim3(64 * kEncWaves), lds, s,
                           V, W.tile_offsets, W.vals[fin], W.g2d, W.counters, W.headers, W.hdr_count, W.wpool, (u32)L.pair_cap,
                           W.shards, W.tile_order, alphas, ablate, prio, d, scale_d, fu);
    } else if (fused && n_tiles <= kQuarterMaxTiles) {
#define GWBP_QUARTER(C)                                                                                               \
    hipLaunchKernelGGL(k_blend_scatter_quarter<C>, dim3(4 * n_tiles), dim3(64), 0, s, V, W.tile_offsets, W.vals[fin], W.g2d, \
                       W.counters, W.hdr_count, W.tile_order, alphas, ablate, prio, d, scale_d, fu)
        if (D <= 16)
            GWBP_QUARTER(16);
        else
            GWBP_QUARTER(32);
#undef GWBP_QUARTER
    } else if (fused)
        GWBP_BLEND(kFused);
    else if (L.flags & GWBP_FLAG_NARROW_SCATTER)
        GWBP_BLEND(kStore);
    else
        GWBP_BLEND(kHalves);
#undef GWBP_BLEND
    if (!fused)
        hipLaunchKernelGGL(k_pool_stats, dim3(1), dim3(1), 0, s, W.shards, W.counters,
                           (L.flags & GWBP_FLAG_NARROW_SCATTER) ? 0u : kBlendHalves);
    return check_hip(hipGetLastError(), "blend launch");
}

// gwbp_blend_tokens: the blend whose product is the per-(Gaussian, tile) token-quadrant weight sums (k_blend<kToken>) in the
// workspace's header region (16 of its 64 B per intersection), zeroed for this view's intersections first.
int launch_blend_tokens(const Layout &L, const Ws &W, const ViewDev &V, float *alphas, const int32_t *ymap, const int32_t *xmap,
                        hipStream_t s)
{
    if (!ymap || !xmap)
        return set_error(GWBP_EINVAL, "gwbp_blend_tokens needs both index maps");
    int rc = launch_zero_omega(L, W, s);
    if (rc)
        return rc;
    FusedArgs fu = {};
    fu.ymap = ymap, fu.xmap = xmap, fu.omega = reinterpret_cast<float *>(W.headers), fu.estart = W.dkeys[1], fu.rect = W.rect;
    const int prio = (L.flags & GWBP_FLAG_FRONT_PRIORITY) ? 1 : 0;
    const int n_tiles = V.tile_w * V.tile_h;
    const int fin = sort_passes(n_tiles) & 1;
    const int ablate = profile_knob("GWBP_ABLATE_BLEND");
    hipLaunchKernelGGL(k_blend<kToken>, dim3(n_tiles), dim3(64), 0, s, V, W.tile_offsets, W.vals[fin], W.g2d, W.counters, W.headers,
                       W.hdr_count, W.wpool, (u32)L.pair_cap, W.shards, W.tile_order, alphas, ablate, prio, nullptr, 0.f, fu);
    return check_hip(hipGetLastError(), "blend_tokens launch");
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
