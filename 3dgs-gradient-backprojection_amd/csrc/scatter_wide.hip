// scatter_wide.hip -- k_scatter_wide: the D % 256 == 0, channel-contiguous path of the weighted scatter-accumulate
// (C2 D = 512, C4 D = 768, dino-sized 1024); semantics identical to k_scatter / k_scatter_full.
//
//   F[g, c0:c0+256] += sum_p w_g(p) * feats[p, c0:c0+256]        (backproject.py:127-131 via colors.grad)
//
// Why a second fast path: k_scatter_full spends 4 vector instructions per (pair, 128 channels) -- 2 v_readlane,
// 1 address add, 1 v_pk_fma_f32.  With 4 channels per lane (ds_read_b128 + 2 v_pk_fma_f32) the two v_readlane and the
// address add are shared by 256 channels: 2.5 instructions per (pair, 128 channels).  The price is LDS capacity:
// 256 px x 256 ch x 4 B does not fit, so a work item = (tile, 256-channel chunk) runs in TWO passes over half-tile slabs
// (tile rows 0..7, then 8..15; 128 px x 256 ch = 128 KB each).  The flush traffic must not grow with it -- fp32 atomics
// run memory-side at ~1.3 TB/s chip-wide and the per-(Gaussian, tile) flush sits right at that rate (profiles/
// r4_wide_ablation.txt) -- so a record with entries in both halves is NOT flushed twice: the top pass parks its partial
// sums in a carry row (plain stores into a per-workgroup slice), the bottom pass starts from them and issues the one
// atomic flush.
//
// Slab layout: row = pixel (1 KB), position 4*l + k of a row holds channel c0 + 64*k + l.  Lane l reads its 16 B with
// one conflict-free ds_read_b128 and owns channels {l, l+64, l+128, l+192}: the flush is four 256-B contiguous
// atomic wave-instructions with no cross-lane transpose.
//
// Round 4 rewrite of the visit machinery (the arithmetic and the order of every record's sum are unchanged):
//   * The pass's visit list is built BY THIS KERNEL, in LDS, from the blend's 64-B record headers while the slab loads are
//     in flight (thread i takes record i, a ballot compacts the records that have entries in this half): the blend no
//     longer writes half-tile lists, and a visit's descriptor is one broadcast ds_read_b96 (a scalar load of it would have forced
//     every lgkmcnt wait of the loop to zero; the ONE scalar load per batch that round 5 put into the loop is accounted for
//     in the batch's counted waits, see GWBP_BATCH_ASM).
//   * Claims are asynchronous: the ds_add_rtn for the visit after next is issued at the top of a visit and read after its
//     first batch; the descriptor read it enables completes under the rest of the visit.
//   * Every VMEM operation of a visit addresses SGPR base + one of two per-lane constants (lane * 4 / 8): no vector
//     address arithmetic, no 64-bit VGPR pairs.
//   * One landing buffer: the carry dwords of the next visit (and the L2 warm-up of its entries) are loaded a whole visit
//     ahead, consumed by the selects that start a visit, and only then re-targeted.  One copy of the visit code per pass.
// Round 5: the entries reach the arithmetic through the SCALAR unit.  A batch of eight pairs is ONE s_load_dwordx16 into an
// aligned SGPR tuple -- entry j = the SGPR pair {w, pix}: v_pk_fma_f32 reads the weight and v_lshl_add_u32 the pixel straight from
// it -- so a pair costs 3 vector instructions (address, two packed FMAs) where rounds 1-4 spent 5 (two v_readlane in front).  The
// structure-preserving ablation had priced the two v_readlane at 5 % of the C2 step, 10 % together with fewer flushes
// (profiles/r5_combined_ablation.txt); round 1 had found scalar-fed entries latency-bound (8.9 ms).  What makes them stream
// (tools/ubench_sload.hip): (i) the visit's lines are pulled into L2 a whole visit ahead by a one-dword-per-line vector load, so
// the scalar load is an L2 hit; (ii) a rolling double buffer of two fixed SGPR tuples, s[68:83] and s[84:99], which the
// compiler never allocates (amdgpu_num_sgpr(76) keeps it inside s0..s67; a CPU test scans the assembly): the load of batch
// b + 1 -- at the end of a visit: the NEXT visit's first batch -- goes out at the top of batch b and has landed when batch b's last
// FMA has waited for lgkmcnt(0); nothing of the stream is ever in flight outside a batch's asm block.  The blend pads both
// half-tile lists of a record to eight entries with {0, kPadPix} (gwbp_dev.h: Header), so a batch needs no remainder handling.
// The batch loop is a run-time loop over two copies of the block (tuple A -> B, B -> A): 6 KB of code where the unrolled
// v_readlane form had 20.
// Every VMEM instruction of a visit is unconditional WITHIN a pass, so the counted s_waitcnt in front of a visit's carry dwords
// is exact: a visit issues its loads (1 warm-up in the top pass, 1 + 4 carry dwords in the bottom pass) and 4 flush operations
// (atomics, or the 4 stores that park a spanning record).  The denominator d is not accumulated here: the blend adds
// every record's weight sum to d itself (gwbp_blend_weights_d), or k_accum_d (scatter.hip) does from the headers.

#include <stdlib.h>

#include <type_traits>

#include "gwbp_dev.h"

namespace gwbp {

#ifdef GWBP_STAMPS
// In-kernel stamps (make PROFILE=1 only, tools/stamp_scatter.py): shader cycles summed over the waves of all workgroups,
// [0] visit table + slab commit incl. the barrier behind them, [1] visit loop, [2] wait for the next item's facts, [3] wait at
// the barrier in front of a round (the other waves' last visits + own prefetch issue), [4] rounds, [5] visits.
__device__ unsigned long long g_wide_prof[8];
#define GWBP_STAMP(x) const unsigned long long x = __builtin_amdgcn_s_memtime()
#else
#define GWBP_STAMP(x)
#endif

namespace {

constexpr int kWide = 256;             // channels per chunk
constexpr int kHalfPix = kTilePix / 2; // pixels per slab
constexpr int kThreads = 1024;
#ifndef GWBP_BIL_UNROLL
#define GWBP_BIL_UNROLL 1
#endif
constexpr int kSlabFloats = kHalfPix * kWide; // 32768 floats = 128 KB
#ifndef GWBP_VISCAP
#define GWBP_VISCAP 1024
#endif
constexpr int kVisCap = GWBP_VISCAP;          // records per round = visit-table capacity (threads 0 .. kVisCap-1 take one record each)
constexpr u32 kTabOff = (u32)kSlabFloats * 4u;         // visit table: kVisCap x 16 B behind the slab
constexpr u32 kCtlOff = kTabOff + (u32)kVisCap * 16u;  // control words: [0,1] claim counters and [2,3] visit counts by round
                                                       // parity, [4,5] item slots
constexpr size_t kLdsBytes = kCtlOff + 32;
#ifndef GWBP_TAIL
#define GWBP_TAIL 0
#endif
constexpr int kTail = GWBP_TAIL; // visits held back for the end of a round (0 = the table is in list order).  Measured with 16 / 32 /
                                 // 48 / 96: the kernel beside the front stage 6-8 % faster (the waves run out of work together), the
                                 // front stage beside it 10 % slower, the step 3.62 -> 3.96 ms -- at a LARGER register allocation; neutral at the
                                 // same one (profiles/r4_wide_ablation.txt, section D): off
constexpr int kShortN = 16;      // ... chosen among the visits of at most this many entries
static_assert(((size_t)kPadPix - (size_t)kHalfPix) * 1024u >= 160u * 1024u, "a padding entry's slab row must lie beyond any LDS allocation in both passes");

// Structure-preserving ablations (make PROFILE=1 ABL=<bits> via tools/build_ablations.sh; results INVALID by design, never in
// the product library): compile-time, so every build keeps the visit's VMEM count and hence its counted waits.
//   1  every flush = plain stores into ONE row of the workgroup's carry slice (no memory-side atomic cost)
//   2  no LDS reads / FMAs
//   4  no slab staging
//   8  with 1: only 3 of 8 visits flush that way (what merging 2 x 2 tile blocks would save)
//  16  parks and resumes all use carry row 0 (no carry traffic beyond L2)
//  32  (rounds 4-5, v_readlane form of the batch loop only: one v_readlane pair per batch of eight pairs -- what entries fed
//      through the scalar unit would save.  It priced the rewrite above, profiles/r5_combined_ablation.txt; no effect any more)
//  64  every flush = plain stores into the record's OWN row of F (the write traffic of a store-then-sum scatter whose partial
//      rows are summed by a later pass: tools/probe_store_then_sum.py)
#if defined(GWBP_PROFILE) && defined(GWBP_ABL)
constexpr int kAbl = GWBP_ABL;
#else
constexpr int kAbl = 0;
#endif

struct Visit { // wave-uniform description of one (record, half) visit
    u32 gid;
    u32 off;  // first entry
    u32 n;    // entries (1..128)
    u32 span; // nonzero: the record has entries in both halves and owns carry row `row`
    u32 row;  // the record's carry row: its rank among its tile's records that have entries in both halves
};

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
struct Land { // what a visit prefetches a whole visit ahead: the L2 warm-up of its entries (value unused) and its carry dwords
    u32 warm;
    float c[4];
};
// All VMEM of the visit loop: address = SGPR pair + per-lane 32-bit offset + immediate.  Tied operands ("+v"): the load
// must land in the registers the struct lives in (scatter_full.hip explains what happens otherwise).
// The L2 warm-up of a visit's entries: lane k touches byte 128 k of the visit's run, `lanes` = the lines the run can reach.  The
// loaded dword is never used -- the entries are consumed through scalar loads, which must find their lines in L2 -- but the
// instruction is issued for every visit (the counted waits stand) and its register stays reserved until it has landed.
__device__ __forceinline__ void load_warm(u32 &dst, u32 lane4, u64 base, u64 lanes)
{
    u64 saved;
    u32 voff;
    asm volatile("v_lshlrev_b32 %2, 5, %3\n\t"
                 "s_mov_b64 %1, exec\n\t"
                 "s_mov_b64 exec, %5\n\t"
                 "global_load_dword %0, %2, %4\n\t"
                 "s_mov_b64 exec, %1"
                 : "+v"(dst), "=&s"(saved), "=&v"(voff)
                 : "v"(lane4), "s"(base), "s"(lanes)
                 : "memory");
}
// sc1: served by L2, never by this CU's L1 (the row was written by another wave of this workgroup one pass earlier)
template <int OFF>
__device__ __forceinline__ void load_c(float &dst, u32 voff, u64 base)
{
    asm volatile("global_load_dword %0, %1, %2 offset:%3 sc1" : "+v"(dst) : "v"(voff), "s"(base), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void store_c(u32 voff, float v, u64 base)
{
    asm volatile("global_store_dword %0, %1, %2 offset:%3" ::"v"(voff), "v"(v), "s"(base), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void atomic_f(u32 voff, float v, u64 base)
{
    asm volatile("global_atomic_add_f32 %0, %1, %2 offset:%3" ::"v"(voff), "v"(v), "s"(base), "n"(OFF) : "memory");
}
// a wave-uniform dword through the vector path (every lane reads the same address): unlike a scalar load it does not force
// the loop's lgkmcnt waits to zero, and unlike a compiler-issued load it is not waited for behind the visit loop's atomics
__device__ __forceinline__ void load_u(u32 &dst, u32 voff, u64 base)
{
    asm volatile("global_load_dword %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(base) : "memory");
}
// Wait for the two asm-issued loads of the next item's facts: everything but the wave's last flush (`few` != 0: the wave ran
// visits and has at most that flush in flight) or everything.  ONE statement with a scalar branch inside: written as two
// statements in an if / else, the two tied outputs met in a phi, and a build with in-kernel stamps (different register
// allocation) resolved it with v_mov copies of the landing registers IN FRONT of the wait of one branch -- copies of
// registers whose loads had not landed (found in round 5: waves without visits then read stale record counts; the product
// build happened to place the copies behind the wait).
template <int N>
__device__ __forceinline__ void wait_info(u32 few, u32 &a, u32 &b)
{
    asm volatile("s_cmp_lg_u32 %2, 0\n\t"
                 "s_cbranch_scc1 1f\n\t"
                 "s_waitcnt vmcnt(0)\n\t"
                 "s_branch 2f\n"
                 "1:\n\t"
                 "s_waitcnt vmcnt(%3)\n"
                 "2:"
                 : "+v"(a), "+v"(b)
                 : "s"(few), "n"(N)
                 : "scc", "memory");
}
template <int N>
__device__ __forceinline__ void wait_land(Land &x)
{
    asm volatile("s_waitcnt vmcnt(%5)" : "+v"(x.warm), "+v"(x.c[0]), "+v"(x.c[1]), "+v"(x.c[2]), "+v"(x.c[3]) : "n"(N) : "memory");
}
// ---- the scalar entry stream -------------------------------------------------------------------------------------------
// Two fixed SGPR tuples (the kernel is compiled with amdgpu_num_sgpr(76): hipcc stays inside s0..s67) and the batch buffer
// v[72:103] (eight float4; named as clobbers, so hipcc keeps nothing alive there across a batch and is free to use the
// registers between batches -- the slab staging lands in them).
#define GWBP_SA 68
#define GWBP_SB 84
#define GWBP_TUPLES                                                                                                   \
    "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85",  \
        "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99"
#define GWBP_FREGS                                                                                                    \
    "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89",  \
        "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103"
#define GWBP_STR2(x) #x
#define GWBP_STR(x) GWBP_STR2(x)
// entry j of tuple T: weight = s[T + 2j] (read as the aligned pair s[T + 2j : T + 2j + 1] with op_sel_hi 0), pixel = s[T + 2j + 1]
#define GWBP_RD(T, j, v0)                                                                                             \
    "v_lshl_add_u32 %[t], s[" GWBP_STR(T) "+" #j "*2+1], 10, %[rb]\n\tds_read_b128 v[" #v0 ":" #v0 "+3], %[t]\n\t"
#define GWBP_FM(T, j, v0, cnt)                                                                                        \
    "s_waitcnt lgkmcnt(" #cnt ")\n\t"                                                                                  \
    "v_pk_fma_f32 %[lo], s[" GWBP_STR(T) "+" #j "*2:" GWBP_STR(T) "+" #j "*2+1], v[" #v0 ":" #v0 "+1], %[lo] op_sel_hi:[0,1,1]\n\t"   \
    "v_pk_fma_f32 %[hi], s[" GWBP_STR(T) "+" #j "*2:" GWBP_STR(T) "+" #j "*2+1], v[" #v0 "+2:" #v0 "+3], %[hi] op_sel_hi:[0,1,1]\n\t"
// One batch: eight pairs from tuple CUR while the load of the next batch flies into tuple NXT.  lgkmcnt: the scalar load may
// return at any time, LDS reads return in order, so "at most 7 - k outstanding" still proves read k complete whatever else
// (the scalar load, an older claim or table read) is in flight; the last wait is lgkmcnt(0): the next tuple has landed too.
#define GWBP_BATCH_ASM(CUR, NXT)                                                                                      \
    "s_load_dwordx16 s[" GWBP_STR(NXT) ":" GWBP_STR(NXT) "+15], %[nx], 0x0\n\t"                                         \
    GWBP_RD(CUR, 0, 72) GWBP_RD(CUR, 1, 76) GWBP_RD(CUR, 2, 80) GWBP_RD(CUR, 3, 84)                                     \
    GWBP_RD(CUR, 4, 88) GWBP_RD(CUR, 5, 92) GWBP_RD(CUR, 6, 96) GWBP_RD(CUR, 7, 100)                                    \
    GWBP_FM(CUR, 0, 72, 7) GWBP_FM(CUR, 1, 76, 6) GWBP_FM(CUR, 2, 80, 5) GWBP_FM(CUR, 3, 84, 4)                         \
    GWBP_FM(CUR, 4, 88, 3) GWBP_FM(CUR, 5, 92, 2) GWBP_FM(CUR, 6, 96, 1) GWBP_FM(CUR, 7, 100, 0)
// par = 0: the batch sits in tuple A and the next one goes to B; par = 1: the other way round.  ONE asm statement with a scalar
// branch inside: as two statements in an if / else hipcc gave each its own copy of the accumulators and moved them there and
// back around every batch (4 v_mov_b64 per batch).
__device__ __forceinline__ void batch_run(u32 par, u64 next, u32 row_base, f32x2_t &lo, f32x2_t &hi)
{
    u32 t;
    asm volatile("s_cmp_lg_u32 %[par], 0\n\t"
                 "s_cbranch_scc1 1f\n\t"
                 GWBP_BATCH_ASM(GWBP_SA, GWBP_SB)
                 "s_branch 2f\n"
                 "1:\n\t"
                 GWBP_BATCH_ASM(GWBP_SB, GWBP_SA)
                 "2:"
                 : [lo] "+v"(lo), [hi] "+v"(hi), [t] "=&v"(t)
                 : [nx] "s"(next), [rb] "v"(row_base), [par] "s"(par)
                 : GWBP_TUPLES, GWBP_FREGS, "scc", "memory");
}
// the first batch of a pass: nothing to overlap it with (one exposed L2 round trip per wave and pass)
template <int P>
__device__ __forceinline__ void batch_prime(u64 first)
{
    if constexpr (P == 0)
        asm volatile("s_load_dwordx16 s[" GWBP_STR(GWBP_SA) ":" GWBP_STR(GWBP_SA) "+15], %0, 0x0\n\ts_waitcnt lgkmcnt(0)" ::"s"(first)
                     : GWBP_TUPLES, "memory");
    else
        asm volatile("s_load_dwordx16 s[" GWBP_STR(GWBP_SB) ":" GWBP_STR(GWBP_SB) "+15], %0, 0x0\n\ts_waitcnt lgkmcnt(0)" ::"s"(first)
                     : GWBP_TUPLES, "memory");
}
// one lane, one LDS atomic, NOT waited for (the wave-aggregation sequence hipcc wraps around a single-lane atomicAdd is ~8
// instructions and waits at once)
// (The lane mask is set INSIDE the statement: as `if (lane == 0) asm(...)` the tied output met its old value in a phi behind
// the branch, which a differently allocated build may resolve with a copy of the register the atomic has not returned into yet
// -- the hazard class of wait_info above.)
__device__ __forceinline__ void claim_issue(u32 &dst, u32 addr, int lane)
{
    (void)lane;
    const u32 one = 1u;
    u64 saved;
    asm volatile("s_mov_b64 %1, exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "ds_add_rtn_u32 %0, %2, %3\n\t"
                 "s_mov_b64 exec, %1"
                 : "+v"(dst), "=&s"(saved)
                 : "v"(addr), "v"(one)
                 : "memory");
}
typedef u32 u32x3_t __attribute__((ext_vector_type(3))); // a native vector: HIP's uint3 is a struct, not an asm operand
// (12 of a table entry's 16 bytes: three registers in flight per wave instead of four -- see the note on lane8 below)
__device__ __forceinline__ void table_issue(u32x3_t &dst, u32 addr)
{
    asm volatile("ds_read_b96 %0, %1" : "+v"(dst) : "v"(addr) : "memory");
}
__device__ __forceinline__ void wait_lds(u32 &a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)::"memory"); }
__device__ __forceinline__ void wait_lds(u32x3_t &a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)::"memory"); }

// An "s" asm operand must really be scalar: hipcc does not insert the v_readfirstlane itself.  In the product build every base
// below is provably wave-uniform (a compile error otherwise, never a silent miscompile); the no-compute ablation keeps them in
// vector registers.
__device__ __forceinline__ u64 sbase(u64 x)
{
    // hipcc does not always prove these bases wave-uniform (it depends on the shape of the surrounding loops), and a vector
    // register in an "s" operand is a compile error at best: force the issue.  v_readfirstlane -> VMEM address operand needs 5
    // wait states and the hazard recogniser does not look inside inline asm, hence the s_nop (tools/check_asm_hazards.py and
    // a CPU test scan the generated code for the pattern).
    x = uniform64(x);
    asm volatile("s_nop 4" : "+s"(x));
    return x;
}

constexpr int kFlush = 4; // VMEM flush operations per visit

template <bool BILINEAR> // a bilinear low-resolution map (gwbp_scatter_bilinear) has a staging loop of its own: own instantiation
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_num_sgpr(76))) void k_scatter_wide(
    ViewDev V, int n_chunks, const u32 *__restrict__ tile_offsets, const u32 *__restrict__ hdr_count,
    const Header *__restrict__ headers, const WPair *__restrict__ wpool, FeatMap M, int D, float scale_f,
    float *__restrict__ F, u32 *__restrict__ queues, float *__restrict__ carry_all, Counters *__restrict__ ctr)
{
    // gwbp_scatter's contract for D % 256 == 0: the view was blended without GWBP_FLAG_NARROW_SCATTER (its headers hold the
    // weight sums that k_accum_d / the blend turn into d).  Refuse otherwise (F untouched, overflow bit 2 raised).
    if (uniform(ctr->blend_kind) != kBlendHalves) {
        if (blockIdx.x == 0 && threadIdx.x == 0)
            atomicOr(&ctr->overflow, kOverflowMismatch);
        return;
    }
    // REGISTER BUDGET, deliberately padded.  The kernel needs ~72 vector registers of its own; the batch buffer v[72:103] of the
    // scalar-fed loop (and, for builds without it, naming v103 here) makes the hardware allocate 104 per lane to each of its
    // four waves per SIMD, which leaves 96: ONE 64-register wave of the front-stage kernels (k_blend, k_radix_scatter) per
    // SIMD beside it.  Measured on one box (C2, three workspaces), allocation -> ms per view:
    // 80 (three front waves per SIMD) 3.74, 88 / 96 (two) 3.72, 104 (one) 3.64, 112 (one) 3.67, 120 (none: the front's
    // kernels wait for the scatter kernel to END) 3.97.  More front waves beside the kernel cost it more than they gain.
    // tests/test_capi_cpu.py pins the allocation.
    asm volatile("" ::: "v103");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    u32 *ctl = reinterpret_cast<u32 *>(lds) + kCtlOff / 4;
    uint4 *table = reinterpret_cast<uint4 *>(lds) + kTabOff / 16;

    // persistent workgroups, per-XCD-class queues: as k_scatter_full
    const u32 xcls = blockIdx.x & 7u;
    const int n_tiles = V.tile_w * V.tile_h;
    const u32 n_items = (u32)((n_tiles - (int)xcls + 7) / 8) * (u32)n_chunks;
    u32 *queue = queues + xcls * 16;
    const int lane = threadIdx.x & 63;
    const u32 lane4 = (u32)lane * 4u;
    const u64 carry = reinterpret_cast<u64>(carry_all + (size_t)blockIdx.x * kCarryRows * kWide); // this workgroup's slice
    const float *feats = M.p;
    if (threadIdx.x == 0) {
        ctl[0] = ctl[1] = ctl[2] = ctl[3] = ctl[6] = ctl[7] = 0;
        ctl[4] = atomicAdd(queue, 1u);
    }
    __syncthreads();
#ifdef GWBP_STAMPS
    unsigned long long prof_acc[6] = {0, 0, 0, 0, 0, 0};
#endif
    // ---- the round in progress (everything here is wave-uniform) ----------------------------------------------------------
    // A round = one pass of an item over up to kVisCap records: (item, phase, rbase).  The loads a round starts with -- its
    // records' headers and, in the first round of a pass, the half-tile slab -- are issued at the END of the previous round,
    // by every wave as it runs out of visits, so that they fly under the other waves' last visits and the barrier.
    u32 k = 0;                       // items this workgroup has started
    u32 item = uniform(ctl[4]);
    int phase = 0;
    u32 rbase = 0;
    int tile = 0, tx = 0, ty = 0, c0 = 0;
    u32 n_rec = 0;
    const Header *hbase = headers;
    u64 f_chunk = 0;
    auto set_item = [&](u32 it, u32 nrec, u32 toff) __attribute__((always_inline)) {
        const int chunk = (int)(it % (u32)n_chunks);
        tile = (int)((it / (u32)n_chunks) * 8u + xcls);
        tx = tile % V.tile_w, ty = tile / V.tile_w;
        c0 = chunk * kWide;
        n_rec = nrec;
        hbase = headers + toff;
        f_chunk = reinterpret_cast<u64>(F + c0);
    };
    auto tile_of = [&](u32 it) __attribute__((always_inline)) -> u32 { return (it / (u32)n_chunks) * 8u + xcls; };
    // registers that carry a round's loads across the end-of-round barrier
    uint4 h0 = make_uint4(0u, 0u, 0u, 0u), h1 = make_uint4(0u, 0u, 0u, 0u);
    constexpr int kUnits = kHalfPix / (kThreads / 64); // 8 pixels per wave
    // ONE 32-register buffer serves as the landing area of the next slab's loads (from the end of a round to the commit at the
    // top of the next) AND as the batch buffer of the visit loop in between: declared separately, hipcc gave them 32 registers
    // each (110 VGPRs -> 112 allocated, ONE 64-register front-stage wave per SIMD beside the kernel; 113+ -> 120: none at all,
    // and k_project / k_radix_scatter of the next views then waited for the scatter kernel to END -- which is what four
    // 'improvements' of the kernel ran into this round).
    static_assert(kUnits == 8, "the slab landing area and the batch buffer are the same eight float4");
    f32x4_t fbuf[kUnits];
    int bil_y0 = 0;     // BILINEAR, lanes 0..7: first texel row and row weight of the next slab's pixel row `lane`, requested with
    float bil_ly = 0.f; // the round's records (a round ahead of the staging loop that blends the texels)
    u32 next_claim = 0; // thread 0: the item after this one, claimed when this one was started
    const int wv = (int)uniform(threadIdx.x >> 6);
    auto stage_issue = [&]() __attribute__((always_inline)) {
        // (a) this round's records
        const u32 rec = rbase + threadIdx.x;
        h0 = make_uint4(0u, 0u, 0u, 0u), h1 = make_uint4(0u, 0u, 0u, 0u);
        if (threadIdx.x < (u32)kVisCap && rec < n_rec) {
            h0 = reinterpret_cast<const uint4 *>(hbase + rec)[0]; // gid, woff[0..2]
            h1 = reinterpret_cast<const uint4 *>(hbase + rec)[1]; // woff[3], counts, wsum, carry row
        }
        // (b) 128 px x 256 ch (first round of a pass only; a second round reuses the slab): wave v stages tile column v of the
        // eight tile rows of this half, one pixel = 4 coalesced dword loads (one per 64-channel group) + one ds_write_b128 per
        // lane.  Pixel addresses are wave-uniform (scalar registers; with index maps scalar loads), the per-lane part is lane * 4.
        if (!BILINEAR && !(kAbl & 4) && n_rec != 0 && rbase == 0) {
            // pixels past the image edge are never referenced by an entry: load a clamped (valid) address
            const int ix = min(tx * kTile + wv, V.W - 1);
            const int64_t xoff = (int64_t)(M.xmap ? M.xmap[ix] : ix) * M.fs_x + c0;
            const float *rows[kUnits];
#pragma unroll
            for (int u = 0; u < kUnits; ++u) { // (row offsets first: with an index map they are loads themselves)
                const int iy = min(ty * kTile + phase * (kTile / 2) + u, V.H - 1);
                rows[u] = feats + ((int64_t)(M.ymap ? M.ymap[iy] : iy) * M.fs_y + xoff);
            }
#pragma unroll
            for (int u = 0; u < kUnits; ++u) {
                const float *src = rows[u] + lane;
                fbuf[u] = f32x4_t{__builtin_nontemporal_load(src), __builtin_nontemporal_load(src + 64), __builtin_nontemporal_load(src + 128),
                                   __builtin_nontemporal_load(src + 192)};
            }
        } else {
            if (BILINEAR && !(kAbl & 4) && n_rec != 0 && rbase == 0 && lane <= kTile / 2) {
                // the staging loop below walks the slab's eight pixel rows; their texel row and weight are table lookups that
                // used to sit, as a dependent round trip, in front of every iteration's sixteen texel loads.  Lane 8: the same
                // for the wave's pixel column (wave v stages column v of the tile)
                const int iy = min(ty * kTile + phase * (kTile / 2) + lane, V.H - 1), ix = min(tx * kTile + wv, V.W - 1);
                const int32_t *imap = lane < kTile / 2 ? M.ymap + iy : M.xmap + ix;
                const float *lmap = lane < kTile / 2 ? M.ly + iy : M.lx + ix;
                bil_y0 = *imap, bil_ly = *lmap;
            }
            // No slab for the next round: say so.  Without this the buffer's OLD contents count as live from one round's end to
            // the next (a conditional redefinition), i.e. right through the visit loop, and its 32 registers cannot double
            // as the loop's batch buffer.
#pragma unroll
            for (int u = 0; u < kUnits; ++u)
                asm volatile("" : "=v"(fbuf[u]));
        }
    };
    if (item < n_items) {
    set_item(item, uniform(hdr_count[tile_of(item)]), uniform(tile_offsets[tile_of(item)]));
    if (threadIdx.x == 0)
        next_claim = atomicAdd(queue, 1u);
    stage_issue();
    // The top of a round -- wait for the previous round's visits, build the visit table, commit the slab -- is issued right
    // BEHIND the loads it consumes (end of the previous round), inside the same loop iteration: with the loads at the end of one
    // iteration and their consumers at the top of the next, the 32 slab registers are loop-carried values that hipcc will not
    // let share registers with the visit loop's batch buffer.
    u32 round = 0, par = 0; // parity selects the claim counter / visit count in use
    auto setup_round = [&]() __attribute__((always_inline)) {
    GWBP_STAMP(ts0);
    __syncthreads(); // the previous round's visits are over: slab, table and the other parity's counters are free
    GWBP_STAMP(tsa);
    par = round & 1u;
    if (threadIdx.x == 0) {
        ctl[par ^ 1u] = 0, ctl[2u + (par ^ 1u)] = 0, ctl[6u + (par ^ 1u)] = 0; // the next round's counters
        if (phase == 0 && rbase == 0)
            ctl[4 + ((k + 1u) & 1u)] = next_claim; // (claimed a round ago: its round trip is over)
    }
    const u32 rec = rbase + threadIdx.x;
    const bool has = threadIdx.x < (u32)kVisCap && rec < n_rec;
    const bool stage = !(kAbl & 4) && n_rec != 0 && rbase == 0;
    if (BILINEAR && stage) {
        // Bilinear low-resolution map (backproject.py:110-112 folded in): every slab value is the blend of four texels
        // (L2 / Infinity-Cache resident: the 480 x 480 x 512 map of the lseg script is 472 MB), in ATen's association.
        // One (pixel, lane) unit per round: 16 dword loads in flight per thread.
        constexpr int kAllB = kHalfPix * 64;
        constexpr int kUnitsB = (kAllB + kThreads - 1) / kThreads;
#pragma unroll GWBP_BIL_UNROLL
        for (int u = 0; u < kUnitsB; ++u) {
            const int idx = min(u * kThreads + (int)threadIdx.x, kAllB - 1);
            static_assert(kThreads == 16 * 64 && kHalfPix == 8 * kTile, "iteration u of the staging loop = pixel row u of the slab");
            // (indices and weights requested by stage_issue a round ago; pix & 15 == wv, pix >> 4 == phase * 8 + u)
            const int y0 = __builtin_amdgcn_readlane(bil_y0, u), x0 = __builtin_amdgcn_readlane(bil_y0, kTile / 2);
            const int y1 = min(y0 + 1, M.lr_h - 1), x1 = min(x0 + 1, M.lr_w - 1);
            const float h1w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bil_ly), u));
            const float w1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bil_ly), kTile / 2));
            const float h0w = 1.0f - h1w, w0 = 1.0f - w1;
            const float *b0 = feats + c0 + lane;
            const float *pa = b0 + y0 * M.fs_y + x0 * M.fs_x, *pb = b0 + y0 * M.fs_y + x1 * M.fs_x;
            const float *pc = b0 + y1 * M.fs_y + x0 * M.fs_x, *pd = b0 + y1 * M.fs_y + x1 * M.fs_x;
            float r[4];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
                r[k4] = h0w * (w0 * pa[64 * k4] + w1 * pb[64 * k4]) + h1w * (w0 * pc[64 * k4] + w1 * pd[64 * k4]);
            if (kAllB % kThreads == 0 || u * kThreads + (int)threadIdx.x < kAllB)
                *reinterpret_cast<float4 *>(lds + (idx >> 6) * kWide + 4 * lane) = make_float4(r[0], r[1], r[2], r[3]);
        }
    }
    const bool stage_plain = !BILINEAR && stage;
    // (a') the visit table: records with entries in this half, compacted wave by wave, in list order.  (Handing the visits out
    // longest first -- eight length classes, one more barrier -- was measured: the barrier wait in front of a round fell from
    // 12 % to 8 % of the wave time, but a visit took 9 % longer (the long, throughput-bound visits then all run together and
    // so do the short, latency-bound ones), and the pipelined step went from 3.74 to 4.04 ms.  Reverted.)
    {
        const u32 cnt = h1.y;
        const u32 ct = (cnt & 0xFFu) + ((cnt >> 8) & 0xFFu), cb = ((cnt >> 16) & 0xFFu) + (cnt >> 24);
        const u32 n = phase ? cb : ct;
        // a record in both halves owns carry row h1.w (its rank among the tile's spanning records, from the blend: the rows a
        // workgroup touches are few and the same for every item, i.e. hot in L2; indexed by the record itself they were twice
        // as many); a tile with more than kCarryRows of them flushes the rest per half
        const u32 crow = h1.w;
        const u32 span = (ct != 0 && cb != 0 && crow < (u32)kCarryRows) ? 0x100u : 0u;
        const bool valid = has && n != 0;
        const u64 m = __ballot(valid);
        if (m != 0ull) { // wave-uniform
            // kTail > 0: the first kTail SHORT visits (<= kShortN entries) are parked at the END of the table, i.e. handed out
            // last -- the waves then run out of work within one short visit of each other (ctl[2 + par] counts the front part,
            // ctl[6 + par] the candidates for the tail)
            u64 mback = 0ull;
            if (kTail > 0) {
                const u64 ms = __ballot(valid && n <= (u32)kShortN);
                if (ms != 0ull) {
                    u32 sbase_ = 0;
                    if (lane == 0)
                        sbase_ = atomicAdd(&ctl[6u + par], (u32)__popcll(ms));
                    sbase_ = uniform(sbase_);
                    // (signed on purpose: written as `kTail > sbase_ ? min(kTail - sbase_, ..) : 0` in unsigned arithmetic, hipcc 7.2
                    // emitted s_sub_i32 + s_min_u32 without the saturation -- waves that arrived after the tail was full parked
                    // ALL their short visits beyond its end, where nobody claims them)
                    const int room = max(kTail - (int)sbase_, 0);
                    const u32 take = (u32)min(room, (int)__popcll(ms));
                    const bool back = valid && n <= (u32)kShortN && mbcnt(ms) < take;
                    mback = __ballot(back);
                    if (back)
                        table[(u32)kVisCap - 1u - (sbase_ + mbcnt(ms))] = make_uint4(h0.x, phase ? h0.w : h0.y, n | span | (crow << 16), 0u);
                }
            }
            const u64 mf = m & ~mback;
            if (mf != 0ull) {
                u32 wbase = 0;
                if (lane == 0)
                    wbase = atomicAdd(&ctl[2u + par], (u32)__popcll(mf));
                wbase = uniform(wbase);
                if ((mf >> lane) & 1ull)
                    table[wbase + mbcnt(mf)] = make_uint4(h0.x, phase ? h0.w : h0.y, n | span | (crow << 16), 0u);
            }
        }
    }
    if (stage_plain) {
#pragma unroll
        for (int u = 0; u < kUnits; ++u) // slab row = pixel (tile row u of this half, column wv)
            *reinterpret_cast<f32x4_t *>(lds + (u * kTile + wv) * kWide + 4 * lane) = fbuf[u];
    }
    __syncthreads();
#ifdef GWBP_STAMPS
    {
        GWBP_STAMP(tsb);
        prof_acc[3] += tsa - ts0, prof_acc[0] += tsb - tsa;
    }
#endif
    };
    setup_round();
#pragma unroll 1
    for (;;) {
    GWBP_STAMP(ts1);
#ifdef GWBP_STAMPS
    u32 n_vis_prof = 0;
#endif
    const u32 n_front = uniform(ctl[2u + par]);
    const u32 nv = n_front + (kTail > 0 ? min(uniform(ctl[6u + par]), (u32)kTail) : 0u);
    // claim index -> table slot (the held-back visits sit at the end of the table, last slot first)
    auto slot_of = [&](u32 h) __attribute__((always_inline)) -> u32 {
        return (kTail > 0 && h >= n_front) ? (u32)kVisCap - 1u - (h - n_front) : h;
    };
    const u32 claim_addr = kCtlOff + 4u * par;
    const u64 f_base = uniform64(f_chunk); // (a loop-carried value: hipcc does not prove it scalar, and an "s" operand must be)
    // The next item's tile facts (record count, first header), fetched under this pass's visits: two loads of the asm-counted
    // kind, older than every visit's operations, so the counted waits never see them.
    // (Issued in every round, unconditionally: a conditional asm load makes hipcc copy its destination behind the branch --
    // before the data has landed.)
    u32 nx_nrec = 0, nx_toff = 0;
    const u32 nx_item = uniform(ctl[4 + ((k + 1u) & 1u)]);
    {
        const u32 t = tile_of(min(nx_item, n_items - 1u));
        load_u(nx_nrec, 0u, reinterpret_cast<u64>(hdr_count + t));
        load_u(nx_toff, 0u, reinterpret_cast<u64>(tile_offsets + t));
    }

    // dynamic LDS starts at address 0 (no static __shared__ in this kernel): slab row r lives at byte r * 1024
    const u32 row_base = (u32)(lane * 16) - (phase ? (u32)(kHalfPix << 10) : 0u);

    auto decode = [&](const uint4 &t) __attribute__((always_inline)) -> Visit {
        Visit r;
        r.gid = uniform(t.x);
        r.off = uniform(t.y);
        const u32 ns = uniform(t.z); // entries | spans both halves << 8 | carry row << 16
        r.n = ns & 0xFFu;
        r.span = ns & 0x100u;
        r.row = ns >> 16;
        return r;
    };
    const u64 wp_base = uniform64(reinterpret_cast<u64>(wpool));
    auto entries_of = [&](const Visit &R) __attribute__((always_inline)) -> u64 { return wp_base + ((u64)R.off << 3); };
    // exactly 1 (top pass) / 5 (bottom pass) VMEM loads: the top pass never resumes a record
    auto prefetch = [&](const Visit &R, Land &x, auto bottom) __attribute__((always_inline)) {
        const u64 eb = sbase(entries_of(R));
        // the run is 8 n bytes from a 64-byte boundary: it can reach into ceil((8 n + 64) / 128) lines of 128 B (1..9)
        const u32 nl = (R.n * 8u + 64u + 127u) >> 7;
        load_warm(x.warm, lane4, eb, (1ull << nl) - 1ull);
        if constexpr (decltype(bottom)::value) {
            // carry dwords of this lane (non-spanning records: row 0, value ignored -- the count must stay exact)
            const u64 cr = sbase(carry + ((u64)((R.span && !(kAbl & 16)) ? R.row : 0u) << 10));
            load_c<0>(x.c[0], lane4, cr);
            load_c<256>(x.c[1], lane4, cr);
            load_c<512>(x.c[2], lane4, cr);
            load_c<768>(x.c[3], lane4, cr);
        }
    };

    f32x2_t acc_lo, acc_hi; // channels {l, l + 64} and {l + 128, l + 192} of the record's sums

    // Visit pipeline.  `cur` is processed, the carry dwords of `nxt` (and the L2 warm-up of its entries) are in flight into the
    // landing buffer (issued at the top of this visit), the descriptor of the visit after `nxt` is read from the table during
    // this visit, its index claimed at the top of it.  The entry stream runs one batch ahead: when a visit starts, its first
    // batch sits in the SGPR tuple of parity `par_s`; its last batch fetches the first batch of `nxt`.
    auto visits = [&](auto bottom) __attribute__((always_inline)) -> bool {
        Land L = {0u, {0.f, 0.f, 0.f, 0.f}};
        u32 cl = 0;
        u32x3_t tn = {0u, 0u, 0u};
        claim_issue(cl, claim_addr, lane);
        wait_lds(cl);
        const u32 h_cur = uniform(cl);
        if (h_cur < nv) {
            claim_issue(cl, claim_addr, lane);
            Visit cur = decode(table[slot_of(h_cur)]);
            prefetch(cur, L, bottom);
            wait_lds(cl);
            u32 h_nxt = uniform(cl);
            Visit nxt = decode(table[slot_of(min(h_nxt, nv - 1u))]);
            wait_land<0>(L); // (one exposed L2 round trip per wave and pass; the steady-state wait below then holds from the start)
            u32 par_s = 0; // which tuple holds the batch about to run
            if (!(kAbl & 2))
                batch_prime<0>(entries_of(cur));
            for (;;) {
                const bool vnxt = h_nxt < nv;
                claim_issue(cl, claim_addr, lane); // the visit after nxt
                // the landing buffer holds cur's data once everything older than the previous visit's flush has landed
                wait_land<kFlush>(L);
#ifdef GWBP_STAMPS
                ++n_vis_prof;
#endif
                const bool resume = decltype(bottom)::value && cur.span;
                acc_lo = resume ? f32x2_t{L.c[0], L.c[1]} : f32x2_t{0.f, 0.f};
                acc_hi = resume ? f32x2_t{L.c[2], L.c[3]} : f32x2_t{0.f, 0.f};
                // The landing buffer's old contents must be DEAD before the prefetch below re-targets it: otherwise hipcc gives
                // the loads fresh registers and reconciles the names with v_mov copies on the loop's back edge -- copies of
                // registers whose data has not arrived yet (found as a wide-vs-narrow mismatch at C2 size only).  This empty
                // volatile asm pins the selects above in front of the (volatile) loads.
                asm volatile("" : "+v"(acc_lo), "+v"(acc_hi));
                // unconditional (nxt is a valid record even when its claim came too late: harmless loads, drained after the
                // loop); nothing but this visit's flush follows before the next visit's wait
                prefetch(nxt, L, bottom);
                u32 h_n2 = nv;
                if (!(kAbl & 2)) {
                    // ceil(n / 8) batches; the run behind the last one is the next visit's first batch (no next visit: this
                    // visit's own first batch once more -- a valid address, never consumed)
                    const u64 e_cur = entries_of(cur), e_nxt = entries_of(vnxt ? nxt : cur);
                    const u32 nb = (cur.n + 7u) >> 3;
                    // the first batch stands outside the loop so that the descriptor read behind it is issued unconditionally
                    // (inside `if (b == 0)` its tied output would meet the loop-carried value in a phi: the hazard class of
                    // wait_info above)
                    batch_run(par_s, nb > 1u ? e_cur + 64u : e_nxt, row_base, acc_lo, acc_hi);
                    par_s ^= 1u;
                    // the claim has returned with the first batch (its last FMA waited for lgkmcnt(0)): read its descriptor, which
                    // lands under the second batch -- or is waited for below
                    wait_lds(cl);
                    h_n2 = uniform(cl);
                    table_issue(tn, kTabOff + 16u * slot_of(min(h_n2, nv - 1u)));
#pragma unroll 1
                    for (u32 b = 1; b < nb; ++b) {
                        const u64 next = (b + 1u < nb) ? e_cur + (u64)((b + 1u) << 6) : e_nxt;
                        batch_run(par_s, next, row_base, acc_lo, acc_hi);
                        par_s ^= 1u;
                    }
                } else {
                    wait_lds(cl);
                    h_n2 = uniform(cl);
                    table_issue(tn, kTabOff + 16u * slot_of(min(h_n2, nv - 1u)));
                }
                // exactly kFlush VMEM operations
                if (!decltype(bottom)::value && cur.span) { // park the partial sums: plain stores, same shape as the atomics
                    const u64 cr = sbase(carry + ((u64)((kAbl & 16) ? 0u : cur.row) << 10));
                    store_c<0>(lane4, acc_lo.x, cr);
                    store_c<256>(lane4, acc_lo.y, cr);
                    store_c<512>(lane4, acc_hi.x, cr);
                    store_c<768>(lane4, acc_hi.y, cr);
                } else {
                    if (scale_f != 1.0f) // wave-uniform; the .sum() reduction of backproject.py:127 needs no scaling
                        acc_lo *= scale_f, acc_hi *= scale_f;
                    if (kAbl & 64) { // ablation: the record's partial row leaves with plain stores (same shape, same row)
                        const u64 fb = sbase(f_base + (u64)cur.gid * (u64)((u32)D * 4u));
                        store_c<0>(lane4, acc_lo.x, fb);
                        store_c<256>(lane4, acc_lo.y, fb);
                        store_c<512>(lane4, acc_hi.x, fb);
                        store_c<768>(lane4, acc_hi.y, fb);
                    } else if (!(kAbl & 1) || ((kAbl & 8) && (cur.gid & 7u) >= 3u)) {
                        const u64 fb = sbase(f_base + (u64)cur.gid * (u64)((u32)D * 4u));
                        atomic_f<0>(lane4, acc_lo.x, fb);
                        atomic_f<256>(lane4, acc_lo.y, fb);
                        atomic_f<512>(lane4, acc_hi.x, fb);
                        atomic_f<768>(lane4, acc_hi.y, fb);
                    } else { // ablation: same VMEM count, no memory-side cost
                        const u64 cr = sbase(carry + ((u64)(kCarryRows - 1) << 10));
                        store_c<0>(lane4, acc_lo.x, cr);
                        store_c<256>(lane4, acc_lo.y, cr);
                        store_c<512>(lane4, acc_hi.x, cr);
                        store_c<768>(lane4, acc_hi.y, cr);
                    }
                }
                wait_lds(tn); // (issued a batch ago unless the visit had only one)
                if (!vnxt)
                    break;
                cur = nxt;
                nxt = decode(make_uint4(tn.x, tn.y, tn.z, 0u));
                h_nxt = h_n2;
            }
            wait_land<kFlush>(L); // the last prefetch still targets the landing registers (the last flush may stay in flight)
            return true;
        }
        return false;
    };
    bool ran = false;
    if (nv != 0)
        ran = phase ? visits(std::true_type{}) : visits(std::false_type{});
    // No drain: a visit leaves at most its own flush in flight.  The carry rows parked in the top pass are in L2 before any
    // wave of the bottom pass loads them, because every wave waits for the header loads it issues BELOW (in-order vmcnt: all
    // its older stores are complete by then) before it reaches the barrier in front of the next round's visits; LDS reads are
    // complete (every visit waited for its own).
    GWBP_STAMP(ts2);
    // (a wave that ran visits has at most its last flush in flight; one that did not may have to wait for older atomics)
    wait_info<kFlush>(ran ? 1u : 0u, nx_nrec, nx_toff);
    GWBP_STAMP(ts3);
#ifdef GWBP_STAMPS
    prof_acc[1] += ts2 - ts1, prof_acc[2] += ts3 - ts2;
    prof_acc[4] += 1ull, prof_acc[5] += (unsigned long long)n_vis_prof;
#endif
    // ---- the next round, and its loads --------------------------------------------------------------------------------------
    if (rbase + (u32)kVisCap < n_rec) {
        rbase += (u32)kVisCap; // more records of this half: same slab
    } else if (phase == 0) {
        phase = 1, rbase = 0;
    } else {
        ++k;
        item = nx_item;
        if (item >= n_items)
            break;
        set_item(item, uniform(nx_nrec), uniform(nx_toff));
        phase = 0, rbase = 0;
        if (threadIdx.x == 0)
            next_claim = atomicAdd(queue, 1u);
    }
    stage_issue();
    ++round;
    setup_round();
    } // round
    } // any item at all
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
    const bool bil = M.bilinear();
    int rc = bil ? ensure_dynamic_lds(reinterpret_cast<const void *>(k_scatter_wide<true>), (int)kLdsBytes, 9)
                 : ensure_dynamic_lds(reinterpret_cast<const void *>(k_scatter_wide<false>), (int)kLdsBytes, 0);
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
    if (bil)
        hipLaunchKernelGGL(k_scatter_wide<true>, dim3(grid), dim3(kThreads), kLdsBytes, s, V, D / kWide, W.tile_offsets,
                           W.hdr_count, W.headers, W.wpool, M, D, scale_f, F, queues, W.carry, W.counters);
    else
        hipLaunchKernelGGL(k_scatter_wide<false>, dim3(grid), dim3(kThreads), kLdsBytes, s, V, D / kWide, W.tile_offsets,
                           W.hdr_count, W.headers, W.wpool, M, D, scale_f, F, queues, W.carry, W.counters);
    return check_hip(hipGetLastError(), "scatter_wide launch");
}

} // namespace gwbp
