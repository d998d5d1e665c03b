// Microbenchmark (round 5): the scatter kernel's batch loop fed by SCALAR loads instead of v_readlane.
//
//   A  the shipping form: a visit's 64 entries {w, pix} arrive as one vector load a visit ahead, each pair costs 2 v_readlane
//      + v_lshl_add + ds_read_b128 + 2 v_pk_fma_f32 (5 vector instructions)
//   B  the entries of a batch of eight pairs arrive as ONE s_load_dwordx16 into an aligned SGPR tuple ({w, pix} = one SGPR pair:
//      the pk_fma reads the weight, the lshl_add the pixel, straight from it): 3 vector instructions per pair.  Rolling double
//      buffer: at the top of a batch `s_waitcnt lgkmcnt(0)` (only this batch's load is outstanding, issued a whole batch ago),
//      then the NEXT batch's load goes out into the other tuple.  A one-dword-per-line vector load a visit ahead warms L2
//      (scalar loads of lines that are still in HBM: the 8.9 ms of round 1).
//   C  B without the L2-warming vector load (what the warm-up is worth)
//
// One persistent-style workgroup per CU (1024 threads, 128 KB slab), every wave streams its own run of visits of 32 pairs
// (4 batches) from a store larger than L2 + Infinity Cache.  build: hipcc -O3 --offload-arch=gfx950 -o ubench_sload ubench_sload.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32;
typedef unsigned long long u64;

__device__ __forceinline__ f32x4_t lds_read_b128(u32 a)
{
#if __HIP_DEVICE_COMPILE__
    return *reinterpret_cast<const __attribute__((address_space(3))) f32x4_t *>((size_t)a);
#else
    (void)a;
    return f32x4_t{0.f, 0.f, 0.f, 0.f};
#endif
}
__device__ __forceinline__ f32x2_t pk_fma(float w, f32x2_t f, f32x2_t acc) { return __builtin_elementwise_fma(f32x2_t{w, w}, f, acc); }

constexpr int kPairsPerVisit = 32;

// one batch of eight pairs from the SGPR tuple A (16 dwords: w0, pix0, w1, pix1, ...), prefetching the next batch into tuple B
#define BATCH_ASM(A, B, NEXT_OFF)                                                                                       \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                          \
    "s_load_dwordx16 s[" #B ":" #B "+15], %[eb], " #NEXT_OFF "\n\t"                                                     \
    "v_lshl_add_u32 v60, s[" #A "+1], 10, %[rb]\n\t ds_read_b128 v[64:67], v60\n\t"                                     \
    "v_lshl_add_u32 v60, s[" #A "+3], 10, %[rb]\n\t ds_read_b128 v[68:71], v60\n\t"                                     \
    "v_lshl_add_u32 v60, s[" #A "+5], 10, %[rb]\n\t ds_read_b128 v[72:75], v60\n\t"                                     \
    "v_lshl_add_u32 v60, s[" #A "+7], 10, %[rb]\n\t ds_read_b128 v[76:79], v60\n\t"                                     \
    "v_lshl_add_u32 v60, s[" #A "+9], 10, %[rb]\n\t ds_read_b128 v[80:83], v60\n\t"                                     \
    "v_lshl_add_u32 v60, s[" #A "+11], 10, %[rb]\n\t ds_read_b128 v[84:87], v60\n\t"                                    \
    "v_lshl_add_u32 v60, s[" #A "+13], 10, %[rb]\n\t ds_read_b128 v[88:91], v60\n\t"                                    \
    "v_lshl_add_u32 v60, s[" #A "+15], 10, %[rb]\n\t ds_read_b128 v[92:95], v60\n\t"                                    \
    "s_waitcnt lgkmcnt(7)\n\t"                                                                                          \
    "v_pk_fma_f32 %[lo], s[" #A ":" #A "+1], v[64:65], %[lo] op_sel_hi:[0,1,1]\n\t"                                     \
    "v_pk_fma_f32 %[hi], s[" #A ":" #A "+1], v[66:67], %[hi] op_sel_hi:[0,1,1]\n\t"                                     \
    "s_waitcnt lgkmcnt(6)\n\t"                                                                                          \
    "v_pk_fma_f32 %[lo], s[" #A "+2:" #A "+3], v[68:69], %[lo] op_sel_hi:[0,1,1]\n\t"                                   \
    "v_pk_fma_f32 %[hi], s[" #A "+2:" #A "+3], v[70:71], %[hi] op_sel_hi:[0,1,1]\n\t"                                   \
    "s_waitcnt lgkmcnt(5)\n\t"                                                                                          \
    "v_pk_fma_f32 %[lo], s[" #A "+4:" #A "+5], v[72:73], %[lo] op_sel_hi:[0,1,1]\n\t"                                   \
    "v_pk_fma_f32 %[hi], s[" #A "+4:" #A "+5], v[74:75], %[hi] op_sel_hi:[0,1,1]\n\t"                                   \
    "s_waitcnt lgkmcnt(4)\n\t"                                                                                          \
    "v_pk_fma_f32 %[lo], s[" #A "+6:" #A "+7], v[76:77], %[lo] op_sel_hi:[0,1,1]\n\t"                                   \
    "v_pk_fma_f32 %[hi], s[" #A "+6:" #A "+7], v[78:79], %[hi] op_sel_hi:[0,1,1]\n\t"                                   \
    "s_waitcnt lgkmcnt(3)\n\t"                                                                                          \
    "v_pk_fma_f32 %[lo], s[" #A "+8:" #A "+9], v[80:81], %[lo] op_sel_hi:[0,1,1]\n\t"                                   \
    "v_pk_fma_f32 %[hi], s[" #A "+8:" #A "+9], v[82:83], %[hi] op_sel_hi:[0,1,1]\n\t"                                   \
    "s_waitcnt lgkmcnt(2)\n\t"                                                                                          \
    "v_pk_fma_f32 %[lo], s[" #A "+10:" #A "+11], v[84:85], %[lo] op_sel_hi:[0,1,1]\n\t"                                 \
    "v_pk_fma_f32 %[hi], s[" #A "+10:" #A "+11], v[86:87], %[hi] op_sel_hi:[0,1,1]\n\t"                                 \
    "s_waitcnt lgkmcnt(1)\n\t"                                                                                          \
    "v_pk_fma_f32 %[lo], s[" #A "+12:" #A "+13], v[88:89], %[lo] op_sel_hi:[0,1,1]\n\t"                                 \
    "v_pk_fma_f32 %[hi], s[" #A "+12:" #A "+13], v[90:91], %[hi] op_sel_hi:[0,1,1]\n\t"                                 \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                          \
    "v_pk_fma_f32 %[lo], s[" #A "+14:" #A "+15], v[92:93], %[lo] op_sel_hi:[0,1,1]\n\t"                                 \
    "v_pk_fma_f32 %[hi], s[" #A "+14:" #A "+15], v[94:95], %[hi] op_sel_hi:[0,1,1]\n\t"

template <int MODE>
__global__ __launch_bounds__(1024) void k(int visits, const uint2 *__restrict__ store, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768; i += 1024)
        lds[i] = (float)(i & 255) * 1e-3f;
    __syncthreads();
    const u32 row_base = (u32)lane * 16u;
    f32x2_t acc_lo = {0.f, 0.f}, acc_hi = {0.f, 0.f};
    // this wave's run of the store: visits x 32 entries, contiguous
    const size_t wave_id = (size_t)blockIdx.x * 16 + wave;
    const uint2 *run = store + wave_id * (size_t)visits * kPairsPerVisit;
    if (MODE == 0) {
        uint2 e_next = run[lane & 31];
        for (int v = 0; v < visits; ++v) {
            const uint2 e = e_next;
            if (v + 1 < visits)
                e_next = run[(size_t)(v + 1) * kPairsPerVisit + (lane & 31)];
            const float ew = __uint_as_float(e.x);
            const u32 ep = e.y;
#pragma unroll
            for (int b = 0; b < kPairsPerVisit / 8; ++b) {
                u32 px[8];
                f32x4_t f[8];
                float w[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    px[j] = (u32)__builtin_amdgcn_readlane((int)ep, 8 * b + j);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    f[j] = lds_read_b128((px[j] << 10) + row_base);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    w[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ew), 8 * b + j));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc_lo = pk_fma(w[j], f[j].xy, acc_lo);
                    acc_hi = pk_fma(w[j], f[j].zw, acc_hi);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        u64 eb = (u64)run;
        eb = ((u64)__builtin_amdgcn_readfirstlane((int)(eb >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)eb);
        u32 warm = 0;
        // prime: batch 0 into tuple A (s[36:51])
        asm volatile("s_load_dwordx16 s[36:51], %[eb], 0x0" ::[eb] "s"(eb)
                     : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "memory");
        for (int v = 0; v < visits; ++v) {
            if (MODE == 1) {
                // L2 warm-up of the NEXT visit's two 128-B lines (lanes 0 and 1), never waited for on its own
                const u32 voff = (u32)(kPairsPerVisit * 8) + (u32)(lane & 1) * 128u;
                asm volatile("global_load_dword %0, %1, %2" : "+v"(warm) : "v"(voff), "s"(eb) : "memory");
            }
            // four batches: A, B, A, B; the last one prefetches the next visit's first batch (the store is padded by one visit)
            asm volatile(BATCH_ASM(36, 52, 0x40) BATCH_ASM(52, 36, 0x80) BATCH_ASM(36, 52, 0xc0) BATCH_ASM(52, 36, 0x100)
                         : [lo] "+v"(acc_lo), [hi] "+v"(acc_hi)
                         : [eb] "s"(eb), [rb] "v"(row_base)
                         : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50",
                           "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65",
                           "s66", "s67", "v60", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76",
                           "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92",
                           "v93", "v94", "v95", "memory");
            eb += (u64)kPairsPerVisit * 8;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(warm)::"memory");
        acc_lo.x += __uint_as_float(warm) * 0.f;
    }
    out[(size_t)blockIdx.x * 1024 + threadIdx.x] = acc_lo.x + acc_lo.y + acc_hi.x + acc_hi.y;
}

int main(int argc, char **argv)
{
    const int visits = argc > 1 ? atoi(argv[1]) : 1300; // per wave; 256 CUs x 16 waves x 1300 x 32 pairs = 1.7e8 (pair, 256 ch) = one C2 view
    int n_cu = 0;
    CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
    const size_t n_waves = (size_t)n_cu * 16;
    const size_t n_ent = n_waves * (size_t)(visits + 2) * kPairsPerVisit;
    uint2 *h = (uint2 *)malloc(n_ent * 8);
    u32 s = 12345u;
    for (size_t i = 0; i < n_ent; ++i) {
        s = s * 1664525u + 1013904223u;
        float w = (float)((s >> 8) & 1023) * 1e-4f;
        h[i].x = *(u32 *)&w;
        h[i].y = (s >> 20) & 127u;
    }
    uint2 *d;
    float *out;
    CHECK(hipMalloc(&d, n_ent * 8 + 4096));
    CHECK(hipMemcpy(d, h, n_ent * 8, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, (size_t)n_cu * 1024 * 4));
    // a 512 MB scratch swept between runs: the store must come from HBM, as in the pipeline
    char *scratch;
    CHECK(hipMalloc(&scratch, 512u << 20));
    const size_t lds = 128 * 1024;
    CHECK(hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float ref = 0.f;
    const char *names[3] = {"A  v_readlane (shipping form)", "B  s_load_dwordx16, rolling double buffer, L2 warmed", "C  B without the warm-up load"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            CHECK(hipMemset(scratch, rep + mode, 512u << 20));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            if (mode == 0)
                hipLaunchKernelGGL(k<0>, dim3(n_cu), dim3(1024), lds, 0, visits, d, out);
            else if (mode == 1)
                hipLaunchKernelGGL(k<1>, dim3(n_cu), dim3(1024), lds, 0, visits, d, out);
            else
                hipLaunchKernelGGL(k<2>, dim3(n_cu), dim3(1024), lds, 0, visits, d, out);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            CHECK(hipGetLastError());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            float o = 0;
            CHECK(hipMemcpy(&o, out + 77, 4, hipMemcpyDeviceToHost));
            if (mode == 0)
                ref = o;
            const double pairs = (double)n_waves * visits * kPairsPerVisit;
            printf("%-55s %.3f ms  %.1f ns per (pair, 256 ch) and wave  C2-equivalent %.2f ms  check %s (%.6g vs %.6g)\n", names[mode], ms,
                   ms * 1e6 / ((double)visits * kPairsPerVisit), ms * 1.74e8 / pairs, (o == ref) ? "same" : "DIFFERENT", o, ref);
        }
    return 0;
}
