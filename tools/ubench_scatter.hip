// Microbenchmark of the scatter inner loop variants (per-pair cost on one CU with 16 waves, 128 KB LDS slab).
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o ubench_scatter ubench_scatter.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ float rl_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ int rl_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

__device__ __forceinline__ float4 lds_read_b128(unsigned a)
{
#if __HIP_DEVICE_COMPILE__
    return *(const __attribute__((address_space(3))) float4 *)(size_t)a;
#else
    (void)a;
    return float4{};
#endif
}

// MODE 0: 2 readlane + lshl_add + ds_read_b64 + pk_fma   (current design)
// MODE 1: lshl_add + ds_read_b64 + pk_fma with VGPR-held w/pix (no readlane; per-lane values, uniform content)
// MODE 2: readlanes only (+ trivial use)
// MODE 3: ds_read_b64 + pk_fma only (address precomputed per lane, fixed)
// MODE 4: like 0 but w/pix broadcast by ds_bpermute instead of readlane
// MODE 5: like 0 but b128 reads (4 ch/lane, 2 pk_fma) on a 64 KB half slab
template <int MODE>
__global__ __launch_bounds__(1024) void k(int iters, const float *__restrict__ wsrc, const int *__restrict__ psrc,
                                          float *__restrict__ out, unsigned long long *__restrict__ clk)
{
    const unsigned long long t_start = __builtin_amdgcn_s_memtime(), r_start = __builtin_amdgcn_s_memrealtime();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768; i += 1024)
        lds[i] = (float)(i & 255) * 1e-3f;
    for (int i = threadIdx.x; i < 4096 + 256; i += 1024) { // per-wave entry tables for modes 6/7
        lds[32768 + i] = (i & 1) ? __int_as_float((i * 37) & 255) : 0.001f * (i % 97);
    }
    __syncthreads();
    float wv = wsrc[(wave * 64 + lane) & 1023];
    int pv = psrc[(wave * 64 + lane) & 1023] & 255;
    const char *slab = (const char *)lds;
    const unsigned lane_base = lane * 8;
    float2 acc = make_float2(0.f, 0.f);
    float2 acc2 = make_float2(0.f, 0.f);
    unsigned long long smask = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            float2 f[8];
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int p = rl_i(pv, 8 * b + j);
                    f[j] = *(const float2 *)(slab + ((p << 9) + lane_base));
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float w = rl_f(wv, 8 * b + j);
                    acc.x = __builtin_fmaf(w, f[j].x, acc.x);
                    acc.y = __builtin_fmaf(w, f[j].y, acc.y);
                }
            } else if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int p = (pv + 8 * b + j) & 255;
                    f[j] = *(const float2 *)(slab + ((p << 9) + lane_base));
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc.x = __builtin_fmaf(wv, f[j].x, acc.x);
                    acc.y = __builtin_fmaf(wv, f[j].y, acc.y);
                }
            } else if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int p = rl_i(pv, 8 * b + j);
                    const float w = rl_f(wv, 8 * b + j);
                    acc.x += w;
                    acc.y += __int_as_float(p);
                }
            } else if (MODE == 3) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    f[j] = *(const float2 *)(slab + (((8 * b + j) << 9) + lane_base));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc.x = __builtin_fmaf(wv, f[j].x, acc.x);
                    acc.y = __builtin_fmaf(wv, f[j].y, acc.y);
                }
            } else if (MODE == 4) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int p = __builtin_amdgcn_ds_bpermute((8 * b + j) << 2, pv);
                    f[j] = *(const float2 *)(slab + ((p << 9) + lane_base));
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float w = __int_as_float(__builtin_amdgcn_ds_bpermute((8 * b + j) << 2, __float_as_int(wv)));
                    acc.x = __builtin_fmaf(w, f[j].x, acc.x);
                    acc.y = __builtin_fmaf(w, f[j].y, acc.y);
                }
            } else if (MODE == 6 || MODE == 7) {
                // entries {w, pix} of this wave live in LDS (written once per 64 entries); broadcast reads bring them
                // into VGPRs as wave-uniform values: no v_readlane at all
                const char *ent = slab + 131072 + wave * 512; // 64 entries x 8 B per wave (extra 8 KB of LDS)
                if (MODE == 6) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float2 e = *(const float2 *)(ent + 8 * (8 * b + j));
                        const int p = __float_as_int(e.y) & 255;
                        f[j] = *(const float2 *)(slab + ((p << 9) + lane_base));
                        acc2.x += e.x;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float2 e = *(const float2 *)(ent + 8 * (8 * b + j));
                        acc.x = __builtin_fmaf(e.x, f[j].x, acc.x);
                        acc.y = __builtin_fmaf(e.x, f[j].y, acc.y);
                    }
                } else {
                    float4 e4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        e4[j] = *(const float4 *)(ent + 16 * (4 * b + j));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int p0 = __float_as_int(e4[j].y) & 255, p1 = __float_as_int(e4[j].w) & 255;
                        f[2 * j] = *(const float2 *)(slab + ((p0 << 9) + lane_base));
                        f[2 * j + 1] = *(const float2 *)(slab + ((p1 << 9) + lane_base));
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc.x = __builtin_fmaf(e4[j].x, f[2 * j].x, acc.x);
                        acc.y = __builtin_fmaf(e4[j].x, f[2 * j].y, acc.y);
                        acc.x = __builtin_fmaf(e4[j].z, f[2 * j + 1].x, acc.x);
                        acc.y = __builtin_fmaf(e4[j].z, f[2 * j + 1].y, acc.y);
                    }
                }
            } else if (MODE == 8 || MODE == 9) {
                // "two pairs per wave-instruction": lanes 0-31 and 32-63 work on DIFFERENT pairs, each half covering 128
                // channels with 4 channels per lane (slab row = 512 B); {w, pix} come per half from the wave's entry
                // table in LDS (two distinct addresses per read: broadcast inside a half), no v_readlane.
                const int h = lane >> 5, s = lane & 31;
                const char *ent = slab + 131072 + wave * 1024 + (it & 1) * 512; // loop-variant: keeps the reads in the loop
                const unsigned ent_off = 131072u + wave * 1024u + (it & 1) * 512u + 8u * h; // dynamic LDS starts at 0
                float4 g[8];
                if (MODE == 8) {
                    float2 e[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(e[j]) : "v"(ent_off), "n"(16 * (8 * b + j)));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // asm reads are invisible to hipcc's waitcnt pass
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        unsigned a;
                        asm volatile("v_lshl_add_u32 %0, %1, 9, %2" : "=v"(a) : "v"(e[j].y), "v"(s * 16));
                        g[j] = lds_read_b128(a); // dynamic LDS starts at 0
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        acc.x = __builtin_fmaf(e[j].x, g[j].x, acc.x);
                        acc.y = __builtin_fmaf(e[j].x, g[j].y, acc.y);
                        acc2.x = __builtin_fmaf(e[j].x, g[j].z, acc2.x);
                        acc2.y = __builtin_fmaf(e[j].x, g[j].w, acc2.y);
                    }
                } else {
                    float4 e4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        e4[j] = *(const float4 *)(ent + 32 * (4 * b + j) + 16 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int p0 = __float_as_int(e4[j].y), p1 = __float_as_int(e4[j].w);
                        g[2 * j] = *(const float4 *)(slab + ((p0 << 9) + s * 16));
                        g[2 * j + 1] = *(const float4 *)(slab + ((p1 << 9) + s * 16));
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc.x = __builtin_fmaf(e4[j].x, g[2 * j].x, acc.x);
                        acc.y = __builtin_fmaf(e4[j].x, g[2 * j].y, acc.y);
                        acc2.x = __builtin_fmaf(e4[j].x, g[2 * j].z, acc2.x);
                        acc2.y = __builtin_fmaf(e4[j].x, g[2 * j].w, acc2.y);
                        acc.x = __builtin_fmaf(e4[j].z, g[2 * j + 1].x, acc.x);
                        acc.y = __builtin_fmaf(e4[j].z, g[2 * j + 1].y, acc.y);
                        acc2.x = __builtin_fmaf(e4[j].z, g[2 * j + 1].z, acc2.x);
                        acc2.y = __builtin_fmaf(e4[j].z, g[2 * j + 1].w, acc2.y);
                    }
                }
            } else if (MODE == 10 || MODE == 11) {
                // mode 5 with the NEXT batch's LDS reads issued before the current batch's FMAs (two batches in flight):
                // 10 = batches of 4 (8 float4 = 32 VGPRs like mode 5), 11 = batches of 8 (64 VGPRs)
                constexpr int KB = MODE == 10 ? 4 : 8;
                constexpr int NBT = 8 / KB; // batches per b-iteration
                float4 ga[KB], gb[KB];
#define UB_ISSUE(idx0, g)                                                                                             \
    _Pragma("unroll") for (int j = 0; j < KB; ++j)                                                                    \
    {                                                                                                                 \
        const int p = rl_i(pv, (idx0) + j) & 31;                                                                      \
        g[j] = *(const float4 *)(slab + ((p << 10) + lane * 16));                                                     \
    }
#define UB_FMA(idx0, g)                                                                                               \
    _Pragma("unroll") for (int j = 0; j < KB; ++j)                                                                    \
    {                                                                                                                 \
        const float w = rl_f(wv, (idx0) + j);                                                                         \
        acc.x = __builtin_fmaf(w, g[j].x, acc.x);                                                                     \
        acc.y = __builtin_fmaf(w, g[j].y, acc.y);                                                                     \
        acc2.x = __builtin_fmaf(w, g[j].z, acc2.x);                                                                   \
        acc2.y = __builtin_fmaf(w, g[j].w, acc2.y);                                                                   \
    }
                if (b == 0) {
                    UB_ISSUE(0, ga)
                }
                if (NBT == 2) {
                    UB_ISSUE(8 * b + 4, gb)
                    UB_FMA(8 * b, ga)
                    UB_ISSUE((8 * b + 8) & 63, ga)
                    UB_FMA(8 * b + 4, gb)
                } else {
                    if (b & 1) {
                        UB_ISSUE((8 * b + 8) & 63, ga)
                        UB_FMA(8 * b, gb)
                    } else {
                        UB_ISSUE((8 * b + 8) & 63, gb)
                        UB_FMA(8 * b, ga)
                    }
                }
#undef UB_ISSUE
#undef UB_FMA
            } else if (MODE == 12 || MODE == 13) {
                // LDS read rate alone: 8 ds_read_b128 per batch at per-lane fixed addresses (12) or at v_readlane'd pixel
                // addresses like mode 5 (13); results are only waited for, never used
                float4 g[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    unsigned a = (unsigned)((((8 * b + j) & 31) << 10) + lane * 16);
                    if (MODE == 13)
                        a = (unsigned)(((rl_i(pv, 8 * b + j) & 31) << 10) + lane * 16);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(g[j]) : "v"(a));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if (MODE == 14) {
                // mode 5 with the pixel index taken from the record's pixel MASK by a scalar bit scan (s_ff1 + s_bitset0 per
                // pair on the scalar unit) instead of a second v_readlane: 4 vector instructions per pair instead of 5
                float4 g[8];
                if (b == 0)
                    smask = __builtin_amdgcn_readfirstlane(pv) * 0x9E3779B97F4A7C15ull | 0xFFFFFFFFull; // >= 32 bits set
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    int p;
                    asm volatile("s_ff1_i32_b64 %0, %1\n\ts_bitset0_b64 %1, %0" : "=&s"(p), "+s"(smask));
                    g[j] = lds_read_b128((unsigned)(((p & 31) << 10) + lane * 16));
                }
                if (b == 3) // 32 pairs per refill in this benchmark (a half-tile record averages 24)
                    smask = __builtin_amdgcn_readfirstlane(pv + b) * 0x9E3779B97F4A7C15ull | 0xFFFFFFFFull;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float w = rl_f(wv, 8 * b + j);
                    acc.x = __builtin_fmaf(w, g[j].x, acc.x);
                    acc.y = __builtin_fmaf(w, g[j].y, acc.y);
                    acc2.x = __builtin_fmaf(w, g[j].z, acc2.x);
                    acc2.y = __builtin_fmaf(w, g[j].w, acc2.y);
                }
            } else if (MODE == 15) {
                // mode 5 with four v_fma_f32 (SGPR weight) instead of two v_pk_fma_f32: same lanes x flops, half the SGPRs per weight
                float4 g[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int p = rl_i(pv, 8 * b + j) & 31;
                    g[j] = *(const float4 *)(slab + ((p << 10) + lane * 16));
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float w = rl_f(wv, 8 * b + j);
                    asm volatile("v_fma_f32 %0, %4, %5, %0\n\tv_fma_f32 %1, %4, %6, %1\n\tv_fma_f32 %2, %4, %7, %2\n\tv_fma_f32 %3, %4, %8, %3"
                                 : "+v"(acc.x), "+v"(acc.y), "+v"(acc2.x), "+v"(acc2.y)
                                 : "s"(w), "v"(g[j].x), "v"(g[j].y), "v"(g[j].z), "v"(g[j].w));
                }
            } else if (MODE == 5) {
                float4 g[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int p = rl_i(pv, 8 * b + j) & 31;
                    g[j] = *(const float4 *)(slab + ((p << 10) + lane * 16));
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float w = rl_f(wv, 8 * b + j);
                    acc.x = __builtin_fmaf(w, g[j].x, acc.x);
                    acc.y = __builtin_fmaf(w, g[j].y, acc.y);
                    acc2.x = __builtin_fmaf(w, g[j].z, acc2.x);
                    acc2.y = __builtin_fmaf(w, g[j].w, acc2.y);
                }
            }
        }
        pv = (pv + 1) & 255; // keep the loop from being hoisted
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc.x + acc.y + acc2.x + acc2.y;
    if (clk && threadIdx.x == 0) { // in-kernel clock: shader cycles (s_memtime) per 100 MHz tick (s_memrealtime)
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t_start;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r_start;
    }
}

static unsigned long long *g_clk = nullptr;
template <int MODE>
void run(const char *name, int iters, const float *w, const int *p, float *out, double chunks_per_group = 1.0)
{
    if (!g_clk)
        CHECK(hipMalloc(&g_clk, 512 * sizeof(unsigned long long)));
    CHECK(hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 16384 + 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    k<MODE><<<256, 1024, 131072 + 16384 + 1024>>>(iters, w, p, out, nullptr);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<MODE><<<256, 1024, 131072 + 16384 + 1024>>>(iters, w, p, out, g_clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // one loop group = one (pair, 128 channels) unit in modes 0-4, 6, 7; two such units in modes 5, 8, 9
    const double pairs_per_cu = (double)iters * 64 * 16 * chunks_per_group; // per CU (16 waves)
    unsigned long long hc[512];
    CHECK(hipMemcpy(hc, g_clk, sizeof(hc), hipMemcpyDeviceToHost));
    double ghz = 0;
    for (int i = 0; i < 256; ++i)
        ghz += (double)hc[2 * i] / (double)hc[2 * i + 1] * 0.1;
    ghz /= 256;
    printf("[%.2f GHz] ", ghz);
    printf("%-34s %8.3f ms  %6.2f ns/pair/wave  %5.2f pairs/us/CU  -> C2 view (1.35M pair-chunks/CU): %.2f ms\n", name, ms,
           ms * 1e6 / (iters * 64.0), pairs_per_cu / (ms * 1e3), 1.35e6 / (pairs_per_cu / ms));
}

int main()
{
    float *w, *out;
    int *p;
    CHECK(hipMalloc(&w, 4096));
    CHECK(hipMalloc(&p, 4096));
    CHECK(hipMalloc(&out, 256 * 1024 * 4));
    float hw[1024];
    int hp[1024];
    for (int i = 0; i < 1024; ++i) hw[i] = 0.001f * (i % 97), hp[i] = (i * 37) & 255;
    CHECK(hipMemcpy(w, hw, 4096, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(p, hp, 4096, hipMemcpyHostToDevice));
    const int iters = 2000;
    run<0>("0 readlane x2 + add + b64 + pkfma", iters, w, p, out);
    run<1>("1 (no readlane) add + b64 + pkfma", iters, w, p, out);
    run<2>("2 readlanes only", iters, w, p, out);
    run<3>("3 b64 + pkfma only", iters, w, p, out);
    run<4>("4 bpermute x2 + add + b64 + pkfma", iters, w, p, out);
    run<5>("5 readlane x2 + add + b128 + 2pkfma", iters, w, p, out, 2.0);
    run<6>("6 LDS bcast b64/pair + add+b64+pkfma", iters, w, p, out);
    run<7>("7 LDS bcast b128/2pairs + ...", iters, w, p, out);
    run<15>("15 mode 5 with 4 v_fma_f32 per pair", iters, w, p, out, 2.0);
    run<10>("10 mode 5, 2 batches of 4 in flight", iters, w, p, out, 2.0);
    run<11>("11 mode 5, 2 batches of 8 in flight", iters, w, p, out, 2.0);
    run<14>("14 mode 5, pixel by scalar bit scan", iters, w, p, out, 2.0);
    run<12>("12 ds_read_b128 only, fixed addresses", iters, w, p, out, 2.0);
    run<13>("13 ds_read_b128 only, readlane addresses", iters, w, p, out, 2.0);
    run<8>("8 2 pairs/instr: b64 ent + add + b128 + 2pkfma", iters, w, p, out, 2.0);
    run<9>("9 2 pairs/instr: b128 ent/2 steps + ...", iters, w, p, out, 2.0);
    return 0;
}
