// Microbenchmark: the scatter inner loop on the matrix cores, four Gaussians per pixel read.
//
//   step = (4 Gaussians, 1 pixel, 256 channels):  1 v_readlane (pixel) + 1 v_lshl_add + 1 ds_read_b128 (1 KB slab row)
//          + 4 x v_mfma_f32_4x4x1_16B_f32 with the A operand BROADCAST from block `abid` (cbsz = 4)
//
// v_mfma_f32_4x4x1 computes 16 independent 4x4 outer products  D_b[i][j] = A_b[i] * B_b[j] + C_b[i][j]  (k = 1: ONE fmaf per
// element, exact fp32).  Lane 4b + i holds A_b[i]; lane 4b + j holds B_b[j]; lane 4b + j, VGPR i holds D_b[i][j].  With
// cbsz = 4 all 16 blocks take A from block `abid`, an IMMEDIATE: a register whose lane 4e + i holds w[entry e][Gaussian i]
// (= a plain coalesced load of sixteen {w0, w1, w2, w3} entries) feeds sixteen steps without any cross-lane traffic.
// B of MFMA c = component c of the lane's float4 (channels 4 lane + c): after the step lane L, accumulator c, VGPR i holds
// F[Gaussian i][channel 4 L + c]: the same lane <-> channel layout as the vector kernel's flush.
//
// Checks the result against the host, then times W waves per CU (4, 8, 16) with the in-kernel clock.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o ubench_mfma_scatter ubench_mfma_scatter.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 lds_read_b128(unsigned a)
{
#if __HIP_DEVICE_COMPILE__
    return *(const __attribute__((address_space(3))) float4 *)(size_t)a;
#else
    (void)a;
    return float4{};
#endif
}

constexpr int kRows = 128; // slab rows (pixels) of 256 channels = 128 KB

template <int E> struct Step {
    __device__ static __forceinline__ void run(f32x4 (&acc)[4], float wv, const float4 &f)
    {
        acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, f.x, acc[0], 4, E, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, f.y, acc[1], 4, E, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, f.z, acc[2], 4, E, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, f.w, acc[3], 4, E, 0);
    }
};

// MODE 0: full step; 1: no MFMA (LDS + address only); 2: MFMA only (fixed B); 3: vector reference (today's loop on ONE of the 4 Gaussians)
template <int WAVES, int MODE>
__global__ __launch_bounds__(WAVES * 64) void k(int iters, const float *__restrict__ wsrc, const int *__restrict__ psrc,
                                                const float *__restrict__ slab_src, float *__restrict__ out,
                                                unsigned long long *__restrict__ clk)
{
    const unsigned long long t_start = __builtin_amdgcn_s_memtime(), r_start = __builtin_amdgcn_s_memrealtime();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < kRows * 256; i += WAVES * 64)
        lds[i] = slab_src[i];
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
        acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *wp = wsrc + (size_t)wave * iters * 64;
    const int *pp = psrc + (size_t)wave * iters * 16;
    const unsigned lane_base = lane * 16;
    float wv = wp[lane];
    int pv = pp[lane & 15];
    float4 f[8], g[8];
#define UB_READS(E0, F, PV)                                                                                           \
    _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                                     \
    {                                                                                                                 \
        const int p = __builtin_amdgcn_readlane(PV, (E0) + j);                                                        \
        F[j] = MODE == 2 ? float4{1.f, 2.f, 3.f, (float)p} : lds_read_b128((p << 10) + lane_base);                    \
    }
#define UB_MATH(E0, F, WV)                                                                                            \
    if (MODE == 0 || MODE == 2) {                                                                                     \
        Step<(E0) + 0>::run(acc, WV, F[0]);                                                                           \
        Step<(E0) + 1>::run(acc, WV, F[1]);                                                                           \
        Step<(E0) + 2>::run(acc, WV, F[2]);                                                                           \
        Step<(E0) + 3>::run(acc, WV, F[3]);                                                                           \
        Step<(E0) + 4>::run(acc, WV, F[4]);                                                                           \
        Step<(E0) + 5>::run(acc, WV, F[5]);                                                                           \
        Step<(E0) + 6>::run(acc, WV, F[6]);                                                                           \
        Step<(E0) + 7>::run(acc, WV, F[7]);                                                                           \
    } else if (MODE == 1) {                                                                                           \
        _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                                 \
        {                                                                                                             \
            acc[0][0] += F[j].x;                                                                                      \
            acc[1][0] += F[j].y;                                                                                      \
            acc[2][0] += F[j].z;                                                                                      \
            acc[3][0] += F[j].w;                                                                                      \
        }                                                                                                             \
    } else {                                                                                                          \
        _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                                 \
        {                                                                                                             \
            const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(WV), 4 * ((E0) + j)));            \
            acc[0][0] = __builtin_fmaf(w, F[j].x, acc[0][0]);                                                         \
            acc[1][0] = __builtin_fmaf(w, F[j].y, acc[1][0]);                                                         \
            acc[2][0] = __builtin_fmaf(w, F[j].z, acc[2][0]);                                                         \
            acc[3][0] = __builtin_fmaf(w, F[j].w, acc[3][0]);                                                         \
        }                                                                                                             \
    }
    // software pipeline: the reads of the next half-batch are in flight while the matrix cores work on the current one,
    // the next sixteen entries' {w} / pixel registers are loaded one iteration ahead
    float wn = wp[(size_t)min(1, iters - 1) * 64 + lane];
    int pn = pp[(size_t)min(1, iters - 1) * 16 + (lane & 15)];
    UB_READS(0, f, pv)
    for (int it = 0; it < iters; ++it) {
        const float wnn = wp[(size_t)min(it + 2, iters - 1) * 64 + lane]; // two iterations ahead: never waited for here
        const int pnn = pp[(size_t)min(it + 2, iters - 1) * 16 + (lane & 15)];
        __builtin_amdgcn_sched_barrier(0);
        UB_READS(8, g, pv)
        __builtin_amdgcn_sched_barrier(0);
        UB_MATH(0, f, wv)
        __builtin_amdgcn_sched_barrier(0);
        UB_READS(0, f, pn)
        __builtin_amdgcn_sched_barrier(0);
        UB_MATH(8, g, wv)
        __builtin_amdgcn_sched_barrier(0);
        wv = wn, pv = pn;
        wn = wnn, pn = pnn;
    }
    // out[wave][gaussian i][channel 4 lane + c]
    float *o = out + ((size_t)blockIdx.x * WAVES + wave) * 4 * 256;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            o[i * 256 + 4 * lane + c] = acc[c][i];
    if (clk && threadIdx.x == 0) {
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t_start;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r_start;
    }
}

static unsigned long long *g_clk = nullptr;

template <int WAVES, int MODE>
float run(const char *name, int iters, const float *w, const int *p, const float *slab, float *out, bool print = true)
{
    const int lds = kRows * 256 * 4;
    CHECK(hipFuncSetAttribute((const void *)k<WAVES, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    k<WAVES, MODE><<<256, WAVES * 64, lds>>>(iters, w, p, slab, out, nullptr);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<WAVES, MODE><<<256, WAVES * 64, lds>>>(iters, w, p, slab, out, g_clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long hc[512];
    CHECK(hipMemcpy(hc, g_clk, sizeof(hc), hipMemcpyDeviceToHost));
    double ghz = 0;
    for (int i = 0; i < 256; ++i)
        ghz += (double)hc[2 * i] / (double)hc[2 * i + 1] * 0.1;
    ghz /= 256;
    const double steps_per_cu = (double)iters * 16 * WAVES;
    if (print)
        printf("[%.2f GHz] %-40s waves/CU %2d  %8.3f ms  %6.2f ns/step/wave  %6.1f steps/us/CU  (%.1f cycles/step/SIMD)\n", ghz,
               name, WAVES, ms, ms * 1e6 / (iters * 16.0), steps_per_cu / (ms * 1e3),
               ms * 1e-3 * ghz * 1e9 / (steps_per_cu / 4));
    return ms;
}

int main()
{
    const int iters = 1500, max_waves = 16;
    std::vector<float> hw((size_t)max_waves * iters * 64), hs(kRows * 256);
    std::vector<int> hp((size_t)max_waves * iters * 16);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    for (auto &v : hw) { const float r = rnd(); v = r < 0.4f ? 0.f : r * 0.01f; } // 40 % structural zeros like the real blocks
    for (auto &v : hs) v = rnd() * 2.f - 1.f;
    for (auto &v : hp) v = (int)(rnd() * kRows) % kRows;
    float *w, *slab, *out;
    int *p;
    CHECK(hipMalloc(&w, hw.size() * 4));
    CHECK(hipMalloc(&p, hp.size() * 4));
    CHECK(hipMalloc(&slab, hs.size() * 4));
    CHECK(hipMalloc(&out, (size_t)256 * max_waves * 4 * 256 * 4));
    CHECK(hipMalloc(&g_clk, 512 * sizeof(unsigned long long)));
    CHECK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(p, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(slab, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));

    // correctness: wave 0..3 of block 0 against a host fmaf chain in step order
    run<4, 0>("check", iters, w, p, slab, out, false);
    std::vector<float> ho(4 * 4 * 256);
    CHECK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
    long bad = 0;
    double maxd = 0;
    for (int wave = 0; wave < 4; ++wave)
        for (int i = 0; i < 4; ++i)
            for (int ch = 0; ch < 256; ++ch) {
                float a = 0.f;
                for (int it = 0; it < iters; ++it)
                    for (int e = 0; e < 16; ++e) {
                        const float wt = hw[((size_t)wave * iters + it) * 64 + 4 * e + i];
                        const int px = hp[((size_t)wave * iters + it) * 16 + e];
                        a = fmaf(wt, hs[px * 256 + ch], a);
                    }
                const float g = ho[(wave * 4 + i) * 256 + ch];
                if (g != a)
                    ++bad, maxd = fmax(maxd, fabs((double)g - a));
            }
    printf("MFMA 4x4x1 broadcast-A scatter vs host fmaf chain: %ld of %d values differ (max |diff| %.3g)\n", bad, 4 * 4 * 256, maxd);

    run<4, 0>("0 readlane+add+b128+4 mfma4x4x1", iters, w, p, slab, out);
    run<8, 0>("0 readlane+add+b128+4 mfma4x4x1", iters, w, p, slab, out);
    run<16, 0>("0 readlane+add+b128+4 mfma4x4x1", iters, w, p, slab, out);
    run<4, 1>("1 readlane+add+b128 (no mfma)", iters, w, p, slab, out);
    run<8, 1>("1 readlane+add+b128 (no mfma)", iters, w, p, slab, out);
    run<16, 1>("1 readlane+add+b128 (no mfma)", iters, w, p, slab, out);
    run<4, 2>("2 4 mfma4x4x1 only", iters, w, p, slab, out);
    run<8, 2>("2 4 mfma4x4x1 only", iters, w, p, slab, out);
    run<16, 2>("2 4 mfma4x4x1 only", iters, w, p, slab, out);
    run<16, 3>("3 vector loop, ONE Gaussian per step", iters, w, p, slab, out);
    return 0;
}
