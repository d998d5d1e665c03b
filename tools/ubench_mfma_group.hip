// Microbenchmark: the scatter inner loop as a BLOCK-SPARSE product on the matrix cores, 16 Gaussians x 4 pixels per step.
//
//   F[g, c0:c0+128] += sum_p w_g(p) * feats[p, c0:c0+128]   for a GROUP of 16 (Gaussian, tile) records at once:
//   K-step = 4 pixels of the union of the group's footprints
//     A (16 x 4)  = w[record i][pixel k]      one coalesced dword load per lane from a dense per-group table (0 where the
//                                             record has no weight at that pixel)
//     B (4 x 16)  = slab[pixel k][16 n + j]   one ds_read_b32 per 16-channel block n (8 blocks = 128 channels)
//     8 x v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fmaf chain), accumulators = 32 VGPRs per wave
//   group end: 32 atomic wave-instructions (4 records x 64 B each).
// Lane l: i = j = l % 16, k = l / 16; D register v of block n = record 4 * (l / 16) + v, channel 16 n + l % 16.
//
// Times W waves per CU with the in-kernel clock, with MODE 0 full, 1 no MFMA, 2 MFMA only (operands fixed), 3 no flush.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o ubench_mfma_group ubench_mfma_group.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kPix = 256, kCh = 128;

__device__ __forceinline__ float lds_read_b32(unsigned a)
{
#if __HIP_DEVICE_COMPILE__
    return *(const __attribute__((address_space(3))) float *)(size_t)a;
#else
    (void)a;
    return 0.f;
#endif
}

template <int WAVES, int MODE>
__global__ __launch_bounds__(WAVES * 64) void k(int groups_per_wave, int S, const float *__restrict__ apool,
                                                const unsigned *__restrict__ kpix, const unsigned *__restrict__ gids,
                                                const float *__restrict__ slab_src, float *__restrict__ F, int n_rows,
                                                unsigned long long *__restrict__ clk)
{
    const unsigned long long t_start = __builtin_amdgcn_s_memtime(), r_start = __builtin_amdgcn_s_memrealtime();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < kPix * kCh; i += WAVES * 64)
        lds[i] = slab_src[i];
    __syncthreads();
    const int j16 = lane & 15, k4 = lane >> 4;
    const size_t wave_global = (size_t)blockIdx.x * WAVES + wave;
    const unsigned lane_col = (unsigned)j16 * 4u;
    for (int g = 0; g < groups_per_wave; ++g) {
        const size_t grp = wave_global * groups_per_wave + g;
        const size_t ks0 = grp * S;
        f32x4 acc[8];
#pragma unroll
        for (int n = 0; n < 8; ++n)
            acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        // A-store in blocks of 4 K-steps: apool[blk][lane] = float4 {A(4 blk + 0..3)[lane]}, kpix[blk][k4] = the four pixel
        // bytes of this lane's k slot; PF blocks (4 PF K-steps) in flight ahead of the MFMAs
        constexpr int PF = 3;
        const int nblk = (S + 3) >> 2;
        const size_t b0 = grp * (size_t)nblk;
        const float4 *ap = reinterpret_cast<const float4 *>(apool) + b0 * 64 + lane;
        const unsigned *pp = kpix + b0 * 4 + k4;
        float4 aq[PF];
        unsigned pq[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            aq[i] = ap[(size_t)min(i, nblk - 1) * 64];
            pq[i] = pp[(size_t)min(i, nblk - 1) * 4];
        }
        for (int blk = 0; blk < nblk; ++blk) {
            const float4 a4 = aq[0];
            const unsigned p4 = pq[0];
#pragma unroll
            for (int i = 0; i + 1 < PF; ++i)
                aq[i] = aq[i + 1], pq[i] = pq[i + 1];
            aq[PF - 1] = ap[(size_t)min(blk + PF, nblk - 1) * 64];
            pq[PF - 1] = pp[(size_t)min(blk + PF, nblk - 1) * 4];
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (4 * blk + t >= S) // wave-uniform: the group's last block may be short
                    break;
                float b[8];
                const unsigned row = (((p4 >> (8 * t)) & 255u) << 9) + lane_col; // 128 ch x 4 B = 512 B per pixel row
#pragma unroll
                for (int n = 0; n < 8; ++n)
                    b[n] = MODE == 2 ? (float)(n + 1) : lds_read_b32(row + 64u * n);
                if (MODE != 1) {
#pragma unroll
                    for (int n = 0; n < 8; ++n)
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], b[n], acc[n], 0, 0, 0);
                } else {
#pragma unroll
                    for (int n = 0; n < 8; ++n)
                        acc[n][0] += av[t] * b[n];
                }
            }
        }
        if (MODE != 3) {
            // flush: record 4 * k4 + v of this group, channels 16 n + j16
            const uint4 gq = reinterpret_cast<const uint4 *>(gids + grp * 16)[k4];
            const unsigned gv[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float *row = F + (size_t)(gv[v] % (unsigned)n_rows) * 512 + j16;
#pragma unroll
                for (int n = 0; n < 8; ++n)
                    atomicAdd(row + 16 * n, acc[n][v]);
            }
        } else {
            float sum = 0.f;
#pragma unroll
            for (int n = 0; n < 8; ++n)
                sum += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
            if (sum == 123.456f)
                F[lane] = sum;
        }
    }
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t_start;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r_start;
    }
}

template <int WAVES, int MODE>
static void run(const char *name, int n_cu, int gpw, int S, const float *apool, const unsigned *kpix, const unsigned *gids,
                const float *slab, float *F, int n_rows, unsigned long long *clk)
{
    auto kern = k<WAVES, MODE>;
    const size_t lds = (size_t)kPix * kCh * 4;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(n_cu), dim3(WAVES * 64), lds, 0, gpw, S, apool, kpix, gids, slab, F, n_rows, clk);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best)
            best = ms;
    }
    std::vector<unsigned long long> h(2 * n_cu);
    CHECK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0;
    for (int i = 0; i < n_cu; ++i)
        cyc += h[2 * i], rt += h[2 * i + 1];
    cyc /= n_cu, rt /= n_cu;
    const double ksteps_per_simd = (double)WAVES / 4 * gpw * S;
    (void)0;
    const double total_ksteps = (double)n_cu * WAVES * gpw * S;
    // C2 view at 128-channel chunks: 82.4 M pairs / (64 rho) K-steps x 4 chunks
    const double c2_ksteps = 82.4e6 / (64 * 0.36) * 4;
    printf("%-34s W=%2d  %7.3f ms  %6.1f cycles/K-step/SIMD  clock %.2f GHz  -> C2 view (rho 0.36): %.2f ms\n", name, WAVES, best,
           cyc / ksteps_per_simd, cyc / (rt / 100e6) / 1e9, best * c2_ksteps / total_ksteps);
}

int main(int argc, char **argv)
{
    const int S = argc > 1 ? atoi(argv[1]) : 31;
    const float rho = argc > 2 ? (float)atof(argv[2]) : 0.36f;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const int gpw_max = 64;
    const int nblk = (S + 3) / 4;
    const size_t n_groups = (size_t)n_cu * 16 * gpw_max, n_ks = n_groups * nblk * 4;
    std::vector<float> ha(n_ks * 64);                 // [group][blk][lane][t]
    std::vector<unsigned> hp(n_ks), hg(n_groups * 16); // [group][blk][k4] = 4 pixel bytes (t = 0..3)
    srand(7);
    for (auto &v : ha)
        v = (rand() / (float)RAND_MAX) < rho ? rand() / (float)RAND_MAX : 0.f;
    for (auto &v : hp)
        v = (unsigned)(rand() & 255) | (unsigned)(rand() & 255) << 8 | (unsigned)(rand() & 255) << 16 | (unsigned)(rand() & 255) << 24;
    const int n_rows = 1 << 18;
    for (auto &v : hg)
        v = (unsigned)(((unsigned)rand() * 2654435761u) % n_rows);
    std::vector<float> hs(kPix * kCh);
    for (auto &v : hs)
        v = rand() / (float)RAND_MAX - 0.5f;
    float *apool, *slab, *F;
    unsigned *kpix, *gids;
    unsigned long long *clk;
    CHECK(hipMalloc(&apool, ha.size() * 4));
    CHECK(hipMalloc(&kpix, hp.size() * 4));
    CHECK(hipMalloc(&gids, hg.size() * 4));
    CHECK(hipMalloc(&slab, hs.size() * 4));
    CHECK(hipMalloc(&F, (size_t)n_rows * 512 * 4));
    CHECK(hipMalloc(&clk, 2 * n_cu * 8));
    CHECK(hipMemcpy(apool, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(kpix, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(gids, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(slab, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(F, 0, (size_t)n_rows * 512 * 4));
    printf("%d CUs, %d K-steps per group, A density %.2f\n", n_cu, S, rho);
    // correctness of the layout on one group: compare F row sums with the host for W=4 (first wave, first group)
    {
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k<4, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, kPix * kCh * 4));
        hipLaunchKernelGGL((k<4, 0>), dim3(1), dim3(256), (size_t)kPix * kCh * 4, 0, 1, S, apool, kpix, gids, slab, F, n_rows, clk);
        CHECK(hipDeviceSynchronize());
        std::vector<float> hF((size_t)n_rows * 512);
        CHECK(hipMemcpy(hF.data(), F, hF.size() * 4, hipMemcpyDeviceToHost));
        std::vector<double> ref((size_t)n_rows * 128, 0.0);
        for (int w = 0; w < 4; ++w)
            for (int i = 0; i < 16; ++i)
                for (int s = 0; s < S; ++s)
                    for (int kk = 0; kk < 4; ++kk) {
                        const size_t blk = (size_t)w * nblk + s / 4;
                        const int t = s & 3;
                        const double a = ha[(blk * 64 + kk * 16 + i) * 4 + t];
                        const unsigned p = (hp[blk * 4 + kk] >> (8 * t)) & 255u;
                        const unsigned row = hg[(size_t)w * 16 + i] % n_rows;
                        for (int c = 0; c < 128; ++c)
                            ref[(size_t)row * 128 + c] += a * hs[p * kCh + c];
                    }
        double err = 0, mag = 0;
        for (int w = 0; w < 4; ++w)
            for (int i = 0; i < 16; ++i) {
                const unsigned row = hg[(size_t)w * 16 + i] % n_rows;
                for (int c = 0; c < 128; ++c) {
                    err = fmax(err, fabs(hF[(size_t)row * 512 + c] - ref[(size_t)row * 128 + c]));
                    mag = fmax(mag, fabs(ref[(size_t)row * 128 + c]));
                }
            }
        printf("layout check: max |err| %.3g (max |ref| %.3g) %s\n", err, mag, err <= 1e-4 * mag ? "OK" : "MISMATCH");
        {
            const unsigned row = hg[0] % n_rows;
            printf("  row %u: got %g %g %g %g  want %g %g %g %g\n", row, hF[(size_t)row * 512], hF[(size_t)row * 512 + 1],
                   hF[(size_t)row * 512 + 16], hF[(size_t)row * 512 + 127], ref[(size_t)row * 128], ref[(size_t)row * 128 + 1],
                   ref[(size_t)row * 128 + 16], ref[(size_t)row * 128 + 127]);
            size_t nz = 0;
            for (size_t i = 0; i < hF.size(); ++i)
                nz += hF[i] != 0.f;
            printf("  non-zero elements of F: %zu (expected about %d)\n", nz, 64 * 128);
        }
        CHECK(hipMemset(F, 0, (size_t)n_rows * 512 * 4));
    }
#define RUN(Wv, M, name, gpw) run<Wv, M>(name, n_cu, gpw, S, apool, kpix, gids, slab, F, n_rows, clk)
    RUN(16, 0, "full (A load, 8 lds, 8 mfma, flush)", 32);
    RUN(16, 3, "no flush", 32);
    RUN(16, 2, "MFMA only (B fixed)", 32);
    RUN(16, 1, "no MFMA (loads + 8 v_fma)", 32);
    RUN(8, 0, "full", 64);
    RUN(8, 3, "no flush", 64);
    RUN(4, 0, "full", 64);
    RUN(4, 3, "no flush", 64);
    RUN(4, 2, "MFMA only (B fixed)", 64);
    return 0;
}
