// The SUM pass of a store-then-sum scatter, measured at the shapes of C4 and C2 (round-3 review, item 4).
// Instead of one row of float atomics per (Gaussian, tile) record, the scatter kernel would write each record's partial row
// with plain stores; this pass then gives every Gaussian that has records ONE wave, which sums its 1..k partial rows and does
// one plain read-modify-write of F[g, :].  The partial rows of a Gaussian lie where its records were written (tile order), i.e.
// scattered: here they are placed at random.  Shapes (profiles/r3_flush_cache_sim.txt, view 0 of each config):
//   C4: 3,721,145 records, 1,723,388 Gaussians with weight of 5 M, D = 768     C2: 1,810,029 records, 585,522 of 1 M, D = 512
// build: hipcc -O3 --offload-arch=gfx950 -o ubench_sum_pass ubench_sum_pass.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>

typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// one wave per destination; D / 256 f32x4_t per lane and row
template <int D>
__global__ __launch_bounds__(256) void k_sum(int n_dst, const unsigned *__restrict__ dst_gid, const unsigned *__restrict__ row_off,
                                             const unsigned *__restrict__ rows, const f32x4_t *__restrict__ P, f32x4_t *__restrict__ F)
{
    constexpr int Q = D / 256; // f32x4_t per lane
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= n_dst)
        return;
    const unsigned r0 = row_off[wave], r1 = row_off[wave + 1];
    f32x4_t acc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q)
        acc[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const f32x4_t *f = F + (size_t)dst_gid[wave] * (D / 4);
    f32x4_t cur[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) // the F row's read is in flight under the partial rows
        cur[q] = f[q * 64 + lane];
    for (unsigned r = r0; r < r1; ++r) {
        const f32x4_t *p = P + (size_t)rows[r] * (D / 4);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const f32x4_t v = __builtin_nontemporal_load(p + q * 64 + lane);
            acc[q] += v;
        }
    }
    f32x4_t *fo = F + (size_t)dst_gid[wave] * (D / 4);
#pragma unroll
    for (int q = 0; q < Q; ++q)
        fo[q * 64 + lane] = cur[q] + acc[q];
}

#ifdef SUMPASS_LIB
// Library form for tools/probe_store_then_sum.py: a synthetic index of the given shape, the pass launched on the caller's
// stream against the caller's F.
static unsigned *g_gid, *g_off, *g_rows;
static f32x4_t *g_P;
static int g_D;
static size_t g_ndst;
extern "C" int sum_pass_setup(int D, long n_gauss, long n_dst, long n_rec)
{
    std::mt19937_64 rng(1234);
    std::vector<unsigned> gid(n_gauss);
    std::iota(gid.begin(), gid.end(), 0u);
    std::shuffle(gid.begin(), gid.end(), rng);
    gid.resize(n_dst);
    std::vector<unsigned> cnt(n_dst, 1u);
    for (long i = n_dst; i < n_rec; ++i)
        cnt[rng() % n_dst]++;
    std::vector<unsigned> off(n_dst + 1, 0u);
    for (long i = 0; i < n_dst; ++i)
        off[i + 1] = off[i] + cnt[i];
    std::vector<unsigned> rows(n_rec);
    std::iota(rows.begin(), rows.end(), 0u);
    std::shuffle(rows.begin(), rows.end(), rng);
    CHECK(hipMalloc(&g_gid, n_dst * 4));
    CHECK(hipMalloc(&g_off, (n_dst + 1) * 4));
    CHECK(hipMalloc(&g_rows, n_rec * 4));
    CHECK(hipMalloc(&g_P, (size_t)n_rec * D * 4));
    CHECK(hipMemcpy(g_gid, gid.data(), n_dst * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(g_off, off.data(), (n_dst + 1) * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(g_rows, rows.data(), n_rec * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(g_P, 0, (size_t)n_rec * D * 4));
    g_D = D, g_ndst = n_dst;
    return 0;
}
extern "C" int sum_pass_launch(void *F, void *stream)
{
    const int blocks = (int)((g_ndst + 3) / 4);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (g_D == 768)
        k_sum<768><<<blocks, 256, 0, s>>>((int)g_ndst, g_gid, g_off, g_rows, g_P, static_cast<f32x4_t *>(F));
    else if (g_D == 512)
        k_sum<512><<<blocks, 256, 0, s>>>((int)g_ndst, g_gid, g_off, g_rows, g_P, static_cast<f32x4_t *>(F));
    else
        return -1;
    return (int)hipGetLastError();
}
#else
template <int D>
static void run(const char *name, size_t n_gauss, size_t n_dst, size_t n_rec)
{
    std::mt19937_64 rng(1234);
    // destinations: n_dst distinct Gaussians; records per destination: 1 + a remainder spread at random (mean n_rec / n_dst)
    std::vector<unsigned> gid(n_gauss);
    std::iota(gid.begin(), gid.end(), 0u);
    std::shuffle(gid.begin(), gid.end(), rng);
    gid.resize(n_dst);
    std::vector<unsigned> cnt(n_dst, 1u);
    for (size_t i = n_dst; i < n_rec; ++i)
        cnt[rng() % n_dst]++;
    std::vector<unsigned> off(n_dst + 1, 0u);
    for (size_t i = 0; i < n_dst; ++i)
        off[i + 1] = off[i] + cnt[i];
    std::vector<unsigned> rows(n_rec);
    std::iota(rows.begin(), rows.end(), 0u);
    std::shuffle(rows.begin(), rows.end(), rng); // a Gaussian's partial rows lie anywhere in the record-ordered buffer
    unsigned *d_gid, *d_off, *d_rows;
    f32x4_t *P, *F;
    CHECK(hipMalloc(&d_gid, n_dst * 4));
    CHECK(hipMalloc(&d_off, (n_dst + 1) * 4));
    CHECK(hipMalloc(&d_rows, n_rec * 4));
    CHECK(hipMalloc(&P, n_rec * (size_t)D * 4));
    CHECK(hipMalloc(&F, n_gauss * (size_t)D * 4));
    CHECK(hipMemcpy(d_gid, gid.data(), n_dst * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_off, off.data(), (n_dst + 1) * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_rows, rows.data(), n_rec * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(P, 0, n_rec * (size_t)D * 4));
    CHECK(hipMemset(F, 0, n_gauss * (size_t)D * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int blocks = (int)((n_dst + 3) / 4);
    float best = 1e9f, sum = 0.f;
    const int reps = 6;
    for (int it = 0; it < reps + 1; ++it) {
        CHECK(hipEventRecord(e0));
        k_sum<D><<<blocks, 256>>>((int)n_dst, d_gid, d_off, d_rows, P, F);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (it) // (first launch: warm-up)
            best = std::min(best, ms), sum += ms;
    }
    const double bytes = (double)n_rec * D * 4 + 2.0 * (double)n_dst * D * 4;
    printf("%s: %zu partial rows of %d B -> %zu of %zu rows of F: %.3f ms (best %.3f), %.1f GB moved = %.2f TB/s\n", name, n_rec,
           D * 4, n_dst, n_gauss, sum / reps, best, bytes / 1e9, bytes / (sum / reps * 1e-3) / 1e12);
    CHECK(hipFree(d_gid));
    CHECK(hipFree(d_off));
    CHECK(hipFree(d_rows));
    CHECK(hipFree(P));
    CHECK(hipFree(F));
}

int main()
{
    run<768>("C4 sum pass", 5000000, 1723388, 3721145);
    run<512>("C2 sum pass", 1000000, 585522, 1810029);
    return 0;
}
#endif
