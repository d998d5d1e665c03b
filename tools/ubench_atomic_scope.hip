// Microbenchmark: what does a flush of partial rows into a big fp32 matrix cost by memory scope?
//   F[rows, 768] (C4: 5 M rows = 15 GB; here 1 M rows = 3 GB, far beyond L2 + Infinity Cache), 3.7 M flushes of 128-float
//   segments x 6 chunks at random rows, each segment added by one wave (64 lanes x float2), as in k_scatter_full.
//   mode 0: agent-scope atomic add (what atomicAdd compiles to with -munsafe-fp-atomics)
//   mode 1: workgroup-scope atomic add (executes in the XCD's L2 -- only correct if every writer of a row sits on ONE XCD)
//   mode 2: plain load + add + store (only correct for rows with a single writer per launch)
// build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -o ubench_atomic_scope ubench_atomic_scope.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *F, const unsigned *rows, int n_flush, int D)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = gridDim.x * 4;
    const int chunks = D / 128;
    for (int i = wave; i < n_flush * chunks; i += n_waves) {
        const unsigned r = rows[i / chunks];
        float *p = F + (size_t)r * D + (i % chunks) * 128 + lane * 2;
        const float v0 = 1.0f, v1 = 2.0f;
        if (MODE == 0) {
            atomicAdd(p, v0);
            atomicAdd(p + 1, v1);
        } else if (MODE == 1) {
            __hip_atomic_fetch_add(p, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(p + 1, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            float2 o = *reinterpret_cast<float2 *>(p);
            o.x += v0, o.y += v1;
            *reinterpret_cast<float2 *>(p) = o;
        }
    }
}

template <int MODE>
void run(const char *name, float *F, const unsigned *rows, int n_flush, int D)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    k<MODE><<<256 * 8, 256>>>(F, rows, n_flush, D);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<MODE><<<256 * 8, 256>>>(F, rows, n_flush, D);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)n_flush * D * 4;
    printf("%-44s %8.3f ms  %7.1f GB/s of added bytes\n", name, ms, bytes / ms / 1e6);
}

int main(int argc, char **argv)
{
    const int D = 768, n_rows = 1000000, n_flush = argc > 1 ? atoi(argv[1]) : 3700000;
    float *F;
    unsigned *rows, *h = (unsigned *)malloc(sizeof(unsigned) * n_flush);
    unsigned s = 777;
    for (int i = 0; i < n_flush; ++i) { s = s * 1664525u + 1013904223u; h[i] = (s >> 4) % n_rows; }
    CHECK(hipMalloc(&F, (size_t)n_rows * D * 4));
    CHECK(hipMemset(F, 0, (size_t)n_rows * D * 4));
    CHECK(hipMalloc(&rows, sizeof(unsigned) * n_flush));
    CHECK(hipMemcpy(rows, h, sizeof(unsigned) * n_flush, hipMemcpyHostToDevice));
    run<0>("0 agent-scope atomic add (2 x f32 per lane)", F, rows, n_flush, D);
    run<1>("1 workgroup-scope atomic add", F, rows, n_flush, D);
    run<2>("2 plain load + add + store (float2)", F, rows, n_flush, D);
    run<0>("0 again", F, rows, n_flush, D);
    return 0;
}
