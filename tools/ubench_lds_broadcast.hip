// What does an LDS read cost when every lane asks for the SAME address?  (gfx950, 16 waves on one CU, one workgroup)
// Question behind it (round 4): k_scatter_wide could take its per-pair weight / pixel index from a broadcast ds_read_b128
// instead of two v_readlane -- does such a read cost like a quarter of a normal one (one 16-B word for everybody), like a
// normal one (64 lanes x 16 B come back), or more (bank conflicts)?
// build: hipcc -O3 --offload-arch=gfx950 -o ubench_lds_broadcast ubench_lds_broadcast.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// MODE 0: ds_read_b128, lane-distinct contiguous (lane * 16 + row * 1024): the slab read of the scatter loop
// MODE 1: ds_read_b128, same address in every lane
// MODE 2: ds_read_b64,  same address
// MODE 3: ds_read_b32,  same address
// MODE 4: ds_read2_b32, same address (two dwords 4 B apart)
// MODE 5: ds_read_b128, 16 lanes per address (four addresses per wave: what a 4-pair-group layout would read)
// MODE 6: v_readlane_b32 x 2 (the instruction pair the broadcast would replace), for the same clock
template <int MODE>
__global__ __launch_bounds__(1024) void k(int iters, float *__restrict__ out, unsigned long long *__restrict__ clk)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768; i += 1024)
        lds[i] = (float)(i & 255) * 1e-3f;
    __syncthreads();
    unsigned base;
    if (MODE == 0)
        base = (unsigned)lane * 16u + (unsigned)wave * 1024u;
    else if (MODE == 5)
        base = (unsigned)(lane >> 4) * 1024u + (unsigned)wave * 4096u;
    else
        base = (unsigned)wave * 1024u;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    float sacc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // eight reads at eight rows, then one wait: like a batch of the scatter loop
        f32x4_t r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned a = base + (unsigned)(((it + j) & 7) * 8192); // < 128 KB in every mode
            if (MODE == 0 || MODE == 1 || MODE == 5)
                asm volatile("ds_read_b128 %0, %1" : "=v"(r[j]) : "v"(a));
            else if (MODE == 2) {
                f32x2_t t;
                asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(a));
                r[j] = f32x4_t{t.x, t.y, 0.f, 0.f};
            } else if (MODE == 3) {
                float t;
                asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(a));
                r[j] = f32x4_t{t, 0.f, 0.f, 0.f};
            } else if (MODE == 4) {
                f32x2_t t;
                asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(t) : "v"(a));
                r[j] = f32x4_t{t.x, t.y, 0.f, 0.f};
            } else {
                const float v = (float)(lane + it);
                const float s0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
                const float s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j + 8));
                sacc += s0 * s1;
                r[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (MODE != 6)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j)
            asm volatile("" : "+v"(r[j])); // the results are live, nothing is done with them
        acc += r[it & 7];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w + sacc;
    if (threadIdx.x == 0)
        clk[0] = t1 - t0;
}

template <int MODE>
static void run(const char *name, int iters, float *out, unsigned long long *clk)
{
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    k<MODE><<<1, 1024, 131072>>>(iters, out, clk);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    k<MODE><<<1, 1024, 131072>>>(iters, out, clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c;
    CHECK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    const double insts = (double)iters * 8.0 * 16.0 * (MODE == 6 ? 2.0 : 1.0); // wave-instructions issued on the CU
    printf("%-58s %7.3f ns per wave-instruction on the CU (events; = %5.2f cycles at 2.4 GHz), %6.2f s_memtime ticks\n", name,
           (double)ms * 1e6 / insts, (double)ms * 1e6 / insts * 2.4, (double)c / insts);
}

int main()
{
    float *out;
    unsigned long long *clk;
    CHECK(hipMalloc(&out, 4096));
    CHECK(hipMalloc(&clk, 8));
    const int iters = 20000;
    run<0>("0 ds_read_b128, lane-distinct contiguous (slab read)", iters, out, clk);
    run<1>("1 ds_read_b128, same address in every lane", iters, out, clk);
    run<2>("2 ds_read_b64,  same address", iters, out, clk);
    run<3>("3 ds_read_b32,  same address", iters, out, clk);
    run<4>("4 ds_read2_b32, same address", iters, out, clk);
    run<5>("5 ds_read_b128, four addresses (16 lanes each)", iters, out, clk);
    run<6>("6 v_readlane_b32 (per instruction, two per pair)", iters, out, clk);
    return 0;
}
