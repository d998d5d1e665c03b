// Does blockIdx.x % 8 identify the XCD a workgroup runs on when the grid is one 1024-thread, 128 KB-LDS workgroup per CU
// (the shape of the persistent scatter kernels) and another stream is dispatching small kernels meanwhile (the front
// stage of the next view)?  Prints, per class c = blockIdx.x % 8, the XCC ids its workgroups reported.
// build: hipcc -O2 --offload-arch=gfx950 -o xcc_probe xcc_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(1024) void k_big(int *xcc, int spin)
{
    extern __shared__ float lds[];
    // HW_REG_XCC_ID = 20, bits [3:0]
    const int id = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    if (threadIdx.x == 0)
        xcc[blockIdx.x] = id;
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) // stay resident for a while, like a persistent kernel
        a = a * 1.0001f + lds[(threadIdx.x + i) & 1023];
    if (a == 12345.f)
        xcc[0] = -1;
}
__global__ void k_small(float *p)
{
    p[blockIdx.x * blockDim.x + threadIdx.x] += 1.f;
}

int main()
{
    int *xcc;
    float *buf;
    CHECK(hipMalloc(&xcc, 4096 * sizeof(int)));
    CHECK(hipMalloc(&buf, 1 << 24));
    CHECK(hipFuncSetAttribute((const void *)k_big, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipStream_t s0, s1;
    CHECK(hipStreamCreateWithPriority(&s0, hipStreamNonBlocking, 0));
    CHECK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, -1));
    int host[4096];
    for (int round = 0; round < 6; ++round) {
        const bool busy = round >= 2; // rounds 2..5: small kernels on the other stream while the big grid is dispatched
        if (busy)
            for (int i = 0; i < 200; ++i)
                k_small<<<6700, 64, 0, s1>>>(buf);
        k_big<<<256, 1024, 131072, s0>>>(xcc, 20000);
        if (busy)
            for (int i = 0; i < 200; ++i)
                k_small<<<6700, 64, 0, s1>>>(buf);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(host, xcc, 256 * sizeof(int), hipMemcpyDeviceToHost));
        int mixed = 0;
        printf("round %d (%s):", round, busy ? "other stream busy" : "alone");
        for (int c = 0; c < 8; ++c) {
            unsigned mask = 0;
            for (int b = c; b < 256; b += 8)
                mask |= 1u << (host[b] & 15);
            printf(" class %d -> xcc mask 0x%02x;", c, mask);
            mixed += __builtin_popcount(mask) != 1;
        }
        printf("  classes spread over more than one XCD: %d\n", mixed);
    }
    return 0;
}
