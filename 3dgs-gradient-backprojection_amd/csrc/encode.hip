// encode.hip -- k_encode_map: the per-pixel linear encoder of the compressed variant,
//
//   feats16[p, :] = feats512[p, :] @ encoder[512, 16]                         (backproject_compressed.py:127)
//
// as a skinny fp32 GEMM (M = H*W pixels ~ 1.7 M, K = 512, N <= 16) that reads the 3.47 GB map ONCE at HBM rate and
// writes 109 MB.  This is the one dense contraction on the path, so it runs on the matrix cores -- with
// v_mfma_f32_16x16x4_f32, whose result is bit for bit a k-ordered chain of fp32 fmaf (no reduced-precision inputs,
// MI355X_MICROARCH.md "Matrix cores"): 27.8 GFLOP per view = 0.2 ms of MFMA issue, hidden under 0.6 ms of HBM streaming.
// rocBLAS spends ~1.1 ms on this shape (N = 16 is far from its tile sizes).
//
// Wave = 16 pixels x 16 outputs per tile.  A operand (pixels x k): lane (m = lane % 16, q = lane / 16) loads the
// float4 feats[pixel m][16 j + 4 q .. + 3] of k-block j -- 64 B contiguous per pixel over the four q lanes; its component i
// feeds MFMA step i of the block, i.e. the k index of (slot q, step i) is 16 j + 4 q + i.  B operand: the encoder sits
// in LDS re-ordered as [j][i][q][n], so that step (j, i) is one conflict-free 256-B ds_read_b32 row.  Any assignment of k
// indices to MFMA slots is a valid summation order as long as A and B agree.
#include "gwbp_dev.h"

namespace gwbp {

namespace {

constexpr int kEncThreads = 256; // 4 waves per workgroup, 4 workgroups per CU keep ~32 KB of loads in flight per CU
constexpr int kEncN = 16;        // output channels of one MFMA tile (encoders with fewer are zero-padded)
constexpr int kEncPrefetch = 8;  // k-blocks (float4 per lane) in flight

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kEncThreads) void k_encode_map(const float *__restrict__ feats, int64_t fs_y, int64_t fs_x,
                                                            int H, int W, int K, const float *__restrict__ enc,
                                                            int n_out, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float s_enc[]; // [K/16][4][4][16]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = K / 16;
    for (int idx = threadIdx.x; idx < K * kEncN; idx += kEncThreads) {
        // idx = ((j * 4 + i) * 4 + q) * 16 + n  <-  encoder[16 j + 4 q + i][n]
        const int n = idx & 15, q = (idx >> 4) & 3, i = (idx >> 6) & 3, j = idx >> 8;
        const int k = 16 * j + 4 * q + i;
        s_enc[idx] = n < n_out ? enc[(int64_t)k * n_out + n] : 0.f;
    }
    __syncthreads();

    const int m = lane & 15, q = lane >> 4;
    const int64_t n_pix = (int64_t)H * W;
    const int64_t n_tiles = (n_pix + 15) / 16;
    const int64_t stride = (int64_t)gridDim.x * (kEncThreads / 64);
    for (int64_t t = (int64_t)blockIdx.x * (kEncThreads / 64) + wave; t < n_tiles; t += stride) {
        const int64_t p = min(t * 16 + m, n_pix - 1); // the last tile re-reads the last pixel (never stored twice)
        const int64_t y = p / W, x = p - y * W;
        const float4 *src = reinterpret_cast<const float4 *>(feats + y * fs_y + x * fs_x) + q;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float4 a[kEncPrefetch];
#pragma unroll
        for (int u = 0; u < kEncPrefetch; ++u)
            a[u] = src[4 * min(u, nb - 1)];
        for (int j0 = 0; j0 < nb; j0 += kEncPrefetch) {
#pragma unroll
            for (int u = 0; u < kEncPrefetch; ++u) {
                const int j = j0 + u;
                if (j >= nb)
                    break;
                const float4 av = a[u];
                if (j + kEncPrefetch < nb)
                    a[u] = src[4 * (j + kEncPrefetch)];
                const float *b = s_enc + j * 256 + lane; // (q, n) = lane; steps i are 64 floats apart
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b[64], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b[128], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b[192], acc, 0, 0, 0);
            }
        }
        // C layout of the 16x16 tile: lane holds column n = lane % 16, rows 4 * (lane / 16) + r
        if (m < n_out) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t pr = t * 16 + 4 * q + r;
                if (pr < n_pix)
                    out[pr * n_out + m] = acc[r];
            }
        }
    }
}

} // namespace

int launch_encode_map(const float *feats, int64_t fs_y, int64_t fs_x, int H, int W, int K, const float *enc, int n_out,
                      float *out, int workgroups, hipStream_t s)
{
    const size_t lds = (size_t)K * kEncN * sizeof(float);
    if (lds > 64 * 1024) {
        // ensure_dynamic_lds remembers "raised" per (device, slot), not the size: always raise to the largest request this
        // entry point accepts (K = 2048), or a K = 1040 call followed by K = 2048 would launch over the limit
        const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(k_encode_map), 2048 * kEncN * (int)sizeof(float), 6);
        if (rc)
            return rc;
    }
    int n_cu = 0;
    const int rc = device_cus(&n_cu);
    if (rc)
        return rc;
    const int64_t n_tiles = ((int64_t)H * W + 15) / 16;
    // Alone, four workgroups per CU (128 KB of loads in flight per CU) reach 6.0 TB/s.  Beside latency-bound kernels (the
    // next view's front stage, the small-D scatter) that much streaming doubles THEIR memory latency: a caller that
    // overlaps the encoder passes workgroups = one per CU (C5: 2.19 -> 1.96 ms/view; fewer make the encoder the long pole).
    const int per_cu = lds > 40 * 1024 ? 2 : 4;
    int64_t grid = workgroups > 0 ? workgroups : (int64_t)n_cu * per_cu;
    if (grid * (kEncThreads / 64) > n_tiles)
        grid = (n_tiles + kEncThreads / 64 - 1) / (kEncThreads / 64);
    if (grid < 1)
        grid = 1;
    hipLaunchKernelGGL(k_encode_map, dim3((unsigned)grid), dim3(kEncThreads), lds, s, feats, fs_y, fs_x, H, W, K, enc,
                       n_out, out);
    return check_hip(hipGetLastError(), "encode_map launch");
}

} // namespace gwbp
