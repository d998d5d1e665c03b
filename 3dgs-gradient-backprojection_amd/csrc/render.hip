// render.hip -- forward colour render for few channels (D <= 4: RGB, RGB+D, depth) and SH colour evaluation.
//
// "Next" row N3 of SURVEY.md section 8(f): the step BEFORE the hot path in the reference (backproject.py:89-100 renders
// the view with sh_degree=3 and feeds it to the 2-D feature network) and utils.test_proper_pruning (utils.py:316-340).
// k_render_px is the classic tile rasteriser: workgroup = 16x16 tile, thread = pixel, Gaussian records staged in LDS per
// batch, same blend arithmetic as k_blend (so alpha / T / early termination are bit-identical), colours accumulated
// front to back in registers.  It needs only project + bin_sort, not the weight store.
#include "gwbp_dev.h"

namespace gwbp {

__global__ __launch_bounds__(256) void k_render_px(ViewDev V, const u32 *__restrict__ tile_offsets,
                                                   const u32 *__restrict__ vals, const G2D *__restrict__ g2d,
                                                   const float *__restrict__ colors, int D,
                                                   float *__restrict__ out, float *__restrict__ alphas)
{
    __shared__ float4 s_a[256]; // mx, my, opac, -
    __shared__ float4 s_b[256]; // ca, cb, cc, -
    __shared__ float4 s_c[256]; // colour (up to 4 channels)
    const int tile = blockIdx.x;
    const int tx = tile % V.tile_w, ty = tile / V.tile_w;
    const int lane = threadIdx.x & 63;
    const int wave = (int)uniform(threadIdx.x >> 6);
    const int ix = tx * kTile + (lane & 15), iy = ty * kTile + wave * 4 + (lane >> 4);
    const bool inside = ix < V.W && iy < V.H;
    const float px = (float)ix + 0.5f, py = (float)iy + 0.5f;
    const u32 beg = tile_offsets[tile], end = tile_offsets[tile + 1];
    float T = 1.0f;
    bool done = !inside;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (u32 batch = beg; batch < end; batch += 256) {
        if (__syncthreads_count(done) == 256)
            break;
        const u32 bn = min(256u, end - batch);
        if (threadIdx.x < bn) {
            const u32 gid = vals[batch + threadIdx.x];
            const float4 *gp = reinterpret_cast<const float4 *>(g2d + gid);
            s_a[threadIdx.x] = gp[0];
            s_b[threadIdx.x] = gp[1];
            float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
            const float *cp = colors + (size_t)gid * D;
            c.x = cp[0];
            if (D > 1)
                c.y = cp[1];
            if (D > 2)
                c.z = cp[2];
            if (D > 3)
                c.w = cp[3];
            s_c[threadIdx.x] = c;
        }
        __syncthreads();
        for (u32 j = 0; j < bn; ++j) {
            if (__ballot(!done) == 0ull)
                break;
            const float4 a = s_a[j], b = s_b[j], c = s_c[j];
            const float dx = a.x - px, dy = a.y - py;
            const float sigma = __builtin_fmaf(b.y * dx, dy, 0.5f * __builtin_fmaf(b.x * dx, dx, (b.z * dy) * dy));
            const float alpha = __builtin_fminf(kAlphaMax, a.z * exp_neg(-__builtin_fmaxf(sigma, 0.f)));
            const bool ok = !done && (sigma >= 0.f) && (alpha >= kAlphaMin);
            const float next_T = T * (1.0f - alpha);
            const bool term = ok && (next_T <= kTMin);
            const bool valid = ok && !term;
            const float w = valid ? alpha * T : 0.f;
            acc.x = __builtin_fmaf(w, c.x, acc.x);
            acc.y = __builtin_fmaf(w, c.y, acc.y);
            acc.z = __builtin_fmaf(w, c.z, acc.z);
            acc.w = __builtin_fmaf(w, c.w, acc.w);
            T = valid ? next_T : T;
            done = done || term;
        }
    }
    if (inside) {
        float *o = out + ((size_t)iy * V.W + ix) * D;
        o[0] = acc.x;
        if (D > 1)
            o[1] = acc.y;
        if (D > 2)
            o[2] = acc.z;
        if (D > 3)
            o[3] = acc.w;
        if (alphas)
            alphas[(size_t)iy * V.W + ix] = 1.0f - T;
    }
}

// Real spherical harmonics up to degree 3 (the basis every 3DGS code base uses), 3 colour channels, followed by
// gsplat's "+0.5, clamp at 0".  coeffs is [N, K, 3] with K >= (degree+1)^2.
__global__ __launch_bounds__(256) void k_sh_colors(int64_t N, int degree, int K, const float *__restrict__ means,
                                                   const float *__restrict__ coeffs, float cx, float cy, float cz,
                                                   float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N)
        return;
    float x = means[3 * i] - cx, y = means[3 * i + 1] - cy, z = means[3 * i + 2] - cz;
    const float inv = 1.0f / __builtin_fmaxf(__builtin_sqrtf(x * x + y * y + z * z), 1e-12f);
    x *= inv, y *= inv, z *= inv;
    const float *sh = coeffs + (size_t)i * K * 3;
    float b[16];
    b[0] = 0.28209479177387814f;
    if (degree >= 1) {
        b[1] = -0.4886025119029199f * y, b[2] = 0.4886025119029199f * z, b[3] = -0.4886025119029199f * x;
    }
    if (degree >= 2) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        b[4] = 1.0925484305920792f * xy, b[5] = -1.0925484305920792f * yz;
        b[6] = 0.31539156525252005f * (2.f * zz - xx - yy), b[7] = -1.0925484305920792f * xz;
        b[8] = 0.5462742152960396f * (xx - yy);
        if (degree >= 3) {
            b[9] = -0.5900435899266435f * y * (3.f * xx - yy), b[10] = 2.890611442640554f * xy * z;
            b[11] = -0.4570457994644658f * y * (4.f * zz - xx - yy);
            b[12] = 0.3731763325901154f * z * (2.f * zz - 3.f * xx - 3.f * yy);
            b[13] = -0.4570457994644658f * x * (4.f * zz - xx - yy), b[14] = 1.445305721320277f * z * (xx - yy);
            b[15] = -0.5900435899266435f * x * (xx - 3.f * yy);
        }
    }
    const int nb = (degree + 1) * (degree + 1);
    float r = 0.f, g = 0.f, bl = 0.f;
    for (int k = 0; k < nb; ++k) {
        r = __builtin_fmaf(b[k], sh[3 * k], r);
        g = __builtin_fmaf(b[k], sh[3 * k + 1], g);
        bl = __builtin_fmaf(b[k], sh[3 * k + 2], bl);
    }
    out[3 * i] = __builtin_fmaxf(r + 0.5f, 0.f);
    out[3 * i + 1] = __builtin_fmaxf(g + 0.5f, 0.f);
    out[3 * i + 2] = __builtin_fmaxf(bl + 0.5f, 0.f);
}

int launch_render_px(const Ws &W, const ViewDev &V, const float *colors, int D, float *out, float *alphas,
                     hipStream_t s)
{
    const int n_tiles = V.tile_w * V.tile_h;
    const int fin = sort_passes(n_tiles) & 1;
    hipLaunchKernelGGL(k_render_px, dim3(n_tiles), dim3(256), 0, s, V, W.tile_offsets, W.vals[fin], W.g2d, colors, D,
                       out, alphas);
    return check_hip(hipGetLastError(), "render_px launch");
}

int launch_sh_colors(int64_t N, int degree, int K, const float *means, const float *coeffs, const float *campos,
                     float *out, hipStream_t s)
{
    if (N == 0)
        return GWBP_OK;
    hipLaunchKernelGGL(k_sh_colors, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, N, degree, K, means, coeffs,
                       campos[0], campos[1], campos[2], out);
    return check_hip(hipGetLastError(), "sh_colors launch");
}

} // namespace gwbp
