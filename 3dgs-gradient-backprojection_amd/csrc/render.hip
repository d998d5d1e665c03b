// render.hip -- forward colour render for few channels (D <= 32: RGB, RGB+D, depth, the 16-d compressed field) and SH colour
// evaluation.
//
// "Next" row N3 of SURVEY.md section 8(f): the step BEFORE the hot path in the reference (backproject.py:89-100 renders
// the view with sh_degree=3 and feeds it to the 2-D feature network) and utils.test_proper_pruning (utils.py:316-340).
// k_render_px is the classic tile rasteriser: workgroup = 16x16 tile, thread = pixel, Gaussian records staged in LDS per
// batch, same blend arithmetic as k_blend (so alpha / T / early termination are bit-identical), colours accumulated
// front to back in registers.  It needs only project + bin_sort, not the weight store.
#include "gwbp_dev.h"

namespace gwbp {

// CH = channels held per pixel (4, 16 or 32; D <= CH).  Round 5: CH > 4 -- segment_compressed.py:154-165 renders the 16-d
// compressed field this way for every frame; through the weight store (k_blend, then k_render_rows with 8 of 64 lanes at work)
// that cost 0.75 + 1.13 ms at C2 -- and two wave-uniform early-outs taken from k_blend, neither of which changes a bit of the
// result: a candidate whose alpha >= 1/255 ellipse holds no live pixel of the wave's 4 x 16 quarter skips exp and T (sigma
// above ln(255 o) + margin cannot reach 1/255), one that contributes to no pixel skips its colour reads and FMAs.
template <int CH>
__global__ __launch_bounds__(256) void k_render_px(ViewDev V, const u32 *__restrict__ tile_offsets,
                                                   const u32 *__restrict__ vals, const G2D *__restrict__ g2d,
                                                   const float *__restrict__ colors, int D,
                                                   float *__restrict__ out, float *__restrict__ alphas)
{
    __shared__ float4 s_a[256]; // mx, my, opac, ln(255 opac) + margin
    __shared__ float4 s_b[256]; // ca, cb, cc, -
    __shared__ float4 s_c[256][CH / 4]; // colour
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
    float acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c)
        acc[c] = 0.f;
    for (u32 batch = beg; batch < end; batch += 256) {
        if (__syncthreads_count(done) == 256)
            break;
        const u32 bn = min(256u, end - batch);
        if (threadIdx.x < bn) {
            const u32 gid = vals[batch + threadIdx.x];
            const float4 *gp = reinterpret_cast<const float4 *>(g2d + gid);
            float4 a = gp[0];
            // alpha = o exp(-sigma) >= 1/255  <=>  sigma <= ln(255 o); 1e-3 absorbs the error of __logf and exp_neg (k_blend's
            // s_thr); o <= 1/255 gives a negative bound that no sigma >= 0 meets
            a.w = __logf(255.0f * a.z) + 1e-3f;
            s_a[threadIdx.x] = a;
            s_b[threadIdx.x] = gp[1];
            const float *cp = colors + (size_t)gid * D;
            float cv[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c)
                cv[c] = c < D ? cp[c] : 0.f;
#pragma unroll
            for (int c4 = 0; c4 < CH / 4; ++c4)
                s_c[threadIdx.x][c4] = make_float4(cv[4 * c4], cv[4 * c4 + 1], cv[4 * c4 + 2], cv[4 * c4 + 3]);
        }
        __syncthreads();
        for (u32 j = 0; j < bn; ++j) {
            if (__ballot(!done) == 0ull)
                break;
            const float4 a = s_a[j], b = s_b[j];
            const float dx = a.x - px, dy = a.y - py;
            const float sigma = __builtin_fmaf(b.y * dx, dy, 0.5f * __builtin_fmaf(b.x * dx, dx, (b.z * dy) * dy));
            if (__ballot(!done && sigma <= a.w) == 0ull)
                continue; // no live pixel of this quarter inside the alpha >= 1/255 ellipse
            const float alpha = __builtin_fminf(kAlphaMax, a.z * exp_neg(-__builtin_fmaxf(sigma, 0.f)));
            const bool ok = !done && (sigma >= 0.f) && (alpha >= kAlphaMin);
            const float next_T = T * (1.0f - alpha);
            const bool term = ok && (next_T <= kTMin);
            const bool valid = ok && !term;
            const float w = valid ? alpha * T : 0.f;
            T = valid ? next_T : T;
            done = done || term;
            if (__ballot(valid) == 0ull)
                continue; // nobody takes colour from this Gaussian
#pragma unroll
            for (int c4 = 0; c4 < CH / 4; ++c4) {
                const float4 c = s_c[j][c4];
                // (valid ? : instead of a plain w = 0 product: a non-finite colour must reach only the pixels the Gaussian has a
                // weight at, as in k_render_rows, which never multiplies at all where the record has no entry)
                acc[4 * c4] = valid ? __builtin_fmaf(w, c.x, acc[4 * c4]) : acc[4 * c4];
                acc[4 * c4 + 1] = valid ? __builtin_fmaf(w, c.y, acc[4 * c4 + 1]) : acc[4 * c4 + 1];
                acc[4 * c4 + 2] = valid ? __builtin_fmaf(w, c.z, acc[4 * c4 + 2]) : acc[4 * c4 + 2];
                acc[4 * c4 + 3] = valid ? __builtin_fmaf(w, c.w, acc[4 * c4 + 3]) : acc[4 * c4 + 3];
            }
        }
    }
    if (inside) {
        float *o = out + ((size_t)iy * V.W + ix) * D;
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (c < D)
                o[c] = acc[c];
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
#define GWBP_PX(C)                                                                                                    \
    hipLaunchKernelGGL(k_render_px<C>, dim3(n_tiles), dim3(256), 0, s, V, W.tile_offsets, W.vals[fin], W.g2d, colors, D, out, alphas)
    if (D <= 4)
        GWBP_PX(4);
    else if (D <= 16)
        GWBP_PX(16);
    else
        GWBP_PX(32);
#undef GWBP_PX
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
