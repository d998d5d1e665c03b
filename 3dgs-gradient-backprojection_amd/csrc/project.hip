// project.hip -- EWA projection of every Gaussian for one pinhole view + tiles-per-Gaussian count.
//
// Replaces gsplat 1.4.0's fully_fused_projection (first stage of rasterization(), reference call sites
// backproject.py:115,133).  One lane per Gaussian, coalesced AoS loads (12/16/12/4 B per lane, contiguous across the
// wave), one 32-B G2D record + one 8-B tile rectangle written per Gaussian.  Not packed/compacted: culled
// Gaussians keep radius 0 and touch no tile.  Algorithmic bytes: 44 B read + 44 B written per Gaussian.
//
// The arithmetic follows DESIGN.md "arithmetic contract" operation by operation (-ffp-contract=off; FMAs only
// where written), so that means2d / conics / depths / radii / rectangles match the CPU oracle bit for bit.
#include "gwbp_dev.h"

namespace gwbp {

__global__ __launch_bounds__(kScanBlock) void k_project(
    int64_t N, ViewDev V, const float *__restrict__ means, const float *__restrict__ quats,
    const float *__restrict__ scales, const float *__restrict__ opac, G2D *__restrict__ g2d,
    uint2 *__restrict__ rect, u32 *__restrict__ touched, u32 *__restrict__ dkeys, u32 *__restrict__ dvals,
    Counters *__restrict__ ctr, int32_t *__restrict__ o_radii, float *__restrict__ o_means2d, float *__restrict__ o_depths,
    float *__restrict__ o_conics, int tight, int prio)
{
    front_priority(prio);
    const int64_t i = (int64_t)blockIdx.x * kScanBlock + threadIdx.x;
    u32 ntiles = 0;
    G2D g;
    g.mx = g.my = g.opac = g.depth = g.ca = g.cb = g.cc = 0.f;
    g.radius = 0;
    uint2 rc = make_uint2(0u, 0u);

    if (i < N) {
        const float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
        const float x = dot3f(V.R[0], V.R[1], V.R[2], mx, my, mz) + V.t[0];
        const float y = dot3f(V.R[3], V.R[4], V.R[5], mx, my, mz) + V.t[1];
        const float z = dot3f(V.R[6], V.R[7], V.R[8], mx, my, mz) + V.t[2];
        bool ok = !(z < V.near_plane || z > V.far_plane);
        if (ok) {
            const float4 q4 = reinterpret_cast<const float4 *>(quats)[i];
            float qw = q4.x, qx = q4.y, qy = q4.z, qz = q4.w;
            const float n2 = __builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, __builtin_fmaf(qx, qx, qw * qw)));
            const float inv = 1.0f / __builtin_sqrtf(n2);
            qw *= inv, qx *= inv, qy *= inv, qz *= inv;
            const float x2 = qx * qx, y2 = qy * qy, z2 = qz * qz;
            const float xy = qx * qy, xz = qx * qz, yz = qy * qz;
            const float wx = qw * qx, wy = qw * qy, wz = qw * qz;
            const float q00 = 1.f - 2.f * (y2 + z2), q01 = 2.f * (xy - wz), q02 = 2.f * (xz + wy);
            const float q10 = 2.f * (xy + wz), q11 = 1.f - 2.f * (x2 + z2), q12 = 2.f * (yz - wx);
            const float q20 = 2.f * (xz - wy), q21 = 2.f * (yz + wx), q22 = 1.f - 2.f * (x2 + y2);
            const float s0 = scales[3 * i], s1 = scales[3 * i + 1], s2 = scales[3 * i + 2];
            const float m00 = q00 * s0, m01 = q01 * s1, m02 = q02 * s2;
            const float m10 = q10 * s0, m11 = q11 * s1, m12 = q12 * s2;
            const float m20 = q20 * s0, m21 = q21 * s1, m22 = q22 * s2;
            const float S00 = dot3f(m00, m01, m02, m00, m01, m02);
            const float S01 = dot3f(m00, m01, m02, m10, m11, m12);
            const float S02 = dot3f(m00, m01, m02, m20, m21, m22);
            const float S11 = dot3f(m10, m11, m12, m10, m11, m12);
            const float S12 = dot3f(m10, m11, m12, m20, m21, m22);
            const float S22 = dot3f(m20, m21, m22, m20, m21, m22);
            const float A00 = dot3f(V.R[0], V.R[1], V.R[2], S00, S01, S02);
            const float A01 = dot3f(V.R[0], V.R[1], V.R[2], S01, S11, S12);
            const float A02 = dot3f(V.R[0], V.R[1], V.R[2], S02, S12, S22);
            const float A10 = dot3f(V.R[3], V.R[4], V.R[5], S00, S01, S02);
            const float A11 = dot3f(V.R[3], V.R[4], V.R[5], S01, S11, S12);
            const float A12 = dot3f(V.R[3], V.R[4], V.R[5], S02, S12, S22);
            const float A20 = dot3f(V.R[6], V.R[7], V.R[8], S00, S01, S02);
            const float A21 = dot3f(V.R[6], V.R[7], V.R[8], S01, S11, S12);
            const float A22 = dot3f(V.R[6], V.R[7], V.R[8], S02, S12, S22);
            const float C00 = dot3f(A00, A01, A02, V.R[0], V.R[1], V.R[2]);
            const float C01 = dot3f(A00, A01, A02, V.R[3], V.R[4], V.R[5]);
            const float C02 = dot3f(A00, A01, A02, V.R[6], V.R[7], V.R[8]);
            const float C11 = dot3f(A10, A11, A12, V.R[3], V.R[4], V.R[5]);
            const float C12 = dot3f(A10, A11, A12, V.R[6], V.R[7], V.R[8]);
            const float C22 = dot3f(A20, A21, A22, V.R[6], V.R[7], V.R[8]);

            const float Wf = (float)V.W, Hf = (float)V.H;
            const float tan_fovx = 0.5f * Wf / V.fx, tan_fovy = 0.5f * Hf / V.fy;
            const float lim_x_pos = (Wf - V.cx) / V.fx + kClampMargin * tan_fovx;
            const float lim_x_neg = V.cx / V.fx + kClampMargin * tan_fovx;
            const float lim_y_pos = (Hf - V.cy) / V.fy + kClampMargin * tan_fovy;
            const float lim_y_neg = V.cy / V.fy + kClampMargin * tan_fovy;
            const float rz = 1.0f / z;
            const float rz2 = rz * rz;
            const float txc = z * __builtin_fminf(lim_x_pos, __builtin_fmaxf(-lim_x_neg, x * rz));
            const float tyc = z * __builtin_fminf(lim_y_pos, __builtin_fmaxf(-lim_y_neg, y * rz));
            const float J00 = V.fx * rz, J02 = -(V.fx * txc * rz2);
            const float J11 = V.fy * rz, J12 = -(V.fy * tyc * rz2);
            const float B00 = __builtin_fmaf(C02, J02, C00 * J00);
            const float B02 = __builtin_fmaf(C22, J02, C02 * J00);
            const float B10 = __builtin_fmaf(C02, J12, C01 * J11);
            const float B11 = __builtin_fmaf(C12, J12, C11 * J11);
            const float B12 = __builtin_fmaf(C22, J12, C12 * J11);
            float c00 = __builtin_fmaf(J02, B02, J00 * B00);
            const float c01 = __builtin_fmaf(J02, B12, J00 * B10);
            float c11 = __builtin_fmaf(J12, B12, J11 * B11);
            const float u = __builtin_fmaf(V.fx, x * rz, V.cx);
            const float v = __builtin_fmaf(V.fy, y * rz, V.cy);

            c00 += V.eps2d;
            c11 += V.eps2d;
            const float det = c00 * c11 - c01 * c01;
            ok = det > 0.f;
            if (ok) {
                const float inv_det = 1.0f / det;
                const float b = 0.5f * (c00 + c11);
                const float v1 = b + __builtin_sqrtf(__builtin_fmaxf(kRadiusFloor, b * b - det));
                const float radf = __builtin_ceilf(3.f * __builtin_sqrtf(v1));
                ok = (radf > V.radius_clip) && (radf < 1.0e9f);
                ok = ok && !(u + radf <= 0.f || u - radf >= Wf || v + radf <= 0.f || v - radf >= Hf);
                if (ok) {
                    const float ts = (float)kTile;
                    const float tr = radf / ts, tcx = u / ts, tcy = v / ts;
                    const float twf = (float)V.tile_w, thf = (float)V.tile_h;
                    const float fminx = __builtin_fminf(__builtin_fmaxf(__builtin_floorf(tcx - tr), 0.f), twf);
                    const float fminy = __builtin_fminf(__builtin_fmaxf(__builtin_floorf(tcy - tr), 0.f), thf);
                    const float fmaxx = __builtin_fminf(__builtin_fmaxf(__builtin_ceilf(tcx + tr), 0.f), twf);
                    const float fmaxy = __builtin_fminf(__builtin_fmaxf(__builtin_ceilf(tcy + tr), 0.f), thf);
                    u32 x0 = (u32)fminx, y0 = (u32)fminy, x1 = (u32)fmaxx, y1 = (u32)fmaxy;
                    if (tight) {
                        // GWBP_FLAG_TIGHT_BINNING: alpha = o exp(-sigma) >= 1/255 needs sigma <= L = ln(255 o), and
                        // sigma >= dx^2 / (2 Sxx) for every dy (Sxx = c00, the 2-D covariance), so a pixel centre farther
                        // than sqrt(2 L c00) from the mean in x (c11 in y) cannot contribute.  Same bound, same 5 % + 1 px
                        // margin and same 1e-3 slack on L as the strip mask of k_blend (which the parity tests pin).
                        const float L = __logf(255.0f * opac[i]) + 1e-3f;
                        if (!(L > 0.f)) {
                            x1 = x0, y1 = y0; // o <= 1/255: never reaches alpha >= 1/255
                        } else {
                            const float ex = 1.05f * __builtin_sqrtf(2.0f * L * c00) + 1.0f;
                            const float ey = 1.05f * __builtin_sqrtf(2.0f * L * c11) + 1.0f;
                            if (ex == ex && ey == ey) { // degenerate values: keep the 3-sigma square
                                // pixel centres are at integer + 0.5: tiles whose centres fall inside [m - e, m + e]
                                const float lx = __builtin_floorf((u - ex - 0.5f) / ts), hx = __builtin_floorf((u + ex - 0.5f) / ts) + 1.0f;
                                const float ly = __builtin_floorf((v - ey - 0.5f) / ts), hy = __builtin_floorf((v + ey - 0.5f) / ts) + 1.0f;
                                x0 = max(x0, (u32)__builtin_fmaxf(lx, 0.f)), y0 = max(y0, (u32)__builtin_fmaxf(ly, 0.f));
                                x1 = min(x1, (u32)__builtin_fmaxf(hx, 0.f)), y1 = min(y1, (u32)__builtin_fmaxf(hy, 0.f));
                                if (x1 < x0)
                                    x1 = x0;
                                if (y1 < y0)
                                    y1 = y0;
                            }
                        }
                    }
                    g.mx = u, g.my = v, g.opac = opac[i], g.depth = z;
                    g.ca = c11 * inv_det, g.cb = -c01 * inv_det, g.cc = c00 * inv_det;
                    g.radius = (int)radf;
                    rc = make_uint2(x0 | (x1 << 16), y0 | (y1 << 16));
                    ntiles = (x1 - x0) * (y1 - y0);
                }
            }
        }
        // two 16-B stores per lane, contiguous 2 KB per wave
        float4 *gp = reinterpret_cast<float4 *>(g2d + i);
        gp[0] = make_float4(g.mx, g.my, g.opac, g.depth);
        gp[1] = make_float4(g.ca, g.cb, g.cc, __int_as_float(g.radius));
        rect[i] = rc;
        touched[i] = ntiles;
        // depth-sort key: positive float bits order like unsigned integers; culled Gaussians go last
        dkeys[i] = g.radius > 0 ? (u32)__float_as_int(g.depth) : 0xFFFFFFFFu;
        dvals[i] = (u32)i;
        if (o_radii)
            o_radii[i] = g.radius;
        if (o_means2d)
            o_means2d[2 * i] = g.mx, o_means2d[2 * i + 1] = g.my;
        if (o_depths)
            o_depths[i] = g.depth;
        if (o_conics)
            o_conics[3 * i] = g.ca, o_conics[3 * i + 1] = g.cb, o_conics[3 * i + 2] = g.cc;
    }

    // visible-Gaussian count
    __shared__ u32 s_vis[kScanBlock / 64];
    const u64 visb = __ballot(g.radius > 0);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        s_vis[wave] = (u32)__popcll(visb);
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 vis = 0;
#pragma unroll
        for (int w = 0; w < kScanBlock / 64; ++w)
            vis += s_vis[w];
        if (vis)
            atomicAdd(&ctr->n_visible, vis);
    }
}

// Block totals of tiles touched, taken in DEPTH-SORTED Gaussian order (input of the scan that places the emit).
__global__ __launch_bounds__(kScanBlock) void k_sorted_blocksums(int64_t N, const u32 *__restrict__ order,
                                                                 const u32 *__restrict__ touched,
                                                                 u32 *__restrict__ blocksums, int prio)
{
    front_priority(prio);
    const int64_t i = (int64_t)blockIdx.x * kScanBlock + threadIdx.x;
    u32 v = (i < N) ? touched[order[i]] : 0u;
    __shared__ u32 s_sum[kScanBlock / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0)
        s_sum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 tot = 0;
#pragma unroll
        for (int w = 0; w < kScanBlock / 64; ++w)
            tot += s_sum[w];
        blocksums[blockIdx.x] = tot;
    }
}

// Exclusive scan of the per-block totals (single workgroup, carry loop); publishes n_isect.  256 threads, not 1024: a
// 16-wave workgroup needs four free wave slots on every SIMD of ONE CU at the same moment, and beside the persistent
// scatter workgroups the dispatcher can wait milliseconds for that (2.9 ms measured with a 12-wave scatter kernel).
constexpr int kSumThreads = 256;
__global__ __launch_bounds__(kSumThreads) void k_scan_blocksums(int nblk, u32 *__restrict__ blocksums,
                                                                Counters *__restrict__ ctr, u32 isect_cap, int prio)
{
    front_priority(prio);
    __shared__ u32 s_wave[kSumThreads / 64];
    __shared__ u32 s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0)
        s_carry = 0;
    __syncthreads();
    u64 total = 0; // tracked by the last thread in 64 bits to detect wrap
    for (int base = 0; base < nblk; base += kSumThreads) {
        const int i = base + threadIdx.x;
        const u32 v = (i < nblk) ? blocksums[i] : 0u;
        u32 incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 t = __shfl_up(incl, o, 64);
            if (lane >= o)
                incl += t;
        }
        if (lane == 63)
            s_wave[wave] = incl;
        __syncthreads();
        u32 woff = 0;
        for (int w = 0; w < wave; ++w)
            woff += s_wave[w];
        const u32 carry = s_carry;
        if (i < nblk)
            blocksums[i] = carry + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == kSumThreads - 1) {
            total += (u64)woff + incl;
            s_carry = carry + woff + incl;
        }
        __syncthreads();
    }
    if (threadIdx.x == kSumThreads - 1) {
        if (total > (u64)isect_cap) {
            atomicOr(&ctr->overflow, 1u);
            ctr->n_isect = 0; // downstream stages see an empty view; caller must retry with larger caps
        } else {
            ctr->n_isect = (u32)total;
        }
    }
}

// isect_tiles emit in depth-sorted Gaussian order: key = tile id, value = Gaussian index.  Because the Gaussians are
// visited front to back (ties in ascending index), a STABLE sort by tile id alone afterwards yields gsplat's order
// (tile, depth, index) -- 2 byte-passes over the intersections instead of 6 over 64-bit keys.
__global__ __launch_bounds__(kScanBlock) void k_emit(int64_t N, int tile_w, const u32 *__restrict__ order,
                                                     const uint2 *__restrict__ rect,
                                                     const u32 *__restrict__ touched,
                                                     const u32 *__restrict__ blocksums,
                                                     const Counters *__restrict__ ctr, u32 *__restrict__ keys,
                                                     u32 *__restrict__ vals, u32 *__restrict__ estart, int prio)
{
    front_priority(prio);
    if (ctr->overflow & 1u)
        return;
    const int64_t i = (int64_t)blockIdx.x * kScanBlock + threadIdx.x;
    const u32 gid = (i < N) ? order[i] : 0u;
    const u32 cnt = (i < N) ? touched[gid] : 0u;
    __shared__ u32 s_wave[kScanBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t = __shfl_up(incl, o, 64);
        if (lane >= o)
            incl += t;
    }
    if (lane == 63)
        s_wave[wave] = incl;
    __syncthreads();
    u32 woff = 0;
    for (int w = 0; w < wave; ++w)
        woff += s_wave[w];
    if (cnt == 0)
        return;
    u32 pos = blocksums[blockIdx.x] + woff + incl - cnt;
    // where this Gaussian's intersections start: they are written contiguously, row-major over its tile rectangle, so a consumer
    // that knows (gid, tile) finds the emit position by arithmetic -- k_blend<kToken> files its per-record weight sums there and
    // k_token_apply reads a Gaussian's sums back to back (token.hip)
    estart[gid] = pos;
    const uint2 rc = rect[gid];
    const u32 x0 = rc.x & 0xFFFFu, x1 = rc.x >> 16, y0 = rc.y & 0xFFFFu, y1 = rc.y >> 16;
    for (u32 ty = y0; ty < y1; ++ty)
        for (u32 tx = x0; tx < x1; ++tx) {
            keys[pos] = ty * (u32)tile_w + tx;
            vals[pos] = gid;
            ++pos;
        }
}

int launch_emit(const Layout &L, const Ws &W, const ViewDev &V, const u32 *order, hipStream_t s)
{
    const int prio = (L.flags & GWBP_FLAG_FRONT_PRIORITY) ? 1 : 0;
    hipLaunchKernelGGL(k_sorted_blocksums, dim3(L.n_scan_blocks), dim3(kScanBlock), 0, s, L.n, order, W.touched,
                       W.blocksums, prio);
    hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(kSumThreads), 0, s, L.n_scan_blocks, W.blocksums, W.counters,
                       (u32)L.isect_cap, prio);
    hipLaunchKernelGGL(k_emit, dim3(L.n_scan_blocks), dim3(kScanBlock), 0, s, L.n, V.tile_w, order, W.rect, W.touched,
                       W.blocksums, W.counters, W.keys[0], W.vals[0], W.dkeys[1], prio);
    return check_hip(hipGetLastError(), "emit launch");
}

// k_emit alone: blocksums and Counters::n_isect are already in place (k_sort_small's tail)
int launch_emit_scanned(const Layout &L, const Ws &W, const ViewDev &V, const u32 *order, hipStream_t s)
{
    const int prio = (L.flags & GWBP_FLAG_FRONT_PRIORITY) ? 1 : 0;
    hipLaunchKernelGGL(k_emit, dim3(L.n_scan_blocks), dim3(kScanBlock), 0, s, L.n, V.tile_w, order, W.rect, W.touched,
                       W.blocksums, W.counters, W.keys[0], W.vals[0], W.dkeys[1], prio);
    return check_hip(hipGetLastError(), "emit launch");
}

int launch_project(const Layout &L, const Ws &W, const ViewDev &V, const float *means, const float *quats,
                   const float *scales, const float *opac, int32_t *radii, float *means2d, float *depths,
                   float *conics, hipStream_t s)
{
    const int prio = (L.flags & GWBP_FLAG_FRONT_PRIORITY) ? 1 : 0;
    // counters and the pool shard heads / scatter queues are adjacent sub-buffers (g2d follows): one memset node
    int rc = check_hip(hipMemsetAsync(W.counters, 0, L.g2d - L.counters, s),
                       "memset counters");
    if (rc)
        return rc;
    if (L.n == 0)
        return GWBP_OK;
    hipLaunchKernelGGL(k_project, dim3(L.n_scan_blocks), dim3(kScanBlock), 0, s, L.n, V, means, quats, scales, opac,
                       W.g2d, W.rect, W.touched, W.dkeys[0], W.dvals[0], W.counters, radii, means2d, depths, conics,
                       L.flags & GWBP_FLAG_TIGHT_BINNING, prio);
    return check_hip(hipGetLastError(), "project launch");
}

} // namespace gwbp
