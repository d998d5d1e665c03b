// capi.hip -- extern "C" entry points of libgwbp.so (see include/gwbp.h for the contract).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gwbp_dev.h"

namespace gwbp {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_hip(hipError_t e, const char *what)
{
    if (e == hipSuccess)
        return GWBP_OK;
    set_error((int)e, "%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

namespace {
struct DevCache {
    int n_cu;
    unsigned lds_mask;
};
DevCache g_dev[64]; // benign races: every writer stores the same value
DevCache *dev_cache()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return nullptr;
    return &g_dev[dev];
}
} // namespace

int device_cus(int *n_cu)
{
    DevCache *c = dev_cache();
    if (!c)
        return set_error(GWBP_EINVAL, "cannot query the current device");
    if (c->n_cu == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            return set_error(GWBP_EINVAL, "cannot query the device for the persistent scatter grid");
        c->n_cu = v > 0 ? v : 256;
    }
    *n_cu = c->n_cu;
    return GWBP_OK;
}

int ensure_dynamic_lds(const void *func, int bytes, int slot)
{
    DevCache *c = dev_cache();
    if (!c)
        return set_error(GWBP_EINVAL, "cannot query the current device");
    if (c->lds_mask & (1u << slot))
        return GWBP_OK;
    const int rc = check_hip(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes),
                             "dynamic LDS attribute");
    if (rc == GWBP_OK)
        c->lds_mask |= 1u << slot;
    return rc;
}

int profile_knob(const char *name)
{
#ifdef GWBP_PROFILE
    const char *v = getenv(name);
    return v ? atoi(v) : 0;
#else
    (void)name;
    return 0;
#endif
}

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

int make_layout(const gwbp_caps *c, Layout *L)
{
    if (!c || !L)
        return set_error(GWBP_EINVAL, "null caps");
    if (c->n_gaussians < 0 || c->n_gaussians > 0x7FFFFFFFll || c->isect_cap < 1 || c->isect_cap > 0xFFFFF000ll ||
        c->pair_cap < (int64_t)kPage * kShards || c->pair_cap > 0xFFFFF000ll || c->max_width < 1 || c->max_height < 1 ||
        c->max_width > 65535 * kTile || c->max_height > 65535 * kTile)
        return set_error(GWBP_EINVAL, "caps out of range (N=%lld isect_cap=%lld pair_cap=%lld %dx%d)",
                         (long long)c->n_gaussians, (long long)c->isect_cap, (long long)c->pair_cap, c->max_width,
                         c->max_height);
    if (c->flags & ~(GWBP_FLAG_TIGHT_BINNING | GWBP_FLAG_FRONT_PRIORITY | GWBP_FLAG_NARROW_SCATTER | GWBP_FLAG_SPLIT_ENCODER))
        return set_error(GWBP_EINVAL, "unknown caps.flags bits 0x%x", (unsigned)c->flags);
    memset(L, 0, sizeof(*L));
    L->n = c->n_gaussians;
    L->isect_cap = c->isect_cap;
    L->pair_cap = c->pair_cap;
    L->scatter_wgs = c->scatter_workgroups > 0 ? c->scatter_workgroups : 0;
    L->flags = c->flags;
    const int tw = (c->max_width + kTile - 1) / kTile, th = (c->max_height + kTile - 1) / kTile;
    L->max_tiles = tw * th;
    L->n_scan_blocks = (int)((L->n + kScanBlock - 1) / kScanBlock);
    L->n_sort_blocks = (int)(((L->isect_cap > L->n ? L->isect_cap : L->n) + kSortItems - 1) / kSortItems);
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t at = o;
        o += align_up(bytes ? bytes : 1);
        return at;
    };
    L->counters = take(sizeof(Counters));
    L->shards = take((size_t)(kShards + kQueues) * 64);
    L->sweep = take((size_t)kSweepWords * sizeof(u32)); // (inside the range gwbp_project's memset clears: counters .. g2d)
    L->g2d = take((size_t)L->n * sizeof(G2D));
    L->rect = take((size_t)L->n * sizeof(uint2));
    L->touched = take((size_t)L->n * sizeof(u32));
    L->blocksums = take((size_t)(L->n_scan_blocks + 1) * sizeof(u32));
    for (int i = 0; i < 2; ++i) {
        L->dkeys[i] = take((size_t)L->n * sizeof(u32));
        L->dvals[i] = take((size_t)L->n * sizeof(u32));
    }
    L->keys[0] = take((size_t)L->isect_cap * sizeof(u32));
    L->keys[1] = take((size_t)L->isect_cap * sizeof(u32));
    L->vals[0] = take((size_t)L->isect_cap * sizeof(u32));
    L->vals[1] = take((size_t)L->isect_cap * sizeof(u32));
#ifdef GWBP_SORT_ONESWEEP
    // two look-back status buffers [block][256 digits] of 8-B words, used alternately by the passes of a sort level
    L->hist = take((size_t)2 * 256 * L->n_sort_blocks * sizeof(u64));
#else
    L->hist = take((size_t)256 * L->n_sort_blocks * sizeof(u32));
#endif
    L->digit_total = take(256 * sizeof(u32));
    L->tile_offsets = take((size_t)(L->max_tiles + 1) * sizeof(u32));
    L->tile_order = take((size_t)L->max_tiles * sizeof(u32));
    L->hdr_count = take((size_t)L->max_tiles * sizeof(u32));
    L->headers = take((size_t)L->isect_cap * sizeof(Header));
    L->carry = take((size_t)kCarryWgs * kCarryRows * 256 * sizeof(float));
    // + 128 entries of slack behind the pool: k_scatter_wide's L2 warm-up touches one dword per 128-B line of a visit's run and
    // may reach one line past its end (its scalar batch loads stay inside the record's padded lists), and k_render_rows4 reads
    // 16 entries from a visit's first one whatever the visit's length (15 entries = 120 B past the last record at most)
    L->wpool = take((size_t)L->pair_cap * sizeof(WPair) + 1024);
    L->total = o;
    return GWBP_OK;
}

int bind_workspace(const gwbp_caps *caps, void *ws, size_t bytes, Layout *L, Ws *W)
{
    int rc = make_layout(caps, L);
    if (rc)
        return rc;
    if (!ws)
        return set_error(GWBP_EINVAL, "null workspace");
    if ((reinterpret_cast<uintptr_t>(ws) & 255) != 0)
        return set_error(GWBP_EINVAL, "workspace must be 256-B aligned");
    if (bytes < L->total)
        return set_error(GWBP_EWORKSPACE, "workspace too small: have %zu, need %zu", bytes, L->total);
    char *b = static_cast<char *>(ws);
    W->counters = reinterpret_cast<Counters *>(b + L->counters);
    W->shards = reinterpret_cast<u32 *>(b + L->shards);
    W->sweep = reinterpret_cast<u32 *>(b + L->sweep);
    W->g2d = reinterpret_cast<G2D *>(b + L->g2d);
    W->rect = reinterpret_cast<uint2 *>(b + L->rect);
    W->touched = reinterpret_cast<u32 *>(b + L->touched);
    W->blocksums = reinterpret_cast<u32 *>(b + L->blocksums);
    for (int i = 0; i < 2; ++i) {
        W->dkeys[i] = reinterpret_cast<u32 *>(b + L->dkeys[i]);
        W->dvals[i] = reinterpret_cast<u32 *>(b + L->dvals[i]);
    }
    W->keys[0] = reinterpret_cast<u32 *>(b + L->keys[0]);
    W->keys[1] = reinterpret_cast<u32 *>(b + L->keys[1]);
    W->vals[0] = reinterpret_cast<u32 *>(b + L->vals[0]);
    W->vals[1] = reinterpret_cast<u32 *>(b + L->vals[1]);
    W->hist = reinterpret_cast<u32 *>(b + L->hist);
    W->digit_total = reinterpret_cast<u32 *>(b + L->digit_total);
    W->tile_offsets = reinterpret_cast<u32 *>(b + L->tile_offsets);
    W->tile_order = reinterpret_cast<u32 *>(b + L->tile_order);
    W->hdr_count = reinterpret_cast<u32 *>(b + L->hdr_count);
    W->headers = reinterpret_cast<Header *>(b + L->headers);
    W->carry = reinterpret_cast<float *>(b + L->carry);
    W->wpool = reinterpret_cast<WPair *>(b + L->wpool);
    return GWBP_OK;
}

int make_view(const gwbp_view *v, const gwbp_caps *caps, ViewDev *o)
{
    if (!v)
        return set_error(GWBP_EINVAL, "null view");
    if (v->width < 1 || v->height < 1 || v->width > caps->max_width || v->height > caps->max_height)
        return set_error(GWBP_EINVAL, "view %dx%d outside caps %dx%d", v->width, v->height, caps->max_width,
                         caps->max_height);
    const int tw = (v->width + kTile - 1) / kTile, th = (v->height + kTile - 1) / kTile;
    const int max_tiles = ((caps->max_width + kTile - 1) / kTile) * ((caps->max_height + kTile - 1) / kTile);
    if (tw * th > max_tiles)
        return set_error(GWBP_EINVAL, "view has more tiles than the caps allow");
    if (!(v->K[0] > 0.f) || !(v->K[4] > 0.f))
        return set_error(GWBP_EINVAL, "focal lengths must be positive");
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
            o->R[3 * r + c] = v->viewmat[4 * r + c];
        o->t[r] = v->viewmat[4 * r + 3];
    }
    o->fx = v->K[0], o->fy = v->K[4], o->cx = v->K[2], o->cy = v->K[5];
    o->W = v->width, o->H = v->height, o->tile_w = tw, o->tile_h = th;
    o->near_plane = v->near_plane, o->far_plane = v->far_plane, o->eps2d = v->eps2d, o->radius_clip = v->radius_clip;
    return GWBP_OK;
}

static int check_feats(const float *feats, int64_t fs_y, int64_t fs_x, int64_t fs_c, int D)
{
    if (!feats || D < 1 || fs_y < 0 || fs_x < 0 || fs_c < 0)
        return set_error(GWBP_EINVAL, "bad feature map arguments (D=%d strides %lld %lld %lld)", D, (long long)fs_y,
                         (long long)fs_x, (long long)fs_c);
    return GWBP_OK;
}

} // namespace gwbp

using namespace gwbp;

extern "C" {

const char *gwbp_version(void)
{
#ifdef GWBP_PROFILE
    return "libgwbp gfx950 r6 (PROFILE build: ablation knobs live, results may be invalid)";
#else
    return "libgwbp gfx950 r6";
#endif
}
const char *gwbp_last_error_string(void) { return g_err; }

int gwbp_workspace_size(const gwbp_caps *caps, size_t *bytes_host)
{
    Layout L;
    int rc = make_layout(caps, &L);
    if (rc)
        return rc;
    if (!bytes_host)
        return set_error(GWBP_EINVAL, "null bytes_host");
    *bytes_host = L.total;
    return GWBP_OK;
}

int gwbp_project(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                 const float *means, const float *quats, const float *scales, const float *opacities,
                 int32_t *radii, float *means2d, float *depths, float *conics, void *stream)
{
    Layout L;
    Ws W;
    ViewDev V;
    int rc = bind_workspace(caps, workspace, workspace_bytes, &L, &W);
    if (rc)
        return rc;
    if ((rc = make_view(view_host, caps, &V)))
        return rc;
    if (L.n > 0 && (!means || !quats || !scales || !opacities))
        return set_error(GWBP_EINVAL, "null Gaussian parameter pointer");
    if (reinterpret_cast<uintptr_t>(quats) & 15)
        return set_error(GWBP_EINVAL, "quats must be 16-B aligned");
    return launch_project(L, W, V, means, quats, scales, opacities, radii, means2d, depths, conics,
                          static_cast<hipStream_t>(stream));
}

int gwbp_bin_sort(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                  int64_t *isect_ids, int32_t *flatten_ids, int32_t *tile_offsets, void *stream)
{
    Layout L;
    Ws W;
    ViewDev V;
    int rc = bind_workspace(caps, workspace, workspace_bytes, &L, &W);
    if (rc)
        return rc;
    if ((rc = make_view(view_host, caps, &V)))
        return rc;
    return launch_bin_sort(L, W, V, isect_ids, flatten_ids, tile_offsets, static_cast<hipStream_t>(stream));
}

int gwbp_blend_weights(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                       float *alphas, void *stream)
{
    Layout L;
    Ws W;
    ViewDev V;
    int rc = bind_workspace(caps, workspace, workspace_bytes, &L, &W);
    if (rc)
        return rc;
    if ((rc = make_view(view_host, caps, &V)))
        return rc;
    return launch_blend(L, W, V, alphas, nullptr, 0.f, static_cast<hipStream_t>(stream));
}

int gwbp_blend_weights_d(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                         float *alphas, float scale_d, float *d, void *stream)
{
    Layout L;
    Ws W;
    ViewDev V;
    int rc = bind_workspace(caps, workspace, workspace_bytes, &L, &W);
    if (rc)
        return rc;
    if ((rc = make_view(view_host, caps, &V)))
        return rc;
    if (!d)
        return set_error(GWBP_EINVAL, "null d");
    return launch_blend(L, W, V, alphas, d, scale_d, static_cast<hipStream_t>(stream));
}

int gwbp_blend_scatter(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                       const float *feats, int64_t fs_y, int64_t fs_x, int32_t D, float scale_f, float scale_d, float *F,
                       float *d, float *alphas, void *stream)
{
    Layout L;
    Ws W;
    ViewDev V;
    int rc = bind_workspace(caps, workspace, workspace_bytes, &L, &W);
    if (rc)
        return rc;
    if ((rc = make_view(view_host, caps, &V)))
        return rc;
    const FeatMap M{feats, fs_y, fs_x, 1, nullptr, nullptr, nullptr, nullptr, 0, 0};
    return launch_blend(L, W, V, alphas, d, scale_d, static_cast<hipStream_t>(stream), &M, D, scale_f, F);
}

int gwbp_blend_scatter_encoded(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                               const float *feats, int64_t fs_y, int64_t fs_x, int32_t K, const float *encoder, int32_t n_out,
                               float scale_f, float scale_d, float *F, float *d, float *alphas, void *stream)
{
    Layout L;
    Ws W;
    ViewDev V;
    int rc = bind_workspace(caps, workspace, workspace_bytes, &L, &W);
    if (rc)
        return rc;
    if ((rc = make_view(view_host, caps, &V)))
        return rc;
    if (!encoder)
        return set_error(GWBP_EINVAL, "null encoder");
    FeatMap M{feats, fs_y, fs_x, 1, nullptr, nullptr, nullptr, nullptr, 0, 0};
    M.enc = encoder, M.enc_k = K;
    return launch_blend(L, W, V, alphas, d, scale_d, static_cast<hipStream_t>(stream), &M, n_out, scale_f, F);
}

int gwbp_blend_tokens(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                      const int32_t *ymap, const int32_t *xmap, float *alphas, void *stream)
{
    Layout L;
    Ws W;
    ViewDev V;
    int rc = bind_workspace(caps, workspace, workspace_bytes, &L, &W);
    if (rc)
        return rc;
    if ((rc = make_view(view_host, caps, &V)))
        return rc;
    return launch_blend_tokens(L, W, V, alphas, ymap, xmap, static_cast<hipStream_t>(stream));
}

int gwbp_scatter_tokens(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                        const float *tokens, int64_t ts_y, int64_t ts_x, int32_t D, const int32_t *ymap, const int32_t *xmap,
                        float scale_f, float scale_d, float *F, float *d, void *stream)
{
    Layout L;
    Ws W;
    ViewDev V;
    int rc = bind_workspace(caps, workspace, workspace_bytes, &L, &W);
    if (rc)
        return rc;
    if ((rc = make_view(view_host, caps, &V)))
        return rc;
    return launch_token_apply(L, W, V, tokens, ts_y, ts_x, D, ymap, xmap, scale_f, scale_d, F, d, static_cast<hipStream_t>(stream));
}

int gwbp_accumulate_d(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                      float scale_d, float *d, void *stream)
{
    Layout L;
    Ws W;
    ViewDev V;
    int rc = bind_workspace(caps, workspace, workspace_bytes, &L, &W);
    if (rc)
        return rc;
    if ((rc = make_view(view_host, caps, &V)))
        return rc;
    if (!d)
        return set_error(GWBP_EINVAL, "null d");
    return launch_accum_d(L, W, V, scale_d, d, static_cast<hipStream_t>(stream));
}

static int scatter_impl(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                        const FeatMap &M, int32_t D, float scale_f, float scale_d, float *F, float *d, void *stream)
{
    Layout L;
    Ws W;
    ViewDev V;
    int rc = bind_workspace(caps, workspace, workspace_bytes, &L, &W);
    if (rc)
        return rc;
    if ((rc = make_view(view_host, caps, &V)))
        return rc;
    if ((rc = check_feats(M.p, M.fs_y, M.fs_x, M.fs_c, D)))
        return rc;
    if (!F && L.n > 0)
        return set_error(GWBP_EINVAL, "null F");
    return launch_scatter(L, W, V, M, D, scale_f, scale_d, F, d, static_cast<hipStream_t>(stream));
}

int gwbp_scatter(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                 const float *feats, int64_t fs_y, int64_t fs_x, int64_t fs_c, int32_t D, float scale_f,
                 float scale_d, float *F, float *d, void *stream)
{
    const FeatMap M{feats, fs_y, fs_x, fs_c, nullptr, nullptr, nullptr, nullptr, 0, 0};
    return scatter_impl(caps, workspace, workspace_bytes, view_host, M, D, scale_f, scale_d, F, d, stream);
}

int gwbp_scatter_encoded(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                         const float *feats, int64_t fs_y, int64_t fs_x, int32_t K, const float *encoder, int32_t D,
                         float scale_f, float scale_d, float *F, float *d, void *stream)
{
    if (!encoder)
        return set_error(GWBP_EINVAL, "gwbp_scatter_encoded needs the encoder");
    FeatMap M{feats, fs_y, fs_x, 1, nullptr, nullptr, nullptr, nullptr, 0, 0};
    M.enc = encoder, M.enc_k = K;
    return scatter_impl(caps, workspace, workspace_bytes, view_host, M, D, scale_f, scale_d, F, d, stream);
}

int gwbp_scatter_upsampled(const gwbp_caps *caps, void *workspace, size_t workspace_bytes,
                           const gwbp_view *view_host, const float *feats, int64_t fs_y, int64_t fs_x, int64_t fs_c,
                           int32_t D, const int32_t *ymap, const int32_t *xmap, float scale_f, float scale_d, float *F,
                           float *d, void *stream)
{
    if (!ymap || !xmap)
        return set_error(GWBP_EINVAL, "gwbp_scatter_upsampled needs both index maps");
    const FeatMap M{feats, fs_y, fs_x, fs_c, ymap, xmap, nullptr, nullptr, 0, 0};
    return scatter_impl(caps, workspace, workspace_bytes, view_host, M, D, scale_f, scale_d, F, d, stream);
}

int gwbp_scatter_bilinear(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                          const float *feats, int64_t fs_y, int64_t fs_x, int64_t fs_c, int32_t D, int32_t lr_h,
                          int32_t lr_w, const int32_t *y0, const float *ly, const int32_t *x0, const float *lx,
                          float scale_f, float scale_d, float *F, float *d, void *stream)
{
    if (!y0 || !x0 || !ly || !lx || lr_h < 1 || lr_w < 1)
        return set_error(GWBP_EINVAL, "gwbp_scatter_bilinear needs both index maps, both weight maps and the map size");
    const FeatMap M{feats, fs_y, fs_x, fs_c, y0, x0, ly, lx, lr_h, lr_w};
    return scatter_impl(caps, workspace, workspace_bytes, view_host, M, D, scale_f, scale_d, F, d, stream);
}

int gwbp_render(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                const float *colors, int32_t D, float *out, void *stream)
{
    Layout L;
    Ws W;
    ViewDev V;
    int rc = bind_workspace(caps, workspace, workspace_bytes, &L, &W);
    if (rc)
        return rc;
    if ((rc = make_view(view_host, caps, &V)))
        return rc;
    if (!colors || !out || D < 1)
        return set_error(GWBP_EINVAL, "bad render arguments");
    return launch_render(L, W, V, colors, D, out, static_cast<hipStream_t>(stream));
}

int gwbp_render_pixels(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                       const float *colors, int32_t D, float *out, float *alphas, void *stream)
{
    Layout L;
    Ws W;
    ViewDev V;
    int rc = bind_workspace(caps, workspace, workspace_bytes, &L, &W);
    if (rc)
        return rc;
    if ((rc = make_view(view_host, caps, &V)))
        return rc;
    if (!colors || !out || D < 1 || D > 32)
        return set_error(GWBP_EINVAL, "render_pixels needs colors, out and 1 <= D <= 32 (got D=%d)", D);
    return launch_render_px(W, V, colors, D, out, alphas, static_cast<hipStream_t>(stream));
}

int gwbp_sh_colors(int64_t N, int32_t degree, int32_t K, const float *means, const float *coeffs,
                   const float *campos_host, float *out, void *stream)
{
    if (N < 0 || degree < 0 || degree > 3 || K < (degree + 1) * (degree + 1) || !campos_host ||
        (N > 0 && (!means || !coeffs || !out)))
        return set_error(GWBP_EINVAL, "bad sh_colors arguments (N=%lld degree=%d K=%d)", (long long)N, degree, K);
    return launch_sh_colors(N, degree, K, means, coeffs, campos_host, out, static_cast<hipStream_t>(stream));
}

int gwbp_backproject_view(const gwbp_caps *caps, void *workspace, size_t workspace_bytes,
                          const gwbp_view *view_host, const float *means, const float *quats, const float *scales,
                          const float *opacities, const float *feats, int64_t fs_y, int64_t fs_x, int64_t fs_c,
                          int32_t D, float scale_f, float scale_d, float *F, float *d, void *stream)
{
    Layout L;
    Ws W;
    ViewDev V;
    int rc = bind_workspace(caps, workspace, workspace_bytes, &L, &W);
    if (rc)
        return rc;
    if ((rc = make_view(view_host, caps, &V)))
        return rc;
    if ((rc = check_feats(feats, fs_y, fs_x, fs_c, D)))
        return rc;
    if (!F && L.n > 0)
        return set_error(GWBP_EINVAL, "null F");
    if (L.n > 0 && (!means || !quats || !scales || !opacities))
        return set_error(GWBP_EINVAL, "null Gaussian parameter pointer");
    if (reinterpret_cast<uintptr_t>(quats) & 15)
        return set_error(GWBP_EINVAL, "quats must be 16-B aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((rc = launch_project(L, W, V, means, quats, scales, opacities, nullptr, nullptr, nullptr, nullptr, s)))
        return rc;
    if ((rc = launch_bin_sort(L, W, V, nullptr, nullptr, nullptr, s)))
        return rc;
    if ((rc = launch_blend(L, W, V, nullptr, nullptr, 0.f, s)))
        return rc;
    const FeatMap M{feats, fs_y, fs_x, fs_c, nullptr, nullptr, nullptr, nullptr, 0, 0};
    return launch_scatter(L, W, V, M, D, scale_f, scale_d, F, d, s);
}

int gwbp_encode_map(const float *feats, int64_t fs_y, int64_t fs_x, int32_t height, int32_t width, int32_t K,
                    const float *encoder, int32_t n_out, float *out, int32_t workgroups, void *stream)
{
    if (!feats || !encoder || !out || height < 1 || width < 1 || K < 16 || K % 16 != 0 || K > 2048 || n_out < 1 ||
        n_out > 16 || fs_y < 0 || fs_x < 0)
        return set_error(GWBP_EINVAL, "bad encode_map arguments (%dx%d K=%d n_out=%d): K %% 16 == 0, K <= 2048, n_out <= 16",
                         height, width, K, n_out);
    if ((reinterpret_cast<uintptr_t>(feats) & 15) || (fs_y & 3) || (fs_x & 3))
        return set_error(GWBP_EINVAL, "encode_map needs channel-contiguous pixels at 16-B aligned addresses");
    return launch_encode_map(feats, fs_y, fs_x, height, width, K, encoder, n_out, out, workgroups < 0 ? 0 : workgroups,
                             static_cast<hipStream_t>(stream));
}

int gwbp_finalize(int64_t N, int32_t D, const float *F, const float *d, float *out, void *stream)
{
    if (N < 0 || D < 1 || (N > 0 && (!F || !d || !out)))
        return set_error(GWBP_EINVAL, "bad finalize arguments");
    return launch_finalize(N, D, F, d, out, static_cast<hipStream_t>(stream));
}

int gwbp_accumulate_stats(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, gwbp_stats *accum,
                          void *stream)
{
    Layout L;
    Ws W;
    int rc = bind_workspace(caps, workspace, workspace_bytes, &L, &W);
    if (rc)
        return rc;
    if (!accum)
        return set_error(GWBP_EINVAL, "null accum");
    return launch_accum_stats(W, accum, static_cast<hipStream_t>(stream));
}

int gwbp_read_stats(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, gwbp_stats *stats_host,
                    void *stream)
{
    Layout L;
    Ws W;
    int rc = bind_workspace(caps, workspace, workspace_bytes, &L, &W);
    if (rc)
        return rc;
    if (!stats_host)
        return set_error(GWBP_EINVAL, "null stats_host");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((rc = check_hip(hipMemcpyAsync(stats_host, W.counters, sizeof(gwbp_stats), hipMemcpyDeviceToHost, s),
                        "stats copy")))
        return rc;
    return check_hip(hipStreamSynchronize(s), "stats sync");
}

int gwbp_dump_pairs(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                    int64_t cap, int32_t *gid, int32_t *pix, float *w, int64_t *n_host, void *stream)
{
    Layout L;
    Ws W;
    ViewDev V;
    int rc = bind_workspace(caps, workspace, workspace_bytes, &L, &W);
    if (rc)
        return rc;
    if ((rc = make_view(view_host, caps, &V)))
        return rc;
    if (!n_host || cap < 0 || (cap > 0 && (!gid || !pix || !w)))
        return set_error(GWBP_EINVAL, "bad dump arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // scratch counter: reuse digit_total[0..1] (free once the sort has finished)
    u64 *n_dev = reinterpret_cast<u64 *>(W.digit_total);
    if ((rc = check_hip(hipMemsetAsync(n_dev, 0, sizeof(u64), s), "dump memset")))
        return rc;
    if ((rc = launch_dump_pairs(L, W, V, cap, gid, pix, w, n_dev, s)))
        return rc;
    u64 n = 0;
    if ((rc = check_hip(hipMemcpyAsync(&n, n_dev, sizeof(u64), hipMemcpyDeviceToHost, s), "dump copy")))
        return rc;
    if ((rc = check_hip(hipStreamSynchronize(s), "dump sync")))
        return rc;
    *n_host = (int64_t)n;
    return GWBP_OK;
}

} // extern "C"
