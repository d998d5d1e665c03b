/*
 * gwbp.h -- C ABI of libgwbp.so: gradient-weighted feature back-projection on MI355X (gfx950).
 *
 * This is the drop-in boundary for the one hot path this repository implements.  In the reference
 * (JojiJoseph/3dgs-gradient-backprojection) the boundary is the Python operator
 *     gsplat.rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height, ...)
 * called at backproject.py:89,115,133 (and :223,251,271; backproject_compressed.py:102,129,147;
 * utils.py:238,316,329) followed by autograd backward (backproject.py:129,147).  gsplat's own native
 * boundary is a pybind11/torch extension taking at::Tensor; the replacement is a plain C ABI:
 * raw device pointers, sizes, a hipStream_t passed as void*, int status codes, no C++ types, no torch types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host; fp32 arrays are dense row-major
 *   - the caller owns every buffer, including the workspace; the library allocates nothing persistent and
 *     keeps no global mutable state except a thread-local last-error string and two per-device-ordinal caches of
 *     device facts (CU count; "dynamic-LDS limit raised for kernel k") -- work is launched on the CURRENT HIP device,
 *     which must be the device of `stream` and of every pointer
 *   - no environment variable is read (ablation knobs exist only in a -DGWBP_PROFILE build, `make PROFILE=1`)
 *   - every entry point only ENQUEUES work on `stream` (no host synchronisation) unless documented
 *   - return value: 0 = ok, <0 = GWBP_E* (invalid argument / workspace too small), >0 = hipError_t
 *   - quats are (w,x,y,z) and need not be normalised; scales/opacities are post-activation
 *     (backproject.py:55-57); viewmat is row-major 4x4 world->camera [R|t] (utils.py:215-219);
 *     K is row-major 3x3
 *
 * Thread-safety: re-entrant; one host thread per GPU process is the expected caller.
 */
#ifndef GWBP_H
#define GWBP_H

#include <stddef.h>
#include <stdint.h>

/* The library is built with -fvisibility=hidden: the entry points below are its ONLY dynamic symbols (no C++ symbol of the
 * implementation -- launchers taking hipStream_t, kernel host stubs -- crosses the boundary; tests/test_capi_cpu.py compares
 * `nm -D --defined-only` with this header, both directions). */
#if defined(__GNUC__) || defined(__clang__)
#define GWBP_API __attribute__((visibility("default")))
#else
#define GWBP_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define GWBP_OK 0
#define GWBP_EINVAL (-1)    /* bad argument (null pointer, non-positive size, unsupported tile size, ...) */
#define GWBP_EWORKSPACE (-2) /* workspace smaller than gwbp_workspace_size() reports */
#define GWBP_EUNSUPPORTED (-3)

#define GWBP_TILE 16 /* gsplat tile_size default; the tile rectangle rule is part of the numerics */

/* Per-view parameters (host struct, passed by pointer, copied at call time). Mirrors the keyword defaults of
 * gsplat.rasterization(): near_plane=0.01, far_plane=1e10, radius_clip=0.0, eps2d=0.3, tile_size=16. */
typedef struct gwbp_view {
    float viewmat[16]; /* row-major world->camera */
    float K[9];        /* row-major intrinsics */
    int32_t width, height;
    float near_plane, far_plane, eps2d, radius_clip;
} gwbp_view;

/* Capacities the caller chooses for the variable-size intermediates of one view. */
typedef struct gwbp_caps {
    int64_t n_gaussians; /* N */
    int64_t isect_cap;   /* max (Gaussian, tile) intersections per view */
    int64_t pair_cap;    /* max stored weight-store entries (8 B each) per view, incl. page slack */
    int32_t max_width, max_height;
    /* Tuning: number of persistent scatter workgroups (rounded up to a multiple of 8); 0 = one per CU (default and
     * measured optimum on MI355X, also when the next view's front stages overlap on a second stream). */
    int32_t scatter_workgroups;
    /* GWBP_FLAG_* bits; 0 = gsplat's exact tile binning (3-sigma square, what meta["isect_ids"] must show). */
    int32_t flags;
} gwbp_caps;

/* Bin every Gaussian only into the tiles that the bounding box of its alpha >= 1/255 ellipse touches (clipped to the
 * 3-sigma square): low-opacity and elongated Gaussians enter fewer tile lists.  A dropped (Gaussian, tile) pair has no
 * pixel with alpha >= 1/255 (5 % + 1 px margin), so F, d, the weight store and every render are unchanged bit for bit;
 * only n_isect and the sorted intersection lists shrink.  Used by the fused back-projection path. */
#define GWBP_FLAG_TIGHT_BINNING 1
/* The kernels of gwbp_project / gwbp_bin_sort / gwbp_blend_weights run at raised wave priority.  For callers that run
 * them on a second stream beside gwbp_scatter of the previous view when D % 256 == 0 (the 256-channel scatter kernel
 * leaves issue slots the front can only use with priority; with the 128-channel kernel the flag costs time). */
#define GWBP_FLAG_FRONT_PRIORITY 2
/* gwbp_scatter uses the 128-channel kernel even when D % 256 == 0.  The 256-channel kernel needs fewer vector
 * instructions per (pair, channel) but more work per (Gaussian, tile) record: it wins when records are long (C2: 48
 * pairs per record, -5 % per view) and loses when they are short and the flush atomics dominate (C4: 25 pairs per
 * record, +5 %).  The host decides from the first view's gwbp_stats (n_pairs / n_headers).  gwbp_blend_weights skips
 * the half-tile record lists (15-20 % of its time) when the flag is set, so a view blended WITH the flag must be
 * scattered with it; the other direction (blend without, scatter with) is fine. */
#define GWBP_FLAG_NARROW_SCATTER 4
/* gwbp_blend_scatter_encoded runs its producer / consumer form: ONE persistent workgroup per CU whose two encoder waves stream
 * tile after tile through the matrix cores into a ring of encoded tiles in LDS while its fourteen blend waves drain
 * it -- the HBM-bound encoder stream and the issue-bound blend loops run concurrently for the whole launch instead of one after
 * the other in every wave.  Same encoded pixels and weights bit for bit; F and d differ by summation order only.  Pays on images of
 * many tiles per CU (C5: 6700 tiles on 256 CUs); needs gwbp_project of the view to have run on this workspace (the tile counter
 * is a word its memset clears). */
#define GWBP_FLAG_SPLIT_ENCODER 16
/* (bit 3 was GWBP_FLAG_GROUP_SCATTER, an experimental block-sparse scatter on the matrix cores: measured slower than the
 * vector kernels and removed; unknown flag bits are rejected with GWBP_EINVAL.) */

/* Device-resident per-view counters, readable after the stream has drained (gwbp_read_stats). */
typedef struct gwbp_stats {
    uint64_t n_pairs;     /* contributing (Gaussian, pixel) pairs = sum over pixels of #weights */
    uint32_t n_isect;     /* (Gaussian, tile) intersections emitted */
    uint32_t n_visible;   /* Gaussians surviving projection culling */
    uint32_t n_headers;   /* (Gaussian, tile) pairs with at least one contributing pixel */
    uint32_t pool_used;   /* weight-pool entries a uniform shard capacity would need: kShards x fullest shard */
    uint32_t overflow;    /* bit0: isect_cap exceeded, bit1: pair_cap exceeded -> results of this view invalid;
                           * bit4: internal error -- a wave of gwbp_blend_scatter_encoded's producer / consumer form gave up waiting on
                           * its ring of encoded tiles (the launch ends instead of hanging; the view's result is invalid);
                           * bit3: gwbp_blend_tokens met a tile that spans more than 2 x 2 tokens (precondition violated) -> invalid;
                           * bit2: gwbp_scatter / gwbp_accumulate_d asked for the 256-channel kernel on a view that was
                           * blended WITH GWBP_FLAG_NARROW_SCATTER (no half-tile lists / weight sums): that call left
                           * F and d untouched -- scatter again with the flag set */
    uint32_t reserved;    /* what the last blend of this view left: 0 = weight store, 1 = store + half-tile lists, 2 = nothing (gwbp_blend_scatter), 3 = token-quadrant weight sums (gwbp_blend_tokens) */
} gwbp_stats;

/* Library / build identification ("gfx950;<git-less build tag>"). */
GWBP_API const char *gwbp_version(void);
/* Thread-local description of the last non-zero status returned on this thread. */
GWBP_API const char *gwbp_last_error_string(void);

/* Bytes of workspace needed for the given capacities (256-B aligned sub-buffers). */
GWBP_API int gwbp_workspace_size(const gwbp_caps *caps, size_t *bytes_host);

/* ---- stage entry points (replace the stages inside gsplat.rasterization) -------------------- */

/* fully_fused_projection + tiles-per-Gaussian count.  Writes the projected table inside the workspace and,
 * if non-null, user-visible copies: radii[N] int32 (0 = culled), means2d[N,2], depths[N], conics[N,3]. */
GWBP_API int gwbp_project(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                 const float *means, const float *quats, const float *scales, const float *opacities,
                 int32_t *radii, float *means2d, float *depths, float *conics, void *stream);

/* isect_tiles + stable radix sort by (tile, depth) + isect_offset_encode.  Optional outputs:
 * isect_ids[isect_cap] int64 sorted keys, flatten_ids[isect_cap] int32, tile_offsets[tiles+1] int32. */
GWBP_API int gwbp_bin_sort(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                  int64_t *isect_ids, int32_t *flatten_ids, int32_t *tile_offsets, void *stream);

/* rasterize_to_pixels forward, weights only: per tile, front-to-back blend producing the sparse weight store
 * (w = alpha*T per contributing (Gaussian, pixel)) inside the workspace; alphas[H*W] (= 1 - T) optional. */
GWBP_API int gwbp_blend_weights(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                       float *alphas, void *stream);

/* gwbp_blend_weights that also adds the view's denominators, d[g] += scale_d * sum_p w_g(p), while it writes each
 * record (needs caps WITHOUT GWBP_FLAG_NARROW_SCATTER): what a caller that overlaps the front stage of view v+1 with
 * the scatter of view v uses on the front's stream (then d = NULL for gwbp_scatter) -- the whole denominator pass of
 * backproject.py:133-150 costs one 4-B atomic per (Gaussian, tile) record and no kernel of its own. */
GWBP_API int gwbp_blend_weights_d(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                         float *alphas, float scale_d, float *d, void *stream);

/* Blend AND scatter of one view in one kernel, for narrow maps (1 <= D <= 16; on images of at most 4096 tiles, where a wave
 * takes a quarter tile with one pixel per lane, 1 <= D <= 32: backproject_compressed.py:127-165 after its
 * 512 -> 16 encoder; a 3-channel colour gradient): while a tile is blended its 256 pixels x D channels sit in registers,
 * each contributing (Gaussian, tile) record's sums  F[g, :D] += scale_f * sum_p w f[p, :],  d[g] += scale_d * sum_p w  are
 * reduced across the wave and added with one atomic instruction.  No weight store is written (the workspace's store is
 * left EMPTY: a later gwbp_scatter / gwbp_render of this view adds nothing and gwbp_stats.reserved reads 2), no scatter
 * kernel runs.  feats[y * fs_y + x * fs_x + c] full resolution, unit channel stride; d may be NULL; alphas optional.
 * The weights are those of gwbp_blend_weights bit for bit; F and d differ from gwbp_scatter's only by summation order. */
GWBP_API int gwbp_blend_scatter(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                       const float *feats, int64_t fs_y, int64_t fs_x, int32_t D, float scale_f, float scale_d,
                       float *F, float *d, float *alphas, void *stream);

/* gwbp_blend_scatter of feats[H, W, K] @ encoder[K, n_out] (backproject_compressed.py:127-165 in ONE kernel): every tile's wave
 * first streams its 256 pixels x K channels once through the matrix cores (exact fp32, the k-ordered chain of gwbp_encode_map)
 * and keeps the n_out <= 16 outputs per pixel in registers -- no [H, W, n_out] map, no encoder kernel -- then blends and
 * scatters like gwbp_blend_scatter.  feats[y * fs_y + x * fs_x + k]: channel-contiguous 16-B aligned pixels, K % 16 == 0,
 * 16 <= K <= 512; encoder row-major [K, n_out].  Any image size.  Results equal gwbp_encode_map + gwbp_blend_scatter bit for
 * bit in the encoded pixels, hence in every weight and (up to summation order) in F and d. */
GWBP_API int gwbp_blend_scatter_encoded(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                               const float *feats, int64_t fs_y, int64_t fs_x, int32_t K, const float *encoder, int32_t n_out,
                               float scale_f, float scale_d, float *F, float *d, float *alphas, void *stream);

/* The dino variant in TOKEN space (backproject.py:242-289: 64 x 64 x 1024 patch tokens, F.interpolate(mode="nearest") to the
 * view's size, then the back-projection).  All pixels of a token carry the same vector, so F_v[g,:] = sum_t omega_{g,t} tok[t,:]
 * with omega_{g,t} = sum_{p in t} w_g(p).  gwbp_blend_tokens is gwbp_blend_weights whose product is those sums instead of a
 * weight store: per contributing (Gaussian, tile) record the weight sums of the tile's (at most) 2 x 2 tokens, filed at the
 * record's emit position so that the sums of one Gaussian lie back to back.  gwbp_scatter_tokens then adds
 *     F[g,:] += scale_f * sum_t omega_{g,t} tokens[t,:],   d[g] += scale_d * sum_t omega_{g,t}
 * with ONE plain read-modify-write of every row that receives weight: no atomics, no weight store, deterministic.
 * PRECONDITION: ymap[view.height] / xmap[view.width] (int32 device arrays, PyTorch's nearest rule, non-decreasing) send the 16
 * pixels of any tile row / column to at most TWO consecutive tokens, i.e. a token is at least a tile wide and high
 * (16 * lr_w <= width and 16 * lr_h <= height suffice); a tile that violates it sets gwbp_stats.overflow bit 3 and the view's
 * result is invalid -- use gwbp_scatter_upsampled for finer maps.  The weights are gwbp_blend_weights' bit for bit (the alpha map
 * too); F and d equal gwbp_scatter_upsampled's up to summation order.  After gwbp_blend_tokens the workspace holds NO weight
 * store (gwbp_stats.reserved reads 3): gwbp_scatter / gwbp_render of that view add nothing.
 * gwbp_scatter_tokens: tokens[row * ts_y + col * ts_x + c], channel-contiguous 16-B aligned rows, D % 4 == 0 (256-channel chunks,
 * the last one masked; GWBP_EUNSUPPORTED otherwise), views of at most 4096 x 4096 pixels; d may be NULL;
 * needs gwbp_project + gwbp_bin_sort + gwbp_blend_tokens of the same view in this workspace (anything else sets overflow bit 2
 * and leaves F and d untouched). */
GWBP_API int gwbp_blend_tokens(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                               const int32_t *ymap, const int32_t *xmap, float *alphas, void *stream);
GWBP_API int gwbp_scatter_tokens(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                                 const float *tokens, int64_t ts_y, int64_t ts_x, int32_t D, const int32_t *ymap,
                                 const int32_t *xmap, float scale_f, float scale_d, float *F, float *d, void *stream);

/* d[g] += scale_d * sum_p w_g(p) alone, from the per-record weight sums gwbp_blend_weights left in the workspace
 * (needs a blend WITHOUT GWBP_FLAG_NARROW_SCATTER).  A caller that overlaps the front stage of view v+1 with the
 * scatter of view v issues it behind the blend on the front's stream and passes d = NULL to gwbp_scatter: the
 * denominators then cost nothing on the scatter's stream (backproject.py:133-150). */
GWBP_API int gwbp_accumulate_d(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                      float scale_d, float *d, void *stream);

/* What the reference obtains through backward(): F[g,:] += scale_f * sum_p w_g(p) * feats[p,:] and
 * d[g] += scale_d * sum_p w_g(p)   (backproject.py:127-131,145-150; scale = 1 for .sum(), 1/(H*W*D) and
 * 1/(H*W*3) for the dino .mean() variant, backproject.py:263,283).  feats is addressed as
 * feats[y*fs_y + x*fs_x + c*fs_c] (strides in floats; D % 128 == 0 or D <= 64 take the fast kernel; fs_c == 1 stages with 16-B loads).  d may be null. */
GWBP_API int gwbp_scatter(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                 const float *feats, int64_t fs_y, int64_t fs_x, int64_t fs_c, int32_t D, float scale_f,
                 float scale_d, float *F, float *d, void *stream);

/* The compressed variant in one kernel (backproject_compressed.py:127-165): gwbp_scatter of feats[H,W,K] @ encoder[K,D]
 * (D <= 16, K % 16 == 0, K <= 1024, channel-contiguous 16-B aligned pixels: feats[y*fs_y + x*fs_x + k]) without ever
 * writing the [H,W,D] map: every tile's pixels are read once, full width, and multiplied by the encoder (resident in LDS)
 * on the matrix cores while the tile's slab is staged (exact fp32: v_mfma_f32_16x16x4_f32). */
GWBP_API int gwbp_scatter_encoded(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                         const float *feats, int64_t fs_y, int64_t fs_x, int32_t K, const float *encoder, int32_t D,
                         float scale_f, float scale_d, float *F, float *d, void *stream);

/* gwbp_scatter over a LOW-RESOLUTION feature map that the reference would first upsample with
 * F.interpolate(mode="nearest") (dino variant, backproject.py:244-248): pixel (y, x) of the view reads
 * feats[ymap[y]*fs_y + xmap[x]*fs_x + c*fs_c].  ymap[view.height], xmap[view.width]: int32 device arrays (the host
 * side builds them with PyTorch's nearest rule: min(floor(i * in/out), in-1) in fp32).  Same result as
 * gwbp_scatter on the upsampled map, without ever materialising it. */
GWBP_API int gwbp_scatter_upsampled(const gwbp_caps *caps, void *workspace, size_t workspace_bytes,
                           const gwbp_view *view_host, const float *feats, int64_t fs_y, int64_t fs_x, int64_t fs_c,
                           int32_t D, const int32_t *ymap, const int32_t *xmap, float scale_f, float scale_d, float *F,
                           float *d, void *stream);

/* gwbp_scatter over a LOW-RESOLUTION feature map that the reference would first upsample with
 * F.interpolate(mode="bilinear") (lseg variant, backproject.py:110-112; align_corners=False): pixel (y, x) reads
 * h0*(w0*L[y0][x0] + w1*L[y0][x1]) + h1*(w0*L[y1][x0] + w1*L[y1][x1]) with y0 = y0map[y], y1 = min(y0+1, lr_h-1),
 * h1 = ly[y], h0 = 1-h1 (same for x): ATen's UpSampleBilinear2d, computed while the tile's slab is staged.  The maps
 * are device arrays of view.height / view.width entries (engine.bilinear_index builds them like ATen does). */
GWBP_API int gwbp_scatter_bilinear(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                          const float *feats, int64_t fs_y, int64_t fs_x, int64_t fs_c, int32_t D, int32_t lr_h,
                          int32_t lr_w, const int32_t *y0, const float *ly, const int32_t *x0, const float *lx,
                          float scale_f, float scale_d, float *F, float *d, void *stream);

/* Forward render (what rasterization() returns): out[p,:] = sum_g w_g(p) * colors[g,:], out is [H,W,D]. */
GWBP_API int gwbp_render(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                const float *colors, int32_t D, float *out, void *stream);

/* Forward render for 1..32 channels (RGB, RGB+D, depth; round 5: the 16-d compressed field) straight from the sorted tile
 * lists (needs gwbp_project + gwbp_bin_sort of the same view, not the weight store): the render the reference feeds to its
 * 2-D feature network (backproject.py:89-100), compares in utils.test_proper_pruning (utils.py:316-340) and scores per frame
 * in segment_compressed.py:154-165.  alphas[H*W] optional. */
GWBP_API int gwbp_render_pixels(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                       const float *colors, int32_t D, float *out, float *alphas, void *stream);

/* gsplat spherical_harmonics + "+0.5, clamp at 0": coeffs[N,K,3] (K >= (degree+1)^2, degree <= 3), view directions
 * means - campos_host[3]; out[N,3].  What rasterization(..., sh_degree=3) does before rasterising (backproject.py:99). */
GWBP_API int gwbp_sh_colors(int64_t N, int32_t degree, int32_t K, const float *means, const float *coeffs,
                   const float *campos_host, float *out, void *stream);

/* ---- fused entry points --------------------------------------------------------------------------------- */

/* project -> bin_sort -> blend_weights -> scatter for one view: the per-view body of
 * create_feature_field_lseg (backproject.py:115-151) in one call, one blend instead of two. */
GWBP_API int gwbp_backproject_view(const gwbp_caps *caps, void *workspace, size_t workspace_bytes,
                          const gwbp_view *view_host, const float *means, const float *quats, const float *scales,
                          const float *opacities, const float *feats, int64_t fs_y, int64_t fs_x, int64_t fs_c,
                          int32_t D, float scale_f, float scale_d, float *F, float *d, void *stream);

/* backproject_compressed.py:127: out[y, x, :] = feats[y, x, :] @ encoder, encoder [K, n_out] row-major, n_out <= 16,
 * K % 16 == 0, feats[y*fs_y + x*fs_x + c] (channel-contiguous pixels, 16-B aligned), out [H, W, n_out] dense.  The map is
 * read once at HBM rate; exact fp32 (v_mfma_f32_16x16x4_f32 = a k-ordered fmaf chain).  workgroups: 0 = as many as
 * stream fastest alone (6.0 TB/s); a caller that overlaps the encoder with latency-bound kernels passes one per CU. */
GWBP_API int gwbp_encode_map(const float *feats, int64_t fs_y, int64_t fs_x, int32_t height, int32_t width, int32_t K,
                    const float *encoder, int32_t n_out, float *out, int32_t workgroups, void *stream);

/* backproject.py:63,166-169: out = normalize(F / (1e-12 + d)), NaN -> 0.  out may alias F. */
GWBP_API int gwbp_finalize(int64_t N, int32_t D, const float *F, const float *d, float *out, void *stream);

/* Adds this view's counters into `accum` (device, gwbp_stats) -- used by bench/driver to total pairs. */
GWBP_API int gwbp_accumulate_stats(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, gwbp_stats *accum,
                          void *stream);
/* Copies the workspace's counters of the last view to host memory.  SYNCHRONISES `stream`. */
GWBP_API int gwbp_read_stats(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, gwbp_stats *stats_host,
                    void *stream);

/* Debug/test: expand the sparse weight store of the last blended view into triples, sorted by nothing in
 * particular.  gid/pix/w have room for `cap` entries; *n_host receives the count.  SYNCHRONISES. */
GWBP_API int gwbp_dump_pairs(const gwbp_caps *caps, void *workspace, size_t workspace_bytes, const gwbp_view *view_host,
                    int64_t cap, int32_t *gid, int32_t *pix, float *w, int64_t *n_host, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GWBP_H */
