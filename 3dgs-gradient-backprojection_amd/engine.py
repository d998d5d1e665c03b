"""Engine: a per-device workspace + thin methods over the C ABI stages (include/gwbp.h).

Everything is enqueued on torch's current HIP stream of the tensors' device; nothing synchronises except
`stats()` / `dump_pairs()` and the capacity auto-grow in `backproject_view(check=True)`.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import Caps, GwbpError, Stats, check, make_view, ptr

TILE = 16


def nearest_index(n_in: int, n_out: int) -> torch.Tensor:
    """Source index of every output index under torch.nn.functional.interpolate(mode="nearest"):
    min(floor(dst * scale), n_in - 1) with scale = n_in / n_out, all in fp32 (ATen UpSample.h
    nearest_neighbor_compute_source_index; the op the reference applies at backproject.py:244-248)."""
    if n_in < 1 or n_out < 1:
        raise ValueError("sizes must be positive")
    scale = torch.tensor(float(n_in), dtype=torch.float32) / torch.tensor(float(n_out), dtype=torch.float32)
    idx = torch.floor(torch.arange(n_out, dtype=torch.float32) * scale).to(torch.int64).clamp_(max=n_in - 1)
    return idx.to(torch.int32)


def bilinear_index(n_in: int, n_out: int):
    """(lower source index int32[n_out], weight of the upper neighbour float32[n_out]) of
    torch.nn.functional.interpolate(mode="bilinear", align_corners=False), computed like ATen's
    area_pixel_compute_source_index in fp32: src = max(scale * (dst + 0.5) - 0.5, 0), scale = n_in / n_out
    (the op the reference applies at backproject.py:110-112)."""
    if n_in < 1 or n_out < 1:
        raise ValueError("sizes must be positive")
    scale = torch.tensor(float(n_in), dtype=torch.float32) / torch.tensor(float(n_out), dtype=torch.float32)
    src = (scale * (torch.arange(n_out, dtype=torch.float32) + 0.5) - 0.5).clamp_(min=0.0)
    i0 = src.to(torch.int64).clamp_(max=n_in - 1)
    lam = (src - i0.to(torch.float32)).clamp_(0.0, 1.0)
    return i0.to(torch.int32), lam


def _req(t: torch.Tensor, name: str, shape_tail=None) -> torch.Tensor:
    if not t.is_cuda:
        raise GwbpError(f"{name} must be a CUDA/HIP tensor (no CPU fallback exists for this path)")
    if t.dtype != torch.float32:
        raise GwbpError(f"{name} must be float32, got {t.dtype}")
    if shape_tail is not None and tuple(t.shape[1:]) != tuple(shape_tail):
        raise GwbpError(f"{name} has shape {tuple(t.shape)}, expected [N,{','.join(map(str, shape_tail))}]")
    return t.contiguous()


class Engine:
    """Workspace sized for (N Gaussians, max WxH, isect_cap, pair_cap) on one device."""

    def __init__(self, n_gaussians: int, max_width: int, max_height: int, device=None,
                 isect_cap: Optional[int] = None, pair_cap: Optional[int] = None, scatter_workgroups: int = 0,
                 tight_binning: bool = False):
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise GwbpError("Engine needs a HIP device (there is no CPU path)")
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.lib = _lib.lib()
        self._maps: Dict[Tuple[int, int, int, int], Tuple[torch.Tensor, torch.Tensor]] = {}
        self.n = int(n_gaussians)
        self.max_w, self.max_h = int(max_width), int(max_height)
        # Defaults: ~16 tiles per Gaussian and ~128 weights per pixel; both auto-grow on overflow.
        self.isect_cap = int(isect_cap or max(1 << 18, 16 * self.n))
        self.pair_cap = int(pair_cap or max(1 << 20, 128 * self.max_w * self.max_h))
        self.scatter_workgroups = int(scatter_workgroups)  # 0 = one persistent scatter workgroup per CU
        # tight_binning: GWBP_FLAG_TIGHT_BINNING -- same F, d, weights and renders, shorter tile lists; off by default
        # because meta["isect_ids"] of the drop-in operator must show gsplat's 3-sigma binning
        self.tight_binning = bool(tight_binning)
        self._halves = False  # the workspace holds the half-tile lists + weight sums of the view blended last
        self._tokens = None   # (h, w) of the map whose token-quadrant weight sums the workspace holds (blend_tokens)
        self.stream, self._stream_handle = None, None
        self._alloc()

    def set_front_priority(self, on: bool) -> None:
        """GWBP_FLAG_FRONT_PRIORITY for this engine's project / bin_sort / blend_weights launches: raised wave priority.
        Only useful when they run on a second stream beside a D % 256 == 0 scatter (ViewPipeline sets it)."""
        if on:
            self.caps.flags |= _lib.FLAG_FRONT_PRIORITY
        else:
            self.caps.flags &= ~_lib.FLAG_FRONT_PRIORITY

    def set_split_encoder(self, on: bool) -> None:
        """GWBP_FLAG_SPLIT_ENCODER: blend_scatter_encoded as ONE persistent launch of encoder (producer) waves and blend (consumer)
        waves around an LDS ring of encoded tiles (the compressed variant on large images, see gwbp.h)."""
        if on:
            self.caps.flags |= _lib.FLAG_SPLIT_ENCODER
        else:
            self.caps.flags &= ~_lib.FLAG_SPLIT_ENCODER

    def set_narrow_scatter(self, on: bool) -> None:
        """GWBP_FLAG_NARROW_SCATTER: the 128-channel scatter kernel even when D % 256 == 0 (short records, see gwbp.h).
        An Engine starts narrow (k_blend then skips the half-tile lists only the 256-channel kernel reads); whoever
        knows that a D % 256 == 0 scatter follows switches it off BEFORE blend_weights of that view (ViewPipeline,
        the drop-in operator)."""
        if on:
            self.caps.flags |= _lib.FLAG_NARROW_SCATTER
        else:
            self.caps.flags &= ~_lib.FLAG_NARROW_SCATTER

    def _alloc(self):
        # run-time flags survive a re-allocation (grow); a new engine starts narrow
        run = (self.caps.flags & (_lib.FLAG_FRONT_PRIORITY | _lib.FLAG_NARROW_SCATTER | _lib.FLAG_SPLIT_ENCODER)
               if hasattr(self, "caps") else _lib.FLAG_NARROW_SCATTER)
        base = _lib.FLAG_TIGHT_BINNING if self.tight_binning else 0
        self.caps = Caps(self.n, self.isect_cap, self.pair_cap, self.max_w, self.max_h, self.scatter_workgroups, base | run)
        nbytes = C.c_size_t(0)
        self._call("gwbp_workspace_size", C.byref(self.caps), C.byref(nbytes))
        self.ws_bytes = int(nbytes.value)
        self.ws = torch.empty(self.ws_bytes + 256, dtype=torch.uint8, device=self.device)
        off = (-self.ws.data_ptr()) % 256
        self._ws_ptr = C.c_void_p(self.ws.data_ptr() + off)

    def grow(self, stats: Dict[str, int], views: int = 1):
        """Enlarge whichever capacity overflowed (bit0 = isect, bit1 = pairs) and reallocate.  The pair counter keeps
        counting after the pool is exhausted, so the weight store can be sized for what the views really needed
        (`stats` summed over `views` views)."""
        if stats["overflow"] & 1:
            self.isect_cap *= 2
        if stats["overflow"] & 2:
            self.pair_cap = max(2 * self.pair_cap, int(2.5 * stats["n_pairs"] / max(1, views)) + (1 << 16))
        del self.ws
        self._alloc()

    # ---- helpers -----------------------------------------------------------------------------------------
    def _stream(self):
        if self.stream is not None:  # bound by a driver that keeps this engine on one stream (no context switch per call)
            return self._stream_handle
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def bind_stream(self, stream: Optional[torch.cuda.Stream]):
        """Launch this engine's kernels on `stream` whatever torch's current stream is (None: follow the current stream).
        Methods that allocate outputs (want_alphas, render, ...) still allocate on the current stream."""
        self.stream = stream
        self._stream_handle = C.c_void_p(stream.cuda_stream) if stream is not None else None

    def _call(self, name: str, *args):
        """One C-ABI call with this engine's device current: libgwbp launches on the CURRENT HIP device, while the
        workspace, the tensors and the stream handle belong to self.device (a process may hold engines on several)."""
        fn = getattr(self.lib, name)
        if torch.cuda.current_device() == self._dev_index:  # the usual case: no device switch (a context manager costs
            check(fn(*args), name)                           # ~10 us per call, and small scenes are host-bound)
            return
        with torch.cuda.device(self.device):
            check(fn(*args), name)

    def _args(self):
        return C.byref(self.caps), self._ws_ptr, C.c_size_t(self.ws_bytes)

    def view(self, viewmat, K, width, height, **kw):
        return make_view(viewmat, K, int(width), int(height), **kw)

    # ---- stages ------------------------------------------------------------------------------------------
    def project(self, view, means, quats, scales, opacities, want_outputs=False):
        means, quats = _req(means, "means", (3,)), _req(quats, "quats", (4,))
        scales, opacities = _req(scales, "scales", (3,)), _req(opacities, "opacities")
        if means.shape[0] != self.n:
            raise GwbpError(f"engine was sized for {self.n} Gaussians, got {means.shape[0]}")
        out = {}
        if want_outputs:
            out = dict(radii=torch.empty(self.n, dtype=torch.int32, device=self.device),
                       means2d=torch.empty(self.n, 2, device=self.device),
                       depths=torch.empty(self.n, device=self.device),
                       conics=torch.empty(self.n, 3, device=self.device))
        self._call("gwbp_project", *self._args(), C.byref(view), ptr(means), ptr(quats), ptr(scales),
                                    ptr(opacities), ptr(out.get("radii")), ptr(out.get("means2d")),
                                    ptr(out.get("depths")), ptr(out.get("conics")), self._stream())
        return out

    def bin_sort(self, view, want_outputs=False):
        out = {}
        if want_outputs:
            nt = -(-view.width // TILE) * -(-view.height // TILE)
            out = dict(isect_ids=torch.empty(self.isect_cap, dtype=torch.int64, device=self.device),
                       flatten_ids=torch.empty(self.isect_cap, dtype=torch.int32, device=self.device),
                       tile_offsets=torch.empty(nt + 1, dtype=torch.int32, device=self.device))
        self._call("gwbp_bin_sort", *self._args(), C.byref(view), ptr(out.get("isect_ids")),
                                     ptr(out.get("flatten_ids")), ptr(out.get("tile_offsets")), self._stream())
        return out

    def _wide_requested(self) -> bool:
        return not (self.caps.flags & _lib.FLAG_NARROW_SCATTER)

    def blend_weights(self, view, want_alphas=False, d=None, scale_d=1.0):
        """d (optional, float32[N]): also add this view's denominators d[g] += scale_d * sum_p w_g(p) from inside the blend
        (gwbp_blend_weights_d; needs the 256-channel scatter kernel enabled, like accumulate_d)."""
        alphas = torch.empty(view.height, view.width, device=self.device) if want_alphas else None
        self._tokens = None
        self._halves = self._wide_requested()  # k_blend<HALVES> writes the lists only without NARROW_SCATTER
        if d is not None:
            if d.dtype != torch.float32 or not d.is_cuda or d.shape != (self.n,) or not d.is_contiguous():
                raise GwbpError("d must be a contiguous float32 HIP tensor [N]")
            if not self._halves:
                raise GwbpError("blend_weights(d=...) needs the 256-channel scatter kernel enabled "
                                "(set_narrow_scatter(False)): a narrow blend takes no weight sums")
            self._call("gwbp_blend_weights_d", *self._args(), C.byref(view), ptr(alphas), C.c_float(scale_d), ptr(d),
                       self._stream())
            return alphas
        self._call("gwbp_blend_weights", *self._args(), C.byref(view), ptr(alphas), self._stream())
        return alphas

    FUSED_MAX_DIM = 16         # gwbp_blend_scatter holds 4 pixels x 16 channels per lane in registers ...
    FUSED_MAX_DIM_SMALL = 32   # ... or, on images of at most FUSED_SMALL_TILES tiles, one pixel x 32 channels (a wave per
    FUSED_SMALL_TILES = 4096   # quarter tile: four short blend chains per tile instead of one long one)

    @classmethod
    def fused_max_dim(cls, width: int, height: int) -> int:
        tiles = (-(-int(width) // 16)) * (-(-int(height) // 16))
        return cls.FUSED_MAX_DIM_SMALL if tiles <= cls.FUSED_SMALL_TILES else cls.FUSED_MAX_DIM

    @classmethod
    def can_blend_scatter(cls, feats: torch.Tensor) -> bool:
        """Maps gwbp_blend_scatter takes: [H,W,D] float32 at full resolution, D <= 16 (<= 32 on small images), unit channel
        stride, non-negative strides."""
        return (feats.dim() == 3 and feats.is_cuda and feats.dtype == torch.float32
                and 1 <= feats.shape[2] <= cls.fused_max_dim(feats.shape[1], feats.shape[0])
                and feats.stride(2) == 1 and min(feats.stride()) >= 0)

    def blend_scatter(self, view, feats, F, d, scale_f=1.0, scale_d=1.0, want_alphas=False):
        """blend_weights + scatter of one view in ONE kernel for narrow maps (D <= 16: the compressed variant after its
        encoder, backproject_compressed.py:127-165): the tile's pixels sit in registers while it is blended, each
        contributing record is reduced across the wave and added to F / d at once.  No weight store is written: the
        view cannot be scattered or rendered again without a new blend_weights()."""
        if not self.can_blend_scatter(feats):
            raise GwbpError(f"blend_scatter: [H,W,D] float32 map with unit channel stride and D <= {self.FUSED_MAX_DIM} "
                            f"(<= {self.FUSED_MAX_DIM_SMALL} on images of at most {self.FUSED_SMALL_TILES} tiles) required, "
                            f"got {tuple(feats.shape)} strides {tuple(feats.stride())}")
        sy, sx, _, D = self._feat_strides(feats, view)
        self._check_acc(F, d, D)
        alphas = torch.empty(view.height, view.width, device=self.device) if want_alphas else None
        self._halves, self._tokens = False, None  # the store is empty: no scatter kernel has anything to read
        self._call("gwbp_blend_scatter", *self._args(), C.byref(view), ptr(feats), sy, sx, D, C.c_float(scale_f),
                   C.c_float(scale_d), ptr(F), ptr(d), ptr(alphas), self._stream())
        return alphas

    @staticmethod
    def can_blend_scatter_encoded(feats: torch.Tensor, encoder: torch.Tensor) -> bool:
        """Shapes gwbp_blend_scatter_encoded takes: [H,W,K] float32 with channel-contiguous 16-B aligned pixels, K % 16 == 0,
        16 <= K <= 512, at most 16 outputs (the reference's encoder is 512 -> 16, backproject_compressed.py:26,127)."""
        if feats.dim() != 3 or encoder.dim() != 2 or feats.shape[2] != encoder.shape[0]:
            return False
        sy, sx, sc = feats.stride()
        K, n = encoder.shape
        # (a row must span less than 4 GB: the kernel's per-lane column offsets are 32-bit -- a channel slice of a much wider
        # tensor falls back to encode_map, like any other layout the kernel does not take)
        return (feats.is_cuda and feats.dtype == torch.float32 and encoder.dtype == torch.float32 and 1 <= n <= 16
                and K % 16 == 0 and 16 <= K <= 512 and sc == 1 and sy % 4 == 0 and sx % 4 == 0 and sy >= 0 and sx >= K
                and feats.data_ptr() % 16 == 0 and ((feats.shape[1] - 1) * sx + K) * 4 < (1 << 32))

    def blend_scatter_encoded(self, view, feats, encoder, F, d, scale_f=1.0, scale_d=1.0, want_alphas=False):
        """blend_scatter(view, feats @ encoder, ...) of the compressed variant (backproject_compressed.py:127-165) in ONE
        kernel: every tile's wave streams its 256 pixels x K channels once through the matrix cores (exact fp32, the same
        chain as encode_map) into the registers the fused blend + scatter works from -- no [H,W,n] map, no encoder kernel, no
        weight store."""
        if not self.can_blend_scatter_encoded(feats, encoder):
            raise GwbpError("blend_scatter_encoded: [H,W,K] float32 channel-contiguous 16-B aligned map, K % 16 == 0, "
                            f"16 <= K <= 512, <= 16 outputs required, got {tuple(feats.shape)} strides {tuple(feats.stride())} "
                            f"@ {tuple(encoder.shape)}")
        if feats.shape[0] != view.height or feats.shape[1] != view.width:
            raise GwbpError(f"feature map must be [{view.height},{view.width},K], got {tuple(feats.shape)}")
        sy, sx, _ = feats.stride()
        K, n = encoder.shape
        self._check_acc(F, d, n)
        enc = encoder.contiguous()
        alphas = torch.empty(view.height, view.width, device=self.device) if want_alphas else None
        self._halves, self._tokens = False, None  # the store is empty: no scatter kernel has anything to read
        self._call("gwbp_blend_scatter_encoded", *self._args(), C.byref(view), ptr(feats), sy, sx, K, ptr(enc), n,
                   C.c_float(scale_f), C.c_float(scale_d), ptr(F), ptr(d), ptr(alphas), self._stream())
        return alphas

    @staticmethod
    def _feat_strides(feats: torch.Tensor, view, lowres: bool = False) -> Tuple[int, int, int, int]:
        if feats.dim() != 3 or (not lowres and (feats.shape[0] != view.height or feats.shape[1] != view.width)):
            raise GwbpError(f"feature map must be [H,W,D] = [{view.height},{view.width},D], got {tuple(feats.shape)}")
        if feats.dtype != torch.float32 or not feats.is_cuda:
            raise GwbpError("feature map must be a float32 HIP tensor")
        sy, sx, sc = feats.stride()
        if min(sy, sx, sc) < 0:
            raise GwbpError("negative feature-map strides are not supported")
        return sy, sx, sc, feats.shape[2]

    # ---- token space: the dino variant's nearest-upsampled patch-token map (backproject.py:242-249) ------------------
    TOKEN_MIN_DIM = 64  # gwbp_scatter_tokens: one float4 per lane and 256-channel chunk; narrower maps leave 3/4 of a wave idle
    TOKEN_MAX_VIEW = 4096  # ... and keeps per-tile-column / -row tables of 256 entries in LDS (token.hip kTokMaxTiles)

    _TOKEN_GEOMETRY: Dict[Tuple[int, int, int, int], bool] = {}

    @classmethod
    def token_geometry_ok(cls, lr_h: int, lr_w: int, height: int, width: int) -> bool:
        """Does every 16 x 16 tile of a height x width view see at most 2 x 2 texels of an lr_h x lr_w map under
        F.interpolate(mode="nearest")?  Checked on the exact index maps (PyTorch's fp32 rule), tile by tile: the precondition
        of gwbp_blend_tokens.  True whenever a texel is at least a tile wide and high (the 64 x 64 dino tokens at 1600 x 1060)."""
        key = (int(lr_h), int(lr_w), int(height), int(width))
        ok = cls._TOKEN_GEOMETRY.get(key)
        if ok is None:
            ok = True
            for n_in, n_out in ((key[0], key[2]), (key[1], key[3])):
                m = nearest_index(n_in, n_out).to(torch.int64)
                first = m[0::TILE]
                last = m[torch.clamp(torch.arange(0, n_out, TILE) + TILE - 1, max=n_out - 1)]
                ok = ok and int((last - first).max()) <= 1
            cls._TOKEN_GEOMETRY[key] = ok
        return ok

    @classmethod
    def can_scatter_tokens(cls, tokens: torch.Tensor, height: int, width: int) -> bool:
        """Low-resolution maps the token-space path takes: [h, w, D] float32 on the device, D % 4 == 0 and D >= 64 (the 384 / 768 /
        1024 / 1536 channels of the DINOv2 backbones), channel-contiguous
        16-B aligned rows, texels at least a tile wide and high (token_geometry_ok), views of at most 4096 x 4096 pixels.
        Anything else goes through blend_weights + scatter(upsample="nearest")."""
        if tokens.dim() != 3 or not tokens.is_cuda or tokens.dtype != torch.float32:
            return False
        if max(int(height), int(width)) > cls.TOKEN_MAX_VIEW:
            return False
        sy, sx, sc = tokens.stride()
        D = tokens.shape[2]
        return (D >= cls.TOKEN_MIN_DIM and D % 4 == 0 and sc == 1 and sy % 4 == 0 and sx % 4 == 0 and sy >= 0
                and sx >= D and tokens.data_ptr() % 16 == 0
                and cls.token_geometry_ok(tokens.shape[0], tokens.shape[1], int(height), int(width)))

    def blend_tokens(self, view, lr_h: int, lr_w: int, want_alphas=False):
        """blend_weights for a view whose feature map is an lr_h x lr_w map upsampled with mode="nearest" (the dino variant):
        instead of a weight store the blend leaves, per contributing (Gaussian, tile) record, the weight sums of the tile's
        2 x 2 tokens (gwbp_blend_tokens); scatter_tokens() consumes them.  Same weights, same alpha map."""
        if not self.token_geometry_ok(lr_h, lr_w, view.height, view.width):
            raise GwbpError(f"blend_tokens: a {lr_h}x{lr_w} map has texels narrower than a tile at {view.height}x{view.width}; "
                            "use blend_weights + scatter(upsample='nearest')")
        ymap, xmap = self.nearest_maps(lr_h, lr_w, view.height, view.width)
        alphas = torch.empty(view.height, view.width, device=self.device) if want_alphas else None
        self._halves = False  # no weight store: only scatter_tokens can consume this view
        self._tokens = (int(lr_h), int(lr_w))
        self._call("gwbp_blend_tokens", *self._args(), C.byref(view), ptr(ymap), ptr(xmap), ptr(alphas), self._stream())
        return alphas

    def scatter_tokens(self, view, tokens, F, d, scale_f=1.0, scale_d=1.0):
        """F[g,:] += scale_f * sum_t omega_{g,t} tokens[t,:], d[g] += scale_d * sum_t omega_{g,t} from the sums blend_tokens left:
        equals scatter(view, tokens, F, d, upsample="nearest") up to summation order, with one plain read-modify-write of every
        row that receives weight (no atomics: deterministic) and the token map read from L2 / Infinity Cache."""
        if getattr(self, "_tokens", None) != (int(tokens.shape[0]), int(tokens.shape[1])):
            raise GwbpError("scatter_tokens needs blend_tokens(view, h, w) of the same view and map size first")
        if not self.can_scatter_tokens(tokens, view.height, view.width):
            raise GwbpError(f"scatter_tokens: [h,w,D] float32 map with D % 4 == 0, D >= 64, channel-contiguous 16-B aligned rows and texels "
                            f"of at least a tile at a view of at most {self.TOKEN_MAX_VIEW} x {self.TOKEN_MAX_VIEW} pixels required, got "
                            f"{tuple(tokens.shape)} strides {tuple(tokens.stride())} at {view.width} x {view.height}")
        D = tokens.shape[2]
        self._check_acc(F, d, D)
        ymap, xmap = self.nearest_maps(tokens.shape[0], tokens.shape[1], view.height, view.width)
        sy, sx, _ = tokens.stride()
        self._call("gwbp_scatter_tokens", *self._args(), C.byref(view), ptr(tokens), C.c_int64(sy), C.c_int64(sx), D, ptr(ymap),
                   ptr(xmap), C.c_float(scale_f), C.c_float(scale_d), ptr(F), ptr(d), self._stream())

    def accumulate_d(self, view, d, scale_d=1.0):
        """d += scale_d * sum_p w from the blend's per-record weight sums (needs a blend with the wide scatter enabled)."""
        if d is None or d.dtype != torch.float32 or not d.is_cuda or d.shape != (self.n,) or not d.is_contiguous():
            raise GwbpError("d must be a contiguous float32 HIP tensor [N]")
        if not self._halves:
            raise GwbpError("accumulate_d needs a view blended with the 256-channel scatter kernel enabled "
                            "(set_narrow_scatter(False) BEFORE blend_weights): this view's headers hold no weight sums")
        self._call("gwbp_accumulate_d", *self._args(), C.byref(view), C.c_float(scale_d), ptr(d), self._stream())

    def has_weight_sums(self) -> bool:
        """The view in the workspace was blended with the 256-channel scatter kernel enabled: its records carry their weight
        sums (accumulate_d, scatter_uniform)."""
        return bool(self._halves)

    def scatter_uniform(self, view, value: torch.Tensor, F: torch.Tensor) -> None:
        """F[g, :] += value * sum_p w_g(p): the scatter of a map whose every entry is `value` (a 0-d float32 HIP tensor, read
        on the device), from the per-record weight sums -- the backward of `render.sum()`, which is what the reference's
        denominator pass asks for (backproject.py:145-147).  Needs has_weight_sums()."""
        if not self._halves:
            raise GwbpError("scatter_uniform needs a view blended with the 256-channel scatter kernel enabled "
                            "(set_narrow_scatter(False) BEFORE blend_weights): this view's headers hold no weight sums")
        if F.dtype != torch.float32 or not F.is_cuda or F.dim() != 2 or F.shape[0] != self.n:
            raise GwbpError(f"F must be a float32 HIP tensor [{self.n},D]")
        sums = torch.zeros(self.n, device=self.device, dtype=torch.float32)
        saved = self.caps.flags  # (the flag says which kernel the NEXT blend serves; the sums of THIS view are there)
        self.caps.flags = saved & ~_lib.FLAG_NARROW_SCATTER
        try:
            self.accumulate_d(view, sums)
        finally:
            self.caps.flags = saved
        F.add_(sums[:, None] * value.to(torch.float32))

    def scatter(self, view, feats, F, d, scale_f=1.0, scale_d=1.0, upsample: Optional[str] = None):
        """F += scale_f * sum_p w feats[p], d += scale_d * sum_p w from the view's weight store.

        upsample="nearest" / "bilinear": feats is a LOW-RESOLUTION map [h,w,D]; the result equals scattering
        F.interpolate(feats, size=(H,W), mode=...) (dino: backproject.py:244-248; lseg: backproject.py:110-112,
        align_corners=False) without building that map -- the interpolation happens while the tile slabs are staged."""
        if self._tokens is not None:
            raise GwbpError("this view was blended with blend_tokens (no weight store): scatter_tokens() is its consumer; "
                            "blend_weights() first for scatter()")
        if self._wide_requested() and not self._halves:
            # This view was blended WITH GWBP_FLAG_NARROW_SCATTER (no half-tile lists): the 256-channel kernel would read
            # another view's tables.  Scatter it with the 128-channel kernel, which needs only the headers every blend writes.
            saved = self.caps.flags
            self.caps.flags = saved | _lib.FLAG_NARROW_SCATTER
            try:
                return self.scatter(view, feats, F, d, scale_f, scale_d, upsample)
            finally:
                self.caps.flags = saved
        if upsample is None:
            sy, sx, sc, D = self._feat_strides(feats, view)
            self._check_acc(F, d, D)
            self._call("gwbp_scatter", *self._args(), C.byref(view), ptr(feats), C.c_int64(sy), C.c_int64(sx),
                                        C.c_int64(sc), D, C.c_float(scale_f), C.c_float(scale_d), ptr(F), ptr(d),
                                        self._stream())
            return
        if upsample not in ("nearest", "bilinear"):
            raise GwbpError(f"upsample must be None, 'nearest' or 'bilinear', got {upsample!r}")
        sy, sx, sc, D = self._feat_strides(feats, view, lowres=True)
        self._check_acc(F, d, D)
        if upsample == "bilinear":
            y0, ly, x0, lx = self.bilinear_maps(feats.shape[0], feats.shape[1], view.height, view.width)
            self._call("gwbp_scatter_bilinear", *self._args(), C.byref(view), ptr(feats), C.c_int64(sy),
                                                 C.c_int64(sx), C.c_int64(sc), D, int(feats.shape[0]),
                                                 int(feats.shape[1]), ptr(y0), ptr(ly), ptr(x0), ptr(lx),
                                                 C.c_float(scale_f), C.c_float(scale_d), ptr(F), ptr(d),
                                                 self._stream())
            return
        ymap, xmap = self.nearest_maps(feats.shape[0], feats.shape[1], view.height, view.width)
        self._call("gwbp_scatter_upsampled", *self._args(), C.byref(view), ptr(feats), C.c_int64(sy), C.c_int64(sx),
                                              C.c_int64(sc), D, ptr(ymap), ptr(xmap), C.c_float(scale_f),
                                              C.c_float(scale_d), ptr(F), ptr(d), self._stream())

    @staticmethod
    def can_fuse_encoder(feats: torch.Tensor, encoder: torch.Tensor) -> bool:
        """Shapes gwbp_scatter_encoded takes: [H,W,K] float32 with channel-contiguous 16-B aligned pixels, K % 16 == 0,
        K <= 1024, at most 16 outputs."""
        if feats.dim() != 3 or encoder.dim() != 2 or feats.shape[2] != encoder.shape[0]:
            return False
        sy, sx, sc = feats.stride()
        K, n = encoder.shape
        return (feats.is_cuda and feats.dtype == torch.float32 and encoder.dtype == torch.float32 and n <= 16
                and K % 16 == 0 and 16 <= K <= 1024 and sc == 1 and sy % 4 == 0 and sx % 4 == 0 and sy >= 0 and sx >= 0
                and feats.data_ptr() % 16 == 0)

    def scatter_encoded(self, view, feats, encoder, F, d, scale_f=1.0, scale_d=1.0):
        """scatter(view, feats @ encoder, ...) of the compressed variant (backproject_compressed.py:127-165) in ONE kernel:
        the [H,W,K] map is read once, tile by tile, and multiplied by the encoder while the slabs are staged."""
        if not self.can_fuse_encoder(feats, encoder):
            raise GwbpError("scatter_encoded: [H,W,K] float32 channel-contiguous map, K % 16 == 0, K <= 1024, <= 16 outputs")
        if feats.shape[0] != view.height or feats.shape[1] != view.width:
            raise GwbpError(f"feature map must be [{view.height},{view.width},K], got {tuple(feats.shape)}")
        sy, sx, _ = feats.stride()
        K, n = encoder.shape
        self._check_acc(F, d, n)
        enc = encoder.contiguous()
        self._call("gwbp_scatter_encoded", *self._args(), C.byref(view), ptr(feats), sy, sx, K, ptr(enc), n,
                   C.c_float(scale_f), C.c_float(scale_d), ptr(F), ptr(d), self._stream())

    def bilinear_maps(self, h: int, w: int, H: int, W: int):
        """device maps of F.interpolate(mode="bilinear", align_corners=False): (y0[H], ly[H], x0[W], lx[W]); cached."""
        key = ("bilinear", h, w, H, W)
        m = self._maps.get(key)
        if m is None:
            (y0, ly), (x0, lx) = bilinear_index(h, H), bilinear_index(w, W)
            m = tuple(t.to(self.device) for t in (y0, ly, x0, lx))
            self._maps[key] = m
        return m

    def nearest_maps(self, h: int, w: int, H: int, W: int):
        """int32 device index maps of F.interpolate(mode="nearest"): (ymap[H], xmap[W]); cached per geometry."""
        key = (h, w, H, W)
        m = self._maps.get(key)
        if m is None:
            m = tuple(nearest_index(i, o).to(self.device) for i, o in ((h, H), (w, W)))
            self._maps[key] = m
        return m

    def render(self, view, colors):
        colors = _req(colors, "colors")
        D = colors.shape[1]
        out = torch.empty(view.height, view.width, D, device=self.device)
        self._call("gwbp_render", *self._args(), C.byref(view), ptr(colors), D, ptr(out), self._stream())
        return out

    def render_pixels(self, view, colors, want_alphas=True):
        """Pixel-parallel forward render for 1..32 channels; needs project + bin_sort of `view` (not the weight store)."""
        colors = _req(colors, "colors")
        D = colors.shape[1]
        out = torch.empty(view.height, view.width, D, device=self.device)
        alphas = torch.empty(view.height, view.width, device=self.device) if want_alphas else None
        self._call("gwbp_render_pixels", *self._args(), C.byref(view), ptr(colors), D, ptr(out), ptr(alphas),
                                          self._stream())
        return out, alphas

    def sh_colors(self, degree: int, means, coeffs, campos):
        """[N,K,3] SH coefficients -> [N,3] view-dependent colours (+0.5, clamped at 0) on the device."""
        means = _req(means, "means", (3,))
        coeffs = _req(coeffs, "sh coefficients")
        if coeffs.dim() != 3 or coeffs.shape[2] != 3 or coeffs.shape[0] != means.shape[0]:
            raise GwbpError(f"SH coefficients must be [N,K,3], got {tuple(coeffs.shape)}")
        out = torch.empty(means.shape[0], 3, device=self.device)
        cp = (C.c_float * 3)(*[float(v) for v in campos])
        self._call("gwbp_sh_colors", C.c_int64(means.shape[0]), int(degree), coeffs.shape[1], ptr(means), ptr(coeffs),
                                      cp, ptr(out), self._stream())
        return out

    def _check_acc(self, F, d, D):
        if F.dtype != torch.float32 or not F.is_cuda or not F.is_contiguous() or tuple(F.shape) != (self.n, D):
            raise GwbpError(f"F must be a contiguous float32 HIP tensor [{self.n},{D}]")
        if d is not None and (d.dtype != torch.float32 or not d.is_cuda or not d.is_contiguous()
                              or tuple(d.shape) != (self.n,)):
            raise GwbpError(f"d must be a contiguous float32 HIP tensor [{self.n}]")

    def backproject_view(self, view, means, quats, scales, opacities, feats, F, d, scale_f=1.0, scale_d=1.0):
        """Per-view body of create_feature_field_* (backproject.py:115-151), one fused call."""
        sy, sx, sc, D = self._feat_strides(feats, view)
        self._check_acc(F, d, D)
        means, quats = _req(means, "means", (3,)), _req(quats, "quats", (4,))
        scales, opacities = _req(scales, "scales", (3,)), _req(opacities, "opacities")
        self._halves, self._tokens = self._wide_requested(), None
        self._call("gwbp_backproject_view", *self._args(), C.byref(view), ptr(means), ptr(quats), ptr(scales),
                                             ptr(opacities), ptr(feats), C.c_int64(sy), C.c_int64(sx),
                                             C.c_int64(sc), D, C.c_float(scale_f), C.c_float(scale_d), ptr(F),
                                             ptr(d), self._stream())

    @staticmethod
    def can_encode_map(feats: torch.Tensor, encoder: torch.Tensor) -> bool:
        """Shapes gwbp_encode_map takes: [H,W,K] float32 with channel-contiguous 16-B aligned pixels, K % 16 == 0, K <= 2048,
        at most 16 outputs (the reference's encoder is 512 -> 16, backproject_compressed.py:26,127)."""
        if feats.dim() != 3 or encoder.dim() != 2 or feats.shape[2] != encoder.shape[0]:
            return False
        sy, sx, sc = feats.stride()
        K, n = encoder.shape
        return (feats.is_cuda and feats.dtype == torch.float32 and encoder.dtype == torch.float32 and n <= 16 and
                K % 16 == 0 and K <= 2048 and sc == 1 and sy % 4 == 0 and sx % 4 == 0 and feats.data_ptr() % 16 == 0
                and sy >= 0 and sx >= 0)

    def encode_map(self, feats: torch.Tensor, encoder: torch.Tensor, workgroups: int = 0,
                   stream: Optional[torch.cuda.Stream] = None) -> torch.Tensor:
        """feats[H,W,K] @ encoder[K,n] (backproject_compressed.py:127) -> [H,W,n] with the hand-written skinny GEMM
        (gwbp_encode_map: the map is read once at HBM rate, exact fp32 MFMA).  Shapes it does not take (can_encode_map)
        RAISE: the hot stage never falls back to a library GEMM silently -- callers that want one write `feats @ encoder`.
        workgroups: 0 = fastest alone; one per CU when the call overlaps latency-bound kernels on other streams
        (ViewPipeline.encode_ahead).  stream: launch there instead of on this engine's stream (an engine bound to a view's
        stream by bind_stream must not run another view's encoder on it); the output is allocated under that stream."""
        if feats.dim() != 3 or encoder.dim() != 2 or feats.shape[2] != encoder.shape[0]:
            raise GwbpError(f"encode_map: [H,W,K] @ [K,n] expected, got {tuple(feats.shape)} @ {tuple(encoder.shape)}")
        if not self.can_encode_map(feats, encoder):
            raise GwbpError("encode_map: [H,W,K] float32 channel-contiguous 16-B aligned map, K % 16 == 0, K <= 2048, "
                            f"<= 16 outputs required, got {tuple(feats.shape)} strides {tuple(feats.stride())} @ "
                            f"{tuple(encoder.shape)} (use feats @ encoder for other shapes)")
        H, W, K = feats.shape
        n = encoder.shape[1]
        sy, sx, _ = feats.stride()
        enc = encoder.contiguous()
        handle = self._stream() if stream is None else C.c_void_p(stream.cuda_stream)
        if stream is None:
            out = torch.empty(H, W, n, device=feats.device, dtype=torch.float32)
        else:
            with torch.cuda.stream(stream):
                out = torch.empty(H, W, n, device=feats.device, dtype=torch.float32)
        self._call("gwbp_encode_map", ptr(feats), sy, sx, H, W, K, ptr(enc), n, ptr(out), int(workgroups), handle)
        return out

    def finalize(self, F, d, out=None):
        out = torch.empty_like(F) if out is None else out
        self._call("gwbp_finalize", C.c_int64(F.shape[0]), F.shape[1], ptr(F), ptr(d), ptr(out), self._stream())
        return out

    # ---- counters ----------------------------------------------------------------------------------------
    def accumulate_stats(self, accum: torch.Tensor):
        """accum: uint8[32] device tensor holding a gwbp_stats struct (zero-initialised by the caller)."""
        self._call("gwbp_accumulate_stats", *self._args(), ptr(accum), self._stream())

    def stats(self) -> Dict[str, int]:
        st = Stats()
        self._call("gwbp_read_stats", *self._args(), C.byref(st), self._stream())
        return st.as_dict()

    @staticmethod
    def decode_stats(accum: torch.Tensor) -> Dict[str, int]:
        raw = bytes(accum.cpu().numpy().tobytes())
        return Stats.from_buffer_copy(raw).as_dict()

    def dump_pairs(self, view):
        st = self.stats()
        cap = max(int(st["n_pairs"]), 1)
        gid = torch.empty(cap, dtype=torch.int32, device=self.device)
        pix = torch.empty(cap, dtype=torch.int32, device=self.device)
        w = torch.empty(cap, device=self.device)
        n = C.c_int64(0)
        self._call("gwbp_dump_pairs", *self._args(), C.byref(view), C.c_int64(cap), ptr(gid), ptr(pix), ptr(w),
                                       C.byref(n), self._stream())
        k = int(n.value)
        return gid[:k], pix[:k], w[:k]
