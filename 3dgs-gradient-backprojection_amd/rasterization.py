"""Drop-in for `gsplat.rasterization` as the reference calls it.

Reference call sites: backproject.py:89-100 (SH render), :115-125 / :133-143 (zeros colours, differentiated),
utils.py:238-249 (keyword viewmats/Ks, 0-d tensor width/height), click_and_segment.py:241-254 ("RGB+D").
Signature, defaults and return triple follow gsplat 1.4.0.

Forward  = project -> bin/sort -> blend_weights -> render   (all HIP kernels behind the C ABI)
Backward = scatter with v_render_colors as the feature map: colors.grad[g,:] += sum_p w_g(p) v_render[p,:]
           -- exactly what backproject.py:127-131 harvests.  Only `colors` receives a gradient (the reference
           differentiates nothing else on this path); requesting other gradients raises.
"""
from __future__ import annotations

import weakref
from typing import Dict, Optional, Tuple

import torch
from torch.utils import _pytree

from ._lib import GwbpError
from .engine import TILE, Engine

_ENGINES: Dict[Tuple, Engine] = {}
# The reference's harvesting pattern hands the SAME all-zero colour table to every view (backproject.py:67-72): whether a
# table is all zero is found out once per (tensor object, in-place version), not once per call (a 2 GB reduction + a host
# synchronisation per view at C2).
_ZERO_TABLES: Dict[int, Tuple] = {}
# ... and the render of an all-zero table is all zero: one cached [H,W,D] buffer per device and shape instead of a 3.5 GB
# memset per view.  The buffer is handed out as the operator's output; if anybody wrote into it in place (its version
# counter moved) it is zeroed again first.
_ZERO_RENDERS: Dict[Tuple, Tuple] = {}


# rows of a table that are re-checked on EVERY call (see _is_zero_table)
_SAMPLE_ROWS = 257
# colour tables of at most this many channels are rendered by the pixel-parallel kernel (gwbp_render_pixels: no weight store;
# C2: 0.59 ms for RGB, 0.91 ms for 16 channels, where blend + weight-store render take 0.75 + 1.13; the kernel goes up to 32
# channels, but at 1.88 ms it no longer beats that route there)
PIXEL_RENDER_MAX_DIM = 16


def invalidate_zero_table_cache() -> None:
    """Forget every cached "this colour table is all zero" verdict (and the cached zero renders).  Call it after writing into
    a differentiable colour table through `.data`, a raw pointer or anything else that does not move the tensor's version
    counter; in-place torch ops on the tensor itself need nothing."""
    _ZERO_TABLES.clear()
    _ZERO_RENDERS.clear()


def _sample_is_zero(colors: torch.Tensor) -> bool:
    n = colors.shape[0]
    step = max(1, n // _SAMPLE_ROWS)
    return not bool(colors.detach()[::step].any())


def _is_zero_table(colors: torch.Tensor) -> bool:
    """Is the differentiable colour table all zero (the reference's harvesting pattern, backproject.py:67-72)?  The full
    reduction runs once per (tensor object, in-place version).  Writes that do not move the version counter (`.data`, raw
    pointers: old-style optimisers, checkpoint loads) are caught by re-checking a strided sample of ~257 rows on every call
    -- dense rewrites of the table always hit it; a write confined to rows outside the sample does not, which is what
    `invalidate_zero_table_cache()` is for."""
    key = id(colors)
    hit = _ZERO_TABLES.get(key)
    # (the storage address is part of the key: `t.data = other` keeps object and version but swaps the memory)
    if hit is not None and hit[0]() is colors and hit[1] == colors._version and hit[3] == colors.data_ptr():
        if not hit[2] or _sample_is_zero(colors):
            return hit[2]
    z = not bool(colors.detach().any())
    if len(_ZERO_TABLES) > 16:
        _ZERO_TABLES.clear()
    _ZERO_TABLES[key] = (weakref.ref(colors), colors._version, z, colors.data_ptr())
    return z


def _zero_render(dev, h: int, w: int, d: int) -> torch.Tensor:
    """The all-zero [H,W,D] render of an all-zero table: ONE cached storage per device and shape, but a FRESH tensor object per
    call.  autograd hangs the producing node on the returned tensor object itself: handing the same object out twice would
    re-point the first call's output at the second call's node, and two harvest forwards alive before either backward() -- or
    one call with C > 1 cameras -- would send every gradient to the last view's node.  `detach()` shares storage and version
    counter (an in-place write through any handed-out alias is seen here and the buffer zeroed again) and nothing else."""
    key = (str(dev), h, w, d)
    hit = _ZERO_RENDERS.get(key)
    if hit is not None and hit[0]._version == hit[1]:
        return hit[0].detach()
    if hit is not None:
        hit[0].zero_()
        buf = hit[0]
    else:
        if len(_ZERO_RENDERS) >= 2:  # (the reference alternates between [H,W,512] and [H,W,3])
            _ZERO_RENDERS.pop(next(iter(_ZERO_RENDERS)))
        buf = torch.zeros(h, w, d, device=dev, dtype=torch.float32)
    _ZERO_RENDERS[key] = (buf, buf._version)
    return buf.detach()


def _accumulate_node(leaf: torch.Tensor):
    """The AccumulateGrad node of a leaf (what .backward() runs to add into leaf.grad)."""
    with torch.enable_grad():
        return leaf.view_as(leaf).grad_fn.next_functions[0][0]


def _grad_is_unobserved(leaf: torch.Tensor, node) -> bool:
    """May the backward add straight into leaf.grad instead of returning a gradient?  Only when nobody can tell the
    difference: the engine is going to run the leaf's AccumulateGrad node itself (a .backward() call; under
    torch.autograd.grad() the gradient is captured and returned instead, and the engine query raises), and no tensor hook or
    post-accumulate-grad hook is registered on the leaf."""
    if getattr(leaf, "_backward_hooks", None) or getattr(leaf, "_post_accumulate_grad_hooks", None):
        return False
    try:
        return bool(torch._C._will_engine_execute_node(node))
    except (RuntimeError, AttributeError, TypeError):
        # RuntimeError: "... we are currently running autograd.grad()".  The query is a private torch API: if a release
        # renames or re-types it, fall back to the returned-gradient path (always correct, one temporary per view slower).
        return False


def get_engine(device, n: int, width: int, height: int) -> Engine:
    key = (str(device), n, width, height)
    eng = _ENGINES.get(key)
    if eng is None:
        # at most two live workspaces per device (utils.test_proper_pruning alternates between the full and the pruned
        # scene, utils.py:316-340); the least recently created one goes first
        mine = [k for k in _ENGINES if k[0] == key[0]]
        for k in mine[:max(0, len(mine) - 1)]:
            del _ENGINES[k]
        eng = _ENGINES[key] = Engine(n, width, height, device=device)
        eng.generation = 0
    return eng


def _front_key(view, tensors):
    """Identity of a front-stage result: the view parameters and the Gaussian tensors' in-place versions and shapes.  The
    tensors THEMSELVES are kept in the cache entry and compared with `is` (_same_tensors): a (data_ptr, _version) key alone
    would match a NEW temporary that the caching allocator placed at a freed tensor's address (opac.clone() twice from a
    fixed camera: the second call would silently reuse the first one's projection and weight store)."""
    return (bytes(view), tuple((t._version, tuple(t.shape), tuple(t.stride())) for t in tensors))


def _same_tensors(held, tensors) -> bool:
    return held is not None and len(held) == len(tensors) and all(a is b for a, b in zip(held, tensors))


def _run_front(eng: Engine, view, means, quats, scales, opacities, want_alphas, want_meta, want_store=True):
    """project -> sort (-> blend) with auto-grow of the capacities (one host sync per call: this is the
    API-compatible path; the fused driver in backproject.py amortises the check).

    The reference rasterises every view TWICE with the same Gaussians -- zeros [N,512] for the features, zeros [N,3] for the
    denominators (backproject.py:115-125,133-143): the second call finds the first one's projection, sorted lists and
    weight store in the workspace (same view bytes, the SAME tensor objects at the same in-place versions) and skips the
    whole front.  Tensors that are equal but not identical (a fresh .clone() per call) are projected again."""
    gauss = (means, quats, scales, opacities)
    key = _front_key(view, gauss)
    c = getattr(eng, "front_cache", None)
    if (c is not None and c["key"] == key and _same_tensors(c["tensors"], gauss) and (c["store"] or not want_store) and (c["meta"] or not want_meta)
            and (c["alphas"] is not None or not want_alphas) and (c["halves"] or not eng._wide_requested())):
        return c["proj"], c["bins"], (c["alphas"].clone() if want_alphas else None), c["stats"]
    eng.front_cache = None
    while True:
        proj = eng.project(view, means, quats, scales, opacities, want_outputs=want_meta)
        bins = eng.bin_sort(view, want_outputs=want_meta)
        alphas = eng.blend_weights(view, want_alphas=want_alphas) if want_store else None
        st = eng.stats()
        if not st["overflow"]:
            eng.generation += 1
            # `tensors`: strong references -- they pin the storages, so the identity test above cannot be fooled by reuse
            eng.front_cache = dict(key=key, tensors=gauss, proj=proj, bins=bins, alphas=alphas, stats=st,
                                   store=want_store, meta=want_meta, halves=want_store and eng._halves)
            return proj, bins, alphas, st
        eng.grow(st)


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, colors, means, quats, scales, opacities, viewmat, K, width, height, kw, holder, harvest=None):
        dev = means.device
        eng = get_engine(dev, means.shape[0], width, height)
        view = eng.view(viewmat, K, width, height, **kw)
        D = colors.shape[1]
        need_store = colors.requires_grad or D > PIXEL_RENDER_MAX_DIM  # the weight store is only needed for backward / the wide render
        # a backward with D % 256 == 0 channels goes through the 256-channel scatter kernel, which needs the blend's
        # half-tile lists: the flag must be in place before this view's blend
        eng.set_narrow_scatter(not (colors.requires_grad and D % 256 == 0))
        # the reference's harvesting pattern: an all-zero differentiable colour table whose render is only there to be
        # back-propagated through (backproject.py:67-72,115-129,133-147) -- the render of zeros is zeros, for any D
        if harvest is None:
            harvest = colors.requires_grad and _is_zero_table(colors)
        proj, bins, alphas, st = _run_front(eng, view, means, quats, scales, opacities, D > PIXEL_RENDER_MAX_DIM or harvest,
                                            holder is not None, want_store=need_store)
        if harvest:
            out = _zero_render(dev, view.height, view.width, D)  # (alphas: the blend's, kept with the front-stage result)
        elif D <= PIXEL_RENDER_MAX_DIM:  # RGB / RGB+D / depth / the 16-d compressed field (segment_compressed.py:154-165):
            # pixel-parallel rasteriser straight from the sorted tile lists
            out, alphas = eng.render_pixels(view, colors.detach())
        else:
            out = eng.render(view, colors.detach())
        ctx.has_store = need_store
        if holder is not None:
            holder.update(proj=proj, bins=bins, stats=st)
        ctx.eng, ctx.view, ctx.gen = eng, view, eng.generation
        ctx.save_for_backward(means, quats, scales, opacities)
        ctx.shape = colors.shape
        # the leaf whose .grad the backward may add into directly (see backward)
        ctx.leaf = weakref.ref(colors) if (colors.requires_grad and colors.is_leaf) else None
        ctx.leaf_node = _accumulate_node(colors) if ctx.leaf is not None else None
        ctx.mark_non_differentiable(alphas)
        return out, alphas

    @staticmethod
    def backward(ctx, g_out, g_alpha):
        if any(ctx.needs_input_grad[1:7]):
            raise NotImplementedError("only d/d(colors) is implemented: the reference differentiates nothing else "
                                      "(backproject.py:67-72,129,147)")
        eng, view = ctx.eng, ctx.view
        means, quats, scales, opacities = ctx.saved_tensors
        if eng.generation != ctx.gen or eng.n != means.shape[0] or not ctx.has_store:
            # the workspace was reused by another call since forward: rebuild this view's weight store
            eng = get_engine(means.device, means.shape[0], view.width, view.height)
            eng.set_narrow_scatter(ctx.shape[1] % 256 != 0)
            eng.front_cache = None
            _run_front(eng, view, means, quats, scales, opacities, False, False)
        # The reference keeps ONE grad tensor per colour table alive (it clones and zeroes it in place, backproject.py:130-131),
        # so a returned gradient would be ADDED to it by autograd: a 2 GB temporary, its memset and a 6 GB read-modify-write
        # per view at C2.  The scatter kernel accumulates anyway: add straight into the leaf's .grad and hand autograd nothing
        # -- but only where that cannot be observed (_grad_is_unobserved: a plain .backward(), no hooks on the leaf;
        # torch.autograd.grad() and hooked leaves get the returned gradient).
        leaf = ctx.leaf() if ctx.leaf is not None else None
        acc = leaf.grad if leaf is not None else None
        if (acc is not None and acc.dtype == torch.float32 and acc.shape == ctx.shape and acc.is_contiguous()
                and acc.device == means.device and not torch.is_grad_enabled()
                and _grad_is_unobserved(leaf, ctx.leaf_node)):
            _scatter_grad(eng, view, g_out, acc)
            acc[:0].add_(0)  # the kernel wrote through data_ptr: move .grad's version counter like an in-place op would
            return (None,) * 12
        v_colors = torch.zeros(ctx.shape, device=means.device, dtype=torch.float32)
        _scatter_grad(eng, view, g_out, v_colors)
        return (v_colors,) + (None,) * 11


def _scatter_grad(eng: Engine, view, g_out: torch.Tensor, acc: torch.Tensor) -> None:
    """acc[g, :] += sum_p w_g(p) g_out[p, :].  The gradient of `render.sum()` is ONE value expanded over [H,W,D] (all strides
    zero): its scatter is that value times the per-Gaussian weight sums, which a view blended for the 256-channel kernel
    already holds per record (the reference's denominator pass, backproject.py:145-147, right after its 512-channel one)."""
    if g_out.dim() == 3 and g_out.numel() > 0 and not any(g_out.stride()) and eng.has_weight_sums():
        eng.scatter_uniform(view, g_out[0, 0, 0], acc)
    else:
        eng.scatter(view, g_out, acc, None)


# ---- the harvest statement `(render[0] * feats).sum().backward()` without its three [H,W,D] passes --------------------------
# backproject.py:127-129.  `render` is the render of an all-zero table, i.e. zeros: the product is zeros, its sum is 0, and
# the gradient that reaches the rasteriser's backward is feats itself.  Run literally, torch spends 3.8 ms per C2 view on it
# (multiply 1.8, sum 0.9, the multiply's backward 1.1: each a pass over 3.47 GB) -- more than the scatter kernel that does the
# work.  The render of a harvest call is therefore handed out as a tensor SUBCLASS that recognises exactly this statement:
#   render * feats   (either order, Tensor.mul too; same shape / dtype / device, feats a plain tensor without grad)
#       -> _HarvestProduct: no kernel; its data ARE the zeros of the render
#   product.sum() / product.mean()    (the full reductions: lseg backproject.py:127, dino :263)
#       -> a scalar 0 whose backward hands `feats` (times the incoming scalar, read on the host: 1.0 for a plain
#          .backward(); times 1 / numel for the mean) to the rasteriser's backward as the gradient of the render -- no copy
#          when that factor is one, one scaled copy otherwise
# EVERYTHING else -- any other operation on the render or on the product, a feats that requires grad, a broadcasting shape --
# falls back to the literal computation (the product is materialised with torch.mul, history and all), so results never depend
# on the shortcut.  One visible difference: with non-finite values in feats (backproject.py:109 can produce NaN pixels) the
# literal `target` is NaN and this one is 0; nothing in the reference reads it, and the gradient -- NaN at exactly those pixels --
# is the same.
# The two tensor subclasses below lean on private torch entry points; if a release drops one of them the shortcut switches
# itself off at import (every statement is then computed literally -- slower, never wrong) and tests/test_host_logic.py fails
# visibly on the missing name.
PRIVATE_TORCH_APIS = ("torch._C.DisableTorchFunctionSubclass", "torch._C._disabled_torch_function_impl",
                      "torch._C._will_engine_execute_node", "torch.utils._pytree.tree_map")


def missing_private_apis():
    import importlib
    out = []
    for name in PRIVATE_TORCH_APIS:
        mod, _, attr = name.rpartition(".")
        try:
            if not hasattr(importlib.import_module(mod), attr):
                out.append(name)
        except ImportError:
            out.append(name)
    return out


_HARVEST_SHORTCUT = not missing_private_apis()


def set_harvest_shortcut(on: bool) -> None:
    """Switch the recognition of `(render * feats).sum()` on harvest renders on (default) or off (every statement is then
    computed literally by torch: 3.8 ms more per C2 view, same gradients)."""
    global _HARVEST_SHORTCUT
    _HARVEST_SHORTCUT = bool(on)


def _plain(t: torch.Tensor) -> torch.Tensor:
    return t.as_subclass(torch.Tensor) if type(t) is not torch.Tensor else t


class _HarvestSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, render, feats, scale):
        ctx.save_for_backward(feats)  # (saved, not just referenced: an in-place write into feats between the statement and its
        ctx.scale = scale             # backward raises, as it would for the literal product's saved operand)
        return render.new_zeros(())

    @staticmethod
    def backward(ctx, g):
        (feats,) = ctx.saved_tensors
        if torch.is_grad_enabled() or (feats.is_cuda and torch.cuda.is_current_stream_capturing()):
            # backward(create_graph=True) (grad mode is on inside backward only then) or a stream capture: no host read, and
            # the literal product's gradient with its history -- d(sum(render * feats)) / d(render) = g * feats
            return feats * (g * ctx.scale), None, None
        k = float(g) * ctx.scale  # (a host read of one scalar; the loop's other statements synchronise anyway)
        return (feats if k == 1.0 else feats * k), None, None


_MUL_FUNCS = (torch.mul, torch.Tensor.mul, torch.Tensor.__mul__, torch.Tensor.__rmul__, torch.multiply, torch.Tensor.multiply)
_SUM_FUNCS = (torch.sum, torch.Tensor.sum)
_MEAN_FUNCS = (torch.mean, torch.Tensor.mean)  # (the dino variant, backproject.py:263: one scaled copy of feats instead of three passes)
_META_PROPS = ("shape", "dtype", "device", "ndim", "layout", "is_cuda", "is_cpu", "is_sparse", "is_quantized", "is_meta", "names")
_META_METHODS = ("size", "dim", "ndimension", "numel", "nelement", "stride", "element_size", "is_floating_point", "is_complex",
                 "is_contiguous", "get_device", "__len__")


class _HarvestProduct(torch.Tensor):
    """`render * feats` of a harvest render, not computed (see above).  Any use other than the full `.sum()` materialises it."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        me = next((a for a in args if isinstance(a, _HarvestProduct)), None)
        if ((func in _SUM_FUNCS or func in _MEAN_FUNCS) and me is not None and len(args) == 1 and not kwargs
                and torch.is_grad_enabled() and getattr(me, "_harvest", None) is not None):
            render, feats = me._harvest
            return _HarvestSum.apply(_plain(render), feats, 1.0 if func in _SUM_FUNCS else 1.0 / max(1, feats.numel()))
        # metadata never needs the data: properties (shape, dtype, device, ...) and the size-like methods answer from the
        # placeholder; `requires_grad` answers what the literal product would
        name = getattr(func, "__name__", "")
        if name == "__get__":
            prop = getattr(getattr(func, "__self__", None), "__name__", "")
            if prop == "requires_grad":
                return True
            if prop in _META_PROPS:
                with torch._C.DisableTorchFunctionSubclass():
                    return func(*args, **kwargs)
        elif name in _META_METHODS:
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        with torch._C.DisableTorchFunctionSubclass():
            def real(a):
                if isinstance(a, _HarvestProduct) and getattr(a, "_harvest", None) is not None:
                    return torch.mul(_plain(a._harvest[0]), a._harvest[1])  # the literal product, with its history
                return _plain(a) if isinstance(a, torch.Tensor) else a
            # (tree_map: the placeholder may sit inside a list or tuple argument -- torch.stack([p, q]), torch.cat -- where it
            # must not be read as the detached zeros it physically is)
            args, kwargs = _pytree.tree_map(real, (tuple(args), dict(kwargs)))
            return func(*args, **kwargs)


class _HarvestRender(torch.Tensor):
    """[H,W,D] render of an all-zero differentiable colour table (attached to the autograd graph like any output)."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _MUL_FUNCS and len(args) == 2 and not kwargs and torch.is_grad_enabled():
            a, b = args
            r, f = (a, b) if isinstance(a, _HarvestRender) else (b, a)
            if (isinstance(r, _HarvestRender) and type(f) is torch.Tensor and r.requires_grad and not f.requires_grad
                    and f.shape == r.shape and f.dtype == r.dtype and f.device == r.device):
                p = _plain(r).detach().as_subclass(_HarvestProduct)  # (data: the render's zeros; no kernel)
                p._harvest = (r, f)
                return p
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


class _OneCameraBatch(torch.Tensor):
    """render_colors [1,H,W,D] of a one-camera call.  The reference takes `output_for_grad[0]` of it and back-propagates
    (backproject.py:127-129,145-147): autograd's backward of that select is a zero-filled [1,H,W,D] plus a copy of the
    [H,W,D] gradient into it -- 7 GB of traffic per view at C2 for an axis of length one.  `batch[0]` of this class hands back
    the [H,W,D] tensor the batch IS a view of (same storage, same autograd history, no extra node); every other index and
    every torch op behaves as on a plain tensor and returns plain tensors."""
    __torch_function__ = getattr(torch._C, "_disabled_torch_function_impl", torch.Tensor.__torch_function__)

    def __getitem__(self, idx):
        cam0 = getattr(self, "_camera0", None)
        if cam0 is not None and type(idx) is int and idx in (0, -1):
            return cam0
        return super().__getitem__(idx)


def one_camera_batch(render: torch.Tensor) -> torch.Tensor:
    batch = render[None].as_subclass(_OneCameraBatch)
    batch._camera0 = render
    return batch


class LazyMeta(dict):
    """meta dict of gsplat.rasterization (packed=True layout); the packed index tensors are materialised on
    first access because building them (nonzero) synchronises the host."""

    def __init__(self, eager, holders, n_cameras):
        super().__init__(eager)
        self._h, self._c = holders, n_cameras

    _LAZY = ("camera_ids", "gaussian_ids", "radii", "means2d", "depths", "conics", "opacities", "tiles_per_gauss",
             "isect_ids", "flatten_ids", "isect_offsets")

    def _build(self):
        cams, gids, parts = [], [], {k: [] for k in ("radii", "means2d", "depths", "conics")}
        isect, flat, offs, base = [], [], [], 0
        for c, h in enumerate(self._h):
            p, b = h["proj"], h["bins"]
            vis = torch.nonzero(p["radii"] > 0)[:, 0]
            cams.append(torch.full_like(vis, c))
            gids.append(vis)
            for k in parts:
                parts[k].append(p[k][vis])
            n = h["stats"]["n_isect"]
            remap = torch.full((p["radii"].shape[0],), -1, dtype=torch.int64, device=vis.device)
            remap[vis] = torch.arange(vis.numel(), device=vis.device) + base
            # gsplat packs camera | tile | depth with floor(log2(n_tiles)) + 1 tile bits
            isect.append(b["isect_ids"][:n] | (c << 32 + int(b["tile_offsets"].numel() - 1).bit_length()))
            flat.append(remap[b["flatten_ids"][:n].long()].to(torch.int32))
            offs.append(b["tile_offsets"][:-1] + sum(x.numel() for x in flat[:-1]))
            base += vis.numel()
        self["camera_ids"], self["gaussian_ids"] = torch.cat(cams), torch.cat(gids)
        for k in parts:
            self[k] = torch.cat(parts[k])
        self["opacities"] = self._opac[self["gaussian_ids"]]
        self["isect_ids"], self["flatten_ids"] = torch.cat(isect), torch.cat(flat)
        self["isect_offsets"] = torch.stack(offs).reshape(self._c, self["tile_height"], self["tile_width"])
        r = self["radii"].to(torch.float32) / TILE
        m = self["means2d"] / TILE
        tw, th = self["tile_width"], self["tile_height"]
        x0 = torch.clamp(torch.floor(m[:, 0] - r), 0, tw)
        x1 = torch.clamp(torch.ceil(m[:, 0] + r), 0, tw)
        y0 = torch.clamp(torch.floor(m[:, 1] - r), 0, th)
        y1 = torch.clamp(torch.ceil(m[:, 1] + r), 0, th)
        self["tiles_per_gauss"] = ((x1 - x0) * (y1 - y0)).to(torch.int32)

    def __getitem__(self, k):
        if k in self._LAZY and not dict.__contains__(self, k):
            self._build()
        return dict.__getitem__(self, k)

    def __contains__(self, k):
        return k in self._LAZY or dict.__contains__(self, k)


def rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height, near_plane: float = 0.01,
                  far_plane: float = 1e10, radius_clip: float = 0.0, eps2d: float = 0.3,
                  sh_degree: Optional[int] = None, packed: bool = True, tile_size: int = 16, backgrounds=None,
                  render_mode: str = "RGB", sparse_grad: bool = False, absgrad: bool = False,
                  rasterize_mode: str = "classic", channel_chunk: int = 32, distributed: bool = False,
                  camera_model: str = "pinhole", covars=None, want_meta: bool = True):
    """Same call signature / defaults / returns as gsplat.rasterization (1.4.0).

    Returns (render_colors [C,H,W,D'], render_alphas [C,H,W,1], meta).  `channel_chunk`, `packed`, `sparse_grad`
    are accepted for compatibility and have no effect (the kernels are channel-count generic and unpacked).
    """
    if tile_size != TILE:
        raise GwbpError(f"tile_size must be {TILE}: the tile rectangle is part of the numerics")
    if camera_model != "pinhole" or covars is not None or distributed or absgrad or rasterize_mode != "classic":
        raise NotImplementedError("only pinhole / classic / single-process rasterization is on the reference's path")
    if render_mode not in ("RGB", "D", "ED", "RGB+D", "RGB+ED"):
        raise ValueError(render_mode)
    if not means.is_cuda:
        raise GwbpError("rasterization() needs HIP tensors (the reference likewise requires CUDA, backproject.py:314)")
    width, height = int(width), int(height)  # utils.py:247-248 passes 0-d tensors
    C_ = viewmats.shape[0]
    N = means.shape[0]
    kw = dict(near_plane=near_plane, far_plane=far_plane, eps2d=eps2d, radius_clip=radius_clip)

    outs, alphas, holders = [], [], []
    for c in range(C_):
        vm, K = viewmats[c], Ks[c]
        if sh_degree is not None:
            # colors is [N,K,3] (or [C,N,K,3]); view-dependent colour + 0.5, clamped at 0 (gsplat semantics)
            sh = colors if colors.dim() == 3 else colors[c]
            vm_h = vm.detach().cpu()
            campos = (-(vm_h[:3, :3].T @ vm_h[:3, 3])).tolist()
            if sh.requires_grad:
                raise NotImplementedError("gradients w.r.t. SH coefficients are not on the reference's path")
            cols = get_engine(means.device, N, width, height).sh_colors(sh_degree, means, sh, campos)
        else:
            cols = colors if colors.dim() == 2 else colors[c]
        holder = {} if (want_meta or "D" in render_mode) else None
        need_depth = render_mode in ("D", "ED", "RGB+D", "RGB+ED")
        if need_depth:
            z = (means @ vm[:3, :3].T + vm[:3, 3])[:, 2:3]
            cols = z if render_mode in ("D", "ED") else torch.cat([cols, z], dim=1)
        cols = cols.contiguous()
        harvest = bool(cols.requires_grad and _is_zero_table(cols))
        out, alpha = _Rasterize.apply(cols, means, quats, scales, opacities, vm, K, width, height, kw, holder, harvest)
        if _HARVEST_SHORTCUT and harvest and render_mode == "RGB" and backgrounds is None and out.requires_grad:
            out = out.as_subclass(_HarvestRender)  # recognises `(render * feats).sum()`, behaves like a tensor otherwise
        alpha = alpha[..., None]
        if render_mode in ("ED", "RGB+ED"):
            out = torch.cat([out[..., :-1], out[..., -1:] / alpha.clamp_min(1e-10)], dim=-1)
        if backgrounds is not None:
            out = out + (1.0 - alpha) * backgrounds[c]
        outs.append(out)
        alphas.append(alpha)
        holders.append(holder)
    tw, th = -(-width // TILE), -(-height // TILE)
    eager = dict(tile_width=tw, tile_height=th, width=width, height=height, tile_size=TILE, n_cameras=C_)
    meta = LazyMeta(eager, holders, C_) if want_meta else eager
    if want_meta:
        meta._opac = opacities
    if C_ == 1:  # (no copy of a 3.5 GB render: a view with the camera axis in front)
        return one_camera_batch(outs[0]), alphas[0][None], meta
    return torch.stack(outs), torch.stack(alphas), meta
