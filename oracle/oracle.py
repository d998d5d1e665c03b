"""ctypes front-end of the CPU oracle (oracle/gwbp_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product package.  PARITY UNPINNED: see the header of gwbp_oracle.c.

`backproject_oracle` is the counterpart of the per-view loop body + finalise of
create_feature_field_lseg (backproject.py:62-63,83-86,115-151,166-169) with the 2-D feature network
replaced by caller-supplied feature maps.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Callable, Dict, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liborc.so")
    src = os.path.join(_HERE, "gwbp_oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liborc.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        os.environ.setdefault("OMP_NUM_THREADS", str(usable_cores()))  # read when libgomp starts (sort, projection)
        _LIB = C.CDLL(build())
        _LIB.orc_count_isects.restype = C.c_int64
        _LIB.orc_blend_pairs.restype = C.c_int64
        _LIB.orc_exp_neg_export.restype = C.c_float
        _LIB.orc_exp_neg_export.argtypes = [C.c_float]
    return _LIB


def usable_cores() -> int:
    """Threads worth starting: the CPUs this process may run on, capped by the container's CPU quota (cgroup v2
    cpu.max / v1 cfs quota).  On the GPU boxes os.cpu_count() is 256 but the quota is 16 CPUs: 256 OpenMP threads there
    run 5-50x SLOWER than 16."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, n)


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def set_tunables(radius_floor: float = 0.01, exp_ulp: int = 0) -> None:
    """Sensitivity knobs of SURVEY.md 3.3's uncertainty register (tests/test_sensitivity.py); no arguments = contract."""
    lib().orc_set_tunables(C.c_float(radius_floor), C.c_int(exp_ulp))


def exp_neg(x: float) -> float:
    return float(lib().orc_exp_neg_export(C.c_float(x)))


def project(means, quats, scales, viewmat, K, W: int, H: int, near=0.01, far=1e10, eps2d=0.3, radius_clip=0.0,
            tile_size=16) -> Dict[str, np.ndarray]:
    means, quats, scales = _f32(means), _f32(quats), _f32(scales)
    viewmat, K = _f32(viewmat).reshape(16), _f32(K).reshape(9)
    n = means.shape[0]
    out = dict(
        means2d=np.zeros((n, 2), np.float32), depths=np.zeros(n, np.float32), conics=np.zeros((n, 3), np.float32),
        radii=np.zeros(n, np.int32), rect=np.zeros((n, 4), np.int32))
    rc = lib().orc_project(C.c_int64(n), _p(means), _p(quats), _p(scales), _p(viewmat), _p(K), W, H,
                           C.c_float(near), C.c_float(far), C.c_float(eps2d), C.c_float(radius_clip), tile_size,
                           _p(out["means2d"]), _p(out["depths"]), _p(out["conics"]), _p(out["radii"]),
                           _p(out["rect"]))
    if rc:
        raise RuntimeError(f"orc_project failed: {rc}")
    return out


def bin_sort(proj: Dict[str, np.ndarray], W: int, H: int, tile_size=16) -> Dict[str, np.ndarray]:
    tw, th = -(-W // tile_size), -(-H // tile_size)
    n = proj["radii"].shape[0]
    n_isect = int(lib().orc_count_isects(C.c_int64(n), _p(proj["radii"]), _p(proj["rect"])))
    ids = np.zeros(max(n_isect, 1), np.int64)
    flat = np.zeros(max(n_isect, 1), np.int32)
    offs = np.zeros(tw * th + 1, np.int32)
    rc = lib().orc_bin_sort(C.c_int64(n), _p(proj["depths"]), _p(proj["radii"]), _p(proj["rect"]), tw, th,
                            C.c_int64(n_isect), _p(ids), _p(flat), _p(offs))
    if rc:
        raise RuntimeError(f"orc_bin_sort failed: {rc}")
    return dict(n_isect=n_isect, isect_ids=ids[:n_isect], flatten_ids=flat[:n_isect], tile_offsets=offs,
                tile_w=tw, tile_h=th)


def blend_pairs(proj, bins, opacities, W: int, H: int, tile_size=16, want_alphas=False):
    """All contributing (gaussian, pixel, w) triples of one view + optional alpha map."""
    op = _f32(opacities)
    alphas = np.zeros((H, W), np.float32) if want_alphas else None
    args = (W, H, tile_size, _p(bins["tile_offsets"]), _p(np.ascontiguousarray(bins["flatten_ids"])),
            _p(proj["means2d"]), _p(proj["conics"]), _p(op))
    n = int(lib().orc_blend_pairs(*args, C.c_int64(0), None, None, None, None))
    if n < 0:
        raise RuntimeError(f"orc_blend_pairs failed: {n}")
    gid, pix, w = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float32)
    n2 = int(lib().orc_blend_pairs(*args, C.c_int64(n), _p(gid), _p(pix), _p(w), _p(alphas)))
    assert n2 == n
    return gid[:n], pix[:n], w[:n], alphas


def blend_scatter(proj, bins, opacities, feats: np.ndarray, F: np.ndarray, d: np.ndarray, W: int, H: int,
                  tile_size=16, nthreads: Optional[int] = None, want_alphas=False,
                  row_of: Optional[np.ndarray] = None):
    """F[N,D] += sum_p w feats[p,:], d[N] += sum_p w.  F/d float64 (exact sums) or float32 (timing).
    row_of (int32[N], optional): Gaussian g accumulates into row row_of[g] of F/d, < 0 = skipped (F, d then have as
    many rows as the subset: full-size checks of scenes whose whole F would not fit the host)."""
    op = _f32(opacities)
    assert feats.dtype == np.float32 and feats.ndim == 3 and feats.shape[:2] == (H, W)
    D = feats.shape[2]
    assert F.shape[1] == D and F.dtype == d.dtype and F.dtype in (np.float32, np.float64)
    assert F.flags.c_contiguous and d.flags.c_contiguous
    fs = [s // 4 for s in feats.strides]
    alphas = np.zeros((H, W), np.float32) if want_alphas else None
    npairs = C.c_int64(0)
    nt = nthreads or usable_cores()
    if row_of is not None:
        row_of = np.ascontiguousarray(row_of, dtype=np.int32)
        assert row_of.shape == (op.shape[0],) and int(row_of.max()) < F.shape[0]
    rc = lib().orc_blend_scatter(C.c_int64(F.shape[0]), D, W, H, tile_size, _p(bins["tile_offsets"]),
                                 _p(np.ascontiguousarray(bins["flatten_ids"])), _p(proj["means2d"]),
                                 _p(proj["conics"]), _p(op), _p(feats), C.c_int64(fs[0]), C.c_int64(fs[1]),
                                 C.c_int64(fs[2]), int(F.dtype == np.float64), _p(F), _p(d), _p(alphas),
                                 C.byref(npairs), nt, _p(row_of))
    if rc:
        raise RuntimeError(f"orc_blend_scatter failed: {rc}")
    return int(npairs.value), alphas


def render(proj, bins, opacities, colors, W: int, H: int, tile_size=16):
    op, colors = _f32(opacities), _f32(colors)
    D = colors.shape[1]
    out = np.zeros((H, W, D), np.float32)
    alphas = np.zeros((H, W), np.float32)
    rc = lib().orc_render(C.c_int64(colors.shape[0]), D, W, H, tile_size, _p(bins["tile_offsets"]),
                          _p(np.ascontiguousarray(bins["flatten_ids"])), _p(proj["means2d"]), _p(proj["conics"]),
                          _p(op), _p(colors), _p(out), _p(alphas))
    if rc:
        raise RuntimeError(f"orc_render failed: {rc}")
    return out, alphas


def sh_colors(degree: int, means, coeffs, campos) -> np.ndarray:
    means, coeffs, campos = _f32(means), _f32(coeffs), _f32(campos)
    out = np.zeros((means.shape[0], 3), np.float32)
    rc = lib().orc_sh_colors(C.c_int64(means.shape[0]), degree, coeffs.shape[1], _p(means), _p(coeffs), _p(campos), _p(out))
    if rc:
        raise RuntimeError(f"orc_sh_colors failed: {rc}")
    return out


def finalize(F: np.ndarray, d: np.ndarray) -> np.ndarray:
    out = np.zeros(F.shape, np.float32)
    lib().orc_finalize(C.c_int64(F.shape[0]), F.shape[1], int(F.dtype == np.float64), _p(F), _p(d), _p(out))
    return out


def backproject_view(means, quats, scales, opacities, viewmat, K, W, H, feats, F, d, nthreads=None, row_of=None,
                     **kw):
    proj = project(means, quats, scales, viewmat, K, W, H, **kw)
    bins = bin_sort(proj, W, H)
    npairs, _ = blend_scatter(proj, bins, opacities, feats, F, d, W, H, nthreads=nthreads, row_of=row_of)
    return dict(n_pairs=npairs, n_isect=bins["n_isect"], n_vis=int((proj["radii"] > 0).sum()), proj=proj, bins=bins)


def backproject_oracle(means, quats, scales, opacities, viewmats: Sequence, K, W: int, H: int,
                       feature_fn: Callable[[int], np.ndarray], D: int, acc=np.float64, reduction="sum",
                       nthreads=None):
    """Whole loop of create_feature_field_* with supplied feature maps.  Returns out[N,D], F, d, stats."""
    n = np.asarray(means).shape[0]
    F, d = np.zeros((n, D), acc), np.zeros(n, acc)
    stats = []
    for v, vm in enumerate(viewmats):
        feats = _f32(feature_fn(v))
        stats.append({k: val for k, val in backproject_view(means, quats, scales, opacities, vm, K, W, H, feats, F,
                                                            d, nthreads=nthreads).items() if k.startswith("n_")})
    if reduction == "mean":  # backproject.py:263,283 (dino): .mean() instead of .sum()
        F = F / float(H * W * D)
        d = d / float(H * W * 3)
    elif reduction != "sum":
        raise ValueError(reduction)
    return finalize(np.ascontiguousarray(F), np.ascontiguousarray(d)), F, d, stats
