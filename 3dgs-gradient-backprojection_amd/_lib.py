"""ctypes binding of libgwbp.so (the C ABI in include/gwbp.h).

The product path has NO fallback: if the HIP library is missing or fails to load, every operator raises.
`import torch` must happen before the library is loaded so that libgwbp.so binds to the libamdhip64.so.7
that PyTorch-ROCm already mapped (one HIP runtime per process).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import torch  # noqa: F401  (must precede CDLL: see module docstring)

# The view pipeline (backproject.ViewPipeline) runs the front stages of the next views on side streams beside the scatter kernel
# of the current one; that only overlaps if the streams sit on DIFFERENT hardware queues.  The HIP runtime creates 4 by default
# (RCCL takes some of them) and reads GPU_MAX_HW_QUEUES once, when it starts: ask for 8 here, at package import -- which is
# before the first HIP call in every entry point of this repository (bench.py, run_backproject.py, the tests) -- unless the
# caller has chosen a value.  If the runtime was already up when the package was imported the setting comes too late;
# hw_queues_ok() then says so and ViewPipeline refuses to pretend (a pipeline whose streams share a queue runs front + scatter
# back to back: 5.06 instead of 4.27 ms/view at C2 when this was found).
HW_QUEUES_WANTED = 8
_QUEUES_LATE = "GPU_MAX_HW_QUEUES" not in os.environ and torch.cuda.is_initialized()
# SIDE EFFECT OF IMPORTING THIS PACKAGE: os.environ["GPU_MAX_HW_QUEUES"] = "8" unless the variable is already set (child
# processes inherit it, which is what a launcher of per-GPU ranks wants).  When the runtime is already up the variable would
# change nothing in this process, so it is left alone: the environment then still says what the runtime really read.
if not _QUEUES_LATE:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(HW_QUEUES_WANTED))


def hw_queues_late() -> bool:
    """The HIP runtime was already up when the package was imported, and nobody had set GPU_MAX_HW_QUEUES: the request for
    HW_QUEUES_WANTED queues came too late to take effect."""
    return _QUEUES_LATE


def hw_queues_ok() -> bool:
    """False when the pipeline's streams may be sharing hardware queues: the HIP runtime started before this package could ask
    for HW_QUEUES_WANTED queues, or the caller asked for fewer."""
    try:
        return not _QUEUES_LATE and int(os.environ.get("GPU_MAX_HW_QUEUES", "0")) >= HW_QUEUES_WANTED
    except ValueError:
        return False


_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgwbp.so")
CSRC = os.path.join(_HERE, "csrc")

EXPORTS = [
    "gwbp_version", "gwbp_last_error_string", "gwbp_workspace_size", "gwbp_project", "gwbp_bin_sort",
    "gwbp_blend_weights", "gwbp_blend_weights_d", "gwbp_blend_scatter", "gwbp_blend_scatter_encoded", "gwbp_blend_tokens", "gwbp_scatter_tokens", "gwbp_accumulate_d", "gwbp_scatter", "gwbp_scatter_encoded", "gwbp_scatter_upsampled", "gwbp_scatter_bilinear", "gwbp_render", "gwbp_render_pixels", "gwbp_sh_colors",
    "gwbp_backproject_view", "gwbp_encode_map", "gwbp_finalize",
    "gwbp_accumulate_stats", "gwbp_read_stats", "gwbp_dump_pairs",
]


class View(C.Structure):
    _fields_ = [("viewmat", C.c_float * 16), ("K", C.c_float * 9), ("width", C.c_int32), ("height", C.c_int32),
                ("near_plane", C.c_float), ("far_plane", C.c_float), ("eps2d", C.c_float),
                ("radius_clip", C.c_float)]


class Caps(C.Structure):
    _fields_ = [("n_gaussians", C.c_int64), ("isect_cap", C.c_int64), ("pair_cap", C.c_int64),
                ("max_width", C.c_int32), ("max_height", C.c_int32), ("scatter_workgroups", C.c_int32),
                ("flags", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("n_isect", C.c_uint32), ("n_visible", C.c_uint32),
                ("n_headers", C.c_uint32), ("pool_used", C.c_uint32), ("overflow", C.c_uint32),
                ("reserved", C.c_uint32)]

    def as_dict(self):
        # ("reserved" = what the last blend left in the workspace: 0 store, 1 + half-tile lists, 2 nothing)
        return {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "reserved"} | {"blend_kind": int(self.reserved)}


FLAG_TIGHT_BINNING = 1  # GWBP_FLAG_TIGHT_BINNING (include/gwbp.h)
FLAG_FRONT_PRIORITY = 2  # GWBP_FLAG_FRONT_PRIORITY
FLAG_NARROW_SCATTER = 4  # GWBP_FLAG_NARROW_SCATTER
FLAG_SPLIT_ENCODER = 16  # GWBP_FLAG_SPLIT_ENCODER


class GwbpError(RuntimeError):
    pass


PROFILE_LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "tools", "lib", "libgwbp_profile.so"))


def _sources():
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".map")) or f == "Makefile"]
    srcs.append(os.path.join(_HERE, "..", "include", "gwbp.h"))
    srcs.append(os.path.join(_HERE, "..", "tools", "check_asm_hazards.py"))  # the build-time gate of scatter_wide
    return [s for s in srcs if os.path.exists(s)]


def _stale(target: str) -> bool:
    return not os.path.exists(target) or any(os.path.getmtime(s) > os.path.getmtime(target) for s in _sources())


def build(force: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into the in-tree libgwbp.so (hipcc cross-compiles without a GPU)."""
    if force or _stale(LIB_PATH):
        if not os.path.exists("/opt/rocm/bin/hipcc"):
            raise GwbpError("libgwbp.so is stale/missing and hipcc is not available to rebuild it")
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
    return LIB_PATH


def build_profile(force: bool = False) -> str:
    """`make PROFILE=1`: tools/lib/libgwbp_profile.so from the SAME sources as the product library (ablation knobs + in-kernel
    stamps: another register allocation of every kernel).  It is the subject of the differential test of the 256-channel scatter
    kernel against the 128-channel one (tests/test_gpu_parity.py); __graft_entry__.build() keeps it current so that the test
    never runs on a stale binary.  Never loaded by the product path (use_library(..., allow_profile=True) only)."""
    if force or _stale(PROFILE_LIB_PATH):
        if not os.path.exists("/opt/rocm/bin/hipcc"):
            raise GwbpError("libgwbp_profile.so is stale/missing and hipcc is not available to rebuild it")
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j8", "PROFILE=1"])
    return PROFILE_LIB_PATH


_P, _I64, _I32, _F, _SZ = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_size_t
_WS = [C.POINTER(Caps), _P, _SZ]            # caps, workspace, workspace_bytes
_WSV = _WS + [C.POINTER(View)]              # ... + view_host
_MAP = [_P, _I64, _I64, _I64, _I32]         # feats, fs_y, fs_x, fs_c, D
# argument types of every entry point of include/gwbp.h: a Python int passed for an int64_t / size_t / pointer is
# converted (and range-checked) by ctypes instead of being truncated to a C int
ARGTYPES = {
    "gwbp_workspace_size": [C.POINTER(Caps), C.POINTER(C.c_size_t)],
    "gwbp_project": _WSV + [_P] * 8 + [_P],
    "gwbp_bin_sort": _WSV + [_P] * 3 + [_P],
    "gwbp_blend_weights": _WSV + [_P, _P],
    "gwbp_blend_weights_d": _WSV + [_P, _F, _P, _P],
    "gwbp_blend_scatter": _WSV + [_P, _I64, _I64, _I32, _F, _F, _P, _P, _P, _P],
    "gwbp_blend_scatter_encoded": _WSV + [_P, _I64, _I64, _I32, _P, _I32, _F, _F, _P, _P, _P, _P],
    "gwbp_blend_tokens": _WSV + [_P, _P, _P, _P],
    "gwbp_scatter_tokens": _WSV + [_P, _I64, _I64, _I32, _P, _P, _F, _F, _P, _P, _P],
    "gwbp_accumulate_d": _WSV + [_F, _P, _P],
    "gwbp_scatter": _WSV + _MAP + [_F, _F, _P, _P, _P],
    "gwbp_scatter_encoded": _WSV + [_P, _I64, _I64, _I32, _P, _I32, _F, _F, _P, _P, _P],
    "gwbp_scatter_upsampled": _WSV + _MAP + [_P, _P, _F, _F, _P, _P, _P],
    "gwbp_scatter_bilinear": _WSV + _MAP + [_I32, _I32, _P, _P, _P, _P, _F, _F, _P, _P, _P],
    "gwbp_render": _WSV + [_P, _I32, _P, _P],
    "gwbp_render_pixels": _WSV + [_P, _I32, _P, _P, _P],
    "gwbp_sh_colors": [_I64, _I32, _I32, _P, _P, C.POINTER(C.c_float), _P, _P],
    "gwbp_backproject_view": _WSV + [_P] * 4 + _MAP + [_F, _F, _P, _P, _P],
    "gwbp_encode_map": [_P, _I64, _I64, _I32, _I32, _I32, _P, _I32, _P, _I32, _P],
    "gwbp_finalize": [_I64, _I32, _P, _P, _P, _P],
    "gwbp_accumulate_stats": _WS + [_P, _P],
    "gwbp_read_stats": _WS + [C.POINTER(Stats), _P],
    "gwbp_dump_pairs": _WSV + [_I64, _P, _P, _P, C.POINTER(C.c_int64), _P],
}

_lib: Optional[C.CDLL] = None
_lib_path = LIB_PATH
_allow_profile = False


def use_library(path: str, allow_profile: bool = False) -> None:
    """DEVELOPER entry point (tools/, `bench.py --lib`): bind the package to another build of the same C ABI -- an A/B build or
    a PROFILE / ablation build under tools/lib/ -- instead of the in-tree libgwbp.so.  Must be called before the first
    operator.  The product path never looks at the environment for this: nothing but an explicit call can swap the library,
    and a PROFILE build (ablation knobs, possibly invalid results) additionally needs allow_profile=True."""
    global _lib_path, _allow_profile
    if _lib is not None:
        raise GwbpError("use_library() must be called before the library is first used")
    _lib_path, _allow_profile = os.path.abspath(path), bool(allow_profile)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = _lib_path
        if not os.path.exists(path):
            raise GwbpError(f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU or PyTorch fallback for this path)")
        L = C.CDLL(path)
        L.gwbp_version.restype = C.c_char_p
        if b"PROFILE" in L.gwbp_version() and not _allow_profile:
            # a PROFILE build reads ablation knobs from the environment and may produce invalid results: never by accident
            raise GwbpError(f"{path} is a PROFILE build ({L.gwbp_version().decode()}); it is only loaded through "
                            "use_library(path, allow_profile=True) (bench.py --lib, tools/stamp_scatter.py)")
        L.gwbp_last_error_string.restype = C.c_char_p
        for name in EXPORTS[2:]:
            if path != LIB_PATH and not hasattr(L, name):
                continue  # an older A/B build; calling the missing entry point still raises
            fn = getattr(L, name)
            fn.restype = C.c_int
            fn.argtypes = ARGTYPES[name]
        _lib = L
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().gwbp_last_error_string().decode("utf-8", "replace")
        kind = "invalid argument" if rc == -1 else "workspace too small" if rc == -2 else \
            "unsupported" if rc == -3 else f"hipError {rc}"
        raise GwbpError(f"{what} failed ({kind}): {msg}")


def make_view(viewmat, K, width: int, height: int, near_plane=0.01, far_plane=1e10, eps2d=0.3,
              radius_clip=0.0) -> View:
    v = View()
    vm = [float(x) for x in viewmat.detach().reshape(-1).cpu().tolist()]
    kk = [float(x) for x in K.detach().reshape(-1).cpu().tolist()]
    if len(vm) != 16 or len(kk) != 9:
        raise ValueError("viewmat must be 4x4 and K 3x3")
    v.viewmat[:] = vm
    v.K[:] = kk
    v.width, v.height = int(width), int(height)
    v.near_plane, v.far_plane, v.eps2d, v.radius_clip = near_plane, far_plane, eps2d, radius_clip
    return v


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())
