"""No-GPU checks of the C-ABI library: it loads, exports every symbol include/gwbp.h declares, and its pure-host
entry points (workspace sizing, argument validation, error strings) behave.  No compute calls."""
import ctypes as C
import os
import re

import pytest

import gsbp_amd
from gsbp_amd import _lib

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "gwbp.h")


def declared_symbols():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gwbp_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    gsbp_amd.build()
    lib = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 13 and set(syms) == set(_lib.EXPORTS)
    for s in syms:
        assert getattr(lib, s) is not None
    assert b"gfx950" in lib.gwbp_version()


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.View) == 16 * 4 + 9 * 4 + 2 * 4 + 4 * 4
    assert C.sizeof(_lib.Caps) == 8 * 3 + 4 * 4
    assert C.sizeof(_lib.Stats) == 32


def test_workspace_size_and_argument_validation():
    lib = _lib.lib()
    n = C.c_size_t(0)
    caps = _lib.Caps(1_000_000, 16_000_000, 217_088_000, 1600, 1060)
    assert lib.gwbp_workspace_size(C.byref(caps), C.byref(n)) == 0
    # g2d 32 + rect 8 + touched 4 + depth-sort key/value ping-pong 16 B per Gaussian; tile key/value ping-pong 16 B +
    # header 64 B per intersection; 8 B per weight-store entry; the carry slices of
    # the 256-channel scatter (256 workgroups x 1024 rows x 1 KB)
    expect = 1_000_000 * 60 + 16_000_000 * 80 + 217_088_000 * 8 + 256 * 1024 * 1024
    assert expect < n.value < expect * 1.02
    # unknown flag bits are rejected (bit 3 was the removed GWBP_FLAG_GROUP_SCATTER)
    caps_g = _lib.Caps(1_000_000, 16_000_000, 217_088_000, 1600, 1060, 0, 8)
    assert lib.gwbp_workspace_size(C.byref(caps_g), C.byref(C.c_size_t(0))) == -1
    assert b"unknown caps.flags" in lib.gwbp_last_error_string()
    bad = _lib.Caps(-1, 16, 1 << 20, 64, 64)
    assert lib.gwbp_workspace_size(C.byref(bad), C.byref(n)) == -1
    assert b"caps out of range" in lib.gwbp_last_error_string()
    small = _lib.Caps(10, 1 << 16, 1 << 20, 64, 64)
    view = _lib.View()
    view.width, view.height = 64, 64
    view.K[0] = view.K[4] = 50.0
    # null workspace / too-small workspace are rejected before any HIP call
    assert lib.gwbp_project(C.byref(small), None, C.c_size_t(0), C.byref(view), None, None, None, None, None, None,
                            None, None, None) == -1
    buf = (C.c_char * 4096)()
    addr = (C.addressof(buf) + 255) & ~255
    assert lib.gwbp_project(C.byref(small), C.c_void_p(addr), C.c_size_t(1024), C.byref(view), None, None, None, None,
                            None, None, None, None, None) == -2
    assert b"workspace too small" in lib.gwbp_last_error_string()
    assert lib.gwbp_finalize(C.c_int64(-1), 8, None, None, None, None) == -1


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(gsbp_amd.GwbpError):
        gsbp_amd.Engine(10, 64, 64, device="cpu")
    z = torch.zeros(4, 3)
    with pytest.raises(gsbp_amd.GwbpError):
        gsbp_amd.rasterization(z, torch.zeros(4, 4), z, torch.zeros(4), z, torch.eye(4)[None], torch.eye(3)[None], 64, 64)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.dirname(_lib.__file__)
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                assert "oracle" not in txt.replace("CPU oracle", "").replace("the oracle", "").replace(
                    "oracle bit", ""), f


def test_profile_build_is_refused_without_explicit_opt_in(tmp_path, monkeypatch):
    """A library whose gwbp_version() says PROFILE (ablation knobs read from the environment, results may be invalid)
    is only ever loaded through an explicit _lib.use_library(path, allow_profile=True); the environment cannot swap the
    product library at all (GWBP_LIB was a developer knob in the product path)."""
    import subprocess
    src = tmp_path / "fake.c"
    src.write_text('const char *gwbp_version(void) { return "libgwbp gfx950 (PROFILE build)"; }\n'
                   'const char *gwbp_last_error_string(void) { return ""; }\n')
    so = tmp_path / "libfake_profile.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("GWBP_LIB", str(so))          # ignored: the package binds to its own in-tree library
    monkeypatch.setenv("GWBP_ALLOW_PROFILE", "1")
    assert _lib._lib_path == _lib.LIB_PATH
    monkeypatch.setattr(_lib, "_lib_path", _lib.LIB_PATH)
    monkeypatch.setattr(_lib, "_allow_profile", False)
    _lib.use_library(str(so))                          # explicit, but without the PROFILE opt-in: refused
    with pytest.raises(_lib.GwbpError, match="PROFILE"):
        _lib.lib()
    assert _lib._lib is None
    pkg = os.path.dirname(os.path.abspath(_lib.__file__))
    assert not [f for f in os.listdir(pkg) if "profile" in f], "no PROFILE library may sit in the package directory"


def test_inline_asm_vmem_bases_come_from_the_scalar_alu(tmp_path):
    """k_scatter_wide issues its visit loop's loads / stores / atomics through inline asm with SGPR bases.  hipcc's hazard
    recogniser does not look inside inline asm: a base written by v_readfirstlane / v_readlane less than 5 wait states
    earlier would be read stale (tools/check_asm_hazards.py; the first no-compute ablation build faulted on it).  Compile the
    kernels that use the idiom to assembly and scan them."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import check_asm_hazards
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
             "-fhip-fp32-correctly-rounded-divide-sqrt", "-munsafe-fp-atomics", "-S", "--cuda-device-only"]
    for name in ("scatter_wide", "scatter_full"):
        out = tmp_path / f"{name}.s"
        subprocess.check_call([hipcc, *flags, "-o", str(out), os.path.join(_lib.CSRC, f"{name}.hip")],
                              stderr=subprocess.DEVNULL)
        assert check_asm_hazards.scan(str(out)) == []
        if name == "scatter_wide":
            # round 5: the entry stream lives in two fixed SGPR tuples (s[68:83], s[84:99]) ACROSS inline-asm statements, which is
            # only sound while hipcc never allocates them (amdgpu_num_sgpr(76)); the kernel descriptor must still cover them
            assert check_asm_hazards.reserved_sgpr_uses(str(out)) == []
            assert "s_load_dwordx16 s[68:68+15]" in out.read_text() and "s_load_dwordx16 s[84:84+15]" in out.read_text()
            assert set(re.findall(r"\.amdhsa_next_free_sgpr (\d+)", out.read_text())) == {"100"}
            # The 256-channel kernel's register ALLOCATION decides how many front-stage waves fit beside it on a SIMD, and the
            # step time with it (scatter_wide.hip, "REGISTER BUDGET": 104 allocated = one 64-register front wave per SIMD is
            # the measured optimum; 120 shuts the front stage out until the kernel ends).  The full-resolution instantiation must ask
            # for 97..104 registers, the bilinear one (its staging loop holds 16 texel loads) for at most 112.
            txt = out.read_text()
            found = dict(re.findall(r"k_scatter_wideILb([01])E\S*\.num_vgpr, (\d+)", txt))
            assert len(found) == 2 and 97 <= int(found["0"]) <= 104 and 97 <= int(found["1"]) <= 112, found
