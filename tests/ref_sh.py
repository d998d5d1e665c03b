"""Test-only torch reference of the degree <= 3 real spherical-harmonics basis."""
import torch

# Real SH basis constants (degree <= 3), as used by every 3DGS implementation.
_C0 = 0.28209479177387814
_C1 = 0.4886025119029199
_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435)


def spherical_harmonics(degree: int, dirs: torch.Tensor, coeffs: torch.Tensor) -> torch.Tensor:
    """[N,K,3] SH coefficients -> [N,3] colours for `dirs` (normalised here): torch reference of the real SH basis for
    the tests of k_sh_colors / the oracle's orc_sh_colors (the product evaluates SH on the device only)."""
    d = dirs / dirs.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    out = _C0 * coeffs[:, 0]
    if degree >= 1:
        out = out + _C1 * (-y * coeffs[:, 1] + z * coeffs[:, 2] - x * coeffs[:, 3])
    if degree >= 2:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        out = out + _C2[0] * xy * coeffs[:, 4] + _C2[1] * yz * coeffs[:, 5] + _C2[2] * (2 * zz - xx - yy) * coeffs[:, 6] \
            + _C2[3] * xz * coeffs[:, 7] + _C2[4] * (xx - yy) * coeffs[:, 8]
        if degree >= 3:
            out = out + _C3[0] * y * (3 * xx - yy) * coeffs[:, 9] + _C3[1] * xy * z * coeffs[:, 10] \
                + _C3[2] * y * (4 * zz - xx - yy) * coeffs[:, 11] \
                + _C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * coeffs[:, 12] \
                + _C3[4] * x * (4 * zz - xx - yy) * coeffs[:, 13] + _C3[5] * z * (xx - yy) * coeffs[:, 14] \
                + _C3[6] * x * (xx - 3 * yy) * coeffs[:, 15]
    return out
