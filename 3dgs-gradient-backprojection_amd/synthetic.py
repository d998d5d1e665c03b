"""Seeded synthetic scenes, cameras and feature maps for the BASELINE.json configs.

The reference runs on a trained garden scene + COLMAP poses + LSeg features (backproject.py:43-58,74,83-113);
none of those exist offline, so BASELINE.md section 4 / SURVEY.md section 8(d) define statistically similar seeded
inputs.  Everything here is generated on the CPU with explicit torch generators so that the CPU oracle and
the GPU path see bit-identical inputs; the large throughput-only feature maps may be generated on device.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

SCENE_SEED = 1234
CAMERA_SEED = 99
FEATURE_SEED0 = 10_000
ENCODER_SEED = 7


@dataclass(frozen=True)
class Config:
    name: str
    n_gaussians: int
    n_views: int
    width: int
    height: int
    feat_dim: int
    log_scale0: float  # median scale s0 (world units)
    garden: bool  # 70 % volume + 30 % ground disc
    encoder_dim: Optional[int] = None  # backproject_compressed.py: 512 -> 16
    # the 2-D network's OWN map shape, where the reference upsamples it to the view (None: the map is given at view resolution)
    lowres: Optional[tuple] = None     # (h, w) of the network map
    upsample: Optional[str] = None     # "nearest" (dino, backproject.py:244-248) | "bilinear" (lseg, backproject.py:110-112)
    reduction: str = "sum"             # "sum" (lseg, backproject.py:127,145) | "mean" (dino, backproject.py:263,283)
    normalize: bool = True             # L2-normalise over channels (lseg, backproject.py:109); dino's patch tokens are not


# BASELINE.json "configs" in order (C3 = C2 sharded over ranks).
CONFIGS: Dict[str, Config] = {
    "C1": Config("C1", 10_000, 4, 400, 300, 32, 0.03, False),
    "C2": Config("C2", 1_000_000, 200, 1600, 1060, 512, 0.004, True),
    "C4": Config("C4", 5_000_000, 300, 1600, 1060, 768, 0.002, True),
    "C5": Config("C5", 1_000_000, 200, 1600, 1060, 512, 0.004, True, encoder_dim=16),
    # The reference's feature maps AS THE REFERENCE PRODUCES THEM, at C2 geometry (a1 / a7 "as the reference runs them"):
    # dino: 64 x 64 x 1024 patch tokens, nearest-upsampled, .mean() reductions (backproject.py:201,242-249,263,283)
    "DINO64": Config("DINO64", 1_000_000, 200, 1600, 1060, 1024, 0.004, True, lowres=(64, 64), upsample="nearest",
                     reduction="mean", normalize=False),
    # ... and the same loop with the other DINOv2 backbones' widths (vits14 384, vitb14 768, vitg14 1536; 64 x 64 tokens each)
    "DINO64S": Config("DINO64S", 1_000_000, 200, 1600, 1060, 384, 0.004, True, lowres=(64, 64), upsample="nearest",
                      reduction="mean", normalize=False),
    "DINO64B": Config("DINO64B", 1_000_000, 200, 1600, 1060, 768, 0.004, True, lowres=(64, 64), upsample="nearest",
                      reduction="mean", normalize=False),
    "DINO64G": Config("DINO64G", 1_000_000, 200, 1600, 1060, 1536, 0.004, True, lowres=(64, 64), upsample="nearest",
                      reduction="mean", normalize=False),
    # lseg: 480 x 480 x 512 normalised map, bilinearly upsampled, .sum() reductions (backproject.py:102-113,127,145)
    "LSEG480": Config("LSEG480", 1_000_000, 200, 1600, 1060, 512, 0.004, True, lowres=(480, 480), upsample="bilinear"),
    # small shapes used by the parity tests and smoke()
    "T0": Config("T0", 512, 2, 96, 64, 8, 0.06, False),
    "T1": Config("T1", 4_000, 2, 200, 136, 24, 0.04, True),
    # T1 with a dino-shaped map: 8 x 12 x 256 tokens (17 x 16.7 pixel texels: at least a tile), nearest, .mean() -> token space
    "T1D": Config("T1D", 4_000, 2, 200, 136, 256, 0.04, True, lowres=(8, 12), upsample="nearest", reduction="mean",
                  normalize=False),
}


def make_scene(cfg: Config, seed: int = SCENE_SEED) -> Dict[str, torch.Tensor]:
    """Pre-activation splat dict with the reference's key names (utils.py:56-67,105-107)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    n = cfg.n_gaussians
    means = torch.rand(n, 3, generator=g) * 2.0 - 1.0
    if cfg.garden:
        n_ground = int(0.3 * n)
        r = 1.5 * torch.sqrt(torch.rand(n_ground, generator=g))
        th = 2.0 * math.pi * torch.rand(n_ground, generator=g)
        z = -1.0 + 0.1 * torch.rand(n_ground, generator=g)
        means[:n_ground] = torch.stack([r * torch.cos(th), r * torch.sin(th), z], dim=1)
    scaling = math.log(cfg.log_scale0) + 0.5 * torch.randn(n, 3, generator=g)
    rotation = torch.randn(n, 4, generator=g)  # unnormalised on purpose (backproject.py:57)
    opacity = 2.0 * torch.randn(n, generator=g)  # logits
    return {
        "means": means.contiguous(),
        "scaling": scaling.contiguous(),
        "rotation": rotation.contiguous(),
        "opacity": opacity.contiguous(),
    }


def activate(splats: Dict[str, torch.Tensor]):
    """backproject.py:55-57: opacities = sigmoid, scales = exp, quats raw."""
    return (
        splats["means"],
        splats["rotation"],
        torch.exp(splats["scaling"]),
        torch.sigmoid(splats["opacity"]),
    )


def intrinsics(cfg: Config) -> torch.Tensor:
    K = torch.zeros(3, 3)
    K[0, 0] = K[1, 1] = 1.2 * cfg.width
    K[0, 2] = cfg.width / 2.0  # cx = W/2, cy = H/2 exactly (backproject.py:85-86 inverts this)
    K[1, 2] = cfg.height / 2.0
    K[2, 2] = 1.0
    return K


def make_cameras(cfg: Config, seed: int = CAMERA_SEED, n_views: Optional[int] = None) -> torch.Tensor:
    """[V,4,4] world->camera matrices, OpenCV convention (utils.py:215-219 layout [R|t; 0 0 0 1])."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    v = cfg.n_views if n_views is None else n_views
    az = 2.0 * math.pi * torch.rand(v, generator=g)
    el = torch.deg2rad(10.0 + 30.0 * torch.rand(v, generator=g))
    rad = 3.5 * (1.0 + 0.1 * (2.0 * torch.rand(v, generator=g) - 1.0))
    c = torch.stack([rad * torch.cos(el) * torch.cos(az), rad * torch.cos(el) * torch.sin(az), rad * torch.sin(el)], 1)
    fwd = -c / c.norm(dim=1, keepdim=True)  # look at the origin
    up = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd)
    right = torch.linalg.cross(fwd, up)
    right = right / right.norm(dim=1, keepdim=True)
    down = torch.linalg.cross(fwd, right)
    R = torch.stack([right, down, fwd], dim=1)  # rows = right, down, forward
    t = -(R @ c[:, :, None])[:, :, 0]
    vm = torch.zeros(v, 4, 4)
    vm[:, :3, :3] = R
    vm[:, :3, 3] = t
    vm[:, 3, 3] = 1.0
    return vm.contiguous()


def make_feature_map(cfg: Config, view: int, device="cpu", dim: Optional[int] = None) -> torch.Tensor:
    """[H,W,D] fp32, N(0,1) then L2-normalised over D (mimics backproject.py:109); seed = 10000 + view.  Configs with a
    `lowres` shape return the network's own [h,w,D] map (what the reference upsamples to the view)."""
    d = cfg.feat_dim if dim is None else dim
    g = torch.Generator(device=device).manual_seed(FEATURE_SEED0 + view)
    h, w = cfg.lowres if cfg.lowres else (cfg.height, cfg.width)
    f = torch.randn(h, w, d, generator=g, device=device)
    if cfg.normalize:
        f /= f.norm(dim=-1, keepdim=True)
    return f


def upsample_map(cfg: Config, low: torch.Tensor) -> torch.Tensor:
    """The reference's own upsampling of a network map to the view ([h,w,D] -> [H,W,D]): F.interpolate(mode=cfg.upsample)
    (backproject.py:110-112 bilinear, align_corners=False; :244-248 nearest).  Used by checks and the CPU baseline, which
    back-project the materialised map like the reference does."""
    if cfg.upsample is None:
        return low
    kw = {"align_corners": False} if cfg.upsample == "bilinear" else {}
    up = torch.nn.functional.interpolate(low.permute(2, 0, 1)[None], size=(cfg.height, cfg.width), mode=cfg.upsample, **kw)
    return up[0].permute(1, 2, 0)


def make_encoder(cfg: Config, seed: int = ENCODER_SEED) -> torch.Tensor:
    """Random stand-in for encoder_decoder.ckpt's encoder (backproject_compressed.py:26,127)."""
    assert cfg.encoder_dim is not None
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(cfg.feat_dim, cfg.encoder_dim, generator=g) / math.sqrt(cfg.feat_dim)


def view_shard(n_views: int, rank: int, world: int) -> List[int]:
    """Views r, r+R, r+2R, ... (interleaved to balance scene coverage)."""
    return list(range(rank, n_views, world))
