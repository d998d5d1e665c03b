"""MI355X-native gradient-weighted feature back-projection (the hot path of backproject.py of
JojiJoseph/3dgs-gradient-backprojection), hand-written HIP kernels behind a C ABI (include/gwbp.h).

    from gsbp_amd import rasterization            # drop-in for `from gsplat import rasterization`
    from gsbp_amd import create_feature_field     # fused counterpart of create_feature_field_lseg/_dino
"""
from . import synthetic  # noqa: F401
from ._lib import GwbpError, build, lib  # noqa: F401
from .backproject import ViewPipeline, create_feature_field, finalize_reference, prune_mask, reduce_partials, reduce_partials_sharded  # noqa: F401
from .engine import Engine, bilinear_index, nearest_index  # noqa: F401
from . import scene_io  # noqa: F401
from .rasterization import rasterization  # noqa: F401
from .pruning import check_proper_pruning, gradient_mask, prune_by_gradients  # noqa: F401
