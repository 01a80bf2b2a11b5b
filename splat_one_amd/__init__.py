"""splat_one_amd -- MI355X-native 3D-Gaussian-splatting training path for inuex35/splat_one.

Drop-in for the slice of `gsplat` the reference imports
(/root/reference/utils/gsplat_utils/gsplat_trainer.py:42-46):

    from splat_one_amd.rendering import rasterization
    from splat_one_amd.ops import fully_fused_projection, spherical_harmonics, isect_tiles, \
        isect_offset_encode, rasterize_to_pixels

All arithmetic runs in hand-written gfx950 HIP kernels behind the C ABI of
splat_one_amd/lib/libsplat_one_amd.so (include/splat_one_amd.h).  There is no CPU fallback.
"""
__version__ = "0.1.0"

from .rendering import rasterization  # noqa: F401
from .ops import (fully_fused_projection, project_gaussians, spherical_harmonics, isect_tiles,  # noqa: F401
                  isect_offset_encode, rasterize_to_pixels)
