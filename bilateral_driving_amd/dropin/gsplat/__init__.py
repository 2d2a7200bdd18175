"""Import surface of gsplat v1.3.0 as used by bilateral-driving, served by bilateral_driving_amd."""
from bilateral_driving_amd.rendering import rasterization  # noqa: F401
from bilateral_driving_amd.gs_ops import (  # noqa: F401
    fully_fused_projection, isect_tiles, rasterize_to_pixels, spherical_harmonics)

__version__ = "1.3.0+bds.gfx950"
