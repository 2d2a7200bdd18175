"""gsplat.cuda._wrapper names the reference reaches (models/gaussians/basics.py:15)."""
from bilateral_driving_amd.gs_ops import (  # noqa: F401
    fully_fused_projection, isect_tiles, rasterize_to_pixels, spherical_harmonics)
