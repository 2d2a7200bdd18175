from bilateral_driving_amd.rendering import rasterization  # noqa: F401
