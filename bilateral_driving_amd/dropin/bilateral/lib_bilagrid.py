"""Import surface of the reference's bilateral.lib_bilagrid (models/modules.py:13), served by
bilateral_driving_amd.bilagrid / colorcorrect.  The names not built here (CP-4D grids) are present so that the import line
resolves, and raise when used."""
from bilateral_driving_amd.bilagrid import (  # noqa: F401
    BilateralGrid, NeuralBilateralGrid, bilagrid_transform, color_affine_transform, slice, slice_feature, total_variation_loss)
from bilateral_driving_amd.colorcorrect import color_correct  # noqa: F401


def _not_on_hot_path(name):
    def f(*a, **k):
        raise NotImplementedError(f"bilateral.lib_bilagrid.{name} is outside the MI355X hot path built here "
                                  "(CP-4D bilateral grids)")
    f.__name__ = name
    return f


slice4d = _not_on_hot_path("slice4d")


class BilateralGridCP4D:  # noqa: D101
    def __init__(self, *a, **k):
        _not_on_hot_path("BilateralGridCP4D")()
