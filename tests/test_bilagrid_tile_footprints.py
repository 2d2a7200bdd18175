"""Design constants of the tile-with-halo form of the bilateral kernels, checked with the pinned oracle's
own resampling indices (oracle/bilagrid_oracle.py::_src_index = F.interpolate(bilinear, align_corners=False), models/modules.py:409-420,
494-497): how many low-resolution cells a full-resolution tile touches through the up-sampler, how far outside the tile the
down-sampler of those cells reads, and how far outside the tile the pixels lie that contribute to the cells the tile owns in the
backward pass.  These bounds size the LDS arrays and the halo of that kernel; nothing here runs on the GPU."""
import numpy as np
import pytest
import torch

from oracle import bilagrid_oracle as BO

TILE_H, TILE_W = 32, 64


def taps(out_size, in_size):
    i0, i1, _ = BO._src_index(out_size, in_size, torch.float32, "cpu")
    return i0.numpy(), i1.numpy()


@pytest.mark.parametrize("n,tile", [(1080, TILE_H), (1920, TILE_W), (900, TILE_H), (1600, TILE_W), (1280, TILE_H), (57, TILE_H), (91, TILE_W)])
@pytest.mark.parametrize("f", [2, 4, 8])
def test_forward_footprint_of_a_tile(n, tile, f):
    nd = n // f
    if nd < 1:
        pytest.skip("level degenerates")
    up0, up1 = taps(n, nd)            # full-res index -> the two low-res cells its up-sampled map reads
    dn0, dn1 = taps(nd, n)            # low-res cell -> the two full-res pixels its down-sampled colour reads
    for a in range(0, n, tile):
        b = min(a + tile, n)
        cells = np.arange(up0[a:b].min(), up1[a:b].max() + 1)
        assert len(cells) <= tile // f + 2                                   # LDS rows / columns of the tile's low-res maps
        lo, hi = dn0[cells].min(), dn1[cells].max()
        assert a - lo <= f and hi - (b - 1) <= f                             # the down-sampler reads at most f pixels outside the tile


@pytest.mark.parametrize("n,tile", [(1080, TILE_H), (1920, TILE_W), (900, TILE_H), (1600, TILE_W), (1280, TILE_H), (57, TILE_H)])
@pytest.mark.parametrize("f", [2, 4, 8])
def test_backward_ownership_needs_a_halo_of_f_pixels(n, tile, f):
    """Cells are owned by index range: tile t owns cells [t * tile / f, (t + 1) * tile / f) (the last tile also owns the remainder).  Every pixel that sends a gradient to an owned cell through the up-sampler's adjoint lies
    within f pixels of the tile; and the pixels of tile + halo read, forward, cells within tile / f + 4."""
    nd = n // f
    if nd < 1 or tile % f:
        pytest.skip("level degenerates")
    up0, up1 = taps(n, nd)
    n_tiles = (n + tile - 1) // tile
    owned_total = 0
    for t in range(n_tiles):
        a, b = t * tile, min((t + 1) * tile, n)
        c_lo, c_hi = t * tile // f, (nd if t == n_tiles - 1 else min((t + 1) * tile // f, nd))
        if c_lo >= c_hi:
            continue
        owned_total += c_hi - c_lo
        contrib = np.nonzero(((up0 >= c_lo) & (up0 < c_hi)) | ((up1 >= c_lo) & (up1 < c_hi)))[0]
        assert contrib.min() >= a - f and contrib.max() <= b - 1 + f          # the halo of the backward tile
        ha, hb = max(a - f, 0), min(b + f, n)
        cells = np.arange(up0[ha:hb].min(), up1[ha:hb].max() + 1)
        assert len(cells) <= tile // f + 4                                    # low-res maps the tile + halo pixels read forward
    assert owned_total == nd                                                  # the ownership ranges partition the cells
