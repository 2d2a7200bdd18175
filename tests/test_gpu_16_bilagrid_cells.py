"""-m gpu: the cell-aligned bilateral kernels (csrc/bilagrid_cells.hip; bit 0 of bds_set_option(7, ..)) and the one-pass pyramid forward (csrc/bilagrid_tile.hip; bit 1; default 3) against the general
kernels (option 7 = 0), the float64 oracle and the reference's golden vectors -- single-scale (one launch each way) and pyramids,
ragged / tiny images, images smaller than the grid, smooth and noisy guidance (one / every plane bucket of the counting sort)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import bilagrid_oracle as O
from tests.util import rel_err

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def B():
    assert torch.cuda.is_available()
    import bilateral_driving_amd.bilagrid as B
    return B


@pytest.fixture(scope="module")
def L():
    from bilateral_driving_amd import _lib
    return _lib


def _grids(levels, g, amp=0.05):
    out = []
    for (gx, gy, gl) in levels:
        ident = torch.tensor([1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0]).reshape(12, 1, 1, 1).repeat(1, gl, gy, gx)
        out.append(ident + amp * torch.randn(12, gl, gy, gx, generator=g))
    return out


def _run(B, L, cells, rgb, grids, factors, alpha, sky, wt):
    L.set_option(L.OPT_CELLS, cells)
    try:
        rg = rgb.cuda().requires_grad_(True)
        ag = alpha.cuda().requires_grad_(True) if alpha is not None else None
        sg = sky.cuda().requires_grad_(True) if sky is not None else None
        gg = [x.cuda().requires_grad_(True) for x in grids]
        out = B.bilagrid_transform(rg, gg, factors, alpha=ag, sky=sg)
        (out * wt.cuda()).sum().backward()
        torch.cuda.synchronize()
        return out.detach().cpu(), rg.grad.cpu(), (ag.grad.cpu() if ag is not None else None), (sg.grad.cpu() if sg is not None else None), \
            [x.grad.cpu() for x in gg]
    finally:
        L.set_option(L.OPT_CELLS, 3)


SHAPES = [
    (1, 1, [(16, 16, 8)], [1]),
    (2, 3, [(16, 16, 8)], [1]),
    (7, 9, [(16, 16, 8)], [1]),                 # fewer pixels than cells per axis
    (17, 23, [(16, 16, 8)], [1]),
    (135, 240, [(16, 16, 8)], [1]),
    (301, 517, [(16, 16, 8)], [1]),
    (64, 64, [(5, 7, 3)], [1]),
    (40, 56, [(1, 1, 1)], [1]),                 # one node: the appearance-code form
    (33, 47, [(2, 2, 1)], [1]),
    (17, 23, [(2, 2, 1), (4, 4, 2), (8, 8, 4)], [4, 4, 2]),
    (57, 91, [(2, 2, 1), (4, 4, 2), (8, 8, 4)], [4, 4, 2]),
    (270, 480, [(2, 2, 1), (4, 4, 2), (8, 8, 4)], [4, 4, 2]),
    (225, 401, [(2, 2, 1), (4, 4, 2), (8, 8, 4)], [4, 4, 2]),      # sizes the factors do not divide
    (128, 256, [(2, 2, 1), (4, 4, 2), (8, 8, 4), (16, 16, 8)], [8, 4, 4, 2]),
    (96, 160, [(8, 8, 4), (16, 16, 8)], [2, 1]),                    # a full-resolution level inside a pyramid
    # sizes every factor divides: the forward runs as ONE pass over the image (csrc/bilagrid_tile.hip, option bit 1)
    (272, 480, [(2, 2, 1), (4, 4, 2), (8, 8, 4)], [4, 4, 2]),
    (16, 16, [(2, 2, 1), (4, 4, 2), (8, 8, 4)], [4, 4, 2]),         # one partial tile; cells far smaller than a tile (global-memory slice)
    (64, 64, [(2, 2, 1), (4, 4, 2), (8, 8, 4)], [4, 4, 2]),
    (72, 200, [(8, 8, 4)], [2]),                                    # ragged tile grid, one level
    (540, 960, [(2, 2, 1), (4, 4, 2), (8, 8, 4)], [4, 4, 2]),
    (320, 480, [(2, 2, 1), (4, 4, 2), (8, 8, 4), (16, 16, 8)], [8, 4, 4, 2]),
]


@pytest.mark.parametrize("H,W,levels,factors", SHAPES)
@pytest.mark.parametrize("kind", ["noise", "smooth"])
@pytest.mark.parametrize("blend", [False, True])
def test_cell_kernels_equal_general_kernels(B, L, H, W, levels, factors, kind, blend):
    g = torch.Generator().manual_seed(H * 7919 + W)
    if kind == "noise":
        rgb = torch.rand(H, W, 3, generator=g) * 1.3 - 0.1        # guidance outside [0, 1] too: the border clamp
    else:
        yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
        rgb = torch.stack([0.4 + 0.1 * xx, 0.45 + 0.05 * yy, 0.5 - 0.1 * xx * yy], dim=-1)
    alpha = torch.rand(H, W, generator=g) if blend else None
    sky = torch.rand(H, W, 3, generator=g) if blend else None
    grids = _grids(levels, g)
    wt = torch.randn(H, W, 3, generator=g)
    ref = _run(B, L, 0, rgb, grids, factors, alpha, sky, wt)
    for mask in (1, 3):   # cell-aligned kernels alone; plus the one-pass pyramid forward where the shape qualifies
        got = _run(B, L, mask, rgb, grids, factors, alpha, sky, wt)
        assert torch.allclose(got[0], ref[0], rtol=2e-6, atol=2e-6), (mask, float((got[0] - ref[0]).abs().max()))
        # (the colour gradient: a pixel whose guidance lands within rounding of a plane boundary could flip a one-sided derivative;
        # the paths share the pixel arithmetic, so none may)
        assert torch.allclose(got[1], ref[1], rtol=1e-4, atol=1e-5 * float(ref[1].abs().max())), (mask, float((got[1] - ref[1]).abs().max()))
        for a, b in ((got[2], ref[2]), (got[3], ref[3])):
            if b is not None:
                assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max())), mask
        for a, b in zip(got[4], ref[4]):
            assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max())), (mask, float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.parametrize("H,W,levels,factors", [(135, 240, [(16, 16, 8)], [1]), (67, 129, [(16, 16, 8)], [1]), (64, 64, [(5, 7, 3)], [1])])
def test_single_scale_one_launch_vs_oracle(B, L, H, W, levels, factors):
    g = torch.Generator().manual_seed(W)
    rgb = torch.rand(H, W, 3, generator=g) * 1.2
    alpha = torch.rand(H, W, generator=g)
    sky = torch.rand(H, W, 3, generator=g)
    grids = _grids(levels, g)
    wt = torch.randn(H, W, 3, generator=g)
    r64 = rgb.double().requires_grad_(True)
    a64 = alpha.double().requires_grad_(True)
    s64 = sky.double().requires_grad_(True)
    g64 = [x.double().requires_grad_(True) for x in grids]
    ref = O.multiscale_transform(g64, O.sky_blend(r64, a64[..., None], s64), factors)
    (ref * wt.double()).sum().backward()
    out, v_rgb, v_alpha, v_sky, v_grids = _run(B, L, 3, rgb, grids, factors, alpha, sky, wt)
    worst = dict(out=rel_err(out, ref.detach()),
                 v_rgb=float((v_rgb.double() - r64.grad).norm() / r64.grad.norm()),
                 v_alpha=float((v_alpha.double() - a64.grad).norm() / a64.grad.norm()),
                 v_sky=float((v_sky.double() - s64.grad).norm() / s64.grad.norm()),
                 v_grid=float((v_grids[0].double() - g64[0].grad).norm() / g64[0].grad.norm()))
    print("single-scale one-launch path vs float64 oracle, worst relative errors:", worst)
    assert worst["out"] < 1e-4 and max(worst["v_rgb"], worst["v_alpha"], worst["v_sky"], worst["v_grid"]) < 1e-3, worst


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "bilagrid_single_*_f32.npz"))), ids=os.path.basename)
def test_single_scale_one_launch_vs_reference_golden(B, path):
    """The reference's own BilateralAffineTransform outputs and gradients, through the one-launch path (no affine maps requested)."""
    z = np.load(path)
    k = int(z["k_img"])
    rgb = torch.from_numpy(z["rgb"]).float().cuda().requires_grad_(True)
    g = torch.from_numpy(z["grids0"]).float().cuda().requires_grad_(True)
    out = B.bilagrid_transform(rgb, [g[k]], [1])
    err_out = float(np.abs(out.detach().cpu().numpy() - z["out"]).max())
    (out * torch.from_numpy(z["wt"]).float().cuda()).sum().backward()
    scale = max(1.0, float(np.abs(z["v_rgb"]).max()))
    err_rgb = float(np.abs(rgb.grad.cpu().numpy() - z["v_rgb"]).max()) / scale
    ref = z["v_grids0"].copy()
    tvg = None
    if float(z["tv_coef"]) != 0.0:   # the golden's grid gradient includes the TV term
        g2 = torch.from_numpy(z["grids0"]).float().cuda().requires_grad_(True)
        (float(z["tv_coef"]) * B.total_variation_loss(g2)).backward()
        tvg = g2.grad.cpu().numpy()
    got = g.grad.cpu().numpy() + (tvg if tvg is not None else 0.0)
    err_grid = float(np.abs(got - ref).max()) / max(1.0, float(np.abs(ref).max()))
    print(f"{os.path.basename(path)}: max abs err out {err_out:.2e}, v_rgb {err_rgb:.2e} (rel. to max), v_grid {err_grid:.2e} (rel. to max)")
    np.testing.assert_allclose(out.detach().cpu().numpy(), z["out"], rtol=1e-4, atol=1e-5)
    assert err_rgb < 1e-4 and err_grid < 1e-4
