"""-m gpu: the column-strip form of the bilateral transform's full-resolution stage (csrc/bilagrid.hip, "column-strip form": one
wave per strip of 64 pixel columns, up-sampler state in registers, one pass over the image in the backward) against the oracle
(float64 restatement pinned by the reference's goldens) and against the general kernels, on sizes the strips accept."""
import pytest
import torch

from oracle import bilagrid_oracle as O
from tests.util import rel_err

pytestmark = pytest.mark.gpu

L3 = [(2, 2, 1), (4, 4, 2), (8, 8, 4)]


@pytest.fixture()
def env():
    assert torch.cuda.is_available()
    from bilateral_driving_amd import _lib as L
    import bilateral_driving_amd.bilagrid as B
    L.lib()
    L.set_option(L.OPT_STRIPS, 3)      # the strips are opt-in (measured slower than the general kernels in the backward)
    yield B, L
    L.set_option(L.OPT_STRIPS, 0)


def _grids(levels, g):
    out = []
    for (gx, gy, gl) in levels:
        ident = torch.tensor([1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0]).reshape(12, 1, 1, 1).repeat(1, gl, gy, gx)
        out.append(ident + 0.05 * torch.randn(12, gl, gy, gx, generator=g))
    return out


def _uses_strips(B, L, grids, factors, H, W):
    lv = B._levels_struct([x[None].cuda() for x in grids], None, factors)
    return bool(L.lib().bds_bilagrid_ms_uses_strips(len(grids), lv, H, W))


CASES = [
    (96, 160, L3, [4, 4, 2], True),        # several bands, 3 strips
    (64, 256, L3, [4, 4, 2], False),       # no sky blend
    (40, 72, L3, [4, 4, 2], True),         # narrower than two strips, one band
    (80, 136, L3, [8, 4, 2], True),        # F = 8: wider halo, deeper ring
    (136, 200, [(4, 4, 2), (8, 8, 4)], [2, 2], True),   # F = 2
    (270 * 2, 480, L3, [4, 4, 2], True),   # many bands
    (48, 64, [(3, 5, 3)], [4], True),      # one level, odd grid shape
]


@pytest.mark.parametrize("H,W,levels,factors,blend", CASES)
def test_strips_vs_oracle(env, H, W, levels, factors, blend):
    B, L = env
    g = torch.Generator().manual_seed(H * 7 + W)
    rgb = torch.rand(H, W, 3, generator=g) * 1.2
    alpha = torch.rand(H, W, generator=g)
    sky = torch.rand(H, W, 3, generator=g)
    grids = _grids(levels, g)
    assert _uses_strips(B, L, grids, factors, H, W)
    wt = torch.randn(H, W, 3, generator=g)
    r64, a64, s64 = (t.double().requires_grad_(True) for t in (rgb, alpha, sky))
    g64 = [x.double().requires_grad_(True) for x in grids]
    inp = O.sky_blend(r64, a64[..., None], s64) if blend else r64
    ref = O.multiscale_transform(g64, inp, factors)
    (ref * wt.double()).sum().backward()
    rg, ag, sg = (t.cuda().requires_grad_(True) for t in (rgb, alpha, sky))
    gg = [x.cuda().requires_grad_(True) for x in grids]
    out = B.bilagrid_transform(rg, gg, factors, alpha=ag if blend else None, sky=sg if blend else None)
    (out * wt.cuda()).sum().backward()
    assert rel_err(out.detach().cpu(), ref.detach()) < 1e-4
    # element-wise as well as in norm: a missing halo row / column would be a LOCAL error
    d = (rg.grad.cpu().double() - r64.grad).abs()
    assert float(d.max()) <= 2e-4 * float(r64.grad.abs().max()), float(d.max())
    assert float((rg.grad.cpu().double() - r64.grad).norm() / r64.grad.norm()) < 1e-4
    if blend:
        assert float((ag.grad.cpu().double() - a64.grad).norm() / a64.grad.norm()) < 1e-4
        assert float((sg.grad.cpu().double() - s64.grad).norm() / s64.grad.norm()) < 1e-4
    for x, y in zip(gg, g64):
        assert float((x.grad.cpu().double() - y.grad).abs().max()) <= 2e-4 * float(y.grad.abs().max())


@pytest.mark.parametrize("rows", [0, 8, 16, 40])
def test_strips_equal_general_kernels(env, rows):
    """Same configuration through the strips (any band height) and through the general kernels (ablation bit 16): images to rounding
    (same formulas), gradients to summation-order noise."""
    B, L = env
    H, W, factors = 120, 200, [4, 4, 2]
    g = torch.Generator().manual_seed(rows + 1)
    rgb, alpha, sky = torch.rand(H, W, 3, generator=g) * 1.2, torch.rand(H, W, generator=g), torch.rand(H, W, 3, generator=g)
    grids = _grids(L3, g)
    wt = torch.randn(H, W, 3, generator=g).cuda()
    res = []
    for general in (True, False):
        L.set_option(L.OPT_DEBUG, 16 if general else 0)
        L.set_option(L.OPT_STRIP_ROWS, rows)
        try:
            rg, ag, sg = (t.cuda().requires_grad_(True) for t in (rgb, alpha, sky))
            gg = [x.cuda().requires_grad_(True) for x in grids]
            out = B.bilagrid_transform(rg, gg, factors, alpha=ag, sky=sg)
            (out * wt).sum().backward()
            res.append((out.detach(), rg.grad, ag.grad, sg.grad, [x.grad for x in gg]))
        finally:
            L.set_option(L.OPT_DEBUG, 0)
            L.set_option(L.OPT_STRIP_ROWS, 0)
    a, b = res
    assert float((a[0] - b[0]).abs().max()) <= 2e-6 * float(a[0].abs().max())   # (the same formulas; fused multiply-adds may differ)
    for k in (1, 2, 3):
        assert float((a[k] - b[k]).abs().max()) <= 2e-5 * float(a[k].abs().max()), k
    for x, y in zip(a[4], b[4]):
        assert float((x - y).abs().max()) <= 1e-4 * float(x.abs().max())


def test_unaligned_sizes_keep_the_general_kernels(env):
    B, L = env
    g = torch.Generator().manual_seed(0)
    grids = _grids(L3, g)
    assert not _uses_strips(B, L, grids, [4, 4, 2], 270, 480)       # 270 % 4 != 0
    assert not _uses_strips(B, L, _grids([(16, 16, 8)], g), [1], 128, 128)   # factor 1
    assert not _uses_strips(B, L, _grids(L3 + [(16, 16, 8)], g), [8, 4, 4, 2], 128, 256)   # the 16x16x8 grid does not fit
    assert _uses_strips(B, L, grids, [4, 4, 2], 1080, 1920)
