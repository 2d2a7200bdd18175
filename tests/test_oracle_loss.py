"""Oracle of the SSIM term (parity unpinned: pytorch_msssim 1.0.0 is absent, see oracle/loss_oracle.py): pinned by
independent restatements and identities only."""
import torch

from oracle import loss_oracle as LO


def test_window_is_normalised_and_symmetric():
    g = LO.gauss_window()
    assert g.shape == (11,) and abs(float(g.sum()) - 1.0) < 1e-6
    assert torch.allclose(g, g.flip(0)) and float(g[5]) == float(g.max())
    # closed form
    ref = torch.exp(-(torch.arange(-5, 6, dtype=torch.float64) ** 2) / 4.5)
    assert torch.allclose(g.double(), ref / ref.sum(), atol=1e-7)


def test_ssim_map_against_explicit_window_sums():
    g = torch.Generator().manual_seed(0)
    X, Y = torch.rand(1, 3, 17, 14, generator=g, dtype=torch.float64), torch.rand(1, 3, 17, 14, generator=g, dtype=torch.float64)
    assert torch.allclose(LO.ssim_map(X, Y), LO.ssim_map_bruteforce(X, Y), rtol=1e-10, atol=1e-12)


def test_ssim_identities():
    g = torch.Generator().manual_seed(1)
    X = torch.rand(23, 31, 3, generator=g)
    assert abs(float(LO.ssim(X, X)) - 1.0) < 1e-5
    Y = torch.rand(23, 31, 3, generator=g)
    assert abs(float(LO.ssim(X, Y)) - float(LO.ssim(Y, X))) < 1e-6     # symmetric
    assert float(LO.ssim(X, Y)) < 0.2                                   # independent noise is dissimilar
    assert abs(float(LO.ssim_loss(X, Y)) - (1 - float(LO.ssim(X, Y)))) < 1e-7


def test_ssim_gradient_by_finite_differences():
    g = torch.Generator().manual_seed(2)
    X = torch.rand(13, 15, 3, generator=g, dtype=torch.float64)
    Y = torch.rand(13, 15, 3, generator=g, dtype=torch.float64).requires_grad_(True)
    LO.ssim(X, Y).backward()
    eps = 1e-6
    for (i, j, c) in [(0, 0, 0), (6, 7, 1), (12, 14, 2), (3, 10, 0)]:
        Yp, Ym = Y.detach().clone(), Y.detach().clone()
        Yp[i, j, c] += eps; Ym[i, j, c] -= eps
        fd = (float(LO.ssim(X, Yp)) - float(LO.ssim(X, Ym))) / (2 * eps)
        assert abs(fd - float(Y.grad[i, j, c])) < 1e-7 + 1e-5 * abs(fd)


def _load_pixel_case(path, dtype=None):
    import numpy as np
    z = np.load(path)
    t = lambda k: torch.from_numpy(z[k]) if dtype is None else torch.from_numpy(z[k]).to(dtype)
    ego = t("egocar") if z["egocar"].size else None
    return z, dict(rgb=t("rgb"), pixels=t("pixels"), opacity=t("opacity"), sky_masks=t("sky_masks"), depth=t("depth"), lidar=t("lidar"),
                   egocar=ego, w=tuple(float(v) for v in z["w"]), depth_l2=bool(int(z["depth_l2"])))


def test_pixel_loss_oracle_against_reference_goldens(golden_dir):
    """values and gradients of the rgb / sky-mask / depth terms == what the reference's own models/losses.py gave."""
    import glob, os
    files = sorted(glob.glob(os.path.join(golden_dir, "pixel_loss_*.npz")))
    assert len(files) >= 5
    for f in files:
        z, kw = _load_pixel_case(f)
        out = LO.pixel_loss(**kw)
        tol = 1e-12 if "f64" in f else 2e-6
        for k in ("rgb_loss", "sky_loss", "depth_loss", "total"):
            assert abs(float(out[k]) - float(z[k])) <= tol * max(1.0, abs(float(z[k]))), (f, k)
        for k in ("v_rgb", "v_opacity", "v_depth"):
            ref = torch.from_numpy(z[k])
            err = float((out[k] - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)
            assert err <= (1e-12 if "f64" in f else 2e-6), (f, k, err)


def test_densify_stats_oracle_against_the_reference_method(golden_dir):
    import os
    import numpy as np
    z = np.load(os.path.join(golden_dir, "densify_stats.npz"))
    W, H, b = int(z["W"]), int(z["H"]), int(z["batch"])
    st = None
    for call in range(3):
        st = LO.densify_stats_update(st, torch.from_numpy(z[f"absgrad{call}"])[0], torch.from_numpy(z[f"radii{call}"]), W, H, b)
        for name in ("xys_grad_norm", "vis_counts", "max_2Dsize"):
            assert torch.allclose(st[name], torch.from_numpy(z[f"{name}{call}"]), rtol=1e-6, atol=0), (call, name)
    assert float(st["vis_counts"].min()) >= 1.0   # the reference's first call counts every Gaussian as seen once
