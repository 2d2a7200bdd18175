"""Pins oracle/bilagrid_oracle.py against golden vectors produced by the reference's own
Python (oracle/gen_golden_bilateral.py; fixtures in tests/golden/).  CPU only."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import bilagrid_oracle as O
from tests.util import rel_err

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _files(pat):
    fs = sorted(glob.glob(os.path.join(GOLDEN, pat)))
    assert fs, pat
    return fs


def _tol(name):
    return dict(rtol=1e-9, atol=1e-11) if name.endswith("f64.npz") else dict(rtol=2e-4, atol=2e-5)


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("path", _files("bilagrid_ms_*.npz"), ids=os.path.basename)
def test_multiscale_matches_reference(path):
    z = np.load(path)
    tol = _tol(path)
    k = int(z["k_img"])
    factors = [int(f) for f in z["factors"]]
    nl = len(factors)
    H, W = int(z["H"]), int(z["W"])
    rgb = _t(z["rgb"]).clone().requires_grad_(True)
    allg = [_t(z[f"grids{i}"]).clone().requires_grad_(True) for i in range(nl)]
    affs = O.multiscale_affines([g[k] for g in allg], rgb, factors)
    full = z["aff0"].shape[0] == H
    for i, a in enumerate(affs):
        got = a.detach().numpy() if full else a.detach().numpy()[::5, ::7]
        np.testing.assert_allclose(got, z[f"aff{i}"], **tol)
    out = O.compose(affs, rgb)
    np.testing.assert_allclose(out.detach().numpy(), z["out"], **tol)
    tv = O.multiscale_tv(allg)
    np.testing.assert_allclose(float(tv.detach()), float(z["tv"]), rtol=tol["rtol"])
    ((out * _t(z["wt"])).sum() + float(z["tv_coef"]) * tv).backward()
    gt = dict(rtol=tol["rtol"] * 5, atol=tol["atol"] * 50)
    np.testing.assert_allclose(rgb.grad.numpy(), z["v_rgb"], **gt)
    for i in range(nl):
        np.testing.assert_allclose(allg[i].grad.numpy(), z[f"v_grids{i}"], **gt)
    # test branch: neighbour average
    with torch.no_grad():
        near = [int(n) for n in z["near"]]
        ot = O.multiscale_transform([g[near] for g in allg], rgb, factors, neighbours=True)
    np.testing.assert_allclose(ot.numpy(), z["out_test"], **tol)


@pytest.mark.parametrize("path", _files("bilagrid_single_*.npz"), ids=os.path.basename)
def test_single_scale_matches_reference(path):
    z = np.load(path)
    tol = _tol(path)
    k = int(z["k_img"])
    rgb = _t(z["rgb"]).clone().requires_grad_(True)
    g = _t(z["grids0"]).clone().requires_grad_(True)
    a = O.single_scale_affine(g[k], rgb)
    np.testing.assert_allclose(a.detach().numpy(), z["aff0"], **tol)
    out = O.apply_affine(a, rgb)
    np.testing.assert_allclose(out.detach().numpy(), z["out"], **tol)
    tv = O.total_variation_loss(g)
    np.testing.assert_allclose(float(tv.detach()), float(z["tv"]), rtol=tol["rtol"])
    ((out * _t(z["wt"])).sum() + float(z["tv_coef"]) * tv).backward()
    gt = dict(rtol=tol["rtol"] * 5, atol=tol["atol"] * 50)
    np.testing.assert_allclose(rgb.grad.numpy(), z["v_rgb"], **gt)
    np.testing.assert_allclose(g.grad.numpy(), z["v_grids0"], **gt)
    with torch.no_grad():
        near = [int(n) for n in z["near"]]
        at = sum(O.single_scale_affine(g[n], rgb) for n in near) / len(near)
        np.testing.assert_allclose(O.apply_affine(at, rgb).numpy(), z["out_test"], **tol)


@pytest.mark.parametrize("path", _files("bilagrid_points_*.npz"), ids=os.path.basename)
def test_point_slice_matches_reference(path):
    z = np.load(path)
    tol = _tol(path)
    grids = _t(z["grids"]).clone().requires_grad_(True)
    rgb = _t(z["rgb"]).clone().requires_grad_(True)
    xy, idx = _t(z["xy"]), _t(z["idx"])[:, 0]
    aff = torch.zeros(rgb.shape[0], 12, dtype=rgb.dtype)
    for n in range(grids.shape[0]):
        m = idx == n
        aff = aff + m[:, None] * O.slice_grid(grids[n], xy[:, 0], xy[:, 1], O.rgb2gray(rgb))
    np.testing.assert_allclose(aff.detach().numpy(), z["aff"], **tol)
    out = O.apply_affine(aff, rgb)
    np.testing.assert_allclose(out.detach().numpy(), z["out"], **tol)
    (out * _t(z["wt"])).sum().backward()
    gt = dict(rtol=tol["rtol"] * 5, atol=tol["atol"] * 50)
    np.testing.assert_allclose(rgb.grad.numpy(), z["v_rgb"], **gt)
    np.testing.assert_allclose(grids.grad.numpy(), z["v_grids"], **gt)


@pytest.mark.parametrize("path", _files("bilagrid_tv_*.npz"), ids=os.path.basename)
def test_tv_matches_reference(path):
    z = np.load(path)
    tol = _tol(path)
    i = 0
    while f"x{i}" in z:
        x = _t(z[f"x{i}"]).clone().requires_grad_(True)
        tv = O.total_variation_loss(x)
        np.testing.assert_allclose(float(tv.detach()), float(z[f"tv{i}"]), rtol=tol["rtol"])
        if torch.is_tensor(tv) and tv.requires_grad:
            tv.backward()
            np.testing.assert_allclose(x.grad.numpy(), z[f"v_x{i}"], rtol=tol["rtol"] * 5, atol=tol["atol"])
        i += 1
    assert i == 4


def test_feature_grid_slice_and_tv_against_reference_goldens(golden_dir):
    """The channel-generic slice / TV of the oracle vs NeuralBilateralGrid + slice_feature + tv_loss of the reference
    (tests/golden/neural_slice_*.npz, oracle/gen_golden_neural.py): features, TV, d/d(grids), d/d(rgb)."""
    import glob
    files = sorted(glob.glob(os.path.join(golden_dir, "neural_slice_*.npz")))
    assert len(files) >= 5
    for f in files:
        z = np.load(f)
        grids = torch.from_numpy(z["grids"]).requires_grad_(True)
        xy, rgb = torch.from_numpy(z["xy"]), torch.from_numpy(z["rgb"]).requires_grad_(True)
        idx, w = torch.from_numpy(z["idx"]), torch.from_numpy(z["w"])
        if xy.dim() == 3:    # one view
            feats = O.slice_grid(grids[int(idx.reshape(-1)[0])], xy[..., 0], xy[..., 1], O.rgb2gray(rgb))
        else:                # one grid per leading entry
            feats = torch.stack([O.slice_grid(grids[int(idx[b].reshape(-1)[0])], xy[b, ..., 0], xy[b, ..., 1], O.rgb2gray(rgb[b]))
                                 for b in range(xy.shape[0])])
        tv = O.total_variation_loss(grids)
        ((feats * w).sum() + 0.3 * tv).backward()
        # (the reference keeps its BT.601 weights as float32 constants also when run in float64: 5e-8 relative on the guidance)
        tol = 1e-6 if "f64" in f else 3e-5
        assert rel_err(feats, torch.from_numpy(z["feats"])) < tol, f
        assert abs(float(tv) - float(z["tv"])) < tol * max(1.0, float(z["tv"])), f
        assert rel_err(grids.grad, torch.from_numpy(z["v_grids"])) < tol, f
        assert rel_err(rgb.grad, torch.from_numpy(z["v_rgb"])) < tol, f
