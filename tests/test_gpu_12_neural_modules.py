"""Neural bilateral modules (bilateral_driving_amd.modules.NeuralBilateralAffineTransform / MultiScaleNeural...; feature slice on
the HIP kernels, MLP = nn.Linear) against golden vectors produced by the reference's own classes on its own lib_bilagrid
(oracle/gen_golden_neural_modules.py): reference checkpoints load (strict), maps / output / every gradient agree."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "neural_module_*.npz")))


def build(name, z):
    from bilateral_driving_amd.modules import MultiScaleNeuralBilateralAffineTransform, NeuralBilateralAffineTransform
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")}
    if name.startswith("ms"):
        n = sd["bil_grids0.grids"].shape[0]
        grid = []
        i = 0
        while f"bil_grids{i}.grids" in sd:
            _, f, L, gy, gx = sd[f"bil_grids{i}.grids"].shape
            grid.append([gx, gy, L]); i += 1
        mod = MultiScaleNeuralBilateralAffineTransform("Affine", n, grid, feature_dim=f, hidden_dim=sd["affine_network.2.weight"].shape[0])
    else:
        n, f, L, gy, gx = sd["bil_grids.grids"].shape
        mod = NeuralBilateralAffineTransform("Affine", n, gx, gy, L, feature_dim=f, hidden_dim=sd["affine_network.2.weight"].shape[0])
    mod.load_state_dict(sd, strict=True)
    return mod


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[14:-4] for f in FILES])
def test_neural_module_equals_reference_golden(path):
    name = os.path.basename(path)[14:-4]
    z = np.load(path)
    mod = build(name, z)
    assert next(mod.parameters()).is_cuda
    k = int(z["img_idx"])
    if z["neighbours"].size:
        mod.in_test_set = True
        mod.training_indices_for_test = {k: [int(v) for v in z["neighbours"]]}
    rgb = torch.from_numpy(z["rgb"]).cuda().requires_grad_(True)
    H, W, _ = rgb.shape
    infos = {"img_idx": torch.full((H, W), k, dtype=torch.long, device="cuda")}
    from bilateral_driving_amd import mlp_head
    lv = [getattr(mod, f"bil_grids{i}") for i in range(2)] if name.startswith("ms") else [mod.bil_grids]
    assert mlp_head.image_supported(H, W, [g.grids[0] for g in lv], 64)     # transform below is the ONE-kernel form: pinned by the reference too
    A = mod(rgb, infos)
    assert A.shape == (1, H, W, 3, 4)
    np.testing.assert_allclose(A[0].detach().cpu().numpy(), z["maps"], rtol=2e-4, atol=2e-5)
    out = mod.transform(rgb, infos)
    np.testing.assert_allclose(out.detach().cpu().numpy(), z["out"], rtol=2e-4, atol=2e-5)
    tv = mod.tv_loss()
    np.testing.assert_allclose(float(tv.detach()), float(z["tv"]), rtol=1e-5)
    ((out * torch.from_numpy(z["w"]).cuda()).sum() + 0.1 * tv).backward()

    def close(a, b, what):
        a = a.detach().cpu().numpy()
        assert np.linalg.norm(a - b) <= 2e-4 * np.linalg.norm(b) + 1e-6, (what, np.linalg.norm(a - b), np.linalg.norm(b))
    close(rgb.grad, z["v_rgb"], "v_rgb")
    for kname, p in mod.named_parameters():
        close(p.grad, z["grad_" + kname], kname)
    assert list(mod.get_param_groups()) == ["Affine#all"]


def test_no_feature_slice_without_gpu_is_loud():
    """(runs on the GPU box too) CPU tensors are refused by the slice underneath -- there is no fallback."""
    from bilateral_driving_amd import _lib as L
    from bilateral_driving_amd.modules import NeuralBilateralAffineTransform
    mod = NeuralBilateralAffineTransform("Affine", 2, 4, 4, 2, feature_dim=8, hidden_dim=16, device="cpu")
    with pytest.raises(L.BdsError):
        mod(torch.rand(5, 6, 3), {"img_idx": 0})


# ---- the fused head (csrc/mlp_head.hip) against the framework's Linear / Tanh modules on the same device ----------------------------
def _torch_head(feats, rgb, w1, w2, w3, residual):
    A = torch.tanh(torch.tanh(feats @ w1.T) @ w2.T) @ w3.T
    M = A.reshape(-1, 3, 4)
    out = (M[..., :3] @ rgb[..., None])[..., 0] + M[..., 3]
    return (out + rgb if residual else out), A


def _head_inputs(P, F, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    mk = lambda t: t.cuda().requires_grad_(True)
    return (mk(r(P, F) * scale), mk(torch.rand(P, 3, generator=g)), mk(r(64, F) * 0.3), mk(r(64, 64) * 0.2), mk(r(12, 64) * 0.2),
            r(P, 3).cuda(), r(P, 12).cuda())


def _rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


@pytest.mark.parametrize("F", [8, 16, 24, 32])
@pytest.mark.parametrize("P,residual", [(1, True), (31, False), (32, True), (1000, True), (70_001, False), (300_017, True)])
def test_fused_head_equals_framework_modules(F, P, residual):
    """Forward (both outputs) and every gradient; sizes below one tile, not a multiple of the 32-pixel tile, and large enough that
    every wave of the persistent grid owns several tiles and a partial weight gradient."""
    from bilateral_driving_amd import mlp_head
    feats, rgb, w1, w2, w3, v_out, v_aff = _head_inputs(P, F, seed=P + F)
    out, aff = mlp_head.transform_and_maps(feats, rgb, w1, w2, w3, residual=residual)
    ((out * v_out).sum() + (aff * v_aff).sum()).backward()
    got = [t.grad.clone() for t in (feats, rgb, w1, w2, w3)]
    for t in (feats, rgb, w1, w2, w3):
        t.grad = None
    f64 = lambda t: t.detach().double().requires_grad_(True)
    ref_in = [f64(t) for t in (feats, rgb, w1, w2, w3)]
    r_out, r_aff = _torch_head(*ref_in, residual)
    ((r_out * v_out.double()).sum() + (r_aff * v_aff.double()).sum()).backward()
    assert _rel(out.double(), r_out) < 2e-6 and _rel(aff.double(), r_aff) < 2e-6
    # element-wise too: the 12 entries land in the right rows and the right channels
    torch.testing.assert_close(out.double(), r_out.detach(), rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(aff.double(), r_aff.detach(), rtol=1e-4, atol=2e-5)
    for name, g, r in zip(("feats", "rgb", "w1", "w2", "w3"), got, ref_in):
        # fp32 sums over up to 3e5 pixels for the weights: compare with the float64 result at float32 summation accuracy
        assert _rel(g.double(), r.grad) < (3e-5 if name.startswith("w") else 3e-6), (name, _rel(g.double(), r.grad))


def test_fused_head_single_outputs_and_saturation():
    """Only one of the two outputs requested (what the modules do); tanh saturated both ways; deterministic weight gradients."""
    from bilateral_driving_amd import mlp_head
    P, F = 5000, 24
    feats, rgb, w1, w2, w3, v_out, v_aff = _head_inputs(P, F, seed=3, scale=40.0)      # pre-activations of +-100
    out = mlp_head.transform(feats, rgb, w1, w2, w3, residual=True)
    assert torch.isfinite(out).all()
    (out * v_out).sum().backward()
    g_a = [t.grad.clone() for t in (feats, rgb, w1, w2, w3)]
    for t in (feats, rgb, w1, w2, w3):
        t.grad = None
    out2 = mlp_head.transform(feats, rgb, w1, w2, w3, residual=True)
    (out2 * v_out).sum().backward()
    for a, t in zip(g_a, (feats, rgb, w1, w2, w3)):
        assert torch.equal(a, t.grad)                                                  # no atomics anywhere
    r_out, _ = _torch_head(*[t.detach().double() for t in (feats, rgb, w1, w2, w3)], True)
    torch.testing.assert_close(out.double(), r_out, rtol=1e-4, atol=1e-4)
    for t in (feats, rgb, w1, w2, w3):
        t.grad = None
    aff = mlp_head.affine_maps(feats, w1, w2, w3)
    (aff * v_aff).sum().backward()
    assert rgb.grad is None                                                            # the maps alone do not depend on the colour
    ref_in = [t.detach().double().requires_grad_(True) for t in (feats, rgb, w1, w2, w3)]
    _, r_aff = _torch_head(*ref_in, True)
    (r_aff * v_aff.double()).sum().backward()
    for name, t, r in zip(("feats", "w1", "w2", "w3"), (feats, w1, w2, w3), (ref_in[0], ref_in[2], ref_in[3], ref_in[4])):
        assert _rel(t.grad.double(), r.grad) < 3e-5, name


def test_fused_head_refuses_other_sizes_and_cpu():
    from bilateral_driving_amd import _lib as L, mlp_head
    assert not mlp_head.supported(13, 64) and not mlp_head.supported(24, 32) and mlp_head.supported(16, 64)
    feats, rgb, w1, w2, w3, *_ = _head_inputs(10, 24, seed=1)
    with pytest.raises(L.BdsError):
        mlp_head.transform(feats.cpu(), rgb.cpu(), w1.cpu(), w2.cpu(), w3.cpu())
    rc = L.lib().bds_mlp_head_fwd(10, 13, 64, L.ptr(feats), L.ptr(rgb), L.ptr(w1), L.ptr(w2), L.ptr(w3), 1, L.ptr(torch.empty(10, 3, device="cuda")),
                                  None, L.stream())
    assert rc == -1


# ---- the image form of the feature slice (band of the grid staged in LDS) against the point form ---------------------------------------
@pytest.mark.parametrize("H,W,nc,gx,gy,gl", [(11, 14, 24, 6, 5, 4), (1, 300, 8, 16, 16, 8), (75, 130, 24, 16, 16, 8), (40, 700, 13, 7, 9, 3),
                                            (33, 65, 8, 1, 1, 1), (1080, 1920, 24, 16, 16, 8), (540, 960, 16, 16, 16, 8)])
def test_image_slice_equals_point_slice(H, W, nc, gx, gy, gl):
    from bilateral_driving_amd import _lib as L
    from bilateral_driving_amd.bilagrid import _SliceImage, _SlicePoints
    assert L.lib().bds_bilagrid_slice_feat_image_ok(nc, gx, gy, gl)
    g = torch.Generator().manual_seed(H * 7 + W)
    grid = torch.randn(nc, gl, gy, gx, generator=g).cuda().requires_grad_(True)
    rgb = (torch.rand(H, W, 3, generator=g) * 1.3 - 0.15).cuda().requires_grad_(True)      # guidance beyond [0, 1]: border cells
    v = torch.randn(H, W, nc, generator=g).cuda()
    out = _SliceImage.apply(grid, rgb)
    (out * v).sum().backward()
    g_img = (grid.grad.clone(), rgb.grad.clone())
    grid.grad = None; rgb.grad = None
    ys, xs = torch.meshgrid(torch.linspace(0, 1.0, H, device="cuda"), torch.linspace(0, 1.0, W, device="cuda"), indexing="ij")
    xy = torch.stack([xs, ys], dim=-1).reshape(-1, 2)
    ref = _SlicePoints.apply(grid, xy, rgb.reshape(-1, 3)).reshape(H, W, nc)
    (ref * v).sum().backward()
    assert torch.equal(out, ref)                                   # same sampling arithmetic on the same values
    torch.testing.assert_close(g_img[1], rgb.grad, rtol=1e-5, atol=1e-5)
    scale = float(grid.grad.abs().max())
    torch.testing.assert_close(g_img[0], grid.grad, rtol=0, atol=2e-5 * scale)      # sums of up to 2e6 terms in another order


def test_image_slice_refuses_a_band_that_does_not_fit():
    from bilateral_driving_amd import _lib as L
    assert not L.lib().bds_bilagrid_slice_feat_image_ok(48, 16, 16, 8)
    assert L.lib().bds_bilagrid_slice_feat_image_ok(24, 16, 16, 8)
    grid = torch.zeros(48, 8, 16, 16, device="cuda"); rgb = torch.rand(8, 8, 3, device="cuda"); out = torch.empty(8, 8, 48, device="cuda")
    assert L.lib().bds_bilagrid_slice_feat_image_fwd(8, 8, 48, L.ptr(grid), 16, 16, 8, L.ptr(rgb), L.ptr(out), L.stream()) == -1


# ---- slice + head + application as one kernel (bds_neural_image_*) against the two-step path the reference goldens pin ------------------
def _two_step(mod, rgb, infos, **kw):
    A = mod(rgb, infos, **kw)[0]
    return (A[..., :3] @ rgb[..., None])[..., 0] + A[..., 3] + rgb


def _randomise(mod, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in mod.named_parameters():
            p.copy_((torch.randn(p.shape, generator=g) * (0.5 if "grids" in n else 0.25)).cuda())


@pytest.mark.parametrize("kind,H,W", [("single24", 72, 128), ("single24", 37, 301), ("single24", 1, 20), ("single24", 5, 7), ("single8", 40, 333),
                                      ("single16", 33, 64), ("single32", 16, 200), ("ms", 75, 130), ("ms", 64, 1000), ("single24", 540, 960),
                                      ("ms", 1080, 1920)])
def test_fused_image_transform_equals_two_step_path(kind, H, W):
    from bilateral_driving_amd import mlp_head
    from bilateral_driving_amd.modules import MultiScaleNeuralBilateralAffineTransform, NeuralBilateralAffineTransform
    if kind == "ms":
        mod = MultiScaleNeuralBilateralAffineTransform("Affine", 3, [[1, 1, 1], [16, 16, 8]], feature_dim=8, hidden_dim=64)
        levels = [mod.bil_grids0, mod.bil_grids1]
    else:
        gx, gy = (16, 16) if kind == "single24" else (7, 5)
        mod = NeuralBilateralAffineTransform("Affine", 3, gx, gy, 8, feature_dim=int(kind[6:]), hidden_dim=64)
        levels = [mod.bil_grids]
    _randomise(mod, H + W)
    assert mlp_head.image_supported(H, W, [g.grids[0] for g in levels], 64)
    g = torch.Generator().manual_seed(W)
    rgb = (torch.rand(H, W, 3, generator=g) * 1.2 - 0.1).cuda().requires_grad_(True)       # guidance on both sides of its range
    v = torch.randn(H, W, 3, generator=g).cuda()
    infos = {"img_idx": 1}
    out = mod.transform(rgb, infos)
    (out * v).sum().backward()
    got = {n: p.grad.clone() for n, p in mod.named_parameters()}
    got["rgb"] = rgb.grad.clone()
    rgb.grad = None
    for p in mod.parameters():
        p.grad = None
    ref = _two_step(mod, rgb, infos)
    (ref * v).sum().backward()
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=5e-5)
    want = {n: p.grad for n, p in mod.named_parameters()}
    want["rgb"] = rgb.grad
    for n in want:
        scale = float(want[n].abs().max())
        torch.testing.assert_close(got[n], want[n], rtol=0, atol=2e-4 * scale + 1e-7, msg=lambda m, n=n: f"{n}: {m}")
        if "grids" in n:   # only image 1's grid gets a gradient
            assert float(got[n][0].abs().max()) == 0.0 and float(got[n][2].abs().max()) == 0.0 and float(got[n][1].abs().max()) > 0


def test_fused_image_transform_test_branch_and_fallbacks():
    """Test branch: the mean over the neighbour images' grids; shapes outside the kernel's set keep the two-step path."""
    from bilateral_driving_amd import mlp_head
    from bilateral_driving_amd.modules import MultiScaleNeuralBilateralAffineTransform, NeuralBilateralAffineTransform
    H, W = 50, 90
    mod = NeuralBilateralAffineTransform("Affine", 4, 16, 16, 8, feature_dim=24, hidden_dim=64)
    _randomise(mod, 5)
    mod.in_test_set = True
    mod.training_indices_for_test = {2: [1, 3]}
    rgb = torch.rand(H, W, 3).cuda().requires_grad_(True)
    infos = {"img_idx": 2}
    out = mod.transform(rgb, infos)
    out.sum().backward()
    g_fused = mod.bil_grids.grids.grad.clone(); mod.bil_grids.grids.grad = None
    ref = _two_step(mod, rgb, infos)
    ref.sum().backward()
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=5e-5)
    scale = float(mod.bil_grids.grids.grad.abs().max())
    torch.testing.assert_close(g_fused, mod.bil_grids.grids.grad, rtol=0, atol=2e-4 * scale)
    assert float(g_fused[0].abs().max()) == 0.0 and float(g_fused[2].abs().max()) == 0.0      # only the two neighbours
    # gl = 2 is not one of the kernel's shapes: transform still works (slice kernel + head kernel)
    small = NeuralBilateralAffineTransform("Affine", 2, 6, 5, 2, feature_dim=24, hidden_dim=64)
    assert not mlp_head.image_supported(H, W, [small.bil_grids.grids[0]], 64)
    _randomise(small, 6)
    torch.testing.assert_close(small.transform(rgb, {"img_idx": 0}), _two_step(small, rgb, {"img_idx": 0}), rtol=1e-4, atol=5e-5)
    ms = MultiScaleNeuralBilateralAffineTransform("Affine", 2, [[1, 1, 1], [16, 16, 8]], feature_dim=8, hidden_dim=64)
    _randomise(ms, 7)
    a = ms.transform(rgb, {"img_idx": 0}, guidance_factor=[2, 2])       # an explicit guidance factor: two-step path
    torch.testing.assert_close(a, _two_step(ms, rgb, {"img_idx": 0}, guidance_factor=[2, 2]), rtol=1e-4, atol=5e-5)
