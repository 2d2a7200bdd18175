"""-m gpu: evaluation re-renders of Gaussian subsets (fused_view.render_classes: ONE projection + SH + tile-list build, one composite
per mask) against the reference's way -- the whole rasterization() again with opacities * mask (trainers/base.py:392-416,
trainers/scene_graph.py:296-313) -- through this package's rasterization() and through the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _masks(N, seed, dev):
    g = torch.Generator().manual_seed(seed)
    labels = torch.randint(0, 4, (N,), generator=g).to(dev)
    classes = {"Background": 0, "RigidNodes": 1, "DeformableNodes": 2, "SMPLNodes": 3}
    masks = {k: labels == v for k, v in classes.items()}
    masks["Dynamic"] = labels != 0                                   # scene_graph.py:306-313
    masks["Nothing"] = torch.zeros(N, dtype=torch.bool, device=dev)
    return masks


@pytest.mark.parametrize("N,W,H,seed", [(4000, 320, 192, 1), (700, 75, 50, 2)])
def test_class_rerenders_equal_full_rasterization_per_mask(N, W, H, seed):
    import bilateral_driving_amd.gs_ops as ops
    import bilateral_driving_amd.rendering as R
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd.fused_view import fused_view, render_classes
    dev = "cuda"
    cam = Hn.ring_cameras(W, H, yaws_deg=(0.0,), device=dev)[0]
    p = Hn.synthetic_scene(N, seed=seed, device=dev)
    p["means"] = p["means"] * torch.tensor([0.3, 0.3, 1.0], device=dev)
    masks = _masks(N, seed, dev)
    out = render_classes(p, cam.viewmat, cam.K, W, H, masks, cam_pos=cam.cam_pos)
    # the unmasked image is the training forward's
    grids = Hn.make_grids(1, device=dev)
    sky = torch.zeros(H, W, 3, device=dev)
    with torch.no_grad():
        full = fused_view(p, cam.viewmat, cam.K, W, H, grids, sky, Hn.FACTORS_3, img_idx=0, cam_pos=cam.cam_pos)
    assert torch.equal(out["rgb_gaussians"], full["rgb_gaussians"]) and torch.equal(out["opacity"], full["opacity"])
    np.testing.assert_allclose(out["depth"].cpu().numpy(), full["depth"].cpu().numpy(), rtol=1e-6, atol=0)
    # per mask: the reference's recipe
    dirs = p["means"] - cam.cam_pos
    col = torch.clamp(ops.spherical_harmonics(3, dirs, p["sh"]) + 0.5, 0.0, 1.0)
    quats = p["quats"] / p["quats"].norm(dim=-1, keepdim=True)
    opac = torch.sigmoid(p["opacity_logits"])
    acc_alpha = 0
    for name, m in masks.items():
        rr, aa, _ = R.rasterization(p["means"], quats, torch.exp(p["log_scales"]), opac * m, col, cam.viewmat[None], cam.K[None], W, H,
                                    packed=False, near_plane=0.1, render_mode="RGB+ED")
        rgb, depth, alpha = torch.clamp(rr[0][..., :3], max=1.0), rr[0][..., 3:4], aa[0]
        # same survivors in the same order (the general form evaluates activations / SH in separate kernels: rounding-level differences)
        np.testing.assert_allclose(out[name + "_opacity"].cpu().numpy(), alpha.cpu().numpy(), rtol=1e-4, atol=1e-5, err_msg=name)
        np.testing.assert_allclose(out[name + "_rgb"].cpu().numpy(), rgb.cpu().numpy(), rtol=1e-4, atol=1e-5, err_msg=name)
        cov = (alpha > 1e-3).squeeze(-1).cpu().numpy()
        np.testing.assert_allclose(out[name + "_depth"].cpu().numpy()[cov], depth.cpu().numpy()[cov], rtol=1e-4, atol=1e-4, err_msg=name)
        if name in ("Background", "Dynamic"):
            acc_alpha = acc_alpha + alpha
    assert float(out["Nothing_opacity"].abs().max()) == 0.0 and float(out["Nothing_rgb"].abs().max()) == 0.0
    assert float(out["Dynamic_opacity"].max()) > 0.1 and float(out["Background_opacity"].max()) > 0.1
    # a partition's coverages bound the full one: 1-(1-a)(1-b) <= a+b
    assert bool((out["opacity"] <= acc_alpha + 1e-5).all())


def test_class_rerenders_against_oracle():
    from oracle import gs_oracle as O
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd.fused_view import render_classes
    dev = "cuda"
    W, H, N = 96, 64, 600
    cam = Hn.ring_cameras(W, H, yaws_deg=(0.0,), device=dev)[0]
    p = Hn.synthetic_scene(N, seed=5, device=dev)
    p["means"] = p["means"] * torch.tensor([0.3, 0.3, 1.0], device=dev)
    masks = {k: v for k, v in _masks(N, 5, dev).items() if k in ("Background", "Dynamic")}
    out = render_classes(p, cam.viewmat, cam.K, W, H, masks, cam_pos=cam.cam_pos, include_full=False)
    c = {k: v.cpu() for k, v in p.items()}
    dirs = c["means"] - cam.cam_pos.cpu()
    col = torch.clamp(O.spherical_harmonics(3, dirs, c["sh"]) + 0.5, 0.0, 1.0)
    for name, m in masks.items():
        rr, aa, _ = O.rasterization(c["means"], c["quats"] / c["quats"].norm(dim=-1, keepdim=True), torch.exp(c["log_scales"]),
                                    torch.sigmoid(c["opacity_logits"]) * m.cpu(), col, cam.viewmat.cpu()[None], cam.K.cpu()[None], W, H,
                                    near_plane=0.1, render_mode="RGB+ED")
        np.testing.assert_allclose(out[name + "_opacity"].cpu().numpy(), aa[0].numpy(), rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(out[name + "_rgb"].cpu().numpy(), torch.clamp(rr[0][..., :3], max=1.0).numpy(), rtol=1e-4, atol=2e-5)
        covered = aa[0].numpy()[..., 0] > 1e-3
        np.testing.assert_allclose(out[name + "_depth"].cpu().numpy()[covered], rr[0][..., 3:4].numpy()[covered], rtol=2e-4, atol=1e-4)
