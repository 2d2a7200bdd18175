"""The appearance-code baseline (modules.AffineTransform) and reference-format checkpoints on the GPU path, against goldens produced
by the reference's own classes (oracle/gen_golden_checkpoint.py): a checkpoint written by the reference trainer's recipe loads strict
into this package's modules and the fused transforms reproduce the reference modules' outputs."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _close(a, b, what, tol=2e-4):
    a = a.detach().cpu().numpy()
    assert np.linalg.norm(a - b) <= tol * np.linalg.norm(b) + 1e-6, (what, np.linalg.norm(a - b), np.linalg.norm(b))


@pytest.mark.parametrize("name", ["train", "test"])
def test_affine_code_fused_transform_equals_reference(name):
    from bilateral_driving_amd.modules import AffineTransform
    z = np.load(os.path.join(G, f"affine_code_{name}.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")}
    mod = AffineTransform("Affine", n=sd["embedding.weight"].shape[0], embedding_dim=sd["embedding.weight"].shape[1])
    mod.load_state_dict(sd, strict=True)
    mod.in_test_set = bool(z["test"])
    assert next(mod.parameters()).is_cuda
    rgb = torch.from_numpy(z["rgb"]).cuda().requires_grad_(True)
    H, W, _ = rgb.shape
    infos = {"img_idx": torch.full((H, W), int(z["img_idx"]), dtype=torch.long, device="cuda"),
             "viewdirs": torch.zeros(H, W, 3, device="cuda")}
    A = mod(infos)                                     # reference API: per-pixel maps
    np.testing.assert_allclose(A.detach().cpu().numpy(), z["maps"], rtol=1e-5, atol=1e-6)
    out = mod.transform(rgb, infos)                    # fused: decoder once + the bilateral transform kernels on a 1x1x1 grid
    np.testing.assert_allclose(out.detach().cpu().numpy(), z["out"], rtol=1e-5, atol=2e-6)
    (out * torch.from_numpy(z["w"]).cuda()).sum().backward()
    _close(rgb.grad, z["v_rgb"], "v_rgb")
    for k, p in mod.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        _close(g, z["grad_" + k], k)


def test_affine_code_with_sky_blend_equals_composition():
    from bilateral_driving_amd.modules import AffineTransform
    torch.manual_seed(3)
    mod = AffineTransform("Affine", n=3)
    with torch.no_grad():
        for p in mod.parameters():
            p.normal_(0, 0.3)
    H, W = 31, 45
    rgb = (torch.rand(H, W, 3, device="cuda") * 1.3).requires_grad_(True)
    alpha = torch.rand(H, W, 1, device="cuda").requires_grad_(True)
    sky = torch.rand(H, W, 3, device="cuda").requires_grad_(True)
    infos = {"img_idx": torch.full((H, W), 1, dtype=torch.long, device="cuda")}
    out = mod.transform(rgb, infos, alpha=alpha, sky=sky)
    w = torch.randn_like(out)
    g = torch.autograd.grad((out * w).sum(), [rgb, alpha, sky] + list(mod.parameters()))
    A = mod(infos)
    x = torch.clamp(rgb, max=1.0) + sky * (1.0 - alpha)       # trainers/base.py:417 + scene_graph.py:292-294
    ref = (A[..., :3, :3] @ x[..., None] + A[..., :3, 3:])[..., 0]
    gr = torch.autograd.grad((ref * w).sum(), [rgb, alpha, sky] + list(mod.parameters()))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5, atol=2e-6)
    for a, b in zip(g, gr):
        _close(a, b.cpu().numpy(), "grad")


def test_reference_format_checkpoint_loads_and_reproduces_reference_outputs(tmp_path):
    from bilateral_driving_amd.checkpoint import load_checkpoint, save_checkpoint
    from bilateral_driving_amd.modules import AffineTransform, BilateralAffineTransform, MultiScaleBilateralAffineTransform
    io = np.load(os.path.join(G, "checkpoint_ref_format_io.npz"))
    models = {"Affine": MultiScaleBilateralAffineTransform("Affine", n=3, grid=[[2, 2, 1], [4, 4, 2], [8, 8, 4]]),
              "AffineSingle": BilateralAffineTransform("AffineSingle", n=3, grid_X=6, grid_Y=5, grid_W=4),
              "AffineCode": AffineTransform("AffineCode", n=3, embedding_dim=4)}
    step = load_checkpoint(os.path.join(G, "checkpoint_ref_format.pth"), models, strict=True, map_location="cuda")
    assert step == 1234 and all(m.step == 1234 for m in models.values())
    rgb = torch.from_numpy(io["rgb"]).cuda()
    H, W, _ = rgb.shape
    infos = {"img_idx": torch.full((H, W), int(io["img_idx"]), dtype=torch.long, device="cuda")}
    with torch.no_grad():
        np.testing.assert_allclose(models["Affine"].transform(rgb, infos).cpu().numpy(), io["out_ms"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(models["AffineSingle"].transform(rgb, infos).cpu().numpy(), io["out_single"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(models["AffineCode"].transform(rgb, infos).cpu().numpy(), io["out_code"], rtol=1e-5, atol=2e-6)
    # write -> read: same container, same names, same bits
    path = save_checkpoint(models, 77, str(tmp_path))
    assert os.path.basename(path) == "checkpoint_00077.pth"
    back = torch.load(path, map_location="cpu")
    ref = torch.load(os.path.join(G, "checkpoint_ref_format.pth"), map_location="cpu")
    assert back["step"] == 77 and set(back["models"]) == set(ref["models"])
    for cls in ref["models"]:
        assert list(back["models"][cls].keys()) == list(ref["models"][cls].keys()), cls
        for k in ref["models"][cls]:
            assert torch.equal(back["models"][cls][k], ref["models"][cls][k]), (cls, k)
    assert os.path.basename(save_checkpoint(models, 77, str(tmp_path), is_final=True)) == "checkpoint_final.pth"
