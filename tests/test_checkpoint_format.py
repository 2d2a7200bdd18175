"""Host-side checks of the reference-format checkpoint container and of the appearance-code module's reference API (its torch
forward runs anywhere; the fused transform is covered by the -m gpu tests)."""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", ["train", "test"])
def test_affine_code_forward_equals_reference_golden(name):
    from bilateral_driving_amd.modules import AffineTransform
    z = np.load(os.path.join(G, f"affine_code_{name}.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")}
    mod = AffineTransform("Affine", n=sd["embedding.weight"].shape[0], embedding_dim=sd["embedding.weight"].shape[1], device="cpu")
    assert all(float(p.abs().max()) == 0.0 for p in mod.parameters())      # zero_init (modules.py:236-243)
    mod.load_state_dict(sd, strict=True)
    mod.in_test_set = bool(z["test"])
    H, W, _ = z["rgb"].shape
    infos = {"img_idx": torch.full((H, W), int(z["img_idx"]), dtype=torch.long), "viewdirs": torch.zeros(H, W, 3)}
    A = mod(infos)
    np.testing.assert_allclose(A.detach().numpy(), z["maps"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(mod.image_matrix(infos).detach().numpy().reshape(3, 4), z["maps"][0, 0], rtol=1e-6, atol=1e-6)
    assert list(mod.get_param_groups()) == ["Affine#all"]


def test_checkpoint_container_round_trip(tmp_path):
    from bilateral_driving_amd.checkpoint import load_checkpoint, save_checkpoint, state_dict
    from bilateral_driving_amd.modules import AffineTransform
    ref = torch.load(os.path.join(G, "checkpoint_ref_format.pth"), map_location="cpu")
    assert set(ref) == {"models", "step"}
    a = AffineTransform("AffineCode", n=3, device="cpu")
    missing = torch.nn.Linear(2, 2)
    step = load_checkpoint(os.path.join(G, "checkpoint_ref_format.pth"), {"AffineCode": a, "NotInFile": missing}, map_location="cpu")
    assert step == 1234 and a.step == 1234 and missing.step == 1234
    for k, v in ref["models"]["AffineCode"].items():
        assert torch.equal(a.state_dict()[k], v)
    sd = state_dict({"AffineCode": a}, 5)
    assert sd["step"] == 5 and list(sd["models"]["AffineCode"]) == list(ref["models"]["AffineCode"])
    p = save_checkpoint({"AffineCode": a}, 5, str(tmp_path))
    b = AffineTransform("AffineCode", n=3, device="cpu")
    assert load_checkpoint(p, {"AffineCode": b}) == 5
    assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), b.state_dict().values()))
    with pytest.raises(RuntimeError):      # strict: a shape the file does not have
        load_checkpoint(p, {"AffineCode": AffineTransform("AffineCode", n=4, device="cpu")})


@pytest.mark.parametrize("name", ["a", "b"])
def test_inverse_loss_equals_reference_golden(name):
    """MultiScaleBilateralAffineTransform.inverse_loss (modules.py:474-492) in closed form against the reference's per-pixel
    torch.inverse of the composed homogeneous matrix (golden: oracle/gen_golden_inverse_loss.py)."""
    from bilateral_driving_amd.modules import MultiScaleBilateralAffineTransform
    z = np.load(os.path.join(G, f"inverse_loss_{name}.npz"))
    mod = MultiScaleBilateralAffineTransform("Affine", n=2, grid=[[2, 2, 1], [4, 4, 2], [8, 8, 4]], device="cpu")
    maps = [torch.from_numpy(z[f"map{i}"]).requires_grad_(True) for i in range(3)]
    mod.save_matrix = maps
    render = torch.from_numpy(z["render"]).requires_grad_(True)
    loss = mod.inverse_loss(torch.from_numpy(z["gt"]), render)
    np.testing.assert_allclose(float(loss.detach()), float(z["loss"]), rtol=2e-6)
    loss.backward()
    np.testing.assert_allclose(render.grad.numpy(), z["v_render"], rtol=1e-6, atol=1e-9)
    for i, m in enumerate(maps):
        ref = z[f"v_map{i}"]
        assert np.linalg.norm(m.grad.numpy() - ref) <= 2e-5 * np.linalg.norm(ref), i
