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
