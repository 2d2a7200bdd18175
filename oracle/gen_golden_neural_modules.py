#!/usr/bin/env python3
"""Golden vectors for the neural bilateral MODULES by IMPORTING THE REFERENCE:
models/modules.py NeuralBilateralAffineTransform (:595-670) and MultiScaleNeuralBilateralAffineTransform (:672-820) with the
reference's own bilateral/lib_bilagrid.py underneath, applied the way the trainer does (trainers/scene_graph.py:99-106:
rgb' = A rgb + b + rgb, guidance_factor left at None), train branch and test branch (mean over neighbour grids).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_neural_modules.py        (build container only)

tensorly / pytorch3d / nvdiffrast are stubbed (not on this path).  Each file: state_dict, rgb, the 3x4 maps, rgb', and the
autograd gradients of loss = sum(rgb' * w) + 0.1 * tv_loss() w.r.t. every parameter and rgb."""
import os
import sys
import types

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    sys.dont_write_bytecode = True

    def stub(name, **a):
        m = types.ModuleType(name)
        m.__dict__.update(a)
        sys.modules[name] = m
    stub("tensorly", set_backend=lambda *_: None)
    stub("pytorch3d"); stub("pytorch3d.ops", knn_points=None)
    stub("nvdiffrast"); stub("nvdiffrast.torch")
    sys.path.insert(0, "/root/reference/project")
    import models.modules as M
    return M


def run(name, mod, H, W, img_idx, test_neighbours=None):
    g = torch.Generator().manual_seed(len(name))
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.dim() == 5 else 0.25))
    if test_neighbours is not None:
        mod.in_test_set = True
        mod.training_indices_for_test = {img_idx: test_neighbours}
    rgb = (torch.rand(H, W, 3, generator=g) * 1.2 - 0.1).requires_grad_(True)
    infos = {"img_idx": torch.full((H, W), img_idx, dtype=torch.long)}
    A = mod(rgb, infos).reshape(H, W, 3, 4)
    out = (A[..., :3, :3] @ rgb[..., None] + A[..., :3, 3:])[..., 0] + rgb          # scene_graph.py:101-102 / :105-106
    w = torch.randn(out.shape, generator=g)
    tv = mod.tv_loss()
    ((out * w).sum() + 0.1 * tv).backward()
    rec = {"rgb": rgb.detach().numpy(), "w": w.numpy(), "maps": A.detach().numpy(), "out": out.detach().numpy(), "tv": tv.detach().numpy(),
           "v_rgb": rgb.grad.numpy(), "img_idx": np.array(img_idx), "neighbours": np.array(test_neighbours if test_neighbours else [], dtype=np.int64)}
    for k, v in mod.state_dict().items():
        rec["sd_" + k] = v.numpy()
    for k, p in mod.named_parameters():
        rec["grad_" + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, f"neural_module_{name}.npz"), **rec)
    print(name, tuple(A.shape), float(tv), float(out.abs().mean()))


def main():
    M = import_reference()
    run("single_train", M.NeuralBilateralAffineTransform("Affine", 3, 6, 5, 4, feature_dim=24, hidden_dim=64, device="cpu"), 11, 14, 1)
    run("single_test", M.NeuralBilateralAffineTransform("Affine", 4, 6, 5, 4, feature_dim=24, hidden_dim=64, device="cpu"), 11, 14, 2, [1, 3])
    run("ms_train", M.MultiScaleNeuralBilateralAffineTransform("Affine", 3, [[1, 1, 1], [8, 8, 4]], feature_dim=8, hidden_dim=64, device="cpu"),
        12, 10, 2)
    run("ms_test", M.MultiScaleNeuralBilateralAffineTransform("Affine", 4, [[1, 1, 1], [8, 8, 4]], feature_dim=8, hidden_dim=64, device="cpu"),
        12, 10, 1, [0, 2])


if __name__ == "__main__":
    main()
