#!/usr/bin/env python3
"""A reference-format checkpoint + the appearance-code baseline's goldens, produced by IMPORTING THE REFERENCE
(build container only; tensorly / pytorch3d / nvdiffrast stubbed, not on this path):

  * tests/golden/checkpoint_ref_format.pth -- torch.save({"models": {...}, "step": 1234}) exactly as
    BasicTrainer.save_checkpoint writes it (models/trainers/base.py:677-682,739-753), with the reference's own
    MultiScaleBilateralAffineTransform ("Affine"), BilateralAffineTransform ("AffineSingle") and AffineTransform ("AffineCode")
    modules after a few perturbation steps; + inputs and the modules' own outputs, so that a loader can be checked end to end;
  * tests/golden/affine_code_*.npz -- AffineTransform (models/modules.py:213-267) forward applied as the trainer does
    (trainers/scene_graph.py:92-94), values and autograd gradients.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_checkpoint.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden_neural_modules import import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main():
    M = import_reference()
    g = torch.Generator().manual_seed(42)
    H, W, n_img = 24, 36, 3
    ms = M.MultiScaleBilateralAffineTransform("Affine", n=n_img, grid=[[2, 2, 1], [4, 4, 2], [8, 8, 4]], device="cpu")
    single = M.BilateralAffineTransform("AffineSingle", n=n_img, grid_X=6, grid_Y=5, grid_W=4, device="cpu")
    code = M.AffineTransform("AffineCode", n=n_img, embedding_dim=4, pixel_affine=False, device="cpu")
    with torch.no_grad():
        for mod in (ms, single, code):
            for p in mod.parameters():
                p.add_(torch.randn(p.shape, generator=g) * 0.1)
    rgb = torch.rand(H, W, 3, generator=g) * 1.1
    img_idx = 1
    infos = {"img_idx": torch.full((H, W), img_idx, dtype=torch.long)}
    # the trainer's application of each module (trainers/scene_graph.py:92-98,112-117)
    out_ms = rgb
    for aff in ms(rgb, infos):
        aff = aff.reshape(H, W, 3, 4)
        out_ms = (aff[..., :3, :3] @ out_ms[..., None] + aff[..., :3, 3:])[..., 0]
    a1 = single(rgb, infos).reshape(H, W, 3, 4)
    out_single = (a1[..., :3, :3] @ rgb[..., None] + a1[..., :3, 3:])[..., 0]
    a2 = code(infos)
    out_code = (a2[..., :3, :3] @ rgb[..., None] + a2[..., :3, 3:])[..., 0]
    ckpt = {"models": {"Affine": ms.state_dict(), "AffineSingle": single.state_dict(), "AffineCode": code.state_dict()}, "step": 1234}
    torch.save(ckpt, os.path.join(OUT, "checkpoint_ref_format.pth"))
    np.savez_compressed(os.path.join(OUT, "checkpoint_ref_format_io.npz"), rgb=rgb.numpy(), img_idx=np.array(img_idx),
                        out_ms=out_ms.detach().numpy(), out_single=out_single.detach().numpy(), out_code=out_code.detach().numpy())
    print("checkpoint keys:", {k: sorted(v.keys()) for k, v in ckpt["models"].items()})

    # AffineTransform goldens: train branch (img_idx) and test branch (mean code), values + gradients
    for name, test in (("train", False), ("test", True)):
        mod = M.AffineTransform("Affine", n=4, embedding_dim=4, pixel_affine=False, device="cpu")
        with torch.no_grad():
            for p in mod.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)
        mod.in_test_set = test
        Hh, Ww = 13, 17
        x = (torch.rand(Hh, Ww, 3, generator=g) * 1.2 - 0.1).requires_grad_(True)
        inf = {"img_idx": torch.full((Hh, Ww), 2, dtype=torch.long), "viewdirs": torch.randn(Hh, Ww, 3, generator=g)}
        A = mod(inf)
        out = (A[..., :3, :3] @ x[..., None] + A[..., :3, 3:])[..., 0]
        w = torch.randn(out.shape, generator=g)
        (out * w).sum().backward()
        rec = {"rgb": x.detach().numpy(), "w": w.numpy(), "maps": A.detach().numpy(), "out": out.detach().numpy(), "v_rgb": x.grad.numpy(),
               "img_idx": np.array(2), "test": np.array(test)}
        for k, v in mod.state_dict().items():
            rec["sd_" + k] = v.numpy()
        for k, p in mod.named_parameters():
            rec["grad_" + k] = p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        np.savez_compressed(os.path.join(OUT, f"affine_code_{name}.npz"), **rec)
        print(name, tuple(A.shape), float(out.abs().mean()))


if __name__ == "__main__":
    main()
