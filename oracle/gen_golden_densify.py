#!/usr/bin/env python3
"""Golden vectors for the per-step densification statistics, produced by the REFERENCE's own method
VanillaGaussians.after_train (/root/reference/project/models/gaussians/vanilla.py:163-191) fed the way
BasicTrainer.postprocess_per_train_step does (models/trainers/base.py:279-297: absgrad scaled by width/2, height/2 and the
batch size; last_size = max(width, height)).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_densify.py        (build container only)

models.gaussians.vanilla is imported with tensorly / pytorch3d / nvdiffrast / omegaconf stubbed and this repo's drop-in
`gsplat` package in front (none of them is touched by after_train).  Two consecutive calls are recorded (the first call
initialises the accumulators: note that it sets vis_counts to ONE for every Gaussian, visible or not)."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    sys.dont_write_bytecode = True

    def stub(name, **a):
        m = types.ModuleType(name)
        m.__dict__.update(a)
        sys.modules[name] = m
    stub("tensorly", set_backend=lambda *_: None)
    stub("pytorch3d"); stub("pytorch3d.ops", knn_points=None)
    stub("pytorch3d.transforms", matrix_to_quaternion=None, quaternion_to_matrix=None)
    stub("nvdiffrast"); stub("nvdiffrast.torch")
    stub("omegaconf", OmegaConf=type("OmegaConf", (), {}))
    sys.path.insert(0, os.path.join(ROOT, "bilateral_driving_amd", "dropin"))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, "/root/reference/project")
    import models.gaussians.vanilla as V
    return V.VanillaGaussians


def main():
    VG = import_reference()
    N, W, H, batch = 5000, 1920, 1080, 1
    g = torch.Generator().manual_seed(0)
    me = types.SimpleNamespace(num_points=N, filter_mask=torch.ones(N, dtype=torch.bool), xys_grad_norm=None, vis_counts=None,
                               max_2Dsize=None)
    rec = {"N": np.array(N), "W": np.array(W), "H": np.array(H), "batch": np.array(batch)}
    for call in range(3):
        radii = (torch.rand(N, generator=g) * 60).to(torch.int32) * (torch.rand(N, generator=g) < 0.3)
        absgrad = torch.rand(1, N, 2, generator=g) * 1e-3 * (radii > 0)[None, :, None]
        grads = absgrad.clone()                                   # base.py:281-285
        grads[..., 0] *= W / 2.0 * batch
        grads[..., 1] *= H / 2.0 * batch
        VG.after_train(me, radii[None][0], grads[0], max(W, H))   # base.py:290-296 (one class holding every Gaussian)
        rec[f"radii{call}"] = radii.numpy(); rec[f"absgrad{call}"] = absgrad.numpy()
        rec[f"xys_grad_norm{call}"] = me.xys_grad_norm.numpy().copy()
        rec[f"vis_counts{call}"] = me.vis_counts.numpy().copy()
        rec[f"max_2Dsize{call}"] = me.max_2Dsize.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "densify_stats.npz"), **rec)
    print("wrote densify_stats.npz", float(me.xys_grad_norm.sum()), float(me.vis_counts.sum()), float(me.max_2Dsize.max()))


if __name__ == "__main__":
    main()
