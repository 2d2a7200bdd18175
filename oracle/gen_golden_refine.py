#!/usr/bin/env python3
"""Golden vectors for adaptive density control, produced by the REFERENCE's own code: a VanillaGaussians instance
(/root/reference/project/models/gaussians/vanilla.py) whose refinement_after (:205-304) runs split_gaussians / dup_gaussians /
cull_gaussians and the optimiser surgery of models/gaussians/basics.py:162-206 on a torch.optim.Adam with the trainer's named
groups (models/trainers/base.py:201-222).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_refine.py        (build container only)

The module is imported with tensorly / pytorch3d / nvdiffrast / omegaconf stubbed and this repo's drop-in `gsplat` package in
front (quat_to_rotmat is the only gsplat symbol on this path).  Four step numbers exercise the four regimes of the schedule:
  3300  densify (size + screen split, dup) and full cull (opacity, world size, screen size)
  1300  densify, cull by opacity only (step <= reset_alpha_interval)
 16300  no densification (step >= stop_split_at), cull by opacity + world size (step >= stop_screen_size_at)
  3100  neither; opacity reset (step % reset_alpha_interval == refine_interval)
The noise of split_gaussians (torch.randn, :343) is recorded so that the test feeds the same samples."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")

CTRL = dict(warmup_steps=500, reset_alpha_interval=3000, refine_interval=100, sh_degree_interval=1000, n_split_samples=2,
            reset_alpha_value=0.01, densify_grad_thresh=0.0003, densify_size_thresh=0.002, cull_alpha_thresh=0.005,
            cull_scale_thresh=0.5, cull_screen_size=0.15, split_screen_size=0.05, stop_screen_size_at=4000, stop_split_at=15000,
            sh_degree=2)
SCENE_SCALE, NUM_TRAIN_IMAGES, N = 30.0, 150, 700
NAMES = ("xyz", "sh_dc", "sh_rest", "opacity", "scaling", "rotation")
ATTRS = ("_means", "_features_dc", "_features_rest", "_opacities", "_scales", "_quats")


class Cfg(dict):
    __getattr__ = dict.__getitem__


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
    return V


def make_inputs(seed: int):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)
    p = {
        "_means": (r(N, 3) - 0.5) * 40,
        "_features_dc": r(N, 3),
        "_features_rest": (r(N, 8, 3) - 0.5) * 0.2,
        # opacities around the cull threshold (sigmoid(-5.3) = 0.005) and the reset value (sigmoid(-4.6) = 0.01)
        "_opacities": r(N, 1) * 9 - 6.5,
        # world sizes around densify_size_thresh * scene_scale = 0.06 (log -2.8) and cull_scale_thresh * scene_scale = 15 (log 2.7)
        "_scales": r(N, 3) * 7.5 - 4.5,
        "_quats": torch.randn(N, 4, generator=g) * 2,
    }
    small = r(N) < 0.3                    # a third of the set is small in every axis: duplicated rather than split
    p["_scales"][small] = r(int(small.sum()), 3) * 1.6 - 4.5
    stats = {
        "xys_grad_norm": r(N) * 0.004,
        "vis_counts": torch.floor(r(N) * 6) + 1,
        "max_2Dsize": r(N) * 0.2,
    }
    grads = {k: torch.randn(v.shape, generator=g) * 0.01 for k, v in p.items()}
    return p, stats, grads


def main():
    V = import_reference()
    for step in (3300, 1300, 16300, 3100):
        p, stats, grads = make_inputs(step)
        model = V.VanillaGaussians("Background", Cfg(CTRL), scene_scale=SCENE_SCALE, num_train_images=NUM_TRAIN_IMAGES,
                                   device=torch.device("cpu"))
        for a in ATTRS:
            setattr(model, a, torch.nn.Parameter(p[a].clone()))
        groups = [{"params": [getattr(model, a)], "lr": 1e-3, "eps": 1e-15, "weight_decay": 0, "name": model.class_prefix + n}
                  for a, n in zip(ATTRS, NAMES)]
        opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15)          # base.py:222
        for a in ATTRS:
            getattr(model, a).grad = grads[a].clone()
        opt.step()                                                 # populates exp_avg / exp_avg_sq (and moves the parameters)
        rec = {"step": np.array(step), "N": np.array(N), "scene_scale": np.array(SCENE_SCALE),
               "num_train_images": np.array(NUM_TRAIN_IMAGES)}
        for k, v in CTRL.items():
            rec["ctrl_" + k] = np.array(v)
        for a, n in zip(ATTRS, NAMES):
            prm = getattr(model, a)
            rec["in" + a] = prm.detach().numpy().copy()
            rec["in_m" + a] = opt.state[prm]["exp_avg"].numpy().copy()
            rec["in_v" + a] = opt.state[prm]["exp_avg_sq"].numpy().copy()
        for k, v in stats.items():
            setattr(model, k, v.clone())
            rec["in_" + k] = v.numpy().copy()
        model.step = step
        noise = []
        real_randn = torch.randn

        def recording_randn(*a, **k):
            t = real_randn(*a, **k)
            noise.append(t.clone())
            return t
        torch.manual_seed(step)
        torch.randn = recording_randn
        try:
            model.refinement_after(step, opt)
        finally:
            torch.randn = real_randn
        rec["samples"] = (noise[0] if noise else torch.zeros(0, 3)).numpy()
        for a, n in zip(ATTRS, NAMES):
            prm = getattr(model, a)
            st = opt.state[prm]
            rec["out" + a] = prm.detach().numpy().copy()
            rec["out_m" + a] = st["exp_avg"].numpy().copy()
            rec["out_v" + a] = st["exp_avg_sq"].numpy().copy()
            grp = [g for g in opt.param_groups if g["name"] == model.class_prefix + n][0]
            assert grp["params"][0] is prm
        assert model.xys_grad_norm is None and model.vis_counts is None and model.max_2Dsize is None
        np.savez_compressed(os.path.join(OUT, f"refine_step{step}.npz"), **rec)
        print(f"step {step}: {N} -> {model.num_points} points, noise rows {rec['samples'].shape[0]}")


if __name__ == "__main__":
    main()
