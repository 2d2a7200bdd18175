#!/usr/bin/env python3
"""Golden vectors for adaptive density control of the NODE classes, produced by the REFERENCE's own code: a RigidNodes instance
(/root/reference/project/models/nodes/rigid.py; DeformableNodes inherits the same methods) whose refinement_after (:194-293)
carries the per-Gaussian instance id (`point_ids`, :253) through split_gaussians / dup_gaussians (:327-372) and whose
cull_gaussians (:295-325) also drops the Gaussians that left their instance's box (get_out_of_bound_mask, :374-383; the
shipped configs set `cull_out_of_bound: true` for RigidNodes and false for DeformableNodes, configs/omnire_ms_bilateral.yaml:97,135).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_refine_rigid.py        (build container only)

rigid.py is loaded by path (models/nodes/__init__.py would pull trimesh in through the SMPL nodes); everything else is imported
the way gen_golden_refine.py does it.  Regimes:
  3300 oob   densify + full cull + out-of-bound cull (children of a split can leave the box: their decision is per sample)
 16300 oob   cull only (opacity, world size) + out-of-bound cull
  1300 nob   densify, opacity cull, cull_out_of_bound false (the DeformableNodes setting): point_ids carried, nothing else
The noise of split_gaussians (torch.randn, :335) is recorded so that the test feeds the same samples."""
import importlib.util
import os

import numpy as np
import torch

import gen_golden_refine as G

N_INST = 5


def main():
    G.import_reference()
    spec = importlib.util.spec_from_file_location("ref_rigid", "/root/reference/project/models/nodes/rigid.py")
    R = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(R)
    for step, oob in ((3300, True), (16300, True), (1300, False)):
        p, stats, grads = G.make_inputs(step + 7)
        g = torch.Generator().manual_seed(step + 11)
        N = G.N
        point_ids = torch.randint(0, N_INST, (N, 1), generator=g)
        instances_size = torch.rand(N_INST, 3, generator=g) * 30 + 8          # box edge lengths; means are in [-20, 20]
        ctrl = dict(G.CTRL, cull_out_of_bound=oob)
        model = R.RigidNodes(class_name="RigidNodes", ctrl=G.Cfg(ctrl), scene_scale=G.SCENE_SCALE, num_train_images=G.NUM_TRAIN_IMAGES,
                             device=torch.device("cpu"))
        for a in G.ATTRS:
            setattr(model, a, torch.nn.Parameter(p[a].clone()))
        model.point_ids = point_ids.clone()
        model.instances_size = instances_size.clone()
        groups = [{"params": [getattr(model, a)], "lr": 1e-3, "eps": 1e-15, "weight_decay": 0, "name": model.class_prefix + n}
                  for a, n in zip(G.ATTRS, G.NAMES)]
        opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        for a in G.ATTRS:
            getattr(model, a).grad = grads[a].clone()
        opt.step()
        rec = {"step": np.array(step), "N": np.array(N), "scene_scale": np.array(G.SCENE_SCALE),
               "num_train_images": np.array(G.NUM_TRAIN_IMAGES), "in_point_ids": point_ids.numpy().copy(),
               "instances_size": instances_size.numpy().copy()}
        for k, v in ctrl.items():
            rec["ctrl_" + k] = np.array(v)
        for a in G.ATTRS:
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
        for a, n in zip(G.ATTRS, G.NAMES):
            prm = getattr(model, a)
            st = opt.state[prm]
            rec["out" + a] = prm.detach().numpy().copy()
            rec["out_m" + a] = st["exp_avg"].numpy().copy()
            rec["out_v" + a] = st["exp_avg_sq"].numpy().copy()
        rec["out_point_ids"] = model.point_ids.numpy().copy()
        assert model.point_ids.shape[0] == model.num_points
        tag = "oob" if oob else "nob"
        np.savez_compressed(os.path.join(G.OUT, f"refine_rigid_{tag}_step{step}.npz"), **rec)
        print(f"step {step} ({tag}): {N} -> {model.num_points} points, noise rows {rec['samples'].shape[0]}")


if __name__ == "__main__":
    main()
