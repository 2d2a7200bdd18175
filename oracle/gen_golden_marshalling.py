#!/usr/bin/env python3
"""Golden vectors for the input marshalling of the training step, produced by the REFERENCE's own code:
``dataclass_gs`` (/root/reference/project/models/gaussians/basics.py:119-160), ``BasicTrainer.process_camera`` and
``BasicTrainer.collect_gaussians`` (models/trainers/base.py:317-383) (module loaded by file path) called unbound on a stand-in ``self`` with three Gaussian classes
(one of which has no instance in the frame and returns None), and ``VanillaGaussians.get_gaussians`` (models/gaussians/vanilla.py:378-414)
in its sh_degree = 0 branch (the SH branch calls gsplat, which only exists on the GPU here: it is compared on the device instead).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_marshalling.py        (build container only)

models/trainers/base.py imports kornia / viser / nerfview / pytorch_msssim / torchmetrics at module level; none of them is touched
by the three functions, so they are stubbed in sys.modules the way gen_golden_refine.py stubs tensorly / pytorch3d / nvdiffrast."""
import os
import sys
import types

import numpy as np
import torch

import gen_golden_refine as G

KEYS = ("_means", "_scales", "_quats", "_rgbs", "_opacities")
DETACH_SETS = ([], ["means"], ["activated_opacities", "colors"], ["scales", "quats"], ["means", "colors", "scales", "quats", "activated_opacities"])


def stub(name, **a):
    m = types.ModuleType(name)
    m.__dict__.update(a)
    sys.modules[name] = m


def main():
    V = G.import_reference()
    stub("kornia"); stub("viser")
    stub("nerfview", CameraState=object)          # only an annotation of the viewer callback
    stub("pytorch_msssim", SSIM=object)
    stub("torchmetrics"); stub("torchmetrics.image", PeakSignalNoiseRatio=object)
    stub("torchmetrics.image.lpip", LearnedPerceptualImagePatchSimilarity=object)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_trainer_base", "/root/reference/project/models/trainers/base.py")
    B = importlib.util.module_from_spec(spec)      # by path: models/trainers/__init__.py would pull the datasets (cv2) in
    spec.loader.exec_module(B)
    from models.gaussians.basics import dataclass_gs

    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.rand(*s, generator=g)
    sizes = {"Background": 7, "RigidNodes": 0, "DeformableNodes": 4}
    classes = {"Background": 0, "RigidNodes": 1, "DeformableNodes": 2}
    dicts = {}
    for name, n in sizes.items():
        dicts[name] = None if n == 0 else {"_means": r(n, 3), "_opacities": r(n, 1), "_rgbs": r(n, 3), "_scales": r(n, 3), "_quats": r(n, 4)}

    class Fake:
        def __init__(self, d): self.d = d
        def get_gaussians(self, cam): return None if self.d is None else dict(self.d)

    c2w = torch.eye(4) + r(4, 4) * 0.1
    cam_infos = {"camera_to_world": c2w, "intrinsics": r(3, 3), "height": 9, "width": 13}
    shift = lambda c, ids: c + 1.0
    scale = lambda c, ids: c * 2.0
    rec = {}
    for tag, models, novel in (("plain", {}, False), ("refined", {"CamPosePerturb": shift, "CamPose": scale}, False),
                               ("novel", {"CamPosePerturb": shift, "CamPose": scale}, True)):
        me = types.SimpleNamespace(models=models)
        cam = B.BasicTrainer.process_camera(me, cam_infos, torch.tensor([3]), novel_view=novel)
        rec[f"cam_{tag}_c2w"] = cam.camtoworlds.numpy().copy()
        rec[f"cam_{tag}_gt"] = cam.camtoworlds_gt.numpy().copy()
        assert cam.H == 9 and cam.W == 13 and torch.equal(cam.Ks, cam_infos["intrinsics"])
    rec["cam_in_c2w"] = c2w.numpy().copy()
    me = types.SimpleNamespace(models={k: Fake(v) for k, v in dicts.items()}, gaussian_classes=classes, device=torch.device("cpu"),
                               render_dynamic_mask=True)
    gs = B.BasicTrainer.collect_gaussians(me, cam, torch.tensor([3]))
    for k in KEYS:
        rec["cat" + k] = getattr(gs, k).numpy().copy()
    rec["pts_labels"] = me.pts_labels.numpy().copy()
    rec["dynamic_pts_mask"] = me.dynamic_pts_mask.numpy().copy()
    for name, d in dicts.items():
        rec["n_" + name] = np.array(sizes[name])
        if d is not None:
            for k in KEYS:
                rec[f"in_{name}{k}"] = d[k].numpy().copy()
    # which accessor hands out a detached tensor for which detach_keys
    leaf = {k: r(3, 3).requires_grad_(True) for k in KEYS}
    table = []
    for ds in DETACH_SETS:
        o = dataclass_gs(_opacities=leaf["_opacities"], _means=leaf["_means"], _rgbs=leaf["_rgbs"], _scales=leaf["_scales"],
                         _quats=leaf["_quats"], detach_keys=[])
        o.set_grad_controller(list(ds))
        table.append([int(getattr(o, a).requires_grad) for a in ("opacities", "means", "rgbs", "scales", "quats")])
    rec["detach_table"] = np.array(table)
    # get_gaussians, sh_degree = 0
    ctrl = G.Cfg(dict(G.CTRL, sh_degree=0))
    model = V.VanillaGaussians("Background", ctrl, scene_scale=30.0, num_train_images=10, device=torch.device("cpu"))
    p, _, _ = G.make_inputs(77)
    for a in G.ATTRS:
        setattr(model, a, torch.nn.Parameter(p[a][:50].clone()))
    model._features_rest = torch.nn.Parameter(torch.zeros(50, 0, 3))
    model.step = 1234
    out = model.get_gaussians(cam)
    for a in ("_means", "_features_dc", "_opacities", "_scales", "_quats"):
        rec["gg_in" + a] = getattr(model, a).detach().numpy().copy()
    for k in KEYS:
        rec["gg_out" + k] = out[k].detach().numpy().copy()
    np.savez_compressed(os.path.join(G.OUT, "marshalling.npz"), **rec)
    print("labels", rec["pts_labels"], "detach table", table)


if __name__ == "__main__":
    main()
