#!/usr/bin/env python3
"""Golden vectors for the learnable camera pose, produced by the REFERENCE's own code: CameraOptModule
(/root/reference/project/models/modules.py:822-874) with its rotation_6d_to_matrix (utils/geometry.py:83-105): poses out, and the
gradient a loss on inverse(pose) -- the view matrix the rasterizer receives, trainers/base.py:399 -- sends to the embedding.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_camera_opt.py        (build container only)

models/modules.py is imported the way gen_golden_refine.py does it (tensorly / pytorch3d / nvdiffrast stubbed)."""
import os

import numpy as np
import torch

import gen_golden_refine as G


def main():
    G.import_reference()
    import models.modules as M
    g = torch.Generator().manual_seed(3)
    mod = M.CameraOptModule("CamPose", 6, device=torch.device("cpu"))
    with torch.no_grad():
        mod.embeds.weight.copy_(torch.randn(6, 9, generator=g) * 0.1)
    c2w = torch.eye(4).repeat(4, 1, 1)
    c2w[:, :3, :] += torch.randn(4, 3, 4, generator=g) * 0.2
    ids = torch.tensor([5, 0, 2, 2])
    out = mod(c2w, ids)
    Gv = torch.randn(4, 4, 4, generator=g)
    (torch.linalg.inv(out) * Gv).sum().backward()
    rec = {"embeds": mod.embeds.weight.detach().numpy().copy(), "c2w": c2w.numpy(), "ids": ids.numpy(), "out": out.detach().numpy(),
           "v_viewmat": Gv.numpy(), "grad_embeds": mod.embeds.weight.grad.numpy().copy(),
           "state_keys": np.array(sorted(mod.state_dict().keys())), "identity": mod.identity.numpy().copy()}
    zero = M.CameraOptModule("CamPose", 3, device=torch.device("cpu"))
    rec["out_zero_init"] = zero(c2w[:3], torch.tensor([0, 1, 2])).detach().numpy()
    np.savez_compressed(os.path.join(G.OUT, "camera_opt.npz"), **rec)
    print("ok", out.shape, float(mod.embeds.weight.grad.abs().max()))


if __name__ == "__main__":
    main()
