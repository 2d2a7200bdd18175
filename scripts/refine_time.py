#!/usr/bin/env python3
"""Adaptive density control at the benchmark's size (2 M Gaussians, SH degree 3, Adam state): the fused refinement
(bilateral_driving_amd.densify.refinement_after) against the mask-index / cat / mask-again sequence the reference runs
(models/gaussians/vanilla.py:205-334 + basics.py:162-206), written here with plain torch ops on the same device.
Run on the GPU box:  python scripts/refine_time.py"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from test_gpu_11_refine import CTRL, Cfg, GROUPS, synthetic, build_model
from oracle import refine_oracle as RO          # names of the parameters only
from bilateral_driving_amd.densify import refinement_after

N, STEP = 2_000_000, 3300
dev = "cuda"


def torch_sequence(model, opt, samples):
    """Same decisions and tensor operations as the reference's refinement, as a sequence of torch launches."""
    c = model.ctrl_cfg
    names = dict(zip(RO.PARAMS, GROUPS))
    with torch.no_grad():
        avg = model.xys_grad_norm / model.vis_counts
        high = avg > c.densify_grad_thresh
        splits = torch.exp(model._scales).max(dim=-1).values > c.densify_size_thresh * model.scene_scale
        splits |= model.max_2Dsize > c.split_screen_size
        splits &= high
        ns, samps = int(splits.sum().item()), c.n_split_samples
        scaled = torch.exp(model._scales[splits]).repeat(samps, 1) * samples
        q = model._quats[splits]; q = q / q.norm(dim=-1, keepdim=True); q = torch.nn.functional.normalize(q, dim=-1).repeat(samps, 1)
        w, x, y, z = q.unbind(-1)
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z),
                         2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
        new = {"_means": torch.bmm(R, scaled[..., None]).squeeze(-1) + model._means[splits].repeat(samps, 1)}
        for a in ("_features_dc", "_opacities", "_quats"):
            new[a] = getattr(model, a)[splits].repeat(samps, 1)
        new["_features_rest"] = model._features_rest[splits].repeat(samps, 1, 1)
        new["_scales"] = torch.log(torch.exp(model._scales[splits]) / 1.6).repeat(samps, 1)
        model._scales[splits] = torch.log(torch.exp(model._scales[splits]) / 1.6)
        dups = (torch.exp(model._scales).max(dim=-1).values <= c.densify_size_thresh * model.scene_scale) & high
        old = {a: getattr(model, a) for a in RO.PARAMS}
        for a in RO.PARAMS:
            setattr(model, a, torch.nn.Parameter(torch.cat([old[a].detach(), new[a], old[a][dups]], dim=0)))
        m2d = torch.cat([model.max_2Dsize, torch.zeros(samps * ns + int(dups.sum().item()), device=dev)])
        for idx, n in ((torch.where(splits)[0], samps), (torch.where(dups)[0], 1)):
            for a in RO.PARAMS:
                st = opt.state[old[a]]
                for k in ("exp_avg", "exp_avg_sq"):
                    rep = (n,) + tuple(1 for _ in range(st[k].dim() - 1))
                    st[k] = torch.cat([st[k], torch.zeros_like(st[k][idx]).repeat(*rep)], dim=0)
        culls = (torch.sigmoid(model._opacities) < c.cull_alpha_thresh).squeeze()
        culls |= torch.exp(model._scales).max(dim=-1).values > c.cull_scale_thresh * model.scene_scale
        culls |= m2d > c.cull_screen_size
        for a in RO.PARAMS:
            prm = torch.nn.Parameter(getattr(model, a)[~culls].detach())
            st = opt.state.pop(old[a])
            st["exp_avg"] = st["exp_avg"][~culls]; st["exp_avg_sq"] = st["exp_avg_sq"][~culls]
            setattr(model, a, prm)
            opt.state[prm] = st
            [g for g in opt.param_groups if g["name"] == model.class_prefix + names[a]][0]["params"] = [prm]
    return model._means.shape[0]


P, M, V, stats = synthetic(N, seed=1)
sch = RO.schedule(STEP, CTRL, 30.0, 150)
n_split = int(RO.plan(sch, CTRL, P["_scales"], P["_opacities"], stats["xys_grad_norm"], stats["vis_counts"], stats["max_2Dsize"])[0].sum())
samples = torch.randn(2 * n_split, 3, device=dev)
res = {}
for name in ("torch sequence", "fused", "torch sequence", "fused"):
    model, opt = build_model(P, M, V, stats, CTRL, 30.0, 150, STEP, torch.optim.Adam)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if name == "fused":
        refinement_after(model, STEP, opt, samples=samples, verbose=False); n_new = model._means.shape[0]
    else:
        n_new = torch_sequence(model, opt, samples)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    res.setdefault(name, []).append((dt, n_new, model._means.detach().clone()))
    # bytes every formulation has to move at least: read 3 x 59 floats per old Gaussian, write 3 x 59 per new one
    gb = (N + n_new) * 59 * 3 * 4 / 1e9
    print(f"{name:15s} {dt:8.2f} ms   {N} -> {n_new} Gaussians   ({gb / dt * 1e3:.0f} GB/s of the {gb:.2f} GB minimum)")
a, b = res["torch sequence"][-1], res["fused"][-1]
assert a[1] == b[1], (a[1], b[1])
print("same set:", bool(torch.allclose(a[2], b[2], rtol=1e-5, atol=1e-4)), " speed-up %.1fx" % (a[0] / b[0]))
