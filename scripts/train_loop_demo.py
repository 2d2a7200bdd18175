#!/usr/bin/env python3
"""End-to-end optimisation on MI355X with every per-iteration piece of the reference's trainer replaced by this repo's ops:
EnvLight cube-map sky -> fused view (projection, tile lists, SH, compositing, expected depth, clamp + sky blend, multi-scale
bilateral transform) -> image loss (rgb L1 + SSIM + sky-mask BCE + lidar depth) + grid TV -> FusedAdam -> densification
statistics -> every 100 steps split / duplicate / cull with the Adam-state surgery (densify.refinement_after).

Ground truth: a synthetic scene, a random sky cube and non-identity per-image grids render six target views.  The trained
model starts from a thinned, perturbed copy with grey colours, an untrained sky and identity grids; the script prints PSNR
against the targets, which has to rise, and the number of Gaussians, which densification changes.
Run on the GPU box:  python scripts/train_loop_demo.py [steps]"""
import math, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bilateral_driving_amd import harness as Hn
from bilateral_driving_amd.bilagrid import total_variation_loss
from bilateral_driving_amd.densify import refinement_after
from bilateral_driving_amd.envlight import EnvLight
from bilateral_driving_amd.losses import pixel_loss, ssim_loss
from bilateral_driving_amd.optim import DensifyStats, FusedAdam

dev = torch.device("cuda", 0)
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 500
W, H, N_GT = 640, 360, 60_000
torch.manual_seed(0)
cams = Hn.ring_cameras(W, H, device=dev)


def view_dirs(cam):
    jj, ii = torch.meshgrid(torch.arange(W, device=dev) + 0.5, torch.arange(H, device=dev) + 0.5, indexing="xy")
    d = torch.stack([jj, ii, torch.ones_like(jj)], -1) @ torch.linalg.inv(cam.K).T @ torch.linalg.inv(cam.viewmat)[:3, :3].T
    return torch.nn.functional.normalize(d, dim=-1).contiguous()


dirs = [view_dirs(c) for c in cams]
# ---- ground truth ---------------------------------------------------------------------------------------------------
gt = Hn.synthetic_scene(N_GT, seed=1, device=dev)
gt_sky = EnvLight("Sky", resolution=64)
with torch.no_grad():
    low = torch.rand(6, 4, 4, 3, device=dev)
    gt_sky.base.copy_(torch.nn.functional.interpolate(low.permute(0, 3, 1, 2), size=(64, 64), mode="bilinear").permute(0, 2, 3, 1))
gt_grids = Hn.make_grids(len(cams), seed=3, device=dev)
targets, sky_masks, lidar = [], [], []
with torch.no_grad():
    for v, cam in enumerate(cams):
        out = Hn.render_view(gt, cam, gt_grids, v, gt_sky({"viewdirs": dirs[v]}))
        targets.append(out["rgb"].clone())
        sky_masks.append((out["opacity"].squeeze(-1) < 0.5).float())
        keep = (torch.rand(H, W, device=dev) < 0.3) & (out["opacity"].squeeze(-1) > 0.9)
        lidar.append(out["depth"].squeeze(-1) * keep)

# ---- the model, in the reference's parameter layout (models/gaussians/vanilla.py:88-110) -------------------------------
sel = torch.randperm(N_GT, device=dev)[: N_GT // 2]
model = types.SimpleNamespace(
    class_prefix="Background#", scene_scale=30.0, num_train_images=len(cams), step=0, xys_grad_norm=None, vis_counts=None, max_2Dsize=None,
    ctrl_cfg=types.SimpleNamespace(warmup_steps=50, reset_alpha_interval=3000, refine_interval=100, n_split_samples=2, reset_alpha_value=0.01,
                                   densify_grad_thresh=0.0002, densify_size_thresh=0.002, cull_alpha_thresh=0.005, cull_scale_thresh=0.5,
                                   cull_screen_size=0.15, split_screen_size=0.05, stop_screen_size_at=4000, stop_split_at=15000))
P = torch.nn.Parameter
model._means = P(gt["means"][sel] + 0.05 * torch.randn(len(sel), 3, device=dev))
model._features_dc = P(torch.zeros(len(sel), 3, device=dev))
model._features_rest = P(torch.zeros(len(sel), 15, 3, device=dev))
model._opacities = P(torch.full((len(sel), 1), -1.0, device=dev))
model._scales = P(gt["log_scales"][sel] + 0.2)
model._quats = P(gt["quats"][sel].clone())
sky = EnvLight("Sky", resolution=64)
grids = [g.requires_grad_(True) for g in Hn.make_grids(len(cams), seed=0, device=dev)]
with torch.no_grad():
    for g in grids:
        g.copy_(torch.tensor([1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0], device=dev).reshape(1, 12, 1, 1, 1).expand_as(g))
names = dict(_means=("xyz", 1.6e-3), _features_dc=("sh_dc", 2.5e-3), _features_rest=("sh_rest", 1.25e-4), _opacities=("opacity", 5e-2),
             _scales=("scaling", 5e-3), _quats=("rotation", 1e-3))
groups = [{"params": [getattr(model, a)], "name": model.class_prefix + n, "lr": lr, "eps": 1e-15, "weight_decay": 0} for a, (n, lr) in names.items()]
groups += [{"params": [g], "name": f"Affine#grid{i}", "lr": 2e-3, "eps": 1e-15, "weight_decay": 0} for i, g in enumerate(grids)]
groups += [{"params": [sky.base], "name": "Sky#all", "lr": 1e-2, "eps": 1e-15, "weight_decay": 0}]
opt = FusedAdam(groups, lr=0.0, eps=1e-15)
tvw = [0.01 * 0.5 * math.sqrt(g.shape[4] * g.shape[3] * g.shape[2]) for g in grids]
stats = None


def psnr_all():
    with torch.no_grad():
        mse = 0.0
        for v, cam in enumerate(cams):
            out = Hn.render_view(current_params(), cam, grids, v, sky({"viewdirs": dirs[v]}))
            mse += float(((out["rgb"] - targets[v]) ** 2).mean())
    return -10 * math.log10(mse / len(cams))


def current_params():
    return dict(means=model._means, quats=model._quats, log_scales=model._scales, opacity_logits=model._opacities.squeeze(-1),
                sh=torch.cat([model._features_dc[:, None, :], model._features_rest], dim=1))      # vanilla.py:378-414


print(f"step {0:4d}  N {model._means.shape[0]:6d}  PSNR {psnr_all():6.2f} dB")
torch.cuda.synchronize(); t0 = time.perf_counter()
for step in range(1, STEPS + 1):
    model.step = step
    v = step % len(cams)
    opt.zero_grad(set_to_none=True)
    out = Hn.render_view(current_params(), cams[v], grids, v, sky({"viewdirs": dirs[v]}))
    loss = pixel_loss(out["rgb"], out["opacity"], out["depth"], targets[v], sky_masks[v], lidar[v]).sum() + 0.2 * ssim_loss(out["rgb"], targets[v])
    for g, w in zip(grids, tvw):
        loss = loss + total_variation_loss(g, w)
    loss.backward()
    opt.step()
    if stats is None:
        stats = DensifyStats(model._means.shape[0], dev)
    stats.update(out["info"])
    if step % model.ctrl_cfg.refine_interval == 0:
        model.xys_grad_norm, model.vis_counts, model.max_2Dsize = stats.xys_grad_norm, stats.vis_counts, stats.max_2Dsize
        refinement_after(model, step, opt, verbose=False)
        stats = None
        torch.cuda.synchronize()
        print(f"step {step:4d}  N {model._means.shape[0]:6d}  PSNR {psnr_all():6.2f} dB  loss {float(loss):.4f}  "
              f"({(time.perf_counter() - t0) / step * 1e3:.2f} ms/iteration incl. evaluation)")
