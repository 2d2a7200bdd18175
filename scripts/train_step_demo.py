#!/usr/bin/env python3
"""One COMPLETE training iteration on the benchmark scene (2 M Gaussians, 1920x1080): fused view (fwd+bwd) + the reference's
image loss (rgb L1 + SSIM + sky-mask BCE + lidar depth + grid TV) + Adam step on every parameter + densification
statistics.  Prints ms/iteration for the HIP ops of this repo and for the framework-op formulation of loss / optimiser
(the render is the same fused node in both, so the difference isolates what SURVEY.md 8f rank 1-2 add)."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from bilateral_driving_amd import harness as Hn
from bilateral_driving_amd.bilagrid import total_variation_loss
from bilateral_driving_amd.losses import pixel_loss, ssim_loss
from bilateral_driving_amd.optim import DensifyStats, FusedAdam
from oracle import loss_oracle as LO   # measurement script: framework-op SSIM = the oracle's conv2d restatement

dev = torch.device("cuda", 0)
W, H, N = 1920, 1080, 2_000_000
cams = Hn.ring_cameras(W, H, device=dev)
gen = torch.Generator().manual_seed(7)
sky = torch.rand(H, W, 3, generator=gen).to(dev)
pixels = torch.rand(H, W, 3, generator=gen).to(dev)
sky_mask = (torch.rand(H, W, generator=gen) < 0.3).float().to(dev)
lidar = (torch.rand(H, W, generator=gen) * 100 * (torch.rand(H, W, generator=gen) < 0.4)).to(dev)
lrs = dict(means=1.6e-4, quats=1e-3, log_scales=5e-3, opacity_logits=5e-2, sh=1.25e-4)


def run(fused_extras: bool, iters=12):
    params = Hn.synthetic_scene(N, seed=0, device=dev)
    for v in params.values(): v.requires_grad_(True)
    grids = [g.requires_grad_(True) for g in Hn.make_grids(len(cams), device=dev)]
    groups = [{"params": [p], "name": k, "lr": lrs[k], "eps": 1e-15, "weight_decay": 0} for k, p in params.items()]
    groups += [{"params": [g], "name": f"grid{i}", "lr": 2e-3, "eps": 1e-15, "weight_decay": 0} for i, g in enumerate(grids)]
    opt = (FusedAdam if fused_extras else torch.optim.Adam)(groups, lr=0.0, eps=1e-15)
    stats = DensifyStats(N, dev)
    tvw = [0.01 * 0.5 * math.sqrt(g.shape[4] * g.shape[3] * g.shape[2]) for g in grids]

    def it(i):
        v = i % len(cams)
        opt.zero_grad(set_to_none=True)
        out = Hn.render_view(params, cams[v], grids, v, sky)
        rgb, opac, depth = out["rgb"], out["opacity"], out["depth"]
        if fused_extras:
            loss = pixel_loss(rgb, opac, depth, pixels, sky_mask, lidar).sum() + 0.2 * ssim_loss(rgb, pixels)
        else:
            l1 = torch.abs(pixels - rgb).mean()
            bce = F.binary_cross_entropy(opac.squeeze(-1).clamp(0, 1), 1.0 - sky_mask, reduction="none").mean()
            hit = (lidar > 0).float()
            pd, gt = depth.squeeze(-1) * hit, lidar * hit
            m = (gt > 0.01) & (gt < 80) & (pd > 0.0001)
            loss = 0.8 * l1 + 0.2 * (1 - LO.ssim(pixels, rgb)) + 0.05 * bce + 0.01 * F.l1_loss(pd[m], gt[m], reduction="none").mean()
        for g, w in zip(grids, tvw):
            loss = loss + total_variation_loss(g, w)
        loss.backward()
        opt.step()
        if fused_extras:
            stats.update(out["info"])
        else:   # the reference's formulation (base.py:279-297 + vanilla.py:163-191), without its per-class split
            info = out["info"]
            grads = info["means2d"].absgrad.clone(); grads[..., 0] *= W / 2.0; grads[..., 1] *= H / 2.0
            vis = (info["radii"][0] > 0)
            stats.vis_counts[vis] = stats.vis_counts[vis] + 1
            stats.xys_grad_norm[vis] = grads[0].norm(dim=-1)[vis] + stats.xys_grad_norm[vis]
            stats.max_2Dsize[vis] = torch.maximum(stats.max_2Dsize[vis], info["radii"][0][vis] / float(max(W, H)))
        return loss

    for i in range(6): it(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(iters): loss = it(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3, float(loss)


for fe in (True, False, True, False):
    ms, loss = run(fe)
    print(("HIP loss / Adam / stats   " if fe else "framework loss/Adam/stats"), f"{ms:.3f} ms/iteration   (loss {loss:.5f})")
