"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the image-loss terms that follow the hot
path.  Not imported by the product.

SSIM -- **parity unpinned**: the reference calls the external package pytorch_msssim, pinned at 1.0.0 in
/root/reference/requirements.txt:2 and used at /root/reference/project/models/trainers/base.py:15,114,541
(`SSIM(data_range=1.0, size_average=True, channel=3)`, loss = 1 - ssim(gt, pred) on [1,3,H,W]).  The package is neither
vendored in the reference tree nor installed here, so this file restates its published algorithm (pytorch_msssim/ssim.py,
v1.0.0: `_fspecial_gauss_1d`, `gaussian_filter`, `_ssim`, `ssim`) and is pinned only by this repo's own tests
(brute-force window sums, identities, autograd).  Equality with the package itself could not be checked.
L1 is torch's own `(a - b).abs().mean()` (models/trainers/base.py:540)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

WIN_SIZE, WIN_SIGMA, K1, K2 = 11, 1.5, 0.01, 0.03


def gauss_window(size: int = WIN_SIZE, sigma: float = WIN_SIGMA, dtype=torch.float32) -> torch.Tensor:
    """_fspecial_gauss_1d: float arithmetic in the order the package uses."""
    coords = torch.arange(size, dtype=dtype)
    coords -= size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    g /= g.sum()
    return g


def _filter(x: torch.Tensor, win: torch.Tensor) -> torch.Tensor:
    """gaussian_filter: separable VALID correlation over H then W, one group per channel.  x [1,C,H,W]."""
    C = x.shape[1]
    w = win.to(device=x.device, dtype=x.dtype).reshape(1, 1, -1).repeat(C, 1, 1)
    out = F.conv2d(x, w.unsqueeze(-1), groups=C)      # along H  (weight [C,1,11,1])
    out = F.conv2d(out, w.unsqueeze(-2), groups=C)    # along W  (weight [C,1,1,11])
    return out


def ssim_map(X: torch.Tensor, Y: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """_ssim: X, Y [1,C,H,W] -> per-pixel SSIM over the valid region [1,C,H-10,W-10]."""
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    win = gauss_window(dtype=X.dtype)
    mu1, mu2 = _filter(X, win), _filter(Y, win)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = _filter(X * X, win) - mu1_sq
    sigma2_sq = _filter(Y * Y, win) - mu2_sq
    sigma12 = _filter(X * Y, win) - mu1_mu2
    cs_map = (2 * sigma12 + C2) / (sigma1_sq + sigma2_sq + C2)
    return ((2 * mu1_mu2 + C1) / (mu1_sq + mu2_sq + C1)) * cs_map


def ssim(gt_hwc: torch.Tensor, pred_hwc: torch.Tensor) -> torch.Tensor:
    """The trainer's call: images [H,W,3] -> scalar mean SSIM (size_average=True: mean over channels of the
    per-channel spatial means = mean over everything)."""
    X = gt_hwc.permute(2, 0, 1)[None]
    Y = pred_hwc.permute(2, 0, 1)[None]
    m = ssim_map(X, Y)
    return torch.flatten(m, 2).mean(-1).mean()


def ssim_loss(gt_hwc: torch.Tensor, pred_hwc: torch.Tensor) -> torch.Tensor:
    return 1 - ssim(gt_hwc, pred_hwc)


def ssim_map_bruteforce(X: torch.Tensor, Y: torch.Tensor) -> torch.Tensor:
    """Independent check of ssim_map on small inputs: explicit 11x11 window sums in float64, no convolution calls."""
    X, Y = X.double(), Y.double()
    _, C, H, W = X.shape
    w1 = gauss_window(dtype=torch.float64)
    w2 = w1[:, None] * w1[None, :]
    C1, C2 = K1 ** 2, K2 ** 2
    out = torch.zeros(1, C, H - 10, W - 10, dtype=torch.float64)
    for c in range(C):
        for i in range(H - 10):
            for j in range(W - 10):
                px, py = X[0, c, i:i + 11, j:j + 11], Y[0, c, i:i + 11, j:j + 11]
                m1, m2 = (w2 * px).sum(), (w2 * py).sum()
                s1, s2, s12 = (w2 * px * px).sum() - m1 * m1, (w2 * py * py).sum() - m2 * m2, (w2 * px * py).sum() - m1 * m2
                out[0, c, i, j] = ((2 * m1 * m2 + C1) / (m1 * m1 + m2 * m2 + C1)) * ((2 * s12 + C2) / (s1 + s2 + C2))
    return out


# ------------------------------------------------------------------------------------------------------------------
# per-pixel terms of BasicTrainer.compute_losses (/root/reference/project/models/trainers/base.py:518-565) -- PINNED:
# tests/golden/pixel_loss_*.npz are produced by the reference's own models/losses.py (oracle/gen_golden_losses.py).
#   rgb   : |pixels*valid - rgb*valid|.mean()                                             (base.py:533-540)
#   mask  : F.binary_cross_entropy(opacity*valid, (1-sky)*valid).mean()                   (losses.py:82-84, base.py:536-549)
#   depth : DepthLoss(loss_type, normalize=False, use_inverse_depth=False)                (losses.py:92-178, base.py:553-557)
# Values AND gradients are written out in closed form (torch's BCE backward divides by max((1-x)x, float(1e-12)) and its
# forward clamps the logs at -100; autograd through a re-implementation would not reproduce the first).
# ------------------------------------------------------------------------------------------------------------------
def pixel_loss(rgb, pixels, opacity, sky_masks, depth, lidar, egocar=None, w=(0.8, 0.05, 0.01), depth_l2=False,
               max_depth=80.0):
    """rgb/pixels [H,W,3], opacity/depth [H,W,1], sky_masks/lidar/egocar [H,W].
    Returns dict(rgb_loss, sky_loss, depth_loss, total, v_rgb, v_opacity, v_depth) for upstream gradient 1."""
    dt = rgb.dtype
    valid = (1.0 - egocar) if egocar is not None else torch.ones_like(sky_masks)
    P = sky_masks.numel()
    diff = (rgb - pixels) * valid[..., None]
    rgb_loss = w[0] * diff.abs().sum() / (3 * P)
    v_rgb = w[0] / (3 * P) * torch.sign(diff) * valid[..., None]
    x, t = opacity.squeeze(-1) * valid, (1.0 - sky_masks) * valid
    bce = -(t * torch.log(x).clamp(min=-100.0) + (1.0 - t) * torch.log(1.0 - x).clamp(min=-100.0))
    sky_loss = w[1] * bce.sum() / P
    eps = 9.999999960041972e-13   # torch's BCE backward clamps with a FLOAT constant 1e-12, also for float64 inputs
    v_opacity = (w[1] / P * valid * (x - t) / ((1.0 - x) * x).clamp(min=eps))[..., None]
    hit = (lidar > 0).to(dt) * valid
    pred, gt = depth.squeeze(-1) * hit, lidar * hit
    m = (gt > 0.01) & (gt < max_depth) & (pred > 0.0001)
    cnt = m.sum().to(dt)
    e = (pred - gt) * m
    depth_loss = w[2] * ((e * e).sum() if depth_l2 else e.abs().sum()) / cnt
    v_depth = (w[2] / cnt * hit * m * (2.0 * e if depth_l2 else torch.sign(e)))[..., None]
    return dict(rgb_loss=rgb_loss, sky_loss=sky_loss, depth_loss=depth_loss, total=rgb_loss + sky_loss + depth_loss,
                v_rgb=v_rgb, v_opacity=v_opacity, v_depth=v_depth)


# ------------------------------------------------------------------------------------------------------------------
# per-step densification statistics -- PINNED by tests/golden/densify_stats.npz (the reference's own
# VanillaGaussians.after_train, /root/reference/project/models/gaussians/vanilla.py:163-191, fed as
# BasicTrainer.postprocess_per_train_step does, models/trainers/base.py:279-297; oracle/gen_golden_densify.py)
# ------------------------------------------------------------------------------------------------------------------
def densify_stats_update(state, absgrad, radii, width, height, batch_size=1):
    """state: dict(xys_grad_norm, vis_counts, max_2Dsize) or None on the first call; absgrad [N,2], radii [N] int.
    Returns the new state (closed form of the reference's masked-index updates)."""
    g = absgrad.clone()
    g[..., 0] *= width / 2.0 * batch_size
    g[..., 1] *= height / 2.0 * batch_size
    n = g.norm(dim=-1)
    vis = radii > 0
    size = radii.to(torch.float32) / float(max(width, height))
    if state is None:   # first call: norm for EVERY Gaussian, vis_counts = 1 for EVERY Gaussian
        return dict(xys_grad_norm=n.clone(), vis_counts=torch.ones_like(n),
                    max_2Dsize=torch.where(vis, size, torch.zeros_like(size)))
    return dict(xys_grad_norm=torch.where(vis, n + state["xys_grad_norm"], state["xys_grad_norm"]),
                vis_counts=torch.where(vis, state["vis_counts"] + 1, state["vis_counts"]),
                max_2Dsize=torch.where(vis, torch.maximum(state["max_2Dsize"], size), state["max_2Dsize"]))


def reg_losses(pixels, opacity=None, depth=None, rgb=None, dyn_opacity=None, egocar_masks=None, dyn_threshold=0.2):
    """Regularisers of BasicTrainer.compute_losses, transcribed term by term (differentiable torch ops; autograd = gradient oracle):

      * opacity entropy       /root/reference/project/models/trainers/base.py:566-572
      * inverse-depth smoothness  :574-585 -> kornia.losses.inverse_depth_smoothness_loss.  kornia is an external package that is not
        in this image (``import kornia`` fails) and not pinned by the reference: PARITY UNPINNED; restated from kornia's published
        definition (image-gradient-weighted first differences of the inverse depth, mean over x-differences + mean over y-differences)
      * dynamic-region L1     :638-651 (the mask is built from ``outputs["Dynamic_opacity"].data``: no gradient through it)

    Returns the three unweighted terms (0 where the reference would add no term)."""
    terms = []
    if opacity is not None:
        p = torch.clamp(opacity.squeeze(), 1e-6, 1 - 1e-6)
        terms.append((-p * torch.log(p)).mean())
    else:
        terms.append(pixels.new_zeros(()))
    if depth is not None:
        inverse_depth = 1 / (depth.reshape(pixels.shape[0], pixels.shape[1], 1) + 1e-5)
        idepth = inverse_depth[None].repeat(1, 1, 1, 3).permute(0, 3, 1, 2)
        image = pixels[None].permute(0, 3, 1, 2)
        idx, idy = idepth[:, :, :, :-1] - idepth[:, :, :, 1:], idepth[:, :, :-1, :] - idepth[:, :, 1:, :]
        imx, imy = image[:, :, :, :-1] - image[:, :, :, 1:], image[:, :, :-1, :] - image[:, :, 1:, :]
        wx = torch.exp(-torch.mean(torch.abs(imx), dim=1, keepdim=True))
        wy = torch.exp(-torch.mean(torch.abs(imy), dim=1, keepdim=True))
        terms.append(torch.abs(idx * wx).mean() + torch.abs(idy * wy).mean())
    else:
        terms.append(pixels.new_zeros(()))
    dyn = pixels.new_zeros(())
    if dyn_opacity is not None:
        valid = torch.ones_like(pixels[..., 0]) if egocar_masks is None else (1.0 - egocar_masks).to(pixels.dtype)
        gt_rgb, predicted_rgb = pixels * valid[..., None], rgb * valid[..., None]
        mask = (dyn_opacity.detach() > dyn_threshold).reshape(valid.shape) & valid.bool()
        if mask.sum() > 0:
            dyn = torch.abs(gt_rgb[mask] - predicted_rgb[mask]).mean()
    terms.append(dyn)
    return torch.stack(terms)
