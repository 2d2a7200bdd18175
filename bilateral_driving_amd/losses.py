"""Image loss of the training step on the path's outputs (reference:
models/trainers/base.py:518-565 + models/losses.py, affine TV :590-594 + models/modules.py:445,466-472):
``photometric_tv_loss`` (L1 + per-level TV as ONE node accumulating into one device scalar -- the benchmark's loss),
``pixel_loss`` (rgb L1 + sky-mask BCE + lidar depth in one pass each way), ``ssim`` / ``ssim_loss``."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import torch
from torch import Tensor

from . import _lib as L
from .bilagrid import _levels_struct


class _PhotometricTV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb: Tensor, target: Tensor, tv_weights: tuple, grid_grads, *grids: Tensor):
        L.require_gpu(rgb, target, *grids)
        lib, st = L.lib(), L.stream()
        rgb, target = rgb.contiguous(), target.contiguous()
        assert rgb.shape == target.shape and rgb.dtype == torch.float32 and target.dtype == torch.float32
        grids = [g.contiguous() for g in grids]
        out = torch.zeros(1, device=rgb.device, dtype=torch.float32)
        L.check(lib.bds_l1_mean_fwd(rgb.numel(), L.ptr(rgb), L.ptr(target), L.ptr(out), st), "bds_l1_mean_fwd")
        if grids:
            lv = _levels_struct(grids, None, [1] * len(grids))
            wts = (C.c_float * len(grids))(*[float(w) for w in tv_weights])
            L.check(lib.bds_bilagrid_tv_ms_fwd(len(grids), lv, wts, L.ptr(out), st), "bds_bilagrid_tv_ms_fwd")
        ctx.save_for_backward(rgb, target, *grids)
        ctx.tv_weights = tuple(float(w) for w in tv_weights)
        ctx.grid_grads = grid_grads
        return out.reshape(())

    @staticmethod
    def backward(ctx, v_out):
        rgb, target, *grids = ctx.saved_tensors
        lib, st = L.lib(), L.stream()
        v = v_out.reshape(1).to(torch.float32).contiguous()
        v_rgb = None
        if ctx.needs_input_grad[0]:
            v_rgb = torch.empty_like(rgb)
            L.check(lib.bds_l1_mean_bwd(rgb.numel(), L.ptr(rgb), L.ptr(target), L.ptr(v), L.ptr(v_rgb), st), "bds_l1_mean_bwd")
        v_grids = [None] * len(grids)
        need = [ctx.needs_input_grad[4 + i] for i in range(len(grids))]
        if any(need) and ctx.grid_grads is not None:    # add in place to the caller's accumulators, return nothing
            sel = [g for i, g in enumerate(grids) if need[i]]
            sel_v = [a for i, a in enumerate(ctx.grid_grads) if need[i]]
            lv = _levels_struct(sel, sel_v, [1] * len(sel))
            wts = (C.c_float * len(sel))(*[ctx.tv_weights[i] for i in range(len(grids)) if need[i]])
            L.check(lib.bds_bilagrid_tv_ms_bwd(len(sel), lv, wts, L.ptr(v), st), "bds_bilagrid_tv_ms_bwd")
        elif any(need):
            sizes = [(g.numel() + 3) // 4 * 4 if need[i] else 0 for i, g in enumerate(grids)]   # 16-byte aligned slices
            flat = torch.zeros(sum(sizes), device=rgb.device, dtype=torch.float32)               # one fill for all levels
            off, sel, sel_v, sel_w = 0, [], [], []
            for i, g in enumerate(grids):
                if not need[i]:
                    continue
                vg = flat[off:off + g.numel()].view(g.shape)
                off += sizes[i]
                v_grids[i] = vg
                sel.append(g); sel_v.append(vg); sel_w.append(ctx.tv_weights[i])
            lv = _levels_struct(sel, sel_v, [1] * len(sel))
            wts = (C.c_float * len(sel))(*sel_w)
            L.check(lib.bds_bilagrid_tv_ms_bwd(len(sel), lv, wts, L.ptr(v), st), "bds_bilagrid_tv_ms_bwd")
        return (v_rgb, None, None, None, *v_grids)


def loss_slots(dev) -> Tensor:
    """Zeroed accumulator of the one-launch training losses: LOSS_SLOTS slots, LOSS_SLOT_STRIDE floats apart (``slots_value``)."""
    return torch.zeros(L.LOSS_SLOTS * L.LOSS_SLOT_STRIDE, device=dev, dtype=torch.float32)


def slots_value(buf: Tensor) -> Tensor:
    return buf.view(L.LOSS_SLOTS, L.LOSS_SLOT_STRIDE)[:, 0].sum()


def photometric_tv_train(rgb: Tensor, target: Tensor, grids: Sequence[Tensor], tv_weights: Sequence[float], grid_grads: Sequence[Tensor],
                         slots: bool = False):
    """The direct step's loss (no autograd graph; ``fused_view.train_view``): value and gradient of
    mean|rgb - target| + sum_l tv_weights[l] * total_variation(grids[l]) in ONE launch (``bds_l1_tv_train``): returns
    (loss [0-d], v_rgb); the TV gradient is ADDED to ``grid_grads`` (the grids' ``.grad`` slices) with atomics.
    ``slots=True``: the first result is the slotted accumulator instead (``slots_value`` sums it: a caller with a better place for
    that small reduction than right behind this launch)."""
    L.require_gpu(rgb, target, *grids)
    lib, st = L.lib(), L.stream()
    rgb, target = rgb.contiguous(), target.contiguous()
    assert rgb.shape == target.shape and rgb.dtype == torch.float32 and target.dtype == torch.float32
    grids = [g.contiguous() for g in grids]
    assert len(grid_grads) == len(grids) and all(a.numel() == g.numel() and a.is_contiguous() for a, g in zip(grid_grads, grids))
    out = loss_slots(rgb.device)
    v_rgb = torch.empty_like(rgb)
    lv = _levels_struct(grids, list(grid_grads), [1] * len(grids)) if grids else None
    wts = (C.c_float * max(len(grids), 1))(*[float(w) for w in tv_weights])
    L.check(lib.bds_l1_tv_train(rgb.numel(), L.ptr(rgb), L.ptr(target), len(grids), lv, wts, 1.0, L.ptr(out), L.LOSS_SLOTS, L.ptr(v_rgb), st),
            "bds_l1_tv_train")
    return (out if slots else slots_value(out)), v_rgb


def photometric_tv_loss(rgb: Tensor, target: Tensor, grids: Sequence[Tensor], tv_weights: Sequence[float],
                        grid_grads: Sequence[Tensor] = None) -> Tensor:
    """mean|rgb - target| + sum_l tv_weights[l] * total_variation(grids[l])  (grids [n_img,12,L,gy,gx]).
    ``grid_grads`` (optional, one tensor per grid, same shapes): the TV gradient is ADDED to these in place (accumulators that
    already are the grids' ``.grad``, ``dist.FrameExchange.tail_grads()``) and autograd receives no gradient for the grids."""
    assert len(grids) == len(tv_weights)
    if grid_grads is not None:
        assert len(grid_grads) == len(grids) and all(a.numel() == g.numel() and a.is_contiguous() for a, g in zip(grid_grads, grids))
        grid_grads = list(grid_grads)
    return _PhotometricTV.apply(rgb, target, tuple(tv_weights), grid_grads, *grids)


class _SSIM(torch.autograd.Function):
    """mean SSIM(target, pred) of [H,W,C] images: pytorch_msssim.SSIM(data_range=1, size_average=True) as the reference
    trainer calls it (models/trainers/base.py:114,541); gradient w.r.t. pred only (the ground truth is data)."""

    @staticmethod
    def forward(ctx, pred: Tensor, target: Tensor):
        L.require_gpu(pred, target)
        lib, st = L.lib(), L.stream()
        pred, target = pred.contiguous(), target.contiguous()
        assert pred.shape == target.shape and pred.dim() == 3 and pred.dtype == torch.float32 and target.dtype == torch.float32
        H, W, CH = pred.shape
        if H < 11 or W < 11:
            raise ValueError("SSIM needs images of at least 11x11 pixels (window 11, valid region)")
        out = torch.zeros(1, device=pred.device, dtype=torch.float32)
        ws = None
        if ctx.needs_input_grad[0]:
            ws = torch.empty(lib.bds_ssim_workspace_bytes(H, W, CH), device=pred.device, dtype=torch.uint8)
        L.check(lib.bds_ssim_fwd(H, W, CH, L.ptr(target), L.ptr(pred), L.ptr(out), L.ptr(ws), 0 if ws is None else ws.numel(), st),
                "bds_ssim_fwd")
        ctx.save_for_backward(pred, target, ws)
        return out.reshape(())

    @staticmethod
    def backward(ctx, v_out):
        pred, target, ws = ctx.saved_tensors
        if ws is None:
            return None, None
        H, W, CH = pred.shape
        v = v_out.reshape(1).to(torch.float32).contiguous()
        v_pred = torch.empty_like(pred)
        L.check(L.lib().bds_ssim_bwd(H, W, CH, L.ptr(target), L.ptr(pred), L.ptr(ws), ws.numel(), L.ptr(v), L.ptr(v_pred), L.stream()),
                "bds_ssim_bwd")
        return v_pred, None


def ssim(pred: Tensor, target: Tensor) -> Tensor:
    """Mean structural similarity of two [H,W,C] images in [0,1] (11x11 Gaussian window, valid region)."""
    return _SSIM.apply(pred, target)


def ssim_loss(pred: Tensor, target: Tensor) -> Tensor:
    """1 - ssim: the reference's `ssim_loss` before its weight (models/trainers/base.py:541-544)."""
    return 1.0 - ssim(pred, target)


class _PixelLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, opacity, depth, pixels, sky_masks, lidar, egocar, w, depth_l2, max_depth):
        L.require_gpu(rgb, pixels)
        lib, st = L.lib(), L.stream()
        f = lambda t: None if t is None else t.contiguous().to(torch.float32)
        rgb, opacity, depth, pixels, sky_masks, lidar, egocar = (f(t) for t in (rgb, opacity, depth, pixels, sky_masks, lidar, egocar))
        P = rgb.numel() // 3
        assert rgb.shape[-1] == 3 and pixels.shape == rgb.shape
        for t in (opacity, depth, sky_masks, lidar, egocar):
            assert t is None or t.numel() == P, "per-pixel inputs must have H*W elements"
        use_mask = opacity is not None and sky_masks is not None and w[1] != 0.0
        use_depth = depth is not None and lidar is not None and w[2] != 0.0
        sums = torch.empty(4, device=rgb.device, dtype=torch.float32)
        terms = torch.empty(3, device=rgb.device, dtype=torch.float32)
        L.check(lib.bds_pixel_loss_fwd(P, L.ptr(rgb), L.ptr(pixels), L.ptr(opacity) if use_mask else None,
                                       L.ptr(sky_masks) if use_mask else None, L.ptr(depth) if use_depth else None,
                                       L.ptr(lidar) if use_depth else None, L.ptr(egocar), w[0], w[1], w[2], int(depth_l2), max_depth,
                                       L.ptr(sums), L.ptr(terms), st), "bds_pixel_loss_fwd")
        ctx.save_for_backward(rgb, opacity, depth, pixels, sky_masks, lidar, egocar, sums)
        ctx.cfg = (tuple(w), int(depth_l2), float(max_depth), use_mask, use_depth)
        return terms

    @staticmethod
    def backward(ctx, v_terms):
        rgb, opacity, depth, pixels, sky_masks, lidar, egocar, sums = ctx.saved_tensors
        w, depth_l2, max_depth, use_mask, use_depth = ctx.cfg
        P = rgb.numel() // 3
        v_terms = v_terms.contiguous().to(torch.float32)
        v_rgb = torch.empty_like(rgb)
        v_op = torch.empty_like(opacity) if (opacity is not None and ctx.needs_input_grad[1]) else None
        v_dp = torch.empty_like(depth) if (depth is not None and ctx.needs_input_grad[2]) else None
        L.check(L.lib().bds_pixel_loss_bwd(P, L.ptr(rgb), L.ptr(pixels), L.ptr(opacity) if use_mask else None,
                                           L.ptr(sky_masks) if use_mask else None, L.ptr(depth) if use_depth else None,
                                           L.ptr(lidar) if use_depth else None, L.ptr(egocar), w[0], w[1], w[2], depth_l2, max_depth,
                                           L.ptr(sums), L.ptr(v_terms), L.ptr(v_rgb), L.ptr(v_op), L.ptr(v_dp), L.stream()),
                "bds_pixel_loss_bwd")
        return v_rgb, v_op, v_dp, None, None, None, None, None, None, None


def pixel_loss(rgb: Tensor, opacity: Tensor, depth: Tensor, pixels: Tensor, sky_masks: Tensor, lidar_depth: Tensor,
               egocar_masks: Tensor = None, w_rgb: float = 0.8, w_mask: float = 0.05, w_depth: float = 0.01,
               depth_loss_type: str = "l1", max_depth: float = 80.0) -> Tensor:
    """The per-pixel terms of ``BasicTrainer.compute_losses`` (models/trainers/base.py:518-565) in one pass each way:
    returns the three WEIGHTED terms ``[rgb_loss, sky_loss_opacity, depth_loss]`` (sum them for the total; they are what
    the trainer logs).  rgb / pixels [H,W,3]; opacity, depth [H,W,1] or [H,W]; sky_masks, lidar_depth, egocar_masks [H,W].
    Pass ``opacity=None`` / ``depth=None`` to drop a term."""
    if depth_loss_type not in ("l1", "l2"):
        raise NotImplementedError(f"Unknown loss type: {depth_loss_type}")   # DepthLoss (models/losses.py:150) also has smooth_l1
    return _PixelLoss.apply(rgb, opacity, depth, pixels, sky_masks, lidar_depth, egocar_masks,
                            (float(w_rgb), float(w_mask), float(w_depth)), depth_loss_type == "l2", float(max_depth))


class _RegLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, opacity, depth, rgb, pixels, dyn_opacity, egocar, dyn_threshold):
        L.require_gpu(pixels, opacity, depth, rgb)
        f = lambda t: None if t is None else t.contiguous().to(torch.float32)
        opacity, depth, rgb, pixels, dyn_opacity, egocar = (f(t) for t in (opacity, depth, rgb, pixels, dyn_opacity, egocar))
        H, W = pixels.shape[0], pixels.shape[1]
        for t in (opacity, depth, dyn_opacity, egocar):
            assert t is None or t.numel() == H * W, "per-pixel inputs must have H*W elements"
        dev = pixels.device
        sums, terms = torch.empty(5, device=dev, dtype=torch.float32), torch.empty(3, device=dev, dtype=torch.float32)
        L.check(L.lib().bds_reg_loss_fwd(H, W, L.ptr(opacity), L.ptr(depth), L.ptr(pixels), L.ptr(rgb), L.ptr(dyn_opacity), L.ptr(egocar),
                                         dyn_threshold, L.ptr(sums), L.ptr(terms), L.stream()), "bds_reg_loss_fwd")
        ctx.save_for_backward(opacity, depth, rgb, pixels, dyn_opacity, egocar, sums)
        ctx.dyn_threshold = dyn_threshold
        return terms

    @staticmethod
    def backward(ctx, v_terms):
        opacity, depth, rgb, pixels, dyn_opacity, egocar, sums = ctx.saved_tensors
        H, W = pixels.shape[0], pixels.shape[1]
        v_terms = v_terms.contiguous().to(torch.float32)
        v_op = torch.empty_like(opacity) if (opacity is not None and ctx.needs_input_grad[0]) else None
        v_dp = torch.empty_like(depth) if (depth is not None and ctx.needs_input_grad[1]) else None
        v_rgb = torch.empty_like(rgb) if (rgb is not None and dyn_opacity is not None and ctx.needs_input_grad[2]) else None
        L.check(L.lib().bds_reg_loss_bwd(H, W, L.ptr(opacity), L.ptr(depth), L.ptr(pixels), L.ptr(rgb), L.ptr(dyn_opacity), L.ptr(egocar),
                                         ctx.dyn_threshold, L.ptr(sums), L.ptr(v_terms), L.ptr(v_op), L.ptr(v_dp), L.ptr(v_rgb), L.stream()),
                "bds_reg_loss_bwd")
        return v_op, v_dp, v_rgb, None, None, None, None


def reg_losses(pixels: Tensor, opacity: Tensor = None, depth: Tensor = None, rgb: Tensor = None, dyn_opacity: Tensor = None,
               egocar_masks: Tensor = None, dyn_threshold: float = 0.2) -> Tensor:
    """The regularisers of ``BasicTrainer.compute_losses`` (models/trainers/base.py:566-585, 638-659) in one pass each way; returns the
    three UNWEIGHTED terms ``[opacity_entropy, inverse_depth_smoothness, dynamic_region_l1]`` (the trainer multiplies them by
    ``losses.opacity_entropy.w`` / ``losses.inverse_depth_smoothness.w`` / ``losses.dynamic_region.w``).  pixels / rgb [H,W,3];
    opacity, depth, dyn_opacity (``outputs["Dynamic_opacity"]``, treated as data like the reference's ``.data``) [H,W,1] or [H,W].
    Pass ``None`` to drop a term (it is then 0).  The dynamic-region term is 0 when no pixel qualifies (the reference adds none)."""
    return _RegLoss.apply(opacity, depth, rgb, pixels, None if dyn_opacity is None else dyn_opacity.detach(), egocar_masks,
                          float(dyn_threshold))
