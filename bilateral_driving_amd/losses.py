"""Training-step loss on the path's outputs: photometric L1 + TV regulariser of the bilateral grids, as ONE autograd
node whose terms accumulate into one device scalar (reference: models/trainers/base.py:518-529 rgb L1,
:590-594 + models/modules.py:445,466-472 affine TV with per-level weights).  SURVEY.md 8f rank 1 (first slice: L1)."""
from __future__ import annotations

from typing import Sequence

import torch
from torch import Tensor

from . import _lib as L


class _PhotometricTV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb: Tensor, target: Tensor, tv_weights: tuple, *grids: Tensor):
        L.require_gpu(rgb, target, *grids)
        lib, st = L.lib(), L.stream()
        rgb, target = rgb.contiguous(), target.contiguous()
        assert rgb.shape == target.shape and rgb.dtype == torch.float32 and target.dtype == torch.float32
        grids = [g.contiguous() for g in grids]
        out = torch.zeros(1, device=rgb.device, dtype=torch.float32)
        L.check(lib.bds_l1_mean_fwd(rgb.numel(), L.ptr(rgb), L.ptr(target), L.ptr(out), st), "bds_l1_mean_fwd")
        for g, w in zip(grids, tv_weights):
            n, c, gl, gy, gx = g.shape
            assert c == 12
            L.check(lib.bds_bilagrid_tv_fwd(n, gx, gy, gl, L.ptr(g), float(w), L.ptr(out), st), "bds_bilagrid_tv_fwd")
        ctx.save_for_backward(rgb, target, *grids)
        ctx.tv_weights = tuple(float(w) for w in tv_weights)
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
        need = [ctx.needs_input_grad[3 + i] for i in range(len(grids))]
        if any(need):
            sizes = [(g.numel() + 3) // 4 * 4 if need[i] else 0 for i, g in enumerate(grids)]   # 16-byte aligned slices
            flat = torch.zeros(sum(sizes), device=rgb.device, dtype=torch.float32)               # one fill for all levels
            off = 0
            for i, g in enumerate(grids):
                if not need[i]:
                    continue
                vg = flat[off:off + g.numel()].view(g.shape)
                off += sizes[i]
                n, _, gl, gy, gx = g.shape
                L.check(lib.bds_bilagrid_tv_bwd(n, gx, gy, gl, L.ptr(g), ctx.tv_weights[i], L.ptr(v), L.ptr(vg), st),
                        "bds_bilagrid_tv_bwd")
                v_grids[i] = vg
        return (v_rgb, None, None, *v_grids)


def photometric_tv_loss(rgb: Tensor, target: Tensor, grids: Sequence[Tensor], tv_weights: Sequence[float]) -> Tensor:
    """mean|rgb - target| + sum_l tv_weights[l] * total_variation(grids[l])  (grids [n_img,12,L,gy,gx])."""
    assert len(grids) == len(tv_weights)
    return _PhotometricTV.apply(rgb, target, tuple(tv_weights), *grids)


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
