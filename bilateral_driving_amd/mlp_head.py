"""Fused head of the neural bilateral variants: sliced features -> 3-layer tanh MLP -> 3x4 map (-> applied to the pixel), one HIP
kernel each way on the FP32 matrix cores (csrc/mlp_head.hip through ``bds_mlp_head_fwd / _bwd``).

Reference: ``affine_network`` of NeuralBilateralAffineTransform / MultiScaleNeuralBilateralAffineTransform
(/root/reference/project/models/modules.py:621-627, 700-706 -- three bias-free ``nn.Linear`` with ``nn.Tanh`` between them) and the
trainer's application ``A[..., :3] @ rgb + A[..., 3] + rgb`` (models/trainers/scene_graph.py:99-106)."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _lib as L

HIDDEN = 64
FEATURE_WIDTHS = (8, 16, 24, 32)


def supported(in_dim: int, hidden_dim: int) -> bool:
    """The sizes the kernel is built for (every shipped config: hidden 64, feature_dim x levels in {16, 24})."""
    return hidden_dim == HIDDEN and in_dim in FEATURE_WIDTHS


class _MlpHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats: Tensor, rgb: Optional[Tensor], w1: Tensor, w2: Tensor, w3: Tensor, residual: bool, want_out: bool,
                want_affine: bool):
        L.require_gpu(feats, rgb, w1, w2, w3)
        P, F = feats.shape
        assert supported(F, w1.shape[0]) and w1.shape == (HIDDEN, F) and w2.shape == (HIDDEN, HIDDEN) and w3.shape == (12, HIDDEN)
        assert want_out or want_affine
        feats_c = feats.detach().contiguous().float()
        rgb_c = None if rgb is None else rgb.detach().contiguous().float()
        assert not want_out or (rgb_c is not None and rgb_c.shape == (P, 3))
        ws = [w.detach().contiguous().float() for w in (w1, w2, w3)]
        out = torch.empty(P, 3, device=feats.device) if want_out else None
        aff = torch.empty(P, 12, device=feats.device) if want_affine else None
        L.check(L.lib().bds_mlp_head_fwd(P, F, HIDDEN, L.ptr(feats_c), L.ptr(rgb_c), L.ptr(ws[0]), L.ptr(ws[1]), L.ptr(ws[2]),
                                         int(residual), L.ptr(out), L.ptr(aff), L.stream()), "bds_mlp_head_fwd")
        ctx.save_for_backward(feats_c, rgb_c if rgb_c is not None else feats_c.new_empty(0), *ws)
        ctx.cfg = (bool(residual), rgb_c is not None)
        return out, aff

    @staticmethod
    def backward(ctx, v_out, v_aff):
        feats, rgb, w1, w2, w3 = ctx.saved_tensors
        residual, has_rgb = ctx.cfg
        rgb = rgb if has_rgb else None
        P, F = feats.shape
        dev = feats.device
        need = ctx.needs_input_grad
        if v_out is None and v_aff is None:
            return (None,) * 8
        v_out = None if v_out is None else v_out.contiguous().float()
        v_aff = None if v_aff is None else v_aff.contiguous().float()
        v_feats = torch.empty(P, F, device=dev) if need[0] else None
        v_rgb = torch.empty(P, 3, device=dev) if (need[1] and v_out is not None) else None
        v_w = [torch.empty_like(w) if need[2 + i] else None for i, w in enumerate((w1, w2, w3))]
        nb = int(L.lib().bds_mlp_head_bwd_temp_bytes(P, F))
        temp = torch.empty(max(nb, 4), dtype=torch.uint8, device=dev)
        L.check(L.lib().bds_mlp_head_bwd(P, F, HIDDEN, L.ptr(feats), L.ptr(rgb), L.ptr(w1), L.ptr(w2), L.ptr(w3), int(residual),
                                         L.ptr(v_out), L.ptr(v_aff), L.ptr(v_feats), L.ptr(v_rgb), L.ptr(v_w[0]), L.ptr(v_w[1]),
                                         L.ptr(v_w[2]), 0, L.ptr(temp), nb, L.stream()), "bds_mlp_head_bwd")
        if need[1] and v_rgb is None and has_rgb:
            v_rgb = torch.zeros(P, 3, device=dev)
        return v_feats, v_rgb, v_w[0], v_w[1], v_w[2], None, None, None


def affine_maps(feats: Tensor, w1: Tensor, w2: Tensor, w3: Tensor) -> Tensor:
    """[..., F] features -> [..., 12] network outputs (the modules' ``forward``)."""
    lead = feats.shape[:-1]
    _, aff = _MlpHead.apply(feats.reshape(-1, feats.shape[-1]), None, w1, w2, w3, False, False, True)
    return aff.reshape(*lead, 12)


def transform(feats: Tensor, rgb: Tensor, w1: Tensor, w2: Tensor, w3: Tensor, residual: bool = True) -> Tensor:
    """[..., F] features + [..., 3] colours -> the transformed colours; the 3x4 maps never reach memory."""
    out, _ = _MlpHead.apply(feats.reshape(-1, feats.shape[-1]), rgb.reshape(-1, 3), w1, w2, w3, residual, True, False)
    return out.reshape(rgb.shape)


def transform_and_maps(feats: Tensor, rgb: Tensor, w1: Tensor, w2: Tensor, w3: Tensor, residual: bool = True) -> Tuple[Tensor, Tensor]:
    out, aff = _MlpHead.apply(feats.reshape(-1, feats.shape[-1]), rgb.reshape(-1, 3), w1, w2, w3, residual, True, True)
    return out.reshape(rgb.shape), aff.reshape(*rgb.shape[:-1], 12)


# ---- the whole transform of one image, the feature slice folded in -----------------------------------------------------------------------
def _levels_struct(grids: Sequence[Tensor], v_grids=None):
    arr = (L.BdsFeatLevel * len(grids))()
    for i, g in enumerate(grids):
        nch, gl, gy, gx = g.shape
        arr[i].grid = g.data_ptr()
        arr[i].v_grid = None if v_grids is None or v_grids[i] is None else v_grids[i].data_ptr()
        arr[i].gx, arr[i].gy, arr[i].gl, arr[i].nch = gx, gy, gl, nch
    return arr


def image_supported(H: int, W: int, grids: Sequence[Tensor], hidden_dim: int) -> bool:
    """Shapes ``bds_neural_image_*`` is built for (the shipped configs: one 16x16x8 grid with 24 features, or the grids
    [[1,1,1],[16,16,8]] with 8 features each; hidden 64)."""
    if not grids or len(grids) > 2 or not all(g.is_cuda and g.dim() == 4 for g in grids):
        return False
    return bool(L.lib().bds_neural_image_ok(H, W, len(grids), _levels_struct(grids), hidden_dim))


class _NeuralImage(torch.autograd.Function):
    """rgb [H,W,3], w1, w2, w3, one grid [nch,gl,gy,gx] per level -> rgb' [H,W,3] (slice + head + application in one kernel)."""

    @staticmethod
    def forward(ctx, rgb: Tensor, w1: Tensor, w2: Tensor, w3: Tensor, residual: bool, *grids: Tensor):
        L.require_gpu(rgb, w1, w2, w3, *grids)
        H, W, _ = rgb.shape
        rgb_c = rgb.detach().contiguous().float()
        ws = [w.detach().contiguous().float() for w in (w1, w2, w3)]
        gs = [g.detach().contiguous().float() for g in grids]
        out = torch.empty(H, W, 3, device=rgb.device)
        L.check(L.lib().bds_neural_image_fwd(H, W, len(gs), _levels_struct(gs), HIDDEN, L.ptr(rgb_c), L.ptr(ws[0]), L.ptr(ws[1]), L.ptr(ws[2]),
                                             int(residual), L.ptr(out), L.stream()), "bds_neural_image_fwd")
        ctx.save_for_backward(rgb_c, *ws, *gs)
        ctx.residual = bool(residual)
        return out

    @staticmethod
    def backward(ctx, v_out):
        rgb, w1, w2, w3, *gs = ctx.saved_tensors
        H, W, _ = rgb.shape
        dev = rgb.device
        need = ctx.needs_input_grad
        v_out = v_out.contiguous().float()
        v_rgb = torch.empty_like(rgb) if need[0] else None
        v_w = [torch.empty_like(w) if need[1 + i] else None for i, w in enumerate((w1, w2, w3))]
        v_g = [torch.zeros_like(g) if need[5 + i] else None for i, g in enumerate(gs)]
        F_ = sum(g.shape[0] for g in gs)
        nb = int(L.lib().bds_neural_image_bwd_temp_bytes(F_))
        temp = torch.empty(nb, dtype=torch.uint8, device=dev)
        L.check(L.lib().bds_neural_image_bwd(H, W, len(gs), _levels_struct(gs, v_g), HIDDEN, L.ptr(rgb), L.ptr(w1), L.ptr(w2), L.ptr(w3),
                                             int(ctx.residual), L.ptr(v_out), L.ptr(v_rgb), L.ptr(v_w[0]), L.ptr(v_w[1]), L.ptr(v_w[2]), 0,
                                             L.ptr(temp), nb, L.stream()), "bds_neural_image_bwd")
        return (v_rgb, v_w[0], v_w[1], v_w[2], None, *v_g)


def image_transform(rgb: Tensor, grids: Sequence[Tensor], w1: Tensor, w2: Tensor, w3: Tensor, residual: bool = True) -> Tensor:
    """The neural variants' whole ``transform`` of one image: per level the feature grid of that image [nch,gl,gy,gx] (for the
    test branch: the mean of the neighbour images' grids -- the slice is linear in the grid), sliced at the pixel grid, through
    the head and applied.  ``image_supported`` tells whether the shapes are the kernel's."""
    return _NeuralImage.apply(rgb, w1, w2, w3, residual, *grids)
