"""Cube-map sky on MI355X: ``cubemap_sample`` (autograd operator over ``bds_cubemap_fwd/bwd``) and ``EnvLight``, a mirror of the
reference's sky model (/root/reference/project/models/modules.py:176-211: same constructor arguments, parameter name ``base``
[6,res,res,3] initialised to 0.5, ``forward(image_infos)`` reading ``image_infos["viewdirs"]``, ``get_param_groups``) that does
not need nvdiffrast -- a CUDA-only package, which is why the reference cannot evaluate its own sky model on ROCm."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import _lib as L


class _CubemapSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tex: Tensor, dirs: Tensor, rot: Optional[Tensor]):
        L.require_gpu(tex, dirs, rot)
        assert tex.dim() == 4 and tex.shape[0] == 6 and tex.shape[1] == tex.shape[2], "tex is [6,res,res,C]"
        assert dirs.shape[-1] == 3
        res, C = tex.shape[1], tex.shape[3]
        texc = tex.detach().contiguous().float()
        d = dirs.detach().reshape(-1, 3).contiguous().float()
        r = None if rot is None else rot.detach().reshape(9).contiguous().float()
        out = torch.empty(d.shape[0], C, device=tex.device, dtype=torch.float32)
        with L.timed("cubemap_fwd"):
            L.check(L.lib().bds_cubemap_fwd(d.shape[0], res, C, L.ptr(d), L.ptr(r), L.ptr(texc), L.ptr(out), L.stream()), "bds_cubemap_fwd")
        ctx.save_for_backward(d, r)
        ctx.tex_shape = tuple(tex.shape)
        # an image of directions [..., H, W, 3] with at least a few tile rows: the backward pre-sums 16x16 pixel tiles in LDS
        ctx.width = int(dirs.shape[-2]) if dirs.dim() >= 3 and dirs.shape[-2] >= 16 and d.shape[0] // dirs.shape[-2] >= 16 else 0
        return out.reshape(tuple(dirs.shape[:-1]) + (C,))

    @staticmethod
    def backward(ctx, v_out: Tensor):
        d, r = ctx.saved_tensors
        _, res, _, C = ctx.tex_shape
        v_tex = torch.zeros(ctx.tex_shape, device=v_out.device, dtype=torch.float32)
        vo = v_out.reshape(-1, C).contiguous().float()
        with L.timed("cubemap_bwd"):
            L.check(L.lib().bds_cubemap_bwd(d.shape[0], res, C, ctx.width, L.ptr(d), L.ptr(r), L.ptr(vo), L.ptr(v_tex), L.stream()), "bds_cubemap_bwd")
        return v_tex, None, None


def cubemap_sample(tex: Tensor, dirs: Tensor, rot: Optional[Tensor] = None) -> Tensor:
    """Bilinear cube-map lookup: tex [6,res,res,C] along dirs [...,3] (optionally dirs @ rot^T first) -> [...,C].
    Differentiable with respect to ``tex``; directions are data (as the reference's viewdirs)."""
    return _CubemapSample.apply(tex, dirs, rot)


class EnvLight(torch.nn.Module):
    def __init__(self, class_name: str, resolution: int = 1024, device: torch.device = torch.device("cuda"), **kwargs):
        super().__init__()
        self.class_prefix = class_name + "#"
        self.device = device
        self.register_buffer("to_opengl", torch.tensor([[1, 0, 0], [0, 0, 1], [0, -1, 0]], dtype=torch.float32), persistent=False)
        self.base = torch.nn.Parameter(0.5 * torch.ones(6, resolution, resolution, 3))
        self.to(device)

    def forward(self, image_infos) -> Tensor:
        l = image_infos["viewdirs"]
        return cubemap_sample(self.base, l, self.to_opengl)     # the frame change is folded into the lookup (modules.py:196)

    def get_param_groups(self):
        return {self.class_prefix + "all": self.parameters()}
