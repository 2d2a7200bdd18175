"""Host-side mirror of the reference's bilateral colour-transform modules
(/root/reference/project/models/modules.py:275-351 BilateralAffineTransform, :422-593
MultiScaleBilateralAffineTransform): same constructor arguments, attribute / parameter names
(``bil_grids`` / ``bil_grids{i}`` with ``.grids`` and ``.rgb2gray_weight``, module buffer
``rgb2gray_weight``) so that reference checkpoints load, same ``forward`` return values.

Two ways to use them:
  * ``forward(rgb, image_infos)`` -- the reference API: returns the per-pixel 3x4 maps
    ([1,H,W,3,4] or a list of them), differentiable, for the trainer's own matmul composition
    (trainers/scene_graph.py:95-98,112-117);
  * ``transform(rgb, image_infos, alpha=None, sky=None)`` -- the fused fast path: one call that
    slices, up-samples and applies all levels (optionally with the clamp + sky blend in front)
    without materialising the maps.  Numerically equal to forward + composition.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from .bilagrid import BilateralGrid, bilagrid_transform, slice, total_variation_loss


def _img_index(image_infos) -> int:
    # The reference reads int(image_infos["img_idx"][0][0]) (modules.py:319,507): a D2H sync per step when the
    # index lives on the GPU.  A host int / 0-d CPU tensor is accepted too and avoids it.
    v = image_infos["img_idx"]
    if torch.is_tensor(v):
        while v.dim() > 0:
            v = v[0]
        return int(v)
    return int(v)


def fill_matrix_res(H: int, W: int, matrix: Tensor) -> Tensor:
    """modules.py:409-420: bilinear up-sample of the 12 affine channels to (H, W)."""
    B, Hm, Wm, _, _ = matrix.shape
    if Hm == H and Wm == W:
        return matrix
    flat = matrix.flatten(start_dim=-2)
    return F.interpolate(flat.permute(0, 3, 1, 2), size=(H, W), mode="bilinear", align_corners=False).permute(
        0, 2, 3, 1).reshape(B, H, W, 3, 4)


class BilateralAffineTransform(nn.Module):
    def __init__(self, class_name, n, grid_X, grid_Y, grid_W, device="cuda"):
        super().__init__()
        self.bil_grids = BilateralGrid(num=n, grid_X=grid_X, grid_Y=grid_Y, grid_W=grid_W)
        self.register_buffer("rgb2gray_weight", torch.tensor([0.299, 0.587, 0.114], dtype=torch.float32))
        self.class_prefix = class_name + "#"
        self.device = device
        self.in_test_set = False
        self.training_indices_for_test: Dict[int, List[int]] = {}
        self.to(device)

    def tv_loss(self):
        return total_variation_loss(self.bil_grids.grids)

    def _grids_for(self, image_infos) -> Tensor:
        k = _img_index(image_infos)
        if not self.in_test_set:
            return self.bil_grids.grids[k:k + 1]
        near = self.training_indices_for_test[k]
        return self.bil_grids.grids[torch.as_tensor(near, device=self.bil_grids.grids.device)]

    def forward(self, rgb: Tensor, image_infos) -> Tensor:
        """Reference API: the per-pixel 3x4 maps [1,H,W,3,4] (differentiable when grad mode is on)."""
        assert "img_idx" in image_infos
        if not torch.is_grad_enabled():  # evaluation: the fused kernels also emit the maps
            _, maps = bilagrid_transform(rgb, [self._grids_for(image_infos)], [1], return_maps=True)
            return maps[0][None]
        H, W, _ = rgb.shape
        gy, gx = torch.meshgrid(torch.linspace(0, 1.0, H, device=rgb.device), torch.linspace(0, 1.0, W, device=rgb.device),
                                indexing="ij")
        xy = torch.stack([gx, gy], dim=-1).unsqueeze(0)
        k = _img_index(image_infos)
        idxs = [k] if not self.in_test_set else self.training_indices_for_test[k]
        out = 0
        for i in idxs:
            out = out + slice(self.bil_grids, xy, rgb.unsqueeze(0), torch.tensor(i, device=rgb.device, dtype=torch.long))[
                "rgb_affine_mats"]
        return out / len(idxs) if len(idxs) > 1 else out

    def transform(self, rgb: Tensor, image_infos, alpha: Optional[Tensor] = None, sky: Optional[Tensor] = None) -> Tensor:
        return bilagrid_transform(rgb, [self._grids_for(image_infos)], [1], alpha=alpha, sky=sky)

    def get_param_groups(self):
        return {self.class_prefix + "all": self.bil_grids.parameters()}


class MultiScaleBilateralAffineTransform(nn.Module):
    def __init__(self, class_name, n, grid, device="cuda"):
        super().__init__()
        self.grid_size = grid
        self.tv_weight = []
        for i, (gx, gy, gl) in enumerate(grid):
            setattr(self, f"bil_grids{i}", BilateralGrid(num=n, grid_X=gx, grid_Y=gy, grid_W=gl))
            self.tv_weight.append(0.5 * (gx * gy * gl) ** 0.5)
        self.register_buffer("rgb2gray_weight", torch.tensor([0.299, 0.587, 0.114], dtype=torch.float32))
        self.class_prefix = class_name + "#"
        self.device = device
        self.in_test_set = False
        self.training_indices_for_test: Dict[int, List[int]] = {}
        self.save_matrix = None
        self.to(device)

    def tv_loss(self):
        loss = 0
        for i in range(len(self.grid_size)):
            loss = loss + total_variation_loss(getattr(self, f"bil_grids{i}").grids, self.tv_weight[i])
        return loss

    def _grids_for(self, image_infos) -> List[Tensor]:
        k = _img_index(image_infos)
        out = []
        for i in range(len(self.grid_size)):
            g = getattr(self, f"bil_grids{i}").grids
            if not self.in_test_set:
                out.append(g[k:k + 1])
            else:
                near = self.training_indices_for_test[k]
                out.append(g[torch.as_tensor(near, device=g.device)])
        return out

    def get_sample_grid(self, factor, H, W, rgb):
        Hd, Wd = H // factor, W // factor
        lo = F.interpolate(rgb.permute(2, 0, 1).unsqueeze(0), size=(Hd, Wd), mode="bilinear", align_corners=False).permute(
            0, 2, 3, 1).squeeze(0)
        gy, gx = torch.meshgrid(torch.linspace(0, 1.0, Hd, device=rgb.device), torch.linspace(0, 1.0, Wd, device=rgb.device),
                                indexing="ij")
        return torch.stack([gx, gy], dim=-1).unsqueeze(0), lo

    def forward(self, rgb: Tensor, image_infos, guidance_factor: Sequence[int] = (4, 4, 2)) -> List[Tensor]:
        """Reference API: list of per-level maps [1,H,W,3,4] (differentiable)."""
        assert "img_idx" in image_infos
        if guidance_factor is None:
            guidance_factor = [1] * len(self.grid_size)
        if len(guidance_factor) < len(self.grid_size):
            raise IndexError("guidance_factor has fewer entries than grid levels "
                             "(the reference's default [4,4,2] covers 3 levels, modules.py:505)")
        H, W, _ = rgb.shape
        if not torch.is_grad_enabled():
            _, maps = bilagrid_transform(rgb, self._grids_for(image_infos), list(guidance_factor)[:len(self.grid_size)],
                                         return_maps=True)
            out_list = [m[None] for m in maps]
        else:
            k = _img_index(image_infos)
            idxs = [k] if not self.in_test_set else self.training_indices_for_test[k]
            out_list = []
            for i in range(len(self.grid_size)):
                xy, lo = self.get_sample_grid(guidance_factor[i], H, W, rgb)
                acc = 0
                for j in idxs:
                    acc = acc + slice(getattr(self, f"bil_grids{i}"), xy, lo.unsqueeze(0),
                                      torch.tensor(j, device=rgb.device, dtype=torch.long))["rgb_affine_mats"]
                if len(idxs) > 1:
                    acc = acc / len(idxs)
                out_list.append(fill_matrix_res(H, W, acc))
        self.save_matrix = out_list
        return out_list

    def transform(self, rgb: Tensor, image_infos, guidance_factor: Sequence[int] = (4, 4, 2), alpha: Optional[Tensor] = None,
                  sky: Optional[Tensor] = None) -> Tensor:
        """Fused fast path: equals the trainer's composition of forward()'s maps (scene_graph.py:112-117)."""
        if len(guidance_factor) < len(self.grid_size):
            raise IndexError("guidance_factor has fewer entries than grid levels")
        return bilagrid_transform(rgb, self._grids_for(image_infos), list(guidance_factor)[:len(self.grid_size)], alpha=alpha,
                                  sky=sky)

    def get_param_groups(self):
        return {f"{self.class_prefix}grid{i}": getattr(self, f"bil_grids{i}").parameters() for i in range(len(self.grid_size))}
