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

The neural variants (modules.py:595-820) are mirrored too: the feature-grid slice runs on the HIP
kernels (``bds_bilagrid_slice_feat_*``), the three bias-free Linear layers of ``affine_network`` stay ``nn.Linear`` --
[pixels x F] x [F x 64] GEMMs are library (hipBLASLt) work, not something to re-tile by hand.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from . import mlp_head
from .bilagrid import (BilateralGrid, NeuralBilateralGrid, bilagrid_transform, slice, slice_feature, slice_feature_image,
                       total_variation_loss)


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

    def inverse_loss(self, gt: Tensor, render: Tensor) -> Tensor:
        """modules.py:474-492 (weighted by ``losses.affine.w1`` at trainers/base.py:630-632): mean |T^-1(gt) - render| with T the
        composition of the per-pixel 3x4 maps of the LAST ``forward()`` call (``save_matrix``).  The reference inverts one 4x4
        homogeneous matrix per pixel with ``torch.inverse``; the same inverse is formed here in closed form (3x3 adjugate of the
        composed linear part, shift = -inv . shift): identical mathematics, no batched LU over H*W matrices."""
        assert self.save_matrix is not None, "inverse_loss uses the maps of the preceding forward() call"
        lin, shift = None, None
        for arr in self.save_matrix:                       # T <- A_l o T   (hom_arr @ mat, modules.py:479-481)
            A, t = arr[..., :3, :3], arr[..., :3, 3]
            if lin is None:
                lin, shift = A, t
            else:
                lin, shift = A @ lin, (A @ shift[..., None])[..., 0] + t
        a, b, c = lin[..., 0, 0], lin[..., 0, 1], lin[..., 0, 2]
        d, e, f = lin[..., 1, 0], lin[..., 1, 1], lin[..., 1, 2]
        g, h, i = lin[..., 2, 0], lin[..., 2, 1], lin[..., 2, 2]
        c00, c01, c02 = e * i - f * h, c * h - b * i, b * f - c * e
        c10, c11, c12 = f * g - d * i, a * i - c * g, c * d - a * f
        c20, c21, c22 = d * h - e * g, b * g - a * h, a * e - b * d
        det = a * c00 + b * c10 + c * c20
        inv = torch.stack([torch.stack([c00, c01, c02], -1), torch.stack([c10, c11, c12], -1), torch.stack([c20, c21, c22], -1)], -2) \
            / det[..., None, None]
        inv = inv.reshape(gt.shape[0], gt.shape[1], 3, 3)
        inv_shift = -(inv @ shift.reshape(gt.shape[0], gt.shape[1], 3, 1))[..., 0]
        gt_transformed = (inv @ gt[..., None])[..., 0] + inv_shift
        return torch.abs(gt_transformed - render).mean()

    def transform(self, rgb: Tensor, image_infos, guidance_factor: Sequence[int] = (4, 4, 2), alpha: Optional[Tensor] = None,
                  sky: Optional[Tensor] = None) -> Tensor:
        """Fused fast path: equals the trainer's composition of forward()'s maps (scene_graph.py:112-117)."""
        if len(guidance_factor) < len(self.grid_size):
            raise IndexError("guidance_factor has fewer entries than grid levels")
        return bilagrid_transform(rgb, self._grids_for(image_infos), list(guidance_factor)[:len(self.grid_size)], alpha=alpha,
                                  sky=sky)

    def get_param_groups(self):
        return {f"{self.class_prefix}grid{i}": getattr(self, f"bil_grids{i}").parameters() for i in range(len(self.grid_size))}


class AffineTransform(nn.Module):
    """The appearance-code baseline the bilateral grids are compared with (/root/reference/project/models/modules.py:213-267;
    configs/omnire.yaml:245-249): one learnable code per image -> 2-layer ReLU MLP -> one 3x4 colour matrix (+ identity) for the whole
    image.  Same constructor, parameter names (``embedding.weight``, ``decoder.0/2.weight/bias``: reference checkpoints load) and
    ``forward(image_infos)`` return value (per-pixel [*, 3, 4] maps) as the reference.

    ``transform(rgb, image_infos, alpha, sky)`` is the fused fast path for ``pixel_affine=False``: the decoder runs ONCE for the
    image's code (the reference pushes H*W identical rows through it), and the clamp + sky blend + 3x4 application run in the
    bilateral transform's kernels with the matrix as a 1x1x1 grid -- exactly constant under slicing / up-sampling."""

    def __init__(self, class_name: str, n: int, embedding_dim: int = 4, pixel_affine: bool = False, base_mlp_layer_width: int = 64,
                 device="cuda"):
        super().__init__()
        self.class_prefix, self.device = class_name + "#", device
        self.embedding_dim, self.pixel_affine, self.in_test_set = embedding_dim, pixel_affine, False
        # state-dict names fixed by the reference's checkpoints: embedding.weight [n, dim], decoder.0 / decoder.2 (Linear, ReLU, Linear)
        self.embedding = nn.Embedding(n, embedding_dim, dtype=torch.float32)
        code_width = embedding_dim + (2 if pixel_affine else 0)         # per-pixel variant: the pixel coordinates ride along
        self.decoder = nn.Sequential(nn.Linear(code_width, base_mlp_layer_width), nn.ReLU(), nn.Linear(base_mlp_layer_width, 12))
        self.zero_init()
        self.to(device)

    def zero_init(self):
        """Everything zero: the decoder outputs 0 and the transform starts as the identity."""
        with torch.no_grad():
            for t in self.parameters():
                t.zero_()

    def _codes(self, image_infos, lead_shape=None) -> Tensor:
        """Appearance code per element: the image's own, or -- held-out image / no index -- the mean over all images."""
        if "img_idx" in image_infos and not self.in_test_set:
            return self.embedding(image_infos["img_idx"])
        mean = self.embedding.weight.mean(dim=0)
        shape = image_infos["viewdirs"].shape[:-1] if lead_shape is None else lead_shape
        return mean.expand(*shape, self.embedding_dim)

    def forward(self, image_infos) -> Tensor:
        """[*, 3, 4] per element: decoder(code [, pixel coordinate]) reshaped, plus the identity on the 3x3 part."""
        x = self._codes(image_infos)
        if self.pixel_affine:
            x = torch.cat((x, image_infos["pixel_coords"]), dim=-1)
        ident = torch.eye(3, 4, device=x.device, dtype=x.dtype)
        return self.decoder(x).unflatten(-1, (3, 4)) + ident

    def image_matrix(self, image_infos) -> Tensor:
        """[12] the image's colour matrix (row-major 3x4), decoder evaluated once."""
        assert not self.pixel_affine, "per-pixel inputs: use forward()"
        if "img_idx" in image_infos and not self.in_test_set:
            code = self.embedding.weight[_img_index(image_infos)]
        else:
            code = self.embedding.weight.mean(dim=0)
        return self.decoder(code) + torch.eye(3, 4, device=code.device).reshape(12)

    def transform(self, rgb: Tensor, image_infos, alpha: Optional[Tensor] = None, sky: Optional[Tensor] = None) -> Tensor:
        if self.pixel_affine:
            A = self.forward(image_infos)
            blended = rgb if sky is None else torch.clamp(rgb, max=1.0) + sky * (1.0 - alpha)
            return (A[..., :3, :3] @ blended[..., None] + A[..., :3, 3:])[..., 0]
        grid = self.image_matrix(image_infos).reshape(1, 12, 1, 1, 1)
        return bilagrid_transform(rgb, [grid], [1], alpha=alpha, sky=sky)

    def get_param_groups(self):
        return {self.class_prefix + "all": self.parameters()}


def _affine_network(in_dim: int, hidden_dim: int) -> nn.Sequential:
    return nn.Sequential(nn.Linear(in_dim, hidden_dim, bias=False), nn.Tanh(), nn.Linear(hidden_dim, hidden_dim, bias=False), nn.Tanh(),
                         nn.Linear(hidden_dim, 12, bias=False))


def _head_maps(net: nn.Sequential, feats: Tensor, H: int, W: int) -> Tensor:
    """``affine_network(feats).reshape(1, H, W, 3, 4)``: one fused kernel when the sizes are the kernel's (mlp_head.supported),
    the framework's Linear / Tanh modules otherwise."""
    w1, w2, w3 = net[0].weight, net[2].weight, net[4].weight
    if feats.is_cuda and mlp_head.supported(w1.shape[1], w1.shape[0]):
        return mlp_head.affine_maps(feats, w1, w2, w3).reshape(1, H, W, 3, 4)
    return net(feats).reshape(1, H, W, 3, 4)


def _head_transform(net: nn.Sequential, feats: Tensor, rgb: Tensor) -> Tensor:
    """The trainer's application with the residual (scene_graph.py:99-106) fused behind the network."""
    w1, w2, w3 = net[0].weight, net[2].weight, net[4].weight
    if feats.is_cuda and mlp_head.supported(w1.shape[1], w1.shape[0]):
        return mlp_head.transform(feats.reshape(rgb.shape[:-1] + (feats.shape[-1],)), rgb, w1, w2, w3, residual=True)
    A = net(feats).reshape(*rgb.shape[:-1], 3, 4)
    return (A[..., :3] @ rgb[..., None])[..., 0] + A[..., 3] + rgb


def _image_grids(levels: Sequence[NeuralBilateralGrid], idxs: Sequence[int]) -> List[Tensor]:
    """One grid per level for this image: the image's own, or the mean over the neighbour images' grids in the test branch
    (modules.py:651-662 averages the sliced features; the slice is linear in the grid, so the grids can be averaged instead)."""
    if len(idxs) == 1:
        return [g.grids[int(idxs[0])] for g in levels]          # a view: no copy
    return [g.grids[list(idxs)].mean(dim=0) for g in levels]


def _fused_image_ok(net: nn.Sequential, levels: Sequence[NeuralBilateralGrid], rgb: Tensor) -> bool:
    H, W, _ = rgb.shape
    return rgb.is_cuda and mlp_head.image_supported(H, W, [g.grids[0] for g in levels], net[0].weight.shape[0])


def _sliced_features(grids: NeuralBilateralGrid, rgb: Tensor, xy: Optional[Tensor], idxs: Sequence[int]) -> Tensor:
    """Feature slice for one image, or the mean over the neighbour images' grids in the test branch (modules.py:651-662).
    ``xy`` None: the pixel grid of ``rgb`` itself (every call site of the reference) -> the image form of the slice."""
    acc = None
    for i in idxs:
        if xy is None:
            f = slice_feature_image(grids, rgb, i)
        else:
            f = slice_feature(grids, xy, rgb.unsqueeze(0), torch.tensor(i, device=rgb.device, dtype=torch.long))["affine_features"]
        acc = f if acc is None else acc + f
    return acc / len(idxs) if len(idxs) > 1 else acc


class NeuralBilateralAffineTransform(nn.Module):
    """modules.py:595-670: one feature grid per image -> per-pixel features -> 3-layer tanh MLP -> 3x4 map; the trainer adds the
    input back (trainers/scene_graph.py:99-102): ``rgb' = A rgb + b + rgb``."""

    def __init__(self, class_name, n, grid_X, grid_Y, grid_W, feature_dim, hidden_dim, device="cuda"):
        super().__init__()
        self.bil_grids = NeuralBilateralGrid(num=n, grid_X=grid_X, grid_Y=grid_Y, grid_W=grid_W, feature_dim=feature_dim)
        self.register_buffer("rgb2gray_weight", torch.tensor([0.299, 0.587, 0.114], dtype=torch.float32))
        self.affine_network = _affine_network(feature_dim, hidden_dim)
        self.feature_dim = feature_dim
        self.class_prefix = class_name + "#"
        self.device = device
        self.in_test_set = False
        self.training_indices_for_test: Dict[int, List[int]] = {}
        self.to(device)

    def tv_loss(self):
        return total_variation_loss(self.bil_grids.grids)

    def forward(self, rgb: Tensor, image_infos) -> Tensor:
        assert "img_idx" in image_infos
        H, W, _ = rgb.shape
        return _head_maps(self.affine_network, self._features(rgb, image_infos), H, W)

    def _features(self, rgb: Tensor, image_infos) -> Tensor:
        k = _img_index(image_infos)
        H, W, _ = rgb.shape
        idxs = [k] if not self.in_test_set else self.training_indices_for_test[k]
        return _sliced_features(self.bil_grids, rgb, None, idxs)

    def transform(self, rgb: Tensor, image_infos) -> Tensor:
        """forward + the trainer's application with the residual (scene_graph.py:99-102) in one pass: the [H,W,3,4] maps are not
        materialised."""
        assert "img_idx" in image_infos
        if _fused_image_ok(self.affine_network, [self.bil_grids], rgb):      # slice + head + application: one kernel each way
            k = _img_index(image_infos)
            idxs = [k] if not self.in_test_set else self.training_indices_for_test[k]
            net = self.affine_network
            return mlp_head.image_transform(rgb, _image_grids([self.bil_grids], idxs), net[0].weight, net[2].weight, net[4].weight)
        return _head_transform(self.affine_network, self._features(rgb, image_infos), rgb)

    def get_param_groups(self):
        return {self.class_prefix + "all": self.parameters()}


class MultiScaleNeuralBilateralAffineTransform(nn.Module):
    """modules.py:672-820: one feature grid per level; the levels' features (sliced at 1/factor resolution and up-sampled, or at
    full resolution when ``guidance_factor`` is None -- what the trainer passes, scene_graph.py:104) are concatenated in front of
    the MLP."""

    def __init__(self, class_name, n, grid, feature_dim, hidden_dim, device="cuda"):
        super().__init__()
        self.grid_size = grid
        self.tv_weight = []
        for i, (gx, gy, gl) in enumerate(grid):
            setattr(self, f"bil_grids{i}", NeuralBilateralGrid(num=n, grid_X=gx, grid_Y=gy, grid_W=gl, feature_dim=feature_dim))
            self.tv_weight.append(0.5 * (gx * gy * gl) ** 0.5)
        self.register_buffer("rgb2gray_weight", torch.tensor([0.299, 0.587, 0.114], dtype=torch.float32))
        self.feature_dim = feature_dim
        self.affine_network = _affine_network(len(grid) * feature_dim, hidden_dim)
        self.class_prefix = class_name + "#"
        self.device = device
        self.in_test_set = False
        self.training_indices_for_test: Dict[int, List[int]] = {}
        self.save_matrix = None
        self.to(device)

    def tv_loss(self):
        loss = 0
        for i in range(len(self.grid_size)):
            loss = loss + total_variation_loss(getattr(self, f"bil_grids{i}").grids) * self.tv_weight[i]
        return loss

    get_sample_grid = MultiScaleBilateralAffineTransform.get_sample_grid   # the reference's helper of the same name (low-res colour + xy grid)

    def forward(self, rgb: Tensor, image_infos, guidance_factor: Optional[Sequence[int]] = None) -> Tensor:
        assert "img_idx" in image_infos
        H, W, _ = rgb.shape
        return _head_maps(self.affine_network, self._features(rgb, image_infos, guidance_factor), H, W)

    def _features(self, rgb: Tensor, image_infos, guidance_factor: Optional[Sequence[int]] = None) -> Tensor:
        """The levels' features side by side [1, H, W, levels * feature_dim] (modules.py:728-760)."""
        k = _img_index(image_infos)
        H, W, _ = rgb.shape
        idxs = [k] if not self.in_test_set else self.training_indices_for_test[k]
        out_list = []
        for i in range(len(self.grid_size)):
            grids = getattr(self, f"bil_grids{i}")
            if guidance_factor is not None:
                _, lo = self.get_sample_grid(guidance_factor[i], H, W, rgb)     # xy = the pixel grid of the low-res image
                f = _sliced_features(grids, lo, None, idxs)
                B, Hm, Wm, C = f.shape
                if (Hm, Wm) != (H, W):     # fill_matrix_res on the feature channels (modules.py:748)
                    f = F.interpolate(f.permute(0, 3, 1, 2), size=(H, W), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
            else:
                f = _sliced_features(grids, rgb, None, idxs)
            out_list.append(f)
        self.save_matrix = out_list
        return torch.cat(out_list, dim=-1)

    def transform(self, rgb: Tensor, image_infos, guidance_factor: Optional[Sequence[int]] = None) -> Tensor:
        """forward + the trainer's application with the residual (scene_graph.py:103-106), the maps not materialised."""
        assert "img_idx" in image_infos
        levels = [getattr(self, f"bil_grids{i}") for i in range(len(self.grid_size))]
        if guidance_factor is None and _fused_image_ok(self.affine_network, levels, rgb):
            k = _img_index(image_infos)
            idxs = [k] if not self.in_test_set else self.training_indices_for_test[k]
            net = self.affine_network
            return mlp_head.image_transform(rgb, _image_grids(levels, idxs), net[0].weight, net[2].weight, net[4].weight)
        return _head_transform(self.affine_network, self._features(rgb, image_infos, guidance_factor), rgb)

    def get_param_groups(self):
        return {self.class_prefix + "all": self.parameters()}


def rotation_6d_to_matrix(d6: Tensor) -> Tensor:
    """The 6-D rotation representation of Zhou et al. (CVPR 2019) as the reference evaluates it
    (/root/reference/project/utils/geometry.py:83-105): Gram-Schmidt on the two 3-vectors, third row = their cross product; the three
    vectors are the ROWS of the result.  Pinned with CameraOptModule by tests/golden/camera_opt.npz."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = F.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


class CameraOptModule(nn.Module):
    """Mirror of models/modules.py:822-874 (the trainer's ``CamPose`` / ``CamPosePerturb``, applied in process_camera,
    trainers/base.py:324-328): one learnable 9-vector per image -- translation delta + 6-D rotation delta around the identity --
    right-multiplied onto the camera-to-world matrix.  Same parameter / buffer names (``embeds.weight`` [n,9], ``identity``), so
    reference checkpoints load.  It is the consumer of the path's camera-pose gradient: viewmat = inverse(module(c2w, id)), and the
    projection backward returns ``v_viewmat`` (``bds_project_view_bwd_list`` / ``bds_project_bwd``)."""

    def __init__(self, class_name: str, n: int, device="cuda"):
        super().__init__()
        self.class_prefix, self.device = class_name + "#", device
        self.embeds = nn.Embedding(n, 9)                                   # per image: (t, 6-D rotation offset); names fixed by checkpoints
        self.register_buffer("identity", torch.tensor([1.0, 0.0, 0.0, 0.0, 1.0, 0.0]))   # the 6-D code of "no rotation"
        self.zero_init()
        self.to(device)

    def zero_init(self):
        with torch.no_grad():
            self.embeds.weight.zero_()

    def random_init(self, std: float):
        with torch.no_grad():
            self.embeds.weight.normal_(0.0, std)

    def forward(self, camtoworlds: Tensor, embed_ids: Tensor) -> Tensor:
        """camtoworlds [*,4,4] x the image's rigid delta [[R_d, t_d], [0, 1]] on the right, written out by blocks instead of
        assembling the 4x4 delta: the first three columns become A R_d, the last one A t_d + a (A = the first three columns of the
        input, a = its last column; exact for any 4x4 input)."""
        assert camtoworlds.shape[:-2] == embed_ids.shape
        delta = self.embeds(embed_ids)
        R_d = rotation_6d_to_matrix(delta[..., 3:] + self.identity)
        A, a = camtoworlds[..., :, :3], camtoworlds[..., :, 3:]
        return torch.cat((A @ R_d, A @ delta[..., :3, None] + a), dim=-1)

    def get_param_groups(self):
        return {self.class_prefix + "all": self.parameters()}
