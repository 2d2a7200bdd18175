"""Deferred activations of a Gaussian class: what lets the reference's UNMODIFIED call sequence reach the fused one-view node.

The reference's ``VanillaGaussians.get_gaussians`` (/root/reference/project/models/gaussians/vanilla.py:378-414) activates every
Gaussian before anything is culled -- sigmoid, exp, ``quats / quats.norm()``, a concatenation of the two SH parameters and a dense SH
pass over all N -- and hands five dense tensors to ``collect_gaussians`` -> ``render_gaussians`` -> ``gsplat.rasterization``
(models/trainers/base.py:342-408).  ``marshalling.install`` replaces that method by one that returns the same dict with
``LazyField`` placeholders: tensors of the right shape / dtype / device that remember WHERE they come from (``RawGaussians``: the raw
parameters, the active SH degree, the camera centre).  ``rendering.rasterization`` recognises a complete set of placeholders and runs
the raw parameters through one node (activations inside the projection kernel, SH for the visible Gaussians only, straight from the
two SH parameters); ANY other use of a placeholder -- an operator, a method, an index -- first materialises it with the reference's
own expression, autograd graph included, so the trainer's other code paths (several classes concatenated, opacity masks of the
evaluation renders, ``detach_keys``) see ordinary tensors with the reference's values.

Kept lazy: ``torch.cat`` of a single placeholder (base.py:365-366 with one class), ``squeeze`` / ``reshape`` / ``view`` / ``flatten`` of
the element-wise fields (base.py:397 ``gs.opacities.squeeze()``), metadata reads.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor

FIELDS = ("_means", "_opacities", "_scales", "_quats", "_rgbs")
_ELEMENTWISE = ("_opacities", "_scales")      # fields on which a pure reshape commutes with the activation


class RawGaussians:
    """One class's raw parameters for one camera + the reference's expressions for its activated tensors (vanilla.py:383-395)."""

    def __init__(self, means: Tensor, quats: Tensor, log_scales: Tensor, logits: Tensor, features_dc: Tensor, features_rest: Tensor,
                 sh_degree: int, cam_pos: Tensor, step=-1):
        self.means, self.quats, self.log_scales, self.logits = means, quats, log_scales, logits
        self.features_dc, self.features_rest = features_dc, features_rest
        self.sh_degree, self.cam_pos, self.step = int(sh_degree), cam_pos, step
        self._cache: Dict[str, Tensor] = {}

    def materialise(self, field: str) -> Tensor:
        t = self._cache.get(field)
        if t is None:
            if field == "_means":
                t = self.means          # (a placeholder too: base.py:365-366 would otherwise hand the rasterizer a COPY of the parameter)
            elif field == "_opacities":
                t = torch.sigmoid(self.logits)                                              # vanilla.py:177 get_opacity
            elif field == "_scales":
                t = torch.exp(self.log_scales)                                              # vanilla.py:165 get_scaling
            elif field == "_quats":
                t = self.quats / self.quats.norm(dim=-1, keepdim=True)                      # vanilla.py:181 get_quats
            elif field == "_rgbs":
                from .gs_ops import spherical_harmonics
                colors = torch.cat((self.features_dc[:, None, :], self.features_rest), dim=1)   # vanilla.py:382
                viewdirs = self.means.detach() - self.cam_pos                               # vanilla.py:384-385
                viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
                t = torch.clamp(spherical_harmonics(self.sh_degree, viewdirs, colors) + 0.5, 0.0, 1.0)   # vanilla.py:388-389
            else:
                raise KeyError(field)
            # the reference raises on a NaN / Inf in ANY of the activated tensors (vanilla.py:407-412).  The one-view node checks the
            # raw parameters itself (bds_nonfinite_flags_kinds); every OTHER route to an activated tensor -- several classes
            # concatenated, opacity masks, backgrounds, the fall-back of rasterization() -- comes through here, once per field
            v = t.detach()
            if bool((~torch.isfinite(v)).any()):
                what = "NaN" if bool(torch.isnan(v).any()) else "Inf"
                raise ValueError(f"{what} detected in gaussian {field} at step {self.step}")
            self._cache[field] = t
        return t


_META = None


def _meta_funcs():
    global _META
    if _META is None:
        T = torch.Tensor
        _META = {T.shape.__get__, T.dtype.__get__, T.device.__get__, T.ndim.__get__, T.dim, T.size, T.numel, T.is_cuda.__get__,
                 T.requires_grad.__get__, T.is_floating_point, T.ndimension, T.layout.__get__, T.__len__, T.__repr__, T.__hash__,
                 T.is_leaf.__get__, T.grad_fn.__get__, T.is_sparse.__get__, T.nelement}
    return _META


class LazyField(torch.Tensor):
    """Placeholder for ``src.materialise(field)`` (module docstring).  A wrapper tensor without storage: only metadata is real."""

    @staticmethod
    def __new__(cls, src: RawGaussians, field: str, shape):
        assert field in FIELDS
        r = torch.Tensor._make_wrapper_subclass(cls, tuple(shape), dtype=torch.float32, device=src.means.device, requires_grad=False)
        r._src, r._field = src, field
        return r

    def materialise(self) -> Tensor:
        t = self._src.materialise(self._field)
        with torch._C.DisableTorchFunctionSubclass():
            shape = tuple(torch.Tensor.size(self))
        return t if tuple(t.shape) == shape else t.reshape(shape)

    def __repr__(self):
        with torch._C.DisableTorchFunctionSubclass():
            return f"LazyField({self._field}, shape={tuple(torch.Tensor.size(self))})"

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _meta_funcs():
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        # concatenation of ONE placeholder along dim 0 (one Gaussian class in the scene): the placeholder itself
        if func is torch.cat and len(args) >= 1 and isinstance(args[0], (list, tuple)) and len(args[0]) == 1 \
                and isinstance(args[0][0], LazyField) and (args[1] if len(args) > 1 else kwargs.get("dim", 0)) == 0:
            return args[0][0]
        # pure reshapes of an element-wise field commute with its activation
        if args and isinstance(args[0], LazyField) and args[0]._field in _ELEMENTWISE and func in (
                torch.Tensor.squeeze, torch.Tensor.reshape, torch.Tensor.view, torch.Tensor.flatten, torch.squeeze, torch.reshape,
                torch.flatten, torch.Tensor.unsqueeze, torch.unsqueeze):
            me = args[0]
            with torch._C.DisableTorchFunctionSubclass():
                shape = func(torch.empty(tuple(torch.Tensor.size(me)), device="meta"), *args[1:], **kwargs).shape
            return LazyField(me._src, me._field, shape)
        return func(*_materialised(args), **_materialised(kwargs))

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):      # (an operator that got past __torch_function__: same rule)
        return func(*_materialised(args), **_materialised(kwargs or {}))


def _materialised(x):
    if isinstance(x, LazyField):
        return x.materialise()
    if isinstance(x, (list, tuple)):
        return type(x)(_materialised(v) for v in x)
    if isinstance(x, dict):
        return {k: _materialised(v) for k, v in x.items()}
    return x


def materialised(x):
    """``x`` with every placeholder inside replaced by its tensor (lists / tuples / dicts walked)."""
    return _materialised(x)


def lazy_source(means: Tensor, quats, scales, opacities, colors) -> Optional[RawGaussians]:
    """The ``RawGaussians`` behind a COMPLETE set of placeholders handed to ``rasterization`` (all four fields of one source, the means
    that source's own parameter), else None."""
    fields = (("_quats", quats), ("_scales", scales), ("_opacities", opacities), ("_rgbs", colors))
    if not all(isinstance(t, LazyField) and t._field == name for name, t in fields):
        return None
    src = quats._src
    own_means = means is src.means or (isinstance(means, LazyField) and means._field == "_means" and means._src is src)
    if not all(t._src is src for _, t in fields) or not own_means:
        return None
    return src
