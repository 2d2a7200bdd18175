"""Input marshalling of the reference's training step, restated under the reference's own names and argument
meaning.  The reference's versions are trainer METHODS that keep state on ``self``; these are free functions that RETURN that
state, so two call sites differ from the reference's by their return shape (a trainer method wrapping them stores the extra value
on ``self`` and is otherwise unchanged):

  reference (method)                                          here (function)
  ``gs = self.collect_gaussians(cam, image_ids)``             ``gs, pts_labels = collect_gaussians(models, classes, cam, image_ids)``
      (sets self.pts_labels)
  ``outputs, render_fn = self.render_gaussians(gs, cam)``     ``outputs, render_fn, info = render_gaussians(gs, cam, ...)``
      (sets self.info)

Where the reference's code lives:

  ``dataclass_camera`` / ``dataclass_gs``   /root/reference/project/models/gaussians/basics.py:112-160
  ``process_camera``                        models/trainers/base.py:317-340
  ``collect_gaussians``                     models/trainers/base.py:342-383
  ``get_gaussians``                         models/gaussians/vanilla.py:378-414 (the background class: SH colours + activations)
  ``render_gaussians``                      models/trainers/base.py:385-432 (-> ``rasterization``; returns ``results, render_fn, info``)

``fused_view`` / ``harness.render_view`` are the faster entry for the background
class alone (SH evaluated inside the view node); this module is the interface-compatible route for the multi-class scene graph,
where every class hands over already-activated tensors."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Mapping, Optional, Tuple

import torch
from torch import Tensor

from .gs_ops import spherical_harmonics
from .rendering import rasterization


@dataclass
class dataclass_camera:
    camtoworlds: Tensor
    camtoworlds_gt: Tensor
    Ks: Tensor
    H: int
    W: int


@dataclass
class dataclass_gs:
    """The concatenated, activated Gaussians of every class; ``detach_keys`` cuts the gradient of the named inputs (the names are the
    reference's: "activated_opacities", "means", "colors", "scales", "quats")."""
    _opacities: Tensor
    _means: Tensor
    _rgbs: Tensor
    _scales: Tensor
    _quats: Tensor
    detach_keys: List[str]
    extras: Optional[Dict[str, Tensor]] = None

    def set_grad_controller(self, detach_keys):
        self.detach_keys = detach_keys

    def _pick(self, key: str, t: Tensor) -> Tensor:
        return t.detach() if key in self.detach_keys else t

    @property
    def opacities(self):
        return self._pick("activated_opacities", self._opacities)

    @property
    def means(self):
        return self._pick("means", self._means)

    @property
    def rgbs(self):
        return self._pick("colors", self._rgbs)

    @property
    def scales(self):
        return self._pick("scales", self._scales)

    @property
    def quats(self):
        return self._pick("quats", self._quats)


def process_camera(camera_infos: Mapping[str, Tensor], image_ids: Tensor, models: Optional[Mapping[str, Callable]] = None,
                   novel_view: bool = False) -> dataclass_camera:
    """base.py:317-340: the pose refinement modules (``CamPosePerturb``, then ``CamPose``) act on the camera-to-world matrix unless the
    view is a novel one; the ground-truth pose is kept next to it."""
    camtoworlds = camtoworlds_gt = camera_infos["camera_to_world"]
    models = models or {}
    if "CamPosePerturb" in models and not novel_view:
        camtoworlds = models["CamPosePerturb"](camtoworlds, image_ids)
    if "CamPose" in models and not novel_view:
        camtoworlds = models["CamPose"](camtoworlds, image_ids)
    return dataclass_camera(camtoworlds=camtoworlds, camtoworlds_gt=camtoworlds_gt, Ks=camera_infos["intrinsics"],
                            H=camera_infos["height"], W=camera_infos["width"])


def _check_finite(gs: Mapping[str, Tensor], step) -> None:
    """vanilla.py:408-412 raises on NaN / Inf per tensor (two reductions + two host syncs each); one fused reduction over the five
    tensors and one read-back here, the per-tensor message only on the failing path."""
    bad = torch.stack([(~torch.isfinite(v)).any() for v in gs.values()])
    if bool(bad.any()):
        for k, v in gs.items():
            if torch.isnan(v).any():
                raise ValueError(f"NaN detected in gaussian {k} at step {step}")
            if torch.isinf(v).any():
                raise ValueError(f"Inf detected in gaussian {k} at step {step}")


def get_gaussians(model, cam: dataclass_camera) -> Dict[str, Tensor]:
    """VanillaGaussians.get_gaussians (vanilla.py:378-414) for a model with the reference's attributes (``_means``, ``_features_dc``,
    ``_features_rest``, ``_opacities``, ``_scales``, ``_quats``, ``sh_degree``, ``step``, ``ctrl_cfg.sh_degree_interval``): view
    directions from the detached means, SH colours + 0.5 clamped to [0, 1] at the currently active degree, sigmoid / exp /
    normalise activations."""
    colors = torch.cat((model._features_dc[:, None, :], model._features_rest), dim=1)
    if model.sh_degree > 0:
        viewdirs = model._means.detach() - cam.camtoworlds.data[..., :3, 3]
        viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
        n = min(model.step // model.ctrl_cfg.sh_degree_interval, model.sh_degree)
        rgbs = torch.clamp(spherical_harmonics(n, viewdirs, colors) + 0.5, 0.0, 1.0)
    else:
        rgbs = torch.sigmoid(colors[:, 0, :])
    gs = dict(_means=model._means, _opacities=torch.sigmoid(model._opacities), _rgbs=rgbs, _scales=torch.exp(model._scales),
              _quats=model._quats / model._quats.norm(dim=-1, keepdim=True))
    _check_finite(gs, getattr(model, "step", -1))
    return gs


def get_gaussians_lazy(self, cam: dataclass_camera) -> Dict[str, Tensor]:
    """Drop-in for ``VanillaGaussians.get_gaussians`` (vanilla.py:378-414) that DEFERS the activations: the same dict, its five entries
    placeholders (``lazy_gaussians.LazyField``) over the raw parameters.  ``rasterization`` runs a complete set of them through one
    node -- sigmoid / exp / quaternion normalisation inside the projection kernel, SH colours for the visible Gaussians only, read
    from ``_features_dc`` / ``_features_rest`` where they lie, the NaN / Inf check as one launch whose flag arrives with the list
    counts; every other use of a placeholder materialises it with the reference's own expression (``lazy_gaussians``).  A class
    without SH bands (``sh_degree == 0``: sigmoid colours) takes the eager mirror ``get_gaussians``."""
    from .lazy_gaussians import LazyField, RawGaussians
    if self.sh_degree <= 0:
        return get_gaussians(self, cam)
    if getattr(self, "filter_mask", None) is None or self.filter_mask.shape[0] != self._means.shape[0]:
        self.filter_mask = torch.ones(self._means.shape[0], dtype=torch.bool, device=self._means.device)   # vanilla.py:379-380 (all ones: kept)
    n = min(self.step // self.ctrl_cfg.sh_degree_interval, self.sh_degree)           # vanilla.py:387
    src = RawGaussians(self._means, self._quats, self._scales, self._opacities, self._features_dc, self._features_rest, n,
                       cam.camtoworlds.data[..., :3, 3], step=getattr(self, "step", -1))
    N = self._means.shape[0]
    return dict(_means=LazyField(src, "_means", (N, 3)), _opacities=LazyField(src, "_opacities", self._opacities.shape),
                _rgbs=LazyField(src, "_rgbs", (N, 3)), _scales=LazyField(src, "_scales", (N, 3)), _quats=LazyField(src, "_quats", (N, 4)))


def install(gaussian_class) -> None:
    """``install(VanillaGaussians)``: the class's ``get_gaussians`` becomes ``get_gaussians_lazy`` (the reference's file is not
    touched; the same mechanism as assigning ``densify.refinement_after``).  The original stays reachable as
    ``gaussian_class._bds_reference_get_gaussians``; ``uninstall`` puts it back."""
    if getattr(gaussian_class, "_bds_reference_get_gaussians", None) is None:
        gaussian_class._bds_reference_get_gaussians = gaussian_class.__dict__.get("get_gaussians")
    gaussian_class.get_gaussians = get_gaussians_lazy


def uninstall(gaussian_class) -> None:
    ref = getattr(gaussian_class, "_bds_reference_get_gaussians", None)
    if ref is not None:
        gaussian_class.get_gaussians = ref
    elif "get_gaussians" in gaussian_class.__dict__:
        del gaussian_class.get_gaussians
    gaussian_class._bds_reference_get_gaussians = None


def collect_gaussians(models: Mapping[str, object], gaussian_classes: Mapping[str, int], cam: dataclass_camera,
                      image_ids: Optional[Tensor] = None) -> Tuple[dataclass_gs, Tensor]:
    """base.py:342-383: every class's ``get_gaussians(cam)`` dict (a class may return None: no instance in this frame) concatenated
    along N, plus the class label of every Gaussian (int64 [N]; the trainer's ``pts_labels``, from which the per-class opacity masks
    and ``dynamic_pts_mask = (pts_labels != 0).float()`` follow)."""
    keys = ("_means", "_scales", "_quats", "_rgbs", "_opacities")
    parts: Dict[str, List[Tensor]] = {k: [] for k in keys}
    labels: List[Tensor] = []
    for class_name, label in gaussian_classes.items():
        m = models[class_name]
        gs = m.get_gaussians(cam) if hasattr(m, "get_gaussians") else get_gaussians(m, cam)
        if gs is None:
            continue
        for k in keys:
            parts[k].append(gs[k])
        labels.append(torch.full((gs["_means"].shape[0],), label, device=gs["_means"].device))
    cat = {k: torch.cat(v, dim=0) for k, v in parts.items()}
    pts_labels = torch.cat(labels, dim=0)
    return dataclass_gs(_means=cat["_means"], _scales=cat["_scales"], _quats=cat["_quats"], _rgbs=cat["_rgbs"],
                        _opacities=cat["_opacities"], detach_keys=[], extras=None), pts_labels


def render_gaussians(gs: dataclass_gs, cam: dataclass_camera, *, packed: bool = False, absgrad: bool = True, sparse_grad: bool = False,
                     antialiased: bool = False, training: bool = True, **kwargs):
    """base.py:385-432.  Returns ``(results, render_fn, info)``: ``results`` = {"rgb_gaussians" (clamped at 1), "depth", "opacity"};
    ``render_fn(opacity_mask=None, return_info=False)`` re-renders with the opacities multiplied by a per-Gaussian mask (the
    per-class renders of the evaluation path, scene_graph.py:296-313); ``info`` is the trainer's ``self.info``, with
    ``info["means2d"].retain_grad()`` already called when ``training``.  ``kwargs`` go to ``rasterization`` as the reference's do
    (``near_plane``, ``far_plane``, ``render_mode="RGB+ED"``, ``radius_clip``)."""
    def render_fn(opacity_mask=None, return_info=False):
        op = gs.opacities.squeeze()
        renders, alphas, info = rasterization(
            means=gs.means, quats=gs.quats, scales=gs.scales, opacities=op * opacity_mask if opacity_mask is not None else op,
            colors=gs.rgbs, viewmats=torch.linalg.inv(cam.camtoworlds)[None, ...], Ks=cam.Ks[None, ...], width=cam.W, height=cam.H,
            packed=packed, absgrad=absgrad, sparse_grad=sparse_grad, rasterize_mode="antialiased" if antialiased else "classic",
            **kwargs)
        renders = renders[0]
        alphas = alphas[0].squeeze(-1)
        assert renders.shape[-1] == 4, "Must render rgb, depth and alpha"
        rendered_rgb, rendered_depth = torch.split(renders, [3, 1], dim=-1)
        if not return_info:
            return torch.clamp(rendered_rgb, max=1.0), rendered_depth, alphas[..., None]
        return torch.clamp(rendered_rgb, max=1.0), rendered_depth, alphas[..., None], info

    rgb, depth, opacity, info = render_fn(return_info=True)
    if training:
        info["means2d"].retain_grad()
    return {"rgb_gaussians": rgb, "depth": depth, "opacity": opacity}, render_fn, info
