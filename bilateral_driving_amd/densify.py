"""Adaptive density control of one class of Gaussians on the device: drop-in for
``VanillaGaussians.refinement_after`` (/root/reference/project/models/gaussians/vanilla.py:205-304) including the optimiser
surgery of models/gaussians/basics.py:162-206.

    from bilateral_driving_amd.densify import refinement_after
    VanillaGaussians.refinement_after = refinement_after          # or: refinement_after(model, step, optimizer)

``model`` is anything with the reference's attributes: ``_means [N,3] _features_dc [N,3] _features_rest [N,K-1,3]
_opacities [N,1] _scales [N,3] _quats [N,4]`` (``nn.Parameter``), ``ctrl_cfg`` (attribute access to the ``ctrl`` keys),
``scene_scale``, ``num_train_images``, ``step``, ``class_prefix`` and the statistics ``xys_grad_norm / vis_counts / max_2Dsize``
(``optim.DensifyStats`` keeps them on the device).  ``optimizer`` is the trainer's Adam (torch.optim.Adam or
``optim.FusedAdam``) with one named group per parameter (models/trainers/base.py:201-222).

Where the reference runs three rounds of boolean-mask indexing + ``torch.cat`` over 18 tensors and then masks all of them again
for the cull (a few hundred launches and a host sync per mask), this plans the topology change once (``bds_refine_plan``: 3
launches, ONE host read-back of five counts -- the size of the noise tensor has to be known, as in the reference) and writes
each of the 18 arrays exactly once into its final layout.  The result -- order of the rows included -- is the reference's.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor
from torch.nn import Parameter

from . import _lib as L

_ATTRS = ("_means", "_features_dc", "_features_rest", "_opacities", "_scales", "_quats")
_GROUPS = ("xyz", "sh_dc", "sh_rest", "opacity", "scaling", "rotation")       # vanilla.py:193-201


def _group_of(optimizer, name: str):
    for g in optimizer.param_groups:
        if g.get("name") == name:
            return g
    return None


def plan(log_scales: Tensor, logits: Tensor, xys_grad_norm: Optional[Tensor], vis_counts: Optional[Tensor],
         max_2Dsize: Optional[Tensor], *, do_densify: bool, grad_thresh: float, size_thresh: float, split_by_screen: bool,
         split_screen_size: float, do_cull: bool, cull_alpha_thresh: float, cull_by_scale: bool, cull_scale_thresh: float,
         cull_by_screen: bool, cull_screen_size: float, extra_cull: Optional[Tensor] = None):
    """bds_refine_plan: (flags [N] u8, ranks [N,4] i32, totals [5] i64 on the device).  ``extra_cull`` [N] u8: a cull mask over the
    input rows OR-ed into the originals' decision."""
    L.require_gpu(log_scales)
    N, dev = log_scales.shape[0], log_scales.device
    flags = torch.empty(N, dtype=torch.uint8, device=dev)
    ranks = torch.empty(N, 4, dtype=torch.int32, device=dev)
    totals = torch.empty(5, dtype=torch.int64, device=dev)
    nb = int(L.lib().bds_refine_plan_temp_bytes(N))
    temp = torch.empty(nb, dtype=torch.uint8, device=dev)
    f32 = lambda t: None if t is None else t.detach().reshape(-1).contiguous().float()
    xs, vc, m2 = f32(xys_grad_norm), f32(vis_counts), f32(max_2Dsize)
    ls, lg = log_scales.detach().contiguous(), logits.detach().reshape(-1).contiguous()
    if extra_cull is not None:
        assert extra_cull.dtype == torch.uint8 and extra_cull.shape == (N,) and extra_cull.is_contiguous()
    L.check(L.lib().bds_refine_plan(N, L.ptr(xs), L.ptr(vc), L.ptr(m2), L.ptr(ls), L.ptr(lg), L.ptr(extra_cull), int(do_densify),
                                    float(grad_thresh),
                                    float(size_thresh), int(split_by_screen), float(split_screen_size), int(do_cull),
                                    float(cull_alpha_thresh), int(cull_by_scale), float(cull_scale_thresh), int(cull_by_screen),
                                    float(cull_screen_size), L.ptr(flags), L.ptr(ranks), L.ptr(totals), L.ptr(temp), nb, L.stream()),
            "bds_refine_plan")
    return flags, ranks, totals


def _rows(src: Tensor, n_new: int, samps: int, flags, ranks, totals, zero_children: bool) -> Tensor:
    src = src.detach().contiguous()
    N = src.shape[0]
    width = src.numel() // max(N, 1)
    dst = torch.empty((n_new,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
    if n_new == 0 or width == 0:        # everything culled: nothing to write (an empty tensor has no address to hand over)
        return dst
    L.check(L.lib().bds_refine_rows(N, width, samps, L.ptr(flags), L.ptr(ranks), L.ptr(totals), L.ptr(src), L.ptr(dst),
                                    int(zero_children), L.stream()), "bds_refine_rows")
    return dst


def _ctrl_get(ctrl, key: str, default):
    """ctrl keys that only some classes carry (OmegaConf node, dict or attribute bag)."""
    if hasattr(ctrl, "get"):
        v = ctrl.get(key, default)
        return default if v is None else v
    return getattr(ctrl, key, default)


def _move_rows(self, optimizer, flags, ranks, totals, samps: int, n_new: int, geometry: Optional[Dict[str, Tensor]]) -> None:
    """Writes every per-Gaussian array of the class once into the planned layout: the six parameters (``geometry`` holds the two a
    split changes, already computed), the Adam moments of their groups (children start from zero, basics.py:191-201) and -- node
    classes -- the ``point_ids`` column (nodes/rigid.py:253,317: int64, moved as two 32-bit words per row)."""
    new: Dict[str, Tensor] = dict(geometry or {})
    for a in _ATTRS:
        if a not in new:
            new[a] = _rows(getattr(self, a), n_new, samps, flags, ranks, totals, zero_children=False)
    ids = getattr(self, "point_ids", None)
    if ids is not None:
        assert ids.dtype == torch.int64 and ids.shape[0] == flags.shape[0]
        L.require_gpu(ids)
        words = ids.contiguous().view(torch.float32).reshape(ids.shape[0], -1)
        moved = _rows(words, n_new, samps, flags, ranks, totals, zero_children=False)
        self.point_ids = moved.view(torch.int64).reshape((n_new,) + tuple(ids.shape[1:]))
    for a, gname in zip(_ATTRS, _GROUPS):
        prm = Parameter(new[a])
        setattr(self, a, prm)
        group = _group_of(optimizer, self.class_prefix + gname)
        if group is None:
            continue
        old_p = group["params"][0]
        state = optimizer.state.pop(old_p, None)                          # basics.py:162-206
        if state:
            for k in ("exp_avg", "exp_avg_sq"):
                state[k] = _rows(state[k], n_new, samps, flags, ranks, totals, zero_children=True)
            optimizer.state[prm] = state
        group["params"] = [prm]


def spatial_order(means: Tensor, bits: int = 10) -> Tensor:
    """Permutation [N] (int64) that puts the Gaussians in Morton (Z-curve) order of their centres (``bits`` per axis over the
    bounding box).  Rows that lie together in space then lie together in memory, so the ~15 % of the set a camera sees form RUNS
    instead of isolated rows, and every list-driven kernel of a view (SH rows in the record pack, the projection / SH backward, the
    row clear) moves whole cache lines: 2 M Gaussians / six 1080p views 1023 -> 1088 it/s, 5 M Gaussians (c5) 591 -> 710
    (profiles/NOTES.md).  The reference keeps the order of its initial point cloud (lidar sweeps: coherent along the drive; its random
    points: not) and appends split / duplicated Gaussians at the end (vanilla.py:256-262); the order carries no meaning -- apply the
    permutation with ``reorder_rows`` once after initialisation and, when wanted, after densification steps."""
    m = means.detach().float()
    lo, hi = m.amin(0), m.amax(0)
    q = ((m - lo) / (hi - lo).clamp(min=1e-12) * ((1 << bits) - 1)).long().clamp(0, (1 << bits) - 1)
    code = torch.zeros(m.shape[0], dtype=torch.int64, device=m.device)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return torch.argsort(code, stable=True)


@torch.no_grad()
def reorder_rows(self, optimizer, perm: Tensor) -> None:
    """Row ``i`` of every per-Gaussian array of the class becomes old row ``perm[i]``: the six parameters (new ``nn.Parameter``s in
    their optimizer groups, as the reference's own densification replaces them: basics.py:162-206), their Adam moments, ``point_ids``
    of the node classes and the densification statistics -- everything ``refinement_after`` moves.  ``optimizer`` may be None."""
    N = self._means.shape[0]
    assert perm.shape == (N,) and perm.dtype == torch.int64
    for a, gname in zip(_ATTRS, _GROUPS):
        old_p = getattr(self, a)
        prm = Parameter(old_p.detach().index_select(0, perm))
        setattr(self, a, prm)
        group = _group_of(optimizer, self.class_prefix + gname) if optimizer is not None else None
        if group is None:
            continue
        state = optimizer.state.pop(group["params"][0], None)
        if state:
            for k in ("exp_avg", "exp_avg_sq"):
                if torch.is_tensor(state.get(k)) and state[k].shape[:1] == (N,):
                    state[k] = state[k].index_select(0, perm)
            optimizer.state[prm] = state
        group["params"] = [prm]
    for name in ("point_ids", "xys_grad_norm", "vis_counts", "max_2Dsize"):
        t = getattr(self, name, None)
        if torch.is_tensor(t) and t.shape[:1] == (N,):
            setattr(self, name, t.index_select(0, perm))


def out_of_bound_mask(means: Tensor, point_ids: Tensor, instances_size: Tensor) -> Tensor:
    """RigidNodes.get_out_of_bound_mask (nodes/rigid.py:374-383) as a [N] uint8 mask."""
    L.require_gpu(means, point_ids, instances_size)
    N = means.shape[0]
    assert point_ids.dtype == torch.int64 and point_ids.numel() == N
    m = means.detach().contiguous().float()
    sz = instances_size.detach().contiguous().float()
    ids = point_ids.reshape(-1).contiguous()
    mask = torch.empty(N, dtype=torch.uint8, device=means.device)
    L.check(L.lib().bds_refine_out_of_bound(N, L.ptr(m), L.ptr(ids), sz.shape[0], L.ptr(sz), L.ptr(mask), L.stream()),
            "bds_refine_out_of_bound")
    return mask


@torch.no_grad()
def refinement_after(self, step: int, optimizer: torch.optim.Optimizer, samples: Optional[Tensor] = None, verbose: bool = True,
                     sample_fn=None, reorder: bool = False) -> None:
    """Same contract as VanillaGaussians.refinement_after(step, optimizer) and, for a model that carries ``point_ids``
    (+ ``instances_size`` when ``ctrl.cull_out_of_bound``), as RigidNodes / DeformableNodes.refinement_after
    (models/nodes/rigid.py:194-325).  ``samples`` (optional, [n_split_samples * n_split, 3]) replaces the ``torch.randn`` draw of
    split_gaussians (vanilla.py:343) -- the tests feed the reference's recorded noise; ``sample_fn(shape, device)`` (optional)
    produces it instead (view-parallel training: ``dist.broadcast_randn`` gives every rank rank 0's draw, so that the replicas stay
    identical -- SURVEY.md 8e).

    The box test of the node classes is taken on the rows AFTER split / dup (a split child is judged by where its own sample
    fell), so it runs as a second, cull-only plan over the new set; removing the two masks one after the other leaves the
    reference's rows in the reference's order.  ``reorder=True``: a step that changed the set ends with ``reorder_rows`` into
    ``spatial_order`` (the children a split / dup appends at the end go where their neighbours are)."""
    assert step == self.step
    ctrl = self.ctrl_cfg
    if self.step <= ctrl.warmup_steps:
        return
    if getattr(self, "ball_gaussians", False) or getattr(self, "gaussian_2d", False):
        raise NotImplementedError("ball / 2-D Gaussians are not covered by the fused refinement")
    reset_interval = ctrl.reset_alpha_interval
    past = self.step % reset_interval > max(self.num_train_images, ctrl.refine_interval)
    do_densify = bool(self.step < ctrl.stop_split_at and past)
    do_cull = bool(past)
    box_cull = bool(do_cull and _ctrl_get(ctrl, "cull_out_of_bound", False))
    if box_cull:
        assert getattr(self, "point_ids", None) is not None and getattr(self, "instances_size", None) is not None, \
            "cull_out_of_bound needs point_ids and instances_size (models/nodes/rigid.py:378-380)"
    if verbose:
        print(f"Class {self.class_prefix} current points: {self._means.shape[0]} @ step {self.step}")
    if do_densify or do_cull:
        if do_densify:
            assert self.xys_grad_norm is not None and self.vis_counts is not None and self.max_2Dsize is not None
        for a in _ATTRS:
            L.require_gpu(getattr(self, a))
        N = self._means.shape[0]
        samps = int(ctrl.n_split_samples) if do_densify else 0
        by_scale = bool(self.step > reset_interval)
        flags, ranks, totals = plan(
            self._scales, self._opacities, self.xys_grad_norm, self.vis_counts, self.max_2Dsize,
            do_densify=do_densify, grad_thresh=ctrl.densify_grad_thresh, size_thresh=ctrl.densify_size_thresh * self.scene_scale,
            split_by_screen=self.step < ctrl.stop_screen_size_at, split_screen_size=ctrl.split_screen_size, do_cull=do_cull,
            cull_alpha_thresh=ctrl.cull_alpha_thresh, cull_by_scale=by_scale, cull_scale_thresh=ctrl.cull_scale_thresh * self.scene_scale,
            cull_by_screen=by_scale and self.step < ctrl.stop_screen_size_at, cull_screen_size=ctrl.cull_screen_size)
        n_split, n_dup, KO, KS, KD = (int(v) for v in totals.tolist())       # the one host sync (the reference: one per mask)
        n_new = KO + samps * KS + KD
        dev = self._means.device
        if do_densify:
            if samples is None and sample_fn is not None:
                samples = sample_fn((samps * n_split, 3), dev)
            if samples is None:
                samples = torch.randn((samps * n_split, 3), device=dev)       # vanilla.py:343, same draw from the same stream
            samples = samples.to(device=dev, dtype=torch.float32).contiguous()
            assert samples.shape == (samps * n_split, 3)
        means, quats, ls = self._means.detach().contiguous(), self._quats.detach().contiguous(), self._scales.detach().contiguous()
        geometry = {"_means": torch.empty(n_new, 3, device=dev), "_scales": torch.empty(n_new, 3, device=dev)}
        if n_new:
            L.check(L.lib().bds_refine_geometry(N, samps, L.ptr(flags), L.ptr(ranks), L.ptr(totals), L.ptr(samples) if do_densify else None,
                                                L.ptr(means), L.ptr(quats), L.ptr(ls), L.ptr(geometry["_means"]), L.ptr(geometry["_scales"]),
                                                L.stream()), "bds_refine_geometry")
        _move_rows(self, optimizer, flags, ranks, totals, samps, n_new, geometry)
        n_out = 0
        if box_cull and n_new:
            mask = out_of_bound_mask(self._means, self.point_ids, self.instances_size)
            flags2, ranks2, totals2 = plan(self._scales, self._opacities, None, None, None, do_densify=False, grad_thresh=0.0,
                                           size_thresh=0.0, split_by_screen=False, split_screen_size=0.0, do_cull=True,
                                           cull_alpha_thresh=-1.0, cull_by_scale=False, cull_scale_thresh=0.0, cull_by_screen=False,
                                           cull_screen_size=0.0, extra_cull=mask)
            kept = int(totals2[2].item())                                    # second (and last) host read-back: the new size
            n_out = n_new - kept
            if n_out:
                _move_rows(self, optimizer, flags2, ranks2, totals2, 0, kept, None)
        if verbose:
            if do_densify:
                print(f"    Split: {n_split}")
                print(f"      Dup: {n_dup}")
            print(f"     Cull: {N + samps * n_split + n_dup - n_new + n_out}")
        if reorder and self._means.shape[0] > 1:
            reorder_rows(self, optimizer, spatial_order(self._means))
    if verbose:
        print(f"Class {self.class_prefix} left points: {self._means.shape[0]}")
    if self.step % reset_interval == ctrl.refine_interval:                    # vanilla.py:286-299
        prm = self._opacities
        L.require_gpu(prm)
        group = _group_of(optimizer, self.class_prefix + "opacity")
        state = optimizer.state.get(group["params"][0]) if group is not None else None
        m = state["exp_avg"] if state else None
        v = state["exp_avg_sq"] if state else None
        L.check(L.lib().bds_opacity_reset(prm.numel(), L.ptr(prm.data), float(ctrl.reset_alpha_value), L.ptr(m), L.ptr(v), L.stream()),
                "bds_opacity_reset")
    self.xys_grad_norm = None
    self.vis_counts = None
    self.max_2Dsize = None
