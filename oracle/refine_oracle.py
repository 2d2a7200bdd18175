"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement (numpy, float32) of adaptive density control --
VanillaGaussians.refinement_after (/root/reference/project/models/gaussians/vanilla.py:205-304) with split_gaussians
(:336-363), dup_gaussians (:365-376), cull_gaussians (:306-334) and the optimiser surgery dup_in_optim / remove_from_optim
(models/gaussians/basics.py:162-206).  Not imported by the product.

PINNED: tests/test_oracle_refine.py checks it against tests/golden/refine_step*.npz, which hold inputs and outputs of the
reference's own refinement_after (oracle/gen_golden_refine.py).

It is written in the "plan once, write once" form the HIP kernels use (csrc/refine.hip) rather than as the reference's
cat-then-mask sequence, so that passing the golden vectors also proves that formulation: per Gaussian five flags
(split, dup, keep original, keep split children, keep dup child) and the destination of every row in the final layout
[kept originals | kept split children, sample-major | kept dup children]."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

f32 = np.float32
PARAMS = ("_means", "_features_dc", "_features_rest", "_opacities", "_scales", "_quats")


def schedule(step: int, ctrl: dict, scene_scale: float, num_train_images: int) -> dict:
    """The step-dependent switches, all host integers in the reference (vanilla.py:207-215, 224-230, 279, 286, 314-324)."""
    reset_interval = ctrl["reset_alpha_interval"]
    if step <= ctrl["warmup_steps"]:
        return dict(active=False)
    past = step % reset_interval > max(num_train_images, ctrl["refine_interval"])
    return dict(
        active=True,
        do_densify=bool(step < ctrl["stop_split_at"] and past),
        size_thresh=f32(ctrl["densify_size_thresh"] * scene_scale),
        split_by_screen=bool(step < ctrl["stop_screen_size_at"]),
        do_cull=bool(past),
        cull_by_scale=bool(step > reset_interval),
        cull_scale=f32(ctrl["cull_scale_thresh"] * scene_scale),
        cull_by_screen=bool(step > reset_interval and step < ctrl["stop_screen_size_at"]),
        reset_opacity=bool(step % reset_interval == ctrl["refine_interval"]),
    )


def sigmoid(x):
    return (f32(1) / (f32(1) + np.exp(-x.astype(f32)))).astype(f32)


def shrink(ls):
    return np.log(np.exp(ls.astype(f32)) / f32(1.6)).astype(f32)   # vanilla.py:358-359


def plan(sch: dict, ctrl: dict, log_scales, logits, xys_grad_norm, vis_counts, max_2Dsize):
    N = log_scales.shape[0]
    smax = np.exp(log_scales.astype(f32)).max(axis=-1)
    m2d = np.zeros(N, f32) if max_2Dsize is None else max_2Dsize
    split = np.zeros(N, bool); dup = np.zeros(N, bool)
    if sch["do_densify"]:
        high = (xys_grad_norm / vis_counts) > f32(ctrl["densify_grad_thresh"])
        split = smax > sch["size_thresh"]
        if sch["split_by_screen"]:
            split = split | (m2d > f32(ctrl["split_screen_size"]))
        split = split & high
        # the dup mask is taken AFTER split_gaussians has shrunk the split parents in place (vanilla.py:246-250 follows :232-241):
        # a parent just above the size threshold can be split AND duplicated
        smax_post = np.where(split, np.exp(shrink(log_scales)).max(axis=-1), smax)
        dup = (smax_post <= sch["size_thresh"]) & high
    cull_o = np.zeros(N, bool); cull_c = np.zeros(N, bool)
    if sch["do_cull"]:
        cull_o = sigmoid(logits.reshape(N)) < f32(ctrl["cull_alpha_thresh"])
        cull_c = cull_o.copy()
        if sch["cull_by_scale"]:
            ls_post = np.where(split[:, None], shrink(log_scales), log_scales)
            toobig = np.exp(ls_post.astype(f32)).max(axis=-1) > sch["cull_scale"]
            cull_o = cull_o | toobig
            cull_c = cull_c | toobig
            if sch["cull_by_screen"]:
                cull_o = cull_o | (m2d > f32(ctrl["cull_screen_size"]))
    return split, dup, ~cull_o, split & ~cull_c, dup & ~cull_c


def quat_to_rotmat(q):
    q = q / np.linalg.norm(q, axis=-1, keepdims=True).astype(f32)
    q = q / np.maximum(np.linalg.norm(q, axis=-1, keepdims=True), f32(1e-12)).astype(f32)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], axis=-1).reshape(-1, 3, 3).astype(f32)


def refine(step: int, ctrl: dict, scene_scale: float, num_train_images: int, params: Dict[str, np.ndarray],
           exp_avg: Dict[str, np.ndarray], exp_avg_sq: Dict[str, np.ndarray], xys_grad_norm, vis_counts, max_2Dsize,
           samples: Optional[np.ndarray]):
    """Returns (new params, new exp_avg, new exp_avg_sq, n_split).  `samples` [n_split_samples * n_split, 3] is the
    reference's torch.randn draw (vanilla.py:343)."""
    return _refine(step, ctrl, scene_scale, num_train_images, params, exp_avg, exp_avg_sq, xys_grad_norm, vis_counts, max_2Dsize,
                   samples, None, None)[:4]


def out_of_bound(means, point_ids, instances_size):
    """nodes/rigid.py:374-383: a Gaussian whose (object-frame) mean lies outside its instance's box."""
    half = (instances_size[point_ids.reshape(-1)] / f32(2)).astype(f32)
    return (np.abs(means.astype(f32)) > half).any(axis=-1)


def refine_nodes(step: int, ctrl: dict, scene_scale: float, num_train_images: int, params, exp_avg, exp_avg_sq, xys_grad_norm,
                 vis_counts, max_2Dsize, samples, point_ids: np.ndarray, instances_size: Optional[np.ndarray]):
    """RigidNodes / DeformableNodes.refinement_after (nodes/rigid.py:194-293): the same schedule and masks as the background
    class, plus `point_ids` [N,1] (the instance every Gaussian belongs to) carried through split / dup / cull (:253, :317,
    :355-356, :371) and, with ctrl.cull_out_of_bound, the box test of cull_gaussians (:302-303) -- which the reference evaluates
    AFTER the new Gaussians were appended, so a split child is culled by where ITS sampled mean fell, not by its parent.
    Returns (params, exp_avg, exp_avg_sq, n_split, point_ids).  PINNED by tests/golden/refine_rigid_*.npz
    (oracle/gen_golden_refine_rigid.py)."""
    return _refine(step, ctrl, scene_scale, num_train_images, params, exp_avg, exp_avg_sq, xys_grad_norm, vis_counts, max_2Dsize,
                   samples, point_ids, instances_size if ctrl.get("cull_out_of_bound", False) else None)


def _refine(step, ctrl, scene_scale, num_train_images, params, exp_avg, exp_avg_sq, xys_grad_norm, vis_counts, max_2Dsize, samples,
            point_ids, box_sizes):
    sch = schedule(step, ctrl, scene_scale, num_train_images)
    P = {k: v.astype(f32).copy() for k, v in params.items()}
    M = {k: v.astype(f32).copy() for k, v in exp_avg.items()}
    V = {k: v.astype(f32).copy() for k, v in exp_avg_sq.items()}
    ids = None if point_ids is None else point_ids.copy()
    if not sch["active"]:
        return P, M, V, 0, ids
    n_split = 0
    if sch["do_densify"] or sch["do_cull"]:
        samps = int(ctrl["n_split_samples"])
        split, dup, keep_o, keep_s, keep_d = plan(sch, ctrl, P["_scales"], P["_opacities"], xys_grad_norm, vis_counts, max_2Dsize)
        N = split.shape[0]
        n_split = int(split.sum())
        rank = lambda m: np.cumsum(m) - m          # exclusive
        r_split, r_o, r_s, r_d = rank(split), rank(keep_o), rank(keep_s), rank(keep_d)
        KO, KS, KD = int(keep_o.sum()), int(keep_s.sum()), int(keep_d.sum())
        n_new = KO + samps * KS + KD
        ls_post = np.where(split[:, None], shrink(P["_scales"]), P["_scales"]).astype(f32)
        dst_o = r_o[keep_o]
        dst_s = [KO + s * KS + r_s[keep_s] for s in range(samps)]
        dst_d = KO + samps * KS + r_d[keep_d]

        def rows(src, zero_children):
            out = np.zeros((n_new,) + src.shape[1:], src.dtype)
            out[dst_o] = src[keep_o]
            if not zero_children:
                for s in range(samps):
                    out[dst_s[s]] = src[keep_s]
                out[dst_d] = src[keep_d]
            return out
        newP = {k: rows(P[k], False) for k in PARAMS if k not in ("_means", "_scales")}
        newP["_scales"] = rows(ls_post, False)
        means = rows(P["_means"], False)
        if KS:
            R = quat_to_rotmat(P["_quats"][keep_s])
            sc = np.exp(P["_scales"][keep_s].astype(f32))                   # BEFORE the shrink (vanilla.py:344-346)
            for s in range(samps):
                smp = samples[s * n_split + r_split[keep_s]].astype(f32)
                v = (sc * smp).astype(f32)
                means[dst_s[s]] = (np.einsum("nij,nj->ni", R, v).astype(f32) + P["_means"][keep_s]).astype(f32)
        newP["_means"] = means
        P = newP
        M = {k: rows(M[k], True) for k in M}
        V = {k: rows(V[k], True) for k in V}
        if ids is not None:
            ids = rows(ids, False)
        if box_sizes is not None and sch["do_cull"]:
            # second, order-preserving compaction: the union of the two cull masks removed in two rounds leaves the same rows in
            # the same order as the reference's single `culls` mask (rigid.py:300-317)
            keep = ~out_of_bound(P["_means"], ids, box_sizes.astype(f32))
            P = {k: v[keep] for k, v in P.items()}
            M = {k: v[keep] for k, v in M.items()}
            V = {k: v[keep] for k, v in V.items()}
            ids = ids[keep]
    if sch["reset_opacity"]:                                                  # vanilla.py:286-299
        x = np.minimum(sigmoid(P["_opacities"]), f32(ctrl["reset_alpha_value"]))
        P["_opacities"] = np.log(x / (f32(1) - x)).astype(f32)
        M["_opacities"] = np.zeros_like(M["_opacities"])
        V["_opacities"] = np.zeros_like(V["_opacities"])
    return P, M, V, n_split, ids
