"""The replayed frame checked against the eager frame on the SAME parameters, at whatever size the caller runs.

``graph_view.FrameGraph.step()`` (device-side list counts, SH colours in the record pack, loss on the transform's launch, two
streams, in-place gradient rows) and the eager host-count loop (``harness.render_view`` -> ``training_loss`` -> autograd) are the
same kernels in two drivers; the small-size tests tie both to the oracle.  This puts the two side by side at FULL size -- where
capacities, 32-bit offsets and the binned schedule are actually stressed: ``bench.py`` runs it after every timed region
(``selfcheck`` block of its line), ``tests/test_gpu_04`` at BASELINE.json's c3 / c5 sizes.  Reference path being mirrored:
/root/reference/project/tools/train.py:252-283 (one view: forward, losses, backward).
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch
from torch import Tensor

from . import fused_view as FV
from . import harness as Hn

IMAGE_TOL, LOSS_TOL, GRAD_TOL = 0.0, 1e-5, 1e-4


def frame_against_eager(frame, params: Dict[str, Tensor], cams: Sequence[Hn.Camera], grids: Sequence[Tensor], skies: Sequence[Tensor],
                        targets: Sequence[Tensor], factors: Sequence[int], img_indices=None) -> dict:
    """One more ``frame.step()`` and one eager frame over the same views; returns ``{image_bit_equal, image_max_abs_err,
    loss_abs_err, grad_norm_rel_err, sky_grad_rel_err, pose_grad_rel_err, ok}`` (``ok``: images bit-equal, per-view loss within 1e-5,
    norm-relative error of the frame's flat gradient -- every parameter row and the grids, summed over the views -- within 1e-4;
    gradients differ by the order of their float atomics only)."""
    V = len(cams)
    img_indices = list(range(V)) if img_indices is None else list(img_indices)
    assert frame.step() is True, "selfcheck: the frame overflowed its list capacities"
    torch.cuda.synchronize()
    rgb_g = [vg.rgb.detach().clone() for vg in frame.views]
    loss_g = [float(vg.loss) for vg in frame.views]
    grads_g = [view.detach().clone() for view in frame.flat._views]     # (per parameter: its slice of the frame's flat buffer)
    sky_g = [s.grad.detach().clone() if s.grad is not None else None for s in skies]
    pose_g = [c.viewmat.grad.detach().clone() if c.viewmat.grad is not None else None for c in cams]
    flat_params = list(frame.flat.params)
    keep_sh, FV.SH_IN_PACK = FV.SH_IN_PACK, FV.SH_IN_PACK_DEV     # (bit-equal images: both drivers on the pack's SH arithmetic)
    try:
        for t in flat_params + list(skies) + [c.viewmat for c in cams]:
            t.grad = None
        bit_equal, max_abs, loss_err = True, 0.0, 0.0
        for v, cam in enumerate(cams):
            out = Hn.render_view(params, cam, grids, img_indices[v], skies[v], factors=factors)
            loss = Hn.training_loss(out, targets[v], grids, tv_weight=frame.tv_weight)
            loss.backward()
            bit_equal = bit_equal and bool(torch.equal(out["rgb"], rgb_g[v]))
            max_abs = max(max_abs, float((out["rgb"] - rgb_g[v]).abs().max()))
            loss_err = max(loss_err, abs(float(loss) - loss_g[v]))
            del out, loss
        torch.cuda.synchronize()
        num = den = 0.0
        for p, got in zip(flat_params, grads_g):
            ref = p.grad if p.grad is not None else torch.zeros_like(p)
            num += float((got.reshape(-1).double() - ref.reshape(-1).double()).pow(2).sum())
            den += float(ref.reshape(-1).double().pow(2).sum())

        def rel(a, b):
            worst = 0.0
            for x, y in zip(a, b):
                if x is None or y is None:
                    continue
                worst = max(worst, float((x.double() - y.double()).norm() / y.double().norm().clamp(min=1e-30)))
            return worst
        sky_err = rel(sky_g, [s.grad for s in skies])
        pose_err = rel(pose_g, [c.viewmat.grad for c in cams])
    finally:
        FV.SH_IN_PACK = keep_sh
        for t in flat_params + list(skies) + [c.viewmat for c in cams]:
            t.grad = None
        frame._point_grads_at_flat()          # the frame's in-place gradient rows again ...
        for v, vg in enumerate(frame.views):  # ... and the graphs' static sky / pose gradient outputs
            skies[v].grad, cams[v].viewmat.grad = vg.v_sky, vg.v_viewmat
    grad_err = (num / max(den, 1e-300)) ** 0.5
    res = {"views": V, "image_bit_equal": bit_equal, "image_max_abs_err": max_abs, "loss_abs_err": loss_err,
           "grad_norm_rel_err": grad_err, "sky_grad_rel_err": sky_err, "pose_grad_rel_err": pose_err,
           "bounds": {"image": "bit-equal", "loss_abs": LOSS_TOL, "grad_norm_rel": GRAD_TOL}}
    res["ok"] = bool(bit_equal and loss_err <= LOSS_TOL and grad_err <= GRAD_TOL)
    return res
