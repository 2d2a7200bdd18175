"""-m gpu: one refinement step (bilateral_driving_amd.densify.refinement_after -> csrc/refine.hip) against the pinned CPU oracle
(oracle/refine_oracle.py; models/gaussians/vanilla.py:193-376) at RANDOM sizes, steps and controls: before / inside / after the
densification window, on and off the opacity-reset and screen-size schedules, thresholds that split or cull nearly everything or
nothing, one to three split samples, torch.optim.Adam and FusedAdam.  Rows bit-exact where the oracle's are (tests/test_gpu_11
holds the reference's own goldens and the sizes around the scan's segment boundary).

BDS_SWEEP_CASES (default 12) cases."""
import os

import numpy as np
import pytest
import torch

from oracle import refine_oracle as RO
from tests.test_gpu_11_refine import CTRL, build_model, check, synthetic

pytestmark = pytest.mark.gpu
N_CASES = int(os.environ.get("BDS_SWEEP_CASES", "12"))


@pytest.mark.parametrize("seed", list(range(N_CASES)))
def test_refinement_random_case(seed):
    from bilateral_driving_amd.densify import refinement_after
    from bilateral_driving_amd.optim import FusedAdam
    g = np.random.default_rng(5000 + seed)
    N = int(np.exp(g.uniform(np.log(1), np.log(120_000))))
    ctrl = dict(CTRL)
    ctrl["n_split_samples"] = int(g.integers(1, 4))
    ctrl["densify_grad_thresh"] = float(g.choice([0.0, 0.0003, 0.002, 1.0]))
    ctrl["densify_size_thresh"] = float(g.choice([0.0005, 0.002, 0.02]))
    ctrl["cull_alpha_thresh"] = float(g.choice([0.0, 0.005, 0.3]))
    ctrl["cull_scale_thresh"] = float(g.choice([0.05, 0.5, 5.0]))
    ctrl["refine_interval"] = int(g.choice([100, 200]))
    # steps: warm-up, an ordinary refinement, just after an opacity reset, past the screen-size schedule, past the last split
    step = int(g.choice([300, 1300, 3100, 3300, 4100, 6200, 15100, 16300]))
    step -= step % ctrl["refine_interval"]
    scene_scale, n_img = float(g.choice([1.0, 30.0])), int(g.choice([6, 150]))
    P, M, V, stats = synthetic(N, seed=seed * 7 + 1)
    opt_cls = torch.optim.Adam if seed % 2 else FusedAdam
    model, opt = build_model(P, M, V, stats, ctrl, scene_scale, n_img, step, opt_cls)
    sch = RO.schedule(step, ctrl, scene_scale, n_img)
    n_split = 0
    if sch.get("do_densify"):
        n_split = int(RO.plan(sch, ctrl, P["_scales"], P["_opacities"], stats["xys_grad_norm"], stats["vis_counts"], stats["max_2Dsize"])[0].sum())
    ns_ = ctrl["n_split_samples"]
    samples = np.random.default_rng(seed + 1).standard_normal((ns_ * n_split, 3)).astype(np.float32)
    eP, eM, eV, ns = RO.refine(step, ctrl, scene_scale, n_img, P, M, V, stats["xys_grad_norm"], stats["vis_counts"], stats["max_2Dsize"], samples)
    assert ns == n_split
    refinement_after(model, step, opt, samples=torch.from_numpy(samples), verbose=False)
    tag = dict(seed=seed, N=N, step=step, n_split=n_split, rows_after=int(eP["_means"].shape[0]), **{k: ctrl[k] for k in ("n_split_samples", "densify_grad_thresh", "densify_size_thresh", "cull_alpha_thresh", "cull_scale_thresh")})
    print("[refine sweep]", tag)
    if not sch["active"]:      # warm-up (vanilla.py:208-209 returns at once): nothing moves, the statistics keep accumulating
        for a in RO.PARAMS:
            np.testing.assert_array_equal(getattr(model, a).detach().cpu().numpy(), P[a], err_msg=a)
        assert model.xys_grad_norm is not None and model.vis_counts is not None and model.max_2Dsize is not None
        return
    check(model, opt, eP, eM, eV, means_atol=5e-5)
