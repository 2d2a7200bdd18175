#!/usr/bin/env python3
"""What keeping the rows in spatial order costs and buys over a training run's refinements (What's weak 5 / Next 8).

Full-size loop (2 M Gaussians, six 1920x1080 views, the bench scene in Morton order): frames of six views with the densification
statistics of every view accumulated; after every ``--interval`` frames ``densify.refinement_after`` (split / duplicate / cull: the
reference appends a step's children at the END of every array, models/gaussians/vanilla.py:256-262) -- once with ``reorder=False``
(the reference's order) and once with ``reorder=True`` (``reorder_rows`` into ``spatial_order`` after every step that changed the set).
Reported: it/s (fwd + bwd of the frame, graph replay) before the first refinement and after each one, the set's size, and what
``spatial_order`` + ``reorder_rows`` cost at that size with the Adam moments moved along.

    python scripts/refine_reorder_loop.py [--gaussians N] [--interval 100] [--refinements 2] [--workload headline|c5]"""
import argparse, json, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bilateral_driving_amd import _lib as L, harness as Hn
from bilateral_driving_amd.densify import refinement_after, reorder_rows, spatial_order
from bilateral_driving_amd.graph_view import FrameGraph
from bilateral_driving_amd.optim import FusedAdam

ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", type=int, default=2_000_000)
ap.add_argument("--interval", type=int, default=100)
ap.add_argument("--refinements", type=int, default=2)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
a = ap.parse_args()
dev = torch.device("cuda", 0)
W, H, N0 = a.width, a.height, a.gaussians
cams = Hn.ring_cameras(W, H, device=dev)
V = len(cams)
gen = torch.Generator().manual_seed(7)
skies = [torch.rand(H, W, 3, generator=gen).to(dev) for _ in cams]
targets = [torch.rand(H, W, 3, generator=gen).to(dev) for _ in cams]
grids0 = Hn.make_grids(V, device=dev)
base = Hn.synthetic_scene(N0, seed=0, device=dev)
base = {k: v[Hn.spatial_order(base["means"])].contiguous() for k, v in base.items()}
NAMES = dict(_means="xyz", _features_dc="sh_dc", _features_rest="sh_rest", _opacities="opacity", _scales="scaling", _quats="rotation")
LRS = dict(_means=1.6e-4, _features_dc=2.5e-3, _features_rest=1.25e-4, _opacities=5e-2, _scales=5e-3, _quats=1e-3)


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def run(reorder: bool):
    P = torch.nn.Parameter
    m = types.SimpleNamespace(
        class_prefix="Background#", scene_scale=30.0, num_train_images=V, step=0, xys_grad_norm=None, vis_counts=None, max_2Dsize=None,
        # the reference's controls (configs/omnire_ms_bilateral_extended.yaml:57-72)
        ctrl_cfg=types.SimpleNamespace(warmup_steps=0, reset_alpha_interval=3000, refine_interval=a.interval, n_split_samples=2,
                                       reset_alpha_value=0.01, densify_grad_thresh=0.0005, densify_size_thresh=0.003, cull_alpha_thresh=0.005,
                                       cull_scale_thresh=0.5, cull_screen_size=0.15, split_screen_size=0.05, stop_screen_size_at=4000,
                                       stop_split_at=15000))
    m._means, m._quats, m._scales = P(base["means"].clone()), P(base["quats"].clone()), P(base["log_scales"].clone())
    m._opacities = P(base["opacity_logits"].reshape(-1, 1).clone())
    m._features_dc, m._features_rest = P(base["sh"][:, 0, :].contiguous()), P(base["sh"][:, 1:, :].contiguous())
    opt = FusedAdam([{"params": [getattr(m, k)], "name": m.class_prefix + n, "lr": LRS[k], "eps": 1e-15} for k, n in NAMES.items()], lr=0.0, eps=1e-15)
    for k in NAMES:     # moments exist (one step on zero gradients): the refinement moves them along, as in a real run
        getattr(m, k).grad = torch.zeros_like(getattr(m, k))
    opt.step()
    out = {"reorder": reorder, "phases": []}
    for phase in range(a.refinements + 1):
        N = m._means.shape[0]
        p = {"means": m._means.detach().clone().requires_grad_(True), "quats": m._quats.detach().clone().requires_grad_(True),
             "log_scales": m._scales.detach().clone().requires_grad_(True), "opacity_logits": m._opacities.detach().reshape(-1).clone().requires_grad_(True),
             "sh": torch.cat((m._features_dc.detach()[:, None, :], m._features_rest.detach()), dim=1).contiguous().requires_grad_(True)}
        grids = [g.clone().requires_grad_(True) for g in grids0]
        frame = FrameGraph(p, cams, grids, [s.clone() for s in skies], targets)
        stats = [torch.zeros(N, device=dev) for _ in range(3)]
        for _ in range(3):
            assert frame.step() is True
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(a.interval):
            frame.step(wait=False)
            for v in range(V):    # the view's densification statistics (trainers/base.py:279-297)
                L.check(L.lib().bds_densify_stats(N, L.ptr(frame.g2d[v][1]), L.ptr(frame.views[v].out["radii"]), W, H, 1, max(W, H),
                                                  int(it == 0 and v == 0), L.ptr(stats[0]), L.ptr(stats[1]), L.ptr(stats[2]), L.stream()), "stats")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert frame.valid()
        ph = {"phase": phase, "gaussians": N, "iters_per_sec_incl_stats": V * a.interval / dt, "n_visible_mean": sum(c[1] for c in frame.counts()) / V}
        t1 = time.perf_counter()
        for _ in range(20):
            frame.step(wait=False)
        torch.cuda.synchronize()
        ph["iters_per_sec"] = V * 20 / (time.perf_counter() - t1)
        del frame
        if phase < a.refinements:
            m.step = 600 + phase * a.interval      # (past warm-up, inside the densification window of the reset interval)
            m.xys_grad_norm, m.vis_counts, m.max_2Dsize = stats
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            refinement_after(m, m.step, opt, verbose=False, reorder=reorder)
            torch.cuda.synchronize()
            ph["refinement_ms"] = (time.perf_counter() - t2) * 1e3
            ph["gaussians_after"] = m._means.shape[0]
        out["phases"].append(ph)
        print(json.dumps(ph), flush=True)
    # what the reorder itself costs at the final size (parameters + both Adam moments: 3 x 59 floats per row gathered)
    N = m._means.shape[0]
    out["spatial_order_ms"] = timed(lambda: spatial_order(m._means)) * 1e3
    perm = spatial_order(m._means)
    out["reorder_rows_ms"] = timed(lambda: reorder_rows(m, opt, perm)) * 1e3
    out["reorder_rows_gaussians"] = N
    out["reorder_rows_bytes"] = N * 59 * 4 * 3 * 2
    return out


res = {"interval": a.interval, "runs": [run(False), run(True)]}
print("REORDER_LOOP " + json.dumps(res))
