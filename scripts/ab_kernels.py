#!/usr/bin/env python3
"""In-process interleaved A/B timing of kernel variants on the benchmark scene (rule: perf deltas come
from interleaved rounds in ONE process).  Prints workload statistics (M, visible, traversal depth)
and per-variant median / min times.

    python scripts/ab_kernels.py [--gaussians 2000000] [--rounds 7]
"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bilateral_driving_amd import _lib as L  # noqa: E402
from bilateral_driving_amd import harness as Hn  # noqa: E402


def timed(fn, rounds):
    ts = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return ts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=2_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--view", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    L.lib()
    cams = Hn.ring_cameras(a.width, a.height, device=dev)
    params = Hn.synthetic_scene(a.gaussians, seed=0, device=dev)
    for v in params.values():
        v.requires_grad_(True)
    grids = [g.requires_grad_(True) for g in Hn.make_grids(len(cams), device=dev)]
    gen = torch.Generator().manual_seed(7)
    sky = torch.rand(a.height, a.width, 3, generator=gen).to(dev)
    target = torch.rand(a.height, a.width, 3, generator=gen).to(dev)
    cam = cams[a.view]

    Hn.TILE_CULL = False
    out = Hn.render_view(params, cam, grids, a.view, sky)
    info = out["info"]
    M = info["flatten_ids"].numel()
    offs = info["isect_offsets"].reshape(-1).long()
    cnt = torch.diff(torch.cat([offs, torch.tensor([M], device=dev)]))
    # traversal depth per pixel = last blended index - tile start (+1), from the forward's last_ids
    stats = {"N": a.gaussians, "M": M, "visible": int((info["radii"] > 0).sum()), "tiles": int(offs.numel()),
             "isect_per_tile_mean": float(cnt.float().mean()), "isect_per_tile_max": int(cnt.max()),
             "isect_per_tile_p99": float(torch.quantile(cnt.float(), 0.99)),
             "coverage_mean_alpha": float(out["opacity"].mean())}
    print("WORKLOAD", json.dumps(stats))

    loss = Hn.training_loss(out, target, grids)
    res = {}

    def fwd_bwd():
        o = Hn.render_view(params, cam, grids, a.view, sky)
        Hn.training_loss(o, target, grids).backward()

    # whole step, per backward variant x radix variant
    for rb in (3, 2):
        for rx in (2,):
            L.set_option(L.OPT_RASTER_BWD, rb)
            L.set_option(L.OPT_RADIX, rx)
            fwd_bwd()
    order = [(2, 2, True, 1), (3, 2, True, 2), (2, 1, True, 1), (2, 2, False, 1)]
    for r in range(a.rounds):
        for rb, rx, cull, fw in order:
            Hn.TILE_CULL = cull
            L.set_option(L.OPT_RASTER_FWD, fw)
            L.set_option(L.OPT_RASTER_BWD, rb)
            L.set_option(L.OPT_RADIX, rx)
            L.enable_timers(True)
            t = timed(fwd_bwd, 1)[0]
            torch.cuda.synchronize()
            ts = L.timer_summary()
            L.enable_timers(False)
            d = res.setdefault(f"bwd{rb}_radix{rx}_cull{int(cull)}_fwd{fw}", {"step": [], "rasterize_bwd": [], "rasterize_fwd": [], "isect_prepare": [],
                                                     "isect_build": [], "bilagrid_bwd": [], "bilagrid_fwd": [], "sh_fwd": [], "sh_bwd": [],
                                                     "project_fwd": [], "project_bwd": []})
            d["step"].append(t)
            for k in d:
                if k != "step" and k in ts:
                    d[k].append(ts[k][1])
    L.set_option(L.OPT_RASTER_BWD, 2)
    L.set_option(L.OPT_RADIX, 2)
    L.set_option(L.OPT_RASTER_FWD, 1)
    Hn.TILE_CULL = True
    out = Hn.render_view(params, cam, grids, a.view, sky)
    print("CULLED_M", out["info"]["flatten_ids"].numel())
    for name, d in res.items():
        print("AB", name, json.dumps({k: {"median": round(statistics.median(v), 4), "min": round(min(v), 4)} for k, v in d.items() if v}))


if __name__ == "__main__":
    main()
