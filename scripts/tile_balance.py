#!/usr/bin/env python3
"""How well does the one-wave-per-tile composite fill 8 XCDs x 128 SIMDs?

Per-tile work = Gaussians visited (max over the tile's pixels of the termination index).  A small
processor-sharing simulation (4 resident waves per SIMD share its issue slots) gives the makespan of a
dispatch order relative to the perfectly balanced one.  Run on the GPU box:  python scripts/tile_balance.py
"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bilateral_driving_amd import _lib as L, harness as Hn
import bilateral_driving_amd.gs_ops as ops


def tile_work(view: int, N=2_000_000, W=1920, H=1080):
    dev = "cuda"
    cam = Hn.ring_cameras(W, H, device=dev)[view]
    p = Hn.synthetic_scene(N, seed=0, device=dev)
    opac = torch.sigmoid(p["opacity_logits"]); scales = torch.exp(p["log_scales"])
    radii, m2, d, con, _ = ops.fully_fused_projection(p["means"], p["quats"], scales, cam.viewmat[None], cam.K[None], W, H, near_plane=0.1)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    _, _, fids, offs = ops.isect_tiles(m2, radii, d, 16, tw, th, want_isect_ids=False, conics=con, opacities=opac[None])
    M = fids.numel()
    col = torch.rand(1, N, 4, device=dev)
    render = torch.empty(1, H, W, 4, device=dev); alphas = torch.empty(1, H, W, 1, device=dev)
    last = torch.empty(1, H, W, dtype=torch.int32, device=dev)
    L.check(L.lib().bds_rasterize_fwd(1, N, M, 4, L.ptr(m2), L.ptr(con), L.ptr(col), L.ptr(opac[None].contiguous()), None, W, H, 16, tw, th,
                                      L.ptr(offs), L.ptr(fids), L.ptr(render), L.ptr(alphas), L.ptr(last), L.stream()), "fwd")
    Hp, Wp = th * 16, tw * 16
    lid = torch.zeros(Hp, Wp, dtype=torch.long, device=dev); lid[:H, :W] = last[0].long()
    start = offs[0].long()
    depth = (lid.reshape(th, 16, tw, 16).permute(0, 2, 1, 3).reshape(th, tw, 256) - start[..., None] + 1).clamp(min=0)
    work = depth.max(dim=-1).values.reshape(-1).cpu().numpy().astype(np.float64)
    end = torch.cat([offs.reshape(-1)[1:].long(), torch.tensor([M], device=dev)]).cpu().numpy().astype(np.float64)
    return work, end - start.reshape(-1).cpu().numpy(), tw, th


def simulate(jobs_per_xcd, simds=128, slots=4, fixed=40.0, dt=2.0):
    """jobs_per_xcd: list of 8 arrays (work in dispatch order).  Returns makespan."""
    worst = 0.0
    for q in jobs_per_xcd:
        q = list(q + fixed)
        rem = np.zeros((simds, slots)); qi = 0
        # initial fill, round-robin over SIMDs
        for s in range(slots):
            for i in range(simds):
                if qi < len(q): rem[i, s] = q[qi]; qi += 1
        t = 0.0
        while True:
            act = rem > 0
            n = act.sum(axis=1, keepdims=True)
            if n.sum() == 0: break
            rem = np.where(act, rem - dt / np.maximum(n, 1), rem)
            t += dt
            done = act & (rem <= 0)
            if done.any():
                rem[done] = 0
                if qi < len(q):
                    for i, s in zip(*np.nonzero(done)):
                        if qi < len(q): rem[i, s] = q[qi]; qi += 1
        worst = max(worst, t)
    return worst


def xcd_contiguous_order(work):
    n = len(work); per, remn = divmod(n, 8); out = []; s = 0
    for x in range(8):
        c = per + (1 if x < remn else 0); out.append(work[s:s + c]); s += c
    return out


def main():
    for view in (0, 3):
        work, listlen, tw, th = tile_work(view)
        tot = work.sum(); ideal = (tot + 40.0 * len(work)) / (8 * 128)
        print(f"view {view}: tiles {len(work)}  visited sum {tot:.0f}  mean {work.mean():.0f}  p50 {np.median(work):.0f}  p99 {np.percentile(work, 99):.0f}  max {work.max():.0f}"
              f"  list mean {listlen.mean():.0f} max {listlen.max():.0f}")
        cur = xcd_contiguous_order(work)
        print("  per-XCD work share (contiguous ranges):", " ".join(f"{c.sum() / tot * 8:.2f}" for c in cur))
        print(f"  makespan/ideal  current (XCD-contiguous, in order): {simulate(cur) / ideal:.3f}")
        rr = [work[x::8] for x in range(8)]
        print(f"  makespan/ideal  plain round-robin tile = block:     {simulate(rr) / ideal:.3f}")
        order = np.argsort(-work, kind="stable")
        lpt = [work[order][x::8] for x in range(8)]
        print(f"  makespan/ideal  longest-first, dealt over XCDs:     {simulate(lpt) / ideal:.3f}")
        # longest-first inside each XCD's contiguous range (keeps L2 locality of the range)
        lptc = [np.sort(c)[::-1] for c in cur]
        print(f"  makespan/ideal  longest-first inside XCD ranges:    {simulate(lptc) / ideal:.3f}")
        # row-interleaved XCD ownership (tile row r -> XCD r % 8), longest-first inside
        rows = np.arange(len(work)) // tw
        ri = [np.sort(work[rows % 8 == x])[::-1] for x in range(8)]
        print(f"  makespan/ideal  rows dealt to XCDs, longest-first:  {simulate(ri) / ideal:.3f}")
        # forward: the visited count is not known before the launch; the list length is
        print(f"  corr(visited, list length) = {np.corrcoef(work, listlen)[0, 1]:.3f}")
        byl = []; s0 = 0
        for c in cur:
            ll = listlen[s0:s0 + len(c)]; s0 += len(c)
            byl.append(c[np.argsort(-ll, kind="stable")])
        print(f"  makespan/ideal  longest LIST first inside XCD ranges: {simulate(byl) / ideal:.3f}")
        for slots in (2, 8):
            print(f"  slots/SIMD={slots}: current {simulate(cur, slots=slots) / ideal:.3f}  longest-first in range {simulate(lptc, slots=slots) / ideal:.3f}")
        ri2 = [work[rows % 8 == x] for x in range(8)]
        print(f"  makespan/ideal  rows dealt to XCDs, in order:       {simulate(ri2) / ideal:.3f}")


if __name__ == "__main__":
    main()
