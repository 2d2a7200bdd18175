#!/usr/bin/env python3
"""What the compositor kernels' tails cost and what splitting the longest tiles would buy: per-tile work (entries of its exact 16-px
list a tile's wave walks: forward = up to its early exit, backward = up to its deepest blended entry) of the benchmark's six views, then a
list-scheduling simulation of the waves over the chip's wave slots (every XCD its contiguous range of tiles and an eighth of the slots):
spatial order, longest first, and longest first with the longest tiles split into two half-tile waves (each `split_cost` of the tile)."""
import heapq, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bilateral_driving_amd import harness as Hn, gs_ops as ops, _lib as L
dev = torch.device("cuda", 0)
W, H, N = 1920, 1080, 2_000_000
cams = Hn.ring_cameras(W, H, device=dev)
p = Hn.synthetic_scene(N, seed=0, device=dev)
tw, th = math.ceil(W / 16), math.ceil(H / 16)
total = tw * th


def makespan(idx, cost, slots):
    h = [0.0] * slots
    heapq.heapify(h)
    for i in idx:
        heapq.heappush(h, heapq.heappop(h) + cost[i])
    return max(h)


def simulate(work, slots, fixed, split_cost, split_frac):
    cost = [float(w) + fixed for w in work]
    per = total // 8
    out = {}
    for name in ("spatial", "longest first", "longest first + split"):
        span = 0.0
        for x in range(8):
            idx = list(range(x * per, (x + 1) * per if x < 7 else total))
            c = cost
            if name != "spatial":
                idx.sort(key=lambda i: -cost[i])
            if name.endswith("split"):
                k = max(1, int(len(idx) * split_frac))
                c = list(cost)
                extra = []
                for i in idx[:k]:           # the k longest tiles: two half-tile waves each
                    c.append(cost[i] * split_cost); extra.append(len(c) - 1)
                    c[i] = cost[i] * split_cost
                idx = sorted(idx + extra, key=lambda i: -c[i])
            span = max(span, makespan(idx, c, slots // 8))
        out[name] = span
    return out, sum(cost) / slots, max(cost)


with torch.no_grad():
    for v, cam in enumerate(cams[:3]):
        radii, m2, dep, con, _ = ops.fully_fused_projection(p["means"], p["quats"], torch.exp(p["log_scales"]), cam.viewmat[None], cam.K[None], W, H,
                                                            near_plane=0.1)
        op = torch.sigmoid(p["opacity_logits"])[None].contiguous()
        _, _, fids, offs = ops.isect_tiles(m2, radii, dep, 16, tw, th, want_isect_ids=False, conics=con, opacities=op)
        M = fids.numel()
        col = torch.rand(1, N, 4, device=dev)
        rec = torch.empty(N, L.SPLAT_RECORD_FLOATS, device=dev)
        st = L.stream()
        L.check(L.lib().bds_splat_pack(N, 4, None, L.ptr(m2), L.ptr(con), L.ptr(col), L.ptr(op), L.ptr(radii), L.ptr(rec), st), "pack")
        render, alphas = torch.empty(1, H, W, 4, device=dev), torch.empty(1, H, W, 1, device=dev)
        last = torch.zeros(1, H, W, dtype=torch.int32, device=dev)
        L.check(L.lib().bds_rasterize_fwd(1, N, M, 4, L.ptr(rec), None, W, H, 16, 16, tw, th, L.ptr(offs), L.ptr(fids), L.ptr(render), L.ptr(alphas),
                                          None, L.ptr(last), st), "fwd")
        order = ops.bwd_schedule(1, W, H, 16, offs, last)
        torch.cuda.synchronize()
        work = order[1 + total:1 + 2 * total].cpu().tolist()
        o = offs.reshape(-1).cpu().long()
        length = (torch.cat([o[1:], torch.tensor([M])]) - o).tolist()
        ws = sorted(work)
        print(f"view {v}: M {M}, walked entries per tile: mean {sum(work) / total:.0f}, median {ws[total // 2]}, 99 % {ws[int(total * 0.99)]}, max {ws[-1]}; "
              f"list length mean {sum(length) / total:.0f} max {max(length)}")
        for kernel, slots, fixed, sc in (("backward (5 waves / SIMD)", 256 * 4 * 5, 30.0, 0.65), ("forward (7 waves / SIMD)", 256 * 4 * 7, 15.0, 0.62)):
            for frac in (0.02, 0.05):
                res, ideal, longest = simulate(work, slots, fixed, sc, frac)
                print(f"  {kernel}, split {frac:.0%}: ideal {ideal:.0f}, longest tile {longest:.0f} | " +
                      "  ".join(f"{k} {s:.0f} ({s / ideal:.2f}x)" for k, s in res.items()))
