#!/usr/bin/env python3
"""How ragged is per-pixel termination inside a tile?  (decides whether straggler-pixel compaction pays)"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bilateral_driving_amd import _lib as L, harness as Hn
import bilateral_driving_amd.gs_ops as ops

dev = "cuda"
N, W, H = 2_000_000, 1920, 1080
cam = Hn.ring_cameras(W, H, device=dev)[0]
p = Hn.synthetic_scene(N, seed=0, device=dev)
opac = torch.sigmoid(p["opacity_logits"]); scales = torch.exp(p["log_scales"])
radii, m2, d, con, _ = ops.fully_fused_projection(p["means"], p["quats"], scales, cam.viewmat[None], cam.K[None], W, H, near_plane=0.1)
tw, th = math.ceil(W / 16), math.ceil(H / 16)
_, _, fids, offs = ops.isect_tiles(m2, radii, d, 16, tw, th, want_isect_ids=False, conics=con, opacities=opac[None])
M = fids.numel()
col = torch.rand(1, N, 4, device=dev)
render = torch.empty(1, H, W, 4, device=dev); alphas = torch.empty(1, H, W, 1, device=dev); last = torch.empty(1, H, W, dtype=torch.int32, device=dev)
L.check(L.lib().bds_rasterize_fwd(1, N, M, 4, L.ptr(m2), L.ptr(con), L.ptr(col), L.ptr(opac[None].contiguous()), None, W, H, 16, tw, th,
                                  L.ptr(offs), L.ptr(fids), L.ptr(render), L.ptr(alphas), L.ptr(last), L.stream()), "fwd")
Hp, Wp = th * 16, tw * 16
lid = torch.zeros(Hp, Wp, dtype=torch.long, device=dev); lid[:H, :W] = last[0].long()
start = offs[0].long()                                       # [th, tw]
end = torch.cat([offs.reshape(-1)[1:].long(), torch.tensor([M], device=dev)]).reshape(th, tw)
depth = (lid.reshape(th, 16, tw, 16).permute(0, 2, 1, 3).reshape(th, tw, 256) - start[..., None] + 1).clamp(min=0)  # pairs visited per pixel
depth_sorted = depth.sort(dim=-1, descending=True).values
dmax = depth_sorted[..., 0].float(); d65 = depth_sorted[..., 64].float(); d17 = depth_sorted[..., 16].float()
listlen = (end - start).float()
print("M", M, "mean list", float(listlen.mean()), "mean max depth", float(dmax.mean()), "mean pixel depth", float(depth.float().mean()))
print("sum max depth (pairs visited per tile, bwd)", float(dmax.sum()), " sum d65", float(d65.sum()), "sum d17", float(d17.sum()))
now = float(dmax.sum()) * 4
comp = float(d65.sum()) * 4 + float((dmax - d65).sum()) * 1
print("strip-bodies now ~", now, " with 64-straggler compaction ~", comp, " ratio", now / comp)
print("ideal lane utilisation now:", float(depth.float().sum()) / (float(dmax.sum()) * 256))
