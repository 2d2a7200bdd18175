#!/usr/bin/env python3
"""Work statistics of the composite backward on the benchmark scene (measurement tooling): how many listed
(tile, Gaussian) pairs lie in front of the tile's deepest blended entry ("tested"), how many of them have at least one
pixel that blends the Gaussian ("visited"), how many 16x4 row strips / pixel pairs of a visited pair contain such a pixel,
and the number of (pixel, Gaussian) blends P (SURVEY.md appendix B work counter).  Runs on the GPU with torch ops only.

    python scripts/pair_stats.py [--gaussians 2000000] [--width 1920] [--height 1080] [--view 0]
"""
import argparse
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bilateral_driving_amd import _lib as L  # noqa: E402
from bilateral_driving_amd import gs_ops as ops  # noqa: E402
from bilateral_driving_amd import harness as Hn  # noqa: E402


def pair_stats(N, W, H, view=0, params=None):
    """``params``: the scene's raw parameters (default: the ring scene ``Hn.synthetic_scene(N, seed=0)``)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    cam = Hn.ring_cameras(W, H, device=dev)[view]
    p = Hn.synthetic_scene(N, seed=0, device=dev) if params is None else {k: v.detach() for k, v in params.items()}
    N = p["means"].shape[0]
    with torch.no_grad():
        radii, m2, dep, con, _ = ops.fully_fused_projection(p["means"], p["quats"], torch.exp(p["log_scales"]), cam.viewmat[None], cam.K[None],
                                                            W, H, near_plane=0.1)
        op = torch.sigmoid(p["opacity_logits"])[None].contiguous()
        tw, th = math.ceil(W / 16), math.ceil(H / 16)
        _, _, fids, offs = ops.isect_tiles(m2, radii, dep, 16, tw, th, want_isect_ids=False, conics=con, opacities=op)
        M = fids.numel()
        col = torch.rand(1, N, 3, device=dev)
        render, alphas = torch.empty(1, H, W, 3, device=dev), torch.empty(1, H, W, 1, device=dev)
        last = torch.zeros(1, H, W, dtype=torch.int32, device=dev)
        rec = torch.empty(N, L.SPLAT_RECORD_FLOATS, device=dev)
        L.check(L.lib().bds_splat_pack(N, 3, None, L.ptr(m2), L.ptr(con), L.ptr(col), L.ptr(op), None, L.ptr(rec), L.stream()), "pack")
        L.check(L.lib().bds_rasterize_fwd(1, N, M, 3, L.ptr(rec), None, W, H, 16, 16, tw, th, L.ptr(offs), L.ptr(fids), L.ptr(render),
                                          L.ptr(alphas), None, L.ptr(last), L.stream()), "fwd")
        n_tiles = tw * th
        start = offs.reshape(-1).long()
        end = torch.cat([start[1:], torch.tensor([M], device=dev)])
        Hp, Wp = th * 16, tw * 16
        lid = torch.full((Hp, Wp), -1, dtype=torch.long, device=dev)
        lid[:H, :W] = last[0].long()
        inside = torch.zeros(Hp, Wp, dtype=torch.bool, device=dev)
        inside[:H, :W] = True
        lid_t = lid.reshape(th, 16, tw, 16).permute(0, 2, 1, 3).reshape(n_tiles, 256)       # [tile, pixel (row-major in the tile)]
        ins_t = inside.reshape(th, 16, tw, 16).permute(0, 2, 1, 3).reshape(n_tiles, 256)
        tmax = lid_t.max(dim=1).values
        length = (tmax - start + 1).clamp(min=0)
        length = torch.where(end > start, length, torch.zeros_like(length))
        tested = int(length.sum())
        order = torch.argsort(length, descending=True)
        ys, xs = torch.meshgrid(torch.arange(16, device=dev), torch.arange(16, device=dev), indexing="ij")
        visited = strips = pairs2 = pix = quads = 0
        vis_tile = torch.zeros(n_tiles, dtype=torch.long, device=dev)
        B = 48
        m2f, conf, opf = m2[0], con[0], op[0]
        for b0 in range(0, n_tiles, B):
            tl = order[b0:b0 + B]
            Lmax = int(length[tl].max())
            if Lmax == 0:
                break
            k = torch.arange(Lmax, device=dev)
            idx = start[tl][:, None] + k[None, :]                      # [B, L] position in the list
            ok = k[None, :] < length[tl][:, None]
            g = fids[idx.clamp(max=M - 1)].long()
            ty, tx = tl // tw, tl % tw
            px = (tx[:, None] * 16 + xs.reshape(-1)[None, :]).float() + 0.5    # [B,256]
            py = (ty[:, None] * 16 + ys.reshape(-1)[None, :]).float() + 0.5
            dx = m2f[g][:, :, 0:1] - px[:, None, :]
            dy = m2f[g][:, :, 1:2] - py[:, None, :]
            c = conf[g]
            sigma = 0.5 * (c[:, :, 0:1] * dx * dx + c[:, :, 2:3] * dy * dy) + c[:, :, 1:2] * dx * dy
            alpha = torch.clamp(opf[g][:, :, None] * torch.exp(-sigma), max=0.999)
            valid = (sigma >= 0) & (alpha >= 1.0 / 255.0) & (idx[:, :, None] <= lid_t[tl][:, None, :]) & ins_t[tl][:, None, :] & ok[:, :, None]
            pix += int(valid.sum())
            v4 = valid.reshape(valid.shape[0], valid.shape[1], 4, 4, 16)          # [.., strip q (rows 4q..4q+3), row in strip, col]
            anyp = valid.any(dim=2)
            visited += int(anyp.sum())
            vis_tile[tl] = anyp.sum(dim=1)
            strips += int(v4.any(dim=4).any(dim=3).sum())
            # lane-pixel q of the wave kernel = row (lane >> 4) + 4q: pixel pairs (q0,q1), (q2,q3) as a packed kernel would group them
            vq = valid.reshape(valid.shape[0], valid.shape[1], 4, 4, 16)          # rows r = 4*q + l  -> index [q, l]
            pairs2 += int(vq[:, :, 0:2].any(dim=4).any(dim=3).any(dim=2).sum()) + int(vq[:, :, 2:4].any(dim=4).any(dim=3).any(dim=2).sum())
            quads += int(valid.reshape(valid.shape[0], valid.shape[1], 2, 8, 2, 8).any(dim=5).any(dim=3).sum())
        out = dict(gaussians=N, width=W, height=H, view=view, n_visible=int((radii > 0).sum()), isects_listed=M, pairs_tested=tested,
                   pairs_visited=visited, strips_visited=strips, halves_visited=pairs2, quadrants_visited=quads, pixel_blends=pix,
                   mean_strips_per_visited_pair=strips / max(visited, 1), mean_halves_per_visited_pair=pairs2 / max(visited, 1),
                   mean_pixels_per_visited_pair=pix / max(visited, 1))
        vt = vis_tile.float()
        q = torch.tensor([0.5, 0.9, 0.99, 0.999], device=dev)
        out["visited_per_tile"] = dict(mean=float(vt.mean()), max=int(vt.max()), quantiles_50_90_99_999=[float(x) for x in torch.quantile(vt, q)],
                                       tested_max=int(length.max()), tested_mean=float(length.float().mean()))
        # per-SIMD load if tiles are dealt longest-first to 1024 SIMDs x k resident waves (LPT bound): sum / 1024 vs the longest tile
        out["lpt"] = dict(visited_sum_over_1024=float(vt.sum() / 1024), longest_tile=int(vt.max()))
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=2_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--view", type=int, default=0)
    ap.add_argument("--scene", choices=("ring", "lidar"), default="ring")
    a = ap.parse_args()
    params = None
    if a.scene == "lidar":
        params = Hn.lidar_scene(a.gaussians if a.gaussians != 2_000_000 else 1_000_000, seed=0, device=torch.device("cuda", torch.cuda.current_device()))
    print(json.dumps(pair_stats(a.gaussians, a.width, a.height, a.view, params=params)))


if __name__ == "__main__":
    main()
