"""Oracle (TEST INFRASTRUCTURE, not product code): CPU restatement of the rasterizer half of
the hot path -- SH evaluation, 3D->2D projection, tile intersection, (tile|depth) sort,
front-to-back alpha compositing, expected depth -- in differentiable torch ops, so that
autograd of this file is the gradient reference.

PARITY UNPINNED.  The arithmetic being restated lives in gsplat v1.3.0, a pip git
dependency of the reference (/root/reference/README.md:81) that is neither vendored in the
reference tree nor installable in this image (no network, no CUDA).  There are no golden
vectors for it; the only anchors the reference tree itself offers -- its pinhole projection helper
(utils/geometry.py:7-21,39-57) and SH2RGB (models/gaussians/basics.py:84-89) -- pin the world -> camera
transform, means2d / depths and the degree-0 colour (tests/golden/ref_geometry.npz,
oracle/gen_golden_geometry.py), nothing else.  This file follows the published 3DGS / gsplat algorithm
and the reference's call-site contract:

  * project/models/trainers/base.py:393-408     rasterization(...) call + kwargs
  * project/models/trainers/base.py:409-419     split / clamp / alpha post-processing
  * project/models/trainers/base.py:280-297     consumers of meta["means2d"].absgrad, meta["radii"]
  * project/models/gaussians/vanilla.py:383-389 spherical_harmonics(n, dirs, coeffs) + 0.5, clamp
  * project/models/gaussians/basics.py:12-15    import surface
  * project/models/gaussians/basics.py:76-89    C0 = 0.28209479177387814 (the only SH constant the
                                                reference itself pins)

Constants (eps2d 0.3, alpha cut 1/255, alpha max 0.999, T stop 1e-4, 3-sigma radius,
1.3x FOV clamp, radius discriminant floor 0.01) are the published gsplat/3DGS values and are
this repo's own pinned choice.
"""
from __future__ import annotations

import math

import torch

ALPHA_MIN = 1.0 / 255.0
ALPHA_MAX = 0.999
T_STOP = 1e-4


# --------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------
def num_sh_bases(degree: int) -> int:
    return (degree + 1) ** 2


def quat_to_rotmat(quats: torch.Tensor) -> torch.Tensor:
    """wxyz quaternion (normalised inside) -> rotation matrix [..., 3, 3]."""
    q = quats / quats.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack(
        [
            1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y),
        ],
        dim=-1,
    )
    return R.reshape(*quats.shape[:-1], 3, 3)


# --------------------------------------------------------------------------------------
# spherical harmonics (real SH, degree <= 3, Sloan's recurrences as used by gsplat)
# --------------------------------------------------------------------------------------
def sh_bases(degree: int, dirs: torch.Tensor) -> torch.Tensor:
    """dirs [..., 3] (normalised inside) -> basis values [..., (degree+1)^2]."""
    d = dirs / dirs.norm(dim=-1, keepdim=True)
    x, y, z = d.unbind(-1)
    out = [torch.full_like(x, 0.2820947917738781)]
    if degree >= 1:
        out += [-0.48860251190292 * y, 0.48860251190292 * z, -0.48860251190292 * x]
    if degree >= 2:
        z2 = z * z
        fTmp0B = -1.092548430592079 * z
        fC1 = x * x - y * y
        fS1 = 2 * x * y
        out += [0.5462742152960395 * fS1, fTmp0B * y, 0.9461746957575601 * z2 - 0.3153915652525201,
                fTmp0B * x, 0.5462742152960395 * fC1]
    if degree >= 3:
        fTmp0C = -2.285228997322329 * z2 + 0.4570457994644658
        fTmp1B = 1.445305721320277 * z
        fC2 = x * fC1 - y * fS1
        fS2 = x * fS1 + y * fC1
        out += [-0.5900435899266435 * fS2, fTmp1B * fS1, fTmp0C * y,
                z * (1.865881662950577 * z2 - 1.119528997770346), fTmp0C * x, fTmp1B * fC1,
                -0.5900435899266435 * fC2]
    if degree >= 4:
        raise NotImplementedError("reference uses sh_degree <= 3")
    return torch.stack(out, dim=-1)


def sh_bases_closed_form(dirs: torch.Tensor) -> torch.Tensor:
    """Independent closed form of the 16 degree-3 bases used by the
    known-answer test of ``sh_bases``."""
    d = dirs / dirs.norm(dim=-1, keepdim=True)
    x, y, z = d.unbind(-1)
    xx, yy, zz = x * x, y * y, z * z
    C1, C2a, C2b, C2c = 0.4886025119029199, 1.0925484305920792, 0.31539156525252005, 0.5462742152960396
    return torch.stack(
        [
            torch.full_like(x, 0.28209479177387814),
            -C1 * y, C1 * z, -C1 * x,
            C2a * x * y, -C2a * y * z, C2b * (2 * zz - xx - yy), -C2a * x * z, C2c * (xx - yy),
            -0.5900435899266435 * y * (3 * xx - yy), 2.890611442640554 * x * y * z,
            -0.4570457994644658 * y * (4 * zz - xx - yy), 0.3731763325901154 * z * (2 * zz - 3 * xx - 3 * yy),
            -0.4570457994644658 * x * (4 * zz - xx - yy), 1.445305721320277 * z * (xx - yy),
            -0.5900435899266435 * x * (xx - 3 * yy),
        ],
        dim=-1,
    )


def spherical_harmonics(degrees_to_use: int, dirs: torch.Tensor, coeffs: torch.Tensor, masks=None) -> torch.Tensor:
    """dirs [..., 3], coeffs [..., K, 3] -> colours [..., 3] (vanilla.py:388 adds 0.5 and clamps)."""
    nb = num_sh_bases(degrees_to_use)
    assert coeffs.shape[-2] >= nb
    B = sh_bases(degrees_to_use, dirs)  # [..., nb]
    out = (B[..., :, None] * coeffs[..., :nb, :]).sum(-2)
    if masks is not None:
        out = out * masks[..., None]
    return out


# --------------------------------------------------------------------------------------
# projection (one camera)
# --------------------------------------------------------------------------------------
def project(means, quats, scales, viewmat, K, width, height, eps2d=0.3, near_plane=0.01, far_plane=1e10,
            radius_clip=0.0, calc_compensations=False):
    """means [N,3], quats [N,4] wxyz, scales [N,3], viewmat [4,4] world->cam, K [3,3].

    Returns radii [N] int32 (0 = culled), means2d [N,2], depths [N], conics [N,3] (a,b,c of the
    inverse 2D covariance), compensations [N] or None.  Culled entries are zero."""
    dt = means.dtype
    R = viewmat[:3, :3]
    t = viewmat[:3, 3]
    mc = means @ R.T + t
    x, y, z = mc.unbind(-1)
    Rq = quat_to_rotmat(quats)
    M = Rq * scales[:, None, :]
    cov = M @ M.transpose(-1, -2)
    cov_c = R @ cov @ R.T
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    lim_x = 1.3 * (0.5 * width / fx)
    lim_y = 1.3 * (0.5 * height / fy)
    valid = (z > near_plane) & (z < far_plane)
    zs = torch.where(valid, z, torch.ones_like(z))
    rz = 1.0 / zs
    rz2 = rz * rz
    tx = zs * torch.minimum(lim_x, torch.maximum(-lim_x, x * rz))
    ty = zs * torch.minimum(lim_y, torch.maximum(-lim_y, y * rz))
    zero = torch.zeros_like(rz)
    J = torch.stack([fx * rz, zero, -fx * tx * rz2, zero, fy * rz, -fy * ty * rz2], dim=-1).reshape(-1, 2, 3)
    cov2 = J @ cov_c @ J.transpose(-1, -2)
    det_orig = cov2[:, 0, 0] * cov2[:, 1, 1] - cov2[:, 0, 1] * cov2[:, 1, 0]
    cov2 = cov2 + eps2d * torch.eye(2, dtype=dt)
    det = cov2[:, 0, 0] * cov2[:, 1, 1] - cov2[:, 0, 1] * cov2[:, 1, 0]
    valid = valid & (det > 0)
    dets = torch.where(valid, det, torch.ones_like(det))
    inv = torch.stack([cov2[:, 1, 1], -cov2[:, 0, 1], -cov2[:, 1, 0], cov2[:, 0, 0]], dim=-1).reshape(-1, 2, 2) / dets[:, None, None]
    # conic stored as (a, b, c); b takes both off-diagonals' gradient through the symmetric inverse
    conics = torch.stack([inv[:, 0, 0], 0.5 * (inv[:, 0, 1] + inv[:, 1, 0]), inv[:, 1, 1]], dim=-1)
    b = 0.5 * (cov2[:, 0, 0] + cov2[:, 1, 1])
    v1 = b + torch.sqrt(torch.clamp(b * b - det, min=0.01))
    radius = torch.ceil(3.0 * torch.sqrt(v1)).detach()
    valid = valid & (radius > radius_clip)
    m2 = torch.stack([fx * x * rz + cx, fy * y * rz + cy], dim=-1)
    inside = (m2[:, 0] + radius > 0) & (m2[:, 0] - radius < width) & (m2[:, 1] + radius > 0) & (m2[:, 1] - radius < height)
    valid = valid & inside.detach()
    radii = torch.where(valid, radius, torch.zeros_like(radius)).to(torch.int32)
    vf = valid.to(dt)
    means2d = m2 * vf[:, None]
    depths = z * vf
    conics = conics * vf[:, None]
    comp = None
    if calc_compensations:
        comp = torch.sqrt(torch.clamp(det_orig / dets, min=0.0)) * vf
    return radii, means2d, depths, conics, comp


# --------------------------------------------------------------------------------------
# tile intersection + sort (integer work; bit-exact contract)
# --------------------------------------------------------------------------------------
def tile_rect(means2d, radii, tile_size, tile_w, tile_h):
    r = radii.to(means2d.dtype)
    ts = float(tile_size)
    x0 = torch.floor(means2d[:, 0] / ts - r / ts)
    y0 = torch.floor(means2d[:, 1] / ts - r / ts)
    x1 = torch.ceil(means2d[:, 0] / ts + r / ts)
    y1 = torch.ceil(means2d[:, 1] / ts + r / ts)
    x0 = x0.clamp(0, tile_w).long()
    x1 = x1.clamp(0, tile_w).long()
    y0 = y0.clamp(0, tile_h).long()
    y1 = y1.clamp(0, tile_h).long()
    return x0, y0, x1, y1


def isect_tiles(means2d, radii, depths, tile_size, tile_w, tile_h):
    """One camera.  Returns tiles_per_gauss [N] i32, isect_ids [M] i64 (tile id << 32 | fp32 depth
    bits), flatten_ids [M] i32, both sorted by key (stable => ties keep Gaussian-index order)."""
    means2d, depths = means2d.detach().float(), depths.detach().float()
    x0, y0, x1, y1 = tile_rect(means2d, radii, tile_size, tile_w, tile_h)
    vis = radii > 0
    cnt = ((x1 - x0) * (y1 - y0)) * vis
    keys, vals = [], []
    dbits = depths.contiguous().view(torch.int32).long()
    for g in torch.nonzero(cnt > 0)[:, 0].tolist():
        ys = torch.arange(y0[g], y1[g])
        xs = torch.arange(x0[g], x1[g])
        tid = (ys[:, None] * tile_w + xs[None, :]).reshape(-1)
        keys.append((tid << 32) | dbits[g])
        vals.append(torch.full_like(tid, g))
    if keys:
        keys, vals = torch.cat(keys), torch.cat(vals)
        order = torch.sort(keys, stable=True).indices
        keys, vals = keys[order], vals[order]
    else:
        keys = torch.zeros(0, dtype=torch.long)
        vals = torch.zeros(0, dtype=torch.long)
    return cnt.to(torch.int32), keys, vals.to(torch.int32)


def isect_offset_encode(isect_ids, tile_w, tile_h):
    """offsets[ty, tx] = first index in the sorted list whose tile id >= this tile."""
    tid = (isect_ids >> 32).contiguous()
    q = torch.arange(tile_w * tile_h, dtype=torch.long)
    return torch.searchsorted(tid, q, right=False).to(torch.int32).reshape(tile_h, tile_w)


# --------------------------------------------------------------------------------------
# compositing
# --------------------------------------------------------------------------------------
def rasterize_to_pixels(means2d, conics, colors, opacities, width, height, tile_size, isect_offsets, flatten_ids,
                        backgrounds=None, return_unstable=False, margin=1e-4, absgrad_probe=None, cond_margin=0.0, perturb=None):
    """means2d [N,2], conics [N,3], colors [N,D], opacities [N]; per-pixel front-to-back blend.

    Returns render [H,W,D], alphas [H,W,1], last_ids [H,W] (index into the sorted intersection
    list of the last blended Gaussian; 0 if none).  With ``return_unstable`` also a bool map of
    pixels in which some discrete decision (alpha cut, T stop) sits within ``margin`` (relative)
    of its threshold, i.e. where an fp32 reimplementation may legitimately differ.  The margin widens per (Gaussian, pixel) by the
    uncertainty of sigma itself: ``cond_margin`` x (|a| dx^2 / 2 + |c| dy^2 / 2 + |b dx dy|) -- the rounding of a quadratic form whose
    terms cancel (a thin splat far from its centre: terms ~1e5 for a sigma of 5) -- and, with ``perturb`` = a list of (d_means2d [N,2],
    d_conics [N,3]) SIGNED errors of the inputs (an fp32 projection's against this one's), by twice their first-order effect on sigma
    (signed: the conic of a thin splat is mostly off by a common factor, which moves sigma by that factor only).  Both default to none.

    ``absgrad_probe`` (a list): every tile appends (ids, dx, dy) with the per-(Gaussian, pixel) offsets kept as
    graph tensors whose gradient is retained; after ``backward()`` ``absgrad_from_probe`` sums their absolute values --
    the ``means2d.absgrad`` buffer of gsplat (sum over pixels of |dL/dmean2d through that pixel|), which the
    reference's densification reads at project/models/trainers/base.py:280-297."""
    H, W = int(height), int(width)
    D = colors.shape[-1]
    dt = means2d.dtype
    th, tw = isect_offsets.shape
    M = flatten_ids.numel()
    offs = torch.cat([isect_offsets.reshape(-1).long(), torch.tensor([M])])
    render = torch.zeros(H, W, D, dtype=dt)
    alphas = torch.zeros(H, W, 1, dtype=dt)
    last = torch.zeros(H, W, dtype=torch.int32)
    unstable = torch.zeros(H, W, dtype=torch.bool)
    row_blocks = []
    for ty in range(th):
        blk_c, blk_a = [], []
        for tx in range(tw):
            s, e = int(offs[ty * tw + tx]), int(offs[ty * tw + tx + 1])
            ys = torch.arange(ty * tile_size, min((ty + 1) * tile_size, H))
            xs = torch.arange(tx * tile_size, min((tx + 1) * tile_size, W))
            if len(ys) == 0 or len(xs) == 0:
                continue
            py = (ys.to(dt) + 0.5)[:, None].expand(len(ys), len(xs)).reshape(-1)
            px = (xs.to(dt) + 0.5)[None, :].expand(len(ys), len(xs)).reshape(-1)
            P = px.numel()
            if e > s:
                ids = flatten_ids[s:e].long()
                dx = means2d[ids, 0][:, None] - px[None, :]
                dy = means2d[ids, 1][:, None] - py[None, :]
                if absgrad_probe is not None and dx.requires_grad:
                    dx.retain_grad()
                    dy.retain_grad()
                    absgrad_probe.append((ids, dx, dy))
                cn = conics[ids]
                sigma = 0.5 * (cn[:, 0:1] * dx * dx + cn[:, 2:3] * dy * dy) + cn[:, 1:2] * dx * dy
                raw = opacities[ids][:, None] * torch.exp(-sigma)
                alpha = torch.clamp(raw, max=ALPHA_MAX)
                ok = (sigma >= 0) & (alpha >= ALPHA_MIN)
                a = torch.where(ok, alpha, torch.zeros_like(alpha))
                om = 1 - a
                Tin = torch.cumprod(torch.cat([torch.ones(1, P, dtype=dt), om[:-1]], 0), 0)  # T before g
                nextT = Tin * om
                stop = ok & (nextT <= T_STOP)
                live = torch.cumsum(stop.to(torch.int32), 0) == 0  # strictly before the first stop
                a_eff = a * live
                w = a_eff * Tin
                out = (w[:, :, None] * colors[ids][:, None, :]).sum(0)
                Tfin = torch.prod(1 - a_eff, 0)
                blended = (a_eff > 0)
                idxs = torch.arange(s, e)[:, None].expand(-1, P)
                lid = torch.where(blended, idxs, torch.zeros_like(idxs)).max(0).values
                if return_unstable:
                    with torch.no_grad():
                        cs = torch.cumsum(stop.to(torch.int32), 0)
                        considered = (cs == 0) | (stop & (cs == 1))  # pairs evaluated before/at termination
                        ds = torch.zeros_like(sigma)     # absolute uncertainty of sigma = relative uncertainty of alpha
                        if cond_margin > 0:
                            ds = cond_margin * (0.5 * (cn[:, 0:1].abs() * dx * dx + cn[:, 2:3].abs() * dy * dy) + (cn[:, 1:2] * dx * dy).abs())
                        w_t = a / (1 - a)                   # d ln T / d sigma of an entry
                        dT = torch.cumsum(w_t * ds, 0)      # relative uncertainty of T behind each entry
                        ds_p, dT_p = torch.zeros_like(ds), torch.zeros_like(ds)
                        for dm, dc in (perturb or ()):     # signed input errors: twice their first-order effect on sigma and on ln T
                            dm, dc = dm.to(dt)[ids], dc.to(dt)[ids]
                            d1 = (0.5 * (dc[:, 0:1] * dx * dx + dc[:, 2:3] * dy * dy) + dc[:, 1:2] * dx * dy
                                  + (cn[:, 0:1] * dx + cn[:, 1:2] * dy) * dm[:, 0:1] + (cn[:, 1:2] * dx + cn[:, 2:3] * dy) * dm[:, 1:2])
                            ds_p = torch.maximum(ds_p, 2.0 * d1.abs())
                            dT_p = torch.maximum(dT_p, 2.0 * torch.cumsum(w_t * d1, 0).abs())
                        ds, dT = ds + ds_p, dT + dT_p
                        near_a = (torch.abs(alpha - ALPHA_MIN) < (margin + ds) * ALPHA_MIN) & (sigma >= -ds)
                        near_c = torch.abs(raw - ALPHA_MAX) < margin + ds * raw
                        near_t = ok & (torch.abs(nextT - T_STOP) < (margin + dT) * T_STOP)
                        near_s = torch.abs(sigma) < 1e-7 + ds
                        u = ((near_a | near_t | near_s | near_c) & considered).any(0)
                        unstable[ys[0]:ys[-1] + 1, xs[0]:xs[-1] + 1] = u.reshape(len(ys), len(xs))
            else:
                out = torch.zeros(P, D, dtype=dt)
                Tfin = torch.ones(P, dtype=dt)
                lid = torch.zeros(P, dtype=torch.long)
            if backgrounds is not None:
                out = out + Tfin[:, None] * backgrounds[None, :]
            blk_c.append(out.reshape(len(ys), len(xs), D))
            blk_a.append((1 - Tfin).reshape(len(ys), len(xs), 1))
            last[ys[0]:ys[-1] + 1, xs[0]:xs[-1] + 1] = lid.reshape(len(ys), len(xs)).to(torch.int32)
        if blk_c:
            row_blocks.append((torch.cat(blk_c, 1), torch.cat(blk_a, 1)))
    if row_blocks:  # assembled by concatenation so that autograd flows
        render = torch.cat([r[0] for r in row_blocks], 0)
        alphas = torch.cat([r[1] for r in row_blocks], 0)
    if return_unstable:
        return render, alphas, last, unstable
    return render, alphas, last


def absgrad_from_probe(probe, n):
    """[n, 2]: sum over tiles and pixels of |d loss / d (mean2d - pixel)| per Gaussian (call after backward())."""
    out = None
    for ids, dx, dy in probe:
        if dx.grad is None:
            continue
        if out is None:
            out = torch.zeros(n, 2, dtype=dx.dtype)
        out[:, 0].index_add_(0, ids, dx.grad.abs().sum(1))
        out[:, 1].index_add_(0, ids, dy.grad.abs().sum(1))
    return out if out is not None else torch.zeros(n, 2)


def rasterize_pixel_loop(means2d, conics, colors, opacities, width, height, tile_size, isect_offsets, flatten_ids):
    """Literal per-pixel scalar loop (no autograd) -- the independent check of the vectorised
    ``rasterize_to_pixels`` on tiny cases."""
    H, W = int(height), int(width)
    D = colors.shape[-1]
    th, tw = isect_offsets.shape
    M = flatten_ids.numel()
    offs = isect_offsets.reshape(-1).tolist() + [M]
    render = torch.zeros(H, W, D, dtype=means2d.dtype)
    alphas = torch.zeros(H, W, 1, dtype=means2d.dtype)
    last = torch.zeros(H, W, dtype=torch.int32)
    m2, cn, col, op, fl = means2d.tolist(), conics.tolist(), colors.tolist(), opacities.tolist(), flatten_ids.tolist()
    for i in range(H):
        for j in range(W):
            t = (i // tile_size) * tw + (j // tile_size)
            px, py = j + 0.5, i + 0.5
            T, cur = 1.0, 0
            acc = [0.0] * D
            for idx in range(offs[t], offs[t + 1]):
                g = fl[idx]
                dx, dy = m2[g][0] - px, m2[g][1] - py
                sigma = 0.5 * (cn[g][0] * dx * dx + cn[g][2] * dy * dy) + cn[g][1] * dx * dy
                alpha = min(ALPHA_MAX, op[g] * math.exp(-sigma))
                if sigma < 0 or alpha < ALPHA_MIN:
                    continue
                nT = T * (1 - alpha)
                if nT <= T_STOP:
                    break
                vis = alpha * T
                for k in range(D):
                    acc[k] += col[g][k] * vis
                cur = idx
                T = nT
            render[i, j] = torch.tensor(acc)
            alphas[i, j, 0] = 1 - T
            last[i, j] = cur
    return render, alphas, last


# --------------------------------------------------------------------------------------
# the whole rasterization() call (reference call site trainers/base.py:393-408)
# --------------------------------------------------------------------------------------
def rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height, near_plane=0.01,
                  far_plane=1e10, radius_clip=0.0, eps2d=0.3, tile_size=16, backgrounds=None, render_mode="RGB",
                  return_unstable=False, absgrad_probes=None):
    """Single- or multi-camera (looped).  colors [N,D] or [C,N,D]; opacities [N].
    Returns render [C,H,W,D(+1)], alphas [C,H,W,1], meta (means2d/radii/depths/conics [C,N,*], ...).
    ``absgrad_probes``: a list that receives one probe list per camera (see rasterize_to_pixels)."""
    assert render_mode in ("RGB", "D", "ED", "RGB+D", "RGB+ED")
    W, H = int(width), int(height)
    C = viewmats.shape[0]
    tw, th = (W + tile_size - 1) // tile_size, (H + tile_size - 1) // tile_size
    rs, als, metas, uns = [], [], [], []
    for c in range(C):
        radii, m2, dep, con, _ = project(means, quats, scales, viewmats[c], Ks[c], W, H, eps2d, near_plane, far_plane, radius_clip)
        col = colors[c] if colors.dim() == 3 else colors
        if render_mode in ("RGB+D", "RGB+ED"):
            col = torch.cat([col, dep[:, None]], -1)
        elif render_mode in ("D", "ED"):
            col = dep[:, None]
        tpg, iids, fids = isect_tiles(m2, radii, dep, tile_size, tw, th)
        offs = isect_offset_encode(iids, tw, th)
        bg = None if backgrounds is None else backgrounds[c]
        probe = None
        if absgrad_probes is not None:
            probe = []
            absgrad_probes.append(probe)
        res = rasterize_to_pixels(m2, con, col, opacities, W, H, tile_size, offs, fids, bg, return_unstable, absgrad_probe=probe)
        r, a = res[0], res[1]
        if render_mode in ("ED", "RGB+ED"):
            r = torch.cat([r[..., :-1], r[..., -1:] / a.clamp(min=1e-10)], -1)
        rs.append(r)
        als.append(a)
        if return_unstable:
            uns.append(res[3])
        metas.append(dict(radii=radii, means2d=m2, depths=dep, conics=con, tiles_per_gauss=tpg, isect_ids=iids,
                          flatten_ids=fids, isect_offsets=offs, last_ids=res[2]))
    meta = {k: torch.stack([m[k] for m in metas]) for k in ("radii", "means2d", "depths", "conics", "tiles_per_gauss", "isect_offsets", "last_ids")}
    meta["isect_ids"] = [m["isect_ids"] for m in metas]
    meta["flatten_ids"] = [m["flatten_ids"] for m in metas]
    meta.update(width=W, height=H, tile_size=tile_size, tile_width=tw, tile_height=th, n_cameras=C)
    if return_unstable:
        meta["unstable"] = torch.stack(uns)
    return torch.stack(rs), torch.stack(als), meta
