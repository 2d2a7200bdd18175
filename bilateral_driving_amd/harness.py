"""Trainer-side harness for the hot path: the call sequence of the reference's training step,
restated around the MI355X operators, plus the synthetic scenes the benchmark and tests use.

Reference call sequence being mirrored (file:line under /root/reference/project):
  models/gaussians/vanilla.py:378-414       get_gaussians: SH colours (+0.5, clamp), activations
  models/trainers/base.py:385-432           render_gaussians -> rasterization(...), split, clamp(max=1)
  models/trainers/scene_graph.py:286-294    sky blend + affine_transformation
  models/trainers/scene_graph.py:86-120     affine_transformation (bilateral maps applied per pixel)
  models/trainers/base.py:502-516           backward
  models/trainers/base.py:279-297           consumers of info["means2d"].absgrad / info["radii"]

Scene generator: SURVEY.md 8(d) (ring rig, log-uniform ranges, anisotropic scales).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch
from torch import Tensor

from .bilagrid import bilagrid_transform, total_variation_loss
from .fused_view import fused_view
from .gs_ops import (TILE_SIZE, fully_fused_projection, isect_tiles, rasterize_to_pixels, spherical_harmonics)

SIX_CAM_YAWS = (0.0, 55.0, -55.0, 110.0, -110.0, 180.0)
FIVE_CAM_YAWS = (0.0, 45.0, -45.0, 90.0, -90.0)
LEVELS_3 = ((2, 2, 1), (4, 4, 2), (8, 8, 4))   # configs/omnire_ms_bilateral*.yaml model.Affine.params.grid
FACTORS_3 = (4, 4, 2)                          # modules.py:505 default guidance_factor
LEVELS_SINGLE = ((16, 16, 8),)
FACTORS_SINGLE = (1,)
TILE_CULL = True  # exact tile culling in render_view (see gs_ops.isect_tiles); images/gradients are unaffected
FUSED = True      # render_view = one fused autograd node (fused_view.py); False = the reference's call sequence over the drop-in
#                   operators (render_view_api); "ops" = the chain of individual operators (render_view_staged)


@dataclass
class Camera:
    viewmat: Tensor  # [4,4] world -> camera (OpenCV: x right, y down, z forward)
    K: Tensor        # [3,3]
    width: int
    height: int
    cam_pos: Optional[Tensor] = None  # [3] camera centre in world space (= inv(viewmat)[:3, 3]), cached per camera


def ring_cameras(W: int, H: int, yaws_deg: Sequence[float] = SIX_CAM_YAWS, device="cpu", origin=(0.0, 0.0, 0.0)) -> List[Camera]:
    """Ring rig at ``origin`` (default: the world origin); world frame: x forward, y left, z up (driving convention)."""
    fx = 0.5 * W / math.tan(math.radians(35.0))
    K = torch.tensor([[fx, 0, W / 2], [0, fx, H / 2], [0, 0, 1]], dtype=torch.float32, device=device)
    cams = []
    for yaw in yaws_deg:
        a = math.radians(yaw)
        fwd = torch.tensor([math.cos(a), math.sin(a), 0.0])
        right = torch.tensor([math.sin(a), -math.cos(a), 0.0])
        down = torch.tensor([0.0, 0.0, -1.0])
        R = torch.stack([right, down, fwd])  # rows = camera axes in world coordinates
        vm = torch.eye(4)
        vm[:3, :3] = R
        vm[:3, 3] = -R @ torch.tensor([float(o) for o in origin])   # rig at the origin by default
        cams.append(Camera(vm.to(device), K, W, H, torch.linalg.inv(vm)[:3, 3].contiguous().to(device)))
    return cams


def synthetic_scene(N: int, seed: int = 0, device="cpu") -> Dict[str, Tensor]:
    """Raw (pre-activation) Gaussian parameters, as the reference's VanillaGaussians holds them."""
    g = torch.Generator().manual_seed(seed)
    az = torch.rand(N, generator=g) * 2 * math.pi
    rng = torch.exp(torch.rand(N, generator=g) * math.log(80.0 / 2.0) + math.log(2.0))
    hgt = torch.rand(N, generator=g) * 8.0 - 2.0
    means = torch.stack([rng * torch.cos(az), rng * torch.sin(az), hgt], -1)
    log_scales = (torch.rand(N, 3, generator=g) * math.log(0.3 / 0.01) + math.log(0.01)) + torch.log(rng / 10.0)[:, None]
    quats = torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=-1)
    opac_logit = torch.randn(N, generator=g) * 1.5
    sh = torch.empty(N, 16, 3)
    sh[:, 0] = (torch.rand(N, 3, generator=g) - 0.5) / 0.28209479177387814
    sh[:, 1:] = torch.randn(N, 15, 3, generator=g) * 0.05
    out = dict(means=means, log_scales=log_scales, quats=quats, opacity_logits=opac_logit, sh=sh)
    return {k: v.to(device).contiguous() for k, v in out.items()}


def spatial_order(means: Tensor, bits: int = 10) -> Tensor:
    """``densify.spatial_order`` (Morton order of the centres) -- re-exported for scripts that build scenes here."""
    from .densify import spatial_order as _so
    return _so(means, bits)


def reorder_params(params: Dict[str, Tensor], perm: Tensor) -> Dict[str, Tensor]:
    """The parameter dict with its rows permuted (fresh contiguous leaves; ``requires_grad`` kept)."""
    return {k: v.detach()[perm].contiguous().requires_grad_(v.requires_grad) for k, v in params.items()}


def lidar_scene(N: int = 1_000_000, seed: int = 0, device="cpu", opacity: str = "trained") -> Dict[str, Tensor]:
    """A lidar-INITIALISED street scene, the other end of the splat-size distribution from ``synthetic_scene``: the reference seeds
    its Background class with 800 k points sampled from the drive's accumulated lidar sweeps plus 100 k "near" randoms (uniform in
    the scene sphere) and 100 k "far" ones (inverse distance uniform: the sky's stand-ins), and sets every scale to the MEAN DISTANCE
    OF THE 3 NEAREST NEIGHBOURS in that combined set, isotropic, with random rotations
    (/root/reference/project/configs/omnire_ms_bilateral_extended.yaml:75-80, models/trainers/scene_graph.py:156-190,
    models/gaussians/vanilla.py:82-92; kNN: models/gaussians/basics.py:208-224) -- centimetre-sized splats on surfaces, most of them
    smaller than half a 16-px tile, instead of the ring scene's range-proportional ones.  0.8 N lidar points on a 120 m street
    corridor in the rig's frame (x forward, y left, z up; sensor 1.7 m above the road): 45 % road surface (denser near the
    trajectory), 35 % facades of building blocks either side, 20 % clutter (parked cars, poles, tree crowns), 0.1 N + 0.1 N randoms.
    ``opacity``: "trained" = logits N(0, 1.5) on the lidar points, N(-2.5, 1) on the randoms, SH rest N(0, 0.05); "init" = the
    reference's step-0 state (opacity 0.1 everywhere, rest coefficients 0: vanilla.py:96-104)."""
    import numpy as np
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(1234 + seed)
    n_lidar, n_near = int(0.8 * N), int(0.1 * N)
    n_far = N - n_lidar - n_near
    n_road, n_fac = int(0.45 * n_lidar), int(0.35 * n_lidar)
    n_clut = n_lidar - n_road - n_fac
    z0 = -1.7
    # road: x along the drive, lateral spread of a spinning lidar's returns around the trajectory
    road = np.stack([rng.uniform(-20.0, 100.0, n_road), np.clip(rng.normal(0.0, 8.0, n_road), -25.0, 25.0),
                     z0 + rng.normal(0.0, 0.02, n_road)], -1)
    # facades: building blocks of 8-30 m length at 11-16 m either side, returns denser at low heights
    n_blocks = 40
    bx0 = rng.uniform(-20.0, 95.0, n_blocks)
    blen, bside = rng.uniform(8.0, 30.0, n_blocks), rng.choice([-1.0, 1.0], n_blocks)
    bdist, bh = rng.uniform(11.0, 16.0, n_blocks), rng.uniform(6.0, 18.0, n_blocks)
    b = rng.integers(0, n_blocks, n_fac)
    fac = np.stack([bx0[b] + rng.uniform(0.0, 1.0, n_fac) * blen[b], bside[b] * bdist[b] + rng.normal(0.0, 0.03, n_fac),
                    z0 + bh[b] * rng.uniform(0.0, 1.0, n_fac) ** 1.5], -1)
    # clutter: 500 objects -- cars (boxes 4.5 x 1.9 x 1.5), poles (0.2 x 0.2 x 6), tree crowns (spheres r = 2 at 5 m)
    n_obj = 500
    kind = rng.choice(3, n_obj, p=[0.5, 0.25, 0.25])
    ox, oy = rng.uniform(-20.0, 100.0, n_obj), rng.choice([-1.0, 1.0], n_obj) * rng.uniform(3.5, 10.0, n_obj)
    o = rng.integers(0, n_obj, n_clut)
    u = rng.uniform(-1.0, 1.0, (n_clut, 3))
    face = rng.integers(0, 3, n_clut)
    u[np.arange(n_clut), face] = np.sign(u[np.arange(n_clut), face])          # on a face of the unit cube
    half = np.array([[2.25, 0.95, 0.75], [0.1, 0.1, 3.0], [2.0, 2.0, 2.0]])[kind[o]]
    centre_z = np.array([z0 + 0.75, z0 + 3.0, z0 + 5.0])[kind[o]]
    sph = u / np.linalg.norm(u, axis=1, keepdims=True)
    local = np.where((kind[o] == 2)[:, None], sph * 2.0, u * half)
    clut = np.stack([ox[o], oy[o], centre_z], -1) + local
    # randoms (scene_graph.py:165-176): uniform in the scene sphere; inverse distance uniform in (0, 1 / radius), capped at 2 km
    centre, radius = np.array([40.0, 0.0, 0.0]), 70.0
    def unit(n):
        v = rng.normal(size=(n, 3))
        return v / np.linalg.norm(v, axis=1, keepdims=True)
    near = centre + unit(n_near) * radius * rng.uniform(0.0, 1.0, (n_near, 1)) ** (1.0 / 3.0)
    far = centre + unit(n_far) * (radius / rng.uniform(radius / 2000.0, 1.0, (n_far, 1)))
    near[:, 2] = np.abs(near[:, 2] - z0) + z0          # (the reference keeps what some camera sees: nothing under the road)
    far[:, 2] = np.abs(far[:, 2] - z0) + z0
    pts = np.concatenate([road, fac, clut, near, far], 0).astype(np.float32)
    is_random = np.concatenate([np.zeros(n_lidar, bool), np.ones(n_near + n_far, bool)])
    perm = rng.permutation(pts.shape[0])               # (the reference's point order carries no spatial meaning either)
    pts, is_random = pts[perm], is_random[perm]
    d, _ = cKDTree(pts).query(pts, k=4, workers=-1)
    avg = np.maximum(d[:, 1:].mean(-1), 1e-4).astype(np.float32)           # vanilla.py:85-88
    g = torch.Generator().manual_seed(seed)
    means = torch.from_numpy(pts)
    log_scales = torch.log(torch.from_numpy(avg))[:, None].repeat(1, 3)
    quats = torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=-1)
    sh = torch.zeros(N, 16, 3)
    sh[:, 0] = (torch.rand(N, 3, generator=g) - 0.5) / 0.28209479177387814   # RGB2SH of random colours (basics.py:76-89)
    if opacity == "init":
        opac_logit = torch.full((N,), math.log(0.1 / 0.9))
    else:
        # the metre-sized randoms do not survive training as opaque blobs (opacity reset + cull_alpha_thresh, vanilla.py:205-302):
        # they stay as a faint background layer
        opac_logit = torch.where(torch.from_numpy(is_random), torch.randn(N, generator=g) - 2.5, torch.randn(N, generator=g) * 1.5)
        sh[:, 1:] = torch.randn(N, 15, 3, generator=g) * 0.05
    out = dict(means=means, log_scales=log_scales, quats=quats, opacity_logits=opac_logit, sh=sh)
    return {k: v.to(device).contiguous() for k, v in out.items()}


def make_grids(n_images: int, levels=LEVELS_3, seed: int = 0, device="cpu") -> List[Tensor]:
    g = torch.Generator().manual_seed(1000 + seed)
    out = []
    for (gx, gy, gl) in levels:
        ident = torch.tensor([1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0]).reshape(1, 12, 1, 1, 1).repeat(n_images, 1, gl, gy, gx)
        out.append((ident + 0.05 * torch.randn(n_images, 12, gl, gy, gx, generator=g)).to(device).contiguous())
    return out


def render_view(params: Dict[str, Tensor], cam: Camera, grids: Sequence[Tensor], img_idx: int, sky: Tensor,
                factors: Sequence[int] = FACTORS_3, sh_degree: int = 3, near_plane: float = 0.1, far_plane: float = 1e10,
                radius_clip: float = 0.0, eps2d: float = 0.3, grad_arena=None, arena_rows: int = 0, grad_sink=None, list_tile=None,
                caps=None, prep_ws=None):
    """One view's forward (dict(rgb, depth, opacity, rgb_gaussians, info)): a single fused autograd node
    (fused_view.py) by default, or the chain of individual operators (render_view_staged) when FUSED is off."""
    if FUSED == "ops":
        return render_view_staged(params, cam, grids, img_idx, sky, factors, sh_degree, near_plane, far_plane, radius_clip, eps2d)
    if not FUSED:
        return render_view_api(params, cam, grids, img_idx, sky, factors, sh_degree, near_plane, far_plane, radius_clip, eps2d)
    return fused_view(params, cam.viewmat, cam.K, cam.width, cam.height, grids, sky, factors, cam_pos=cam.cam_pos, sh_degree=sh_degree,
                      near_plane=near_plane, far_plane=far_plane, radius_clip=radius_clip, eps2d=eps2d, tile_cull=TILE_CULL,
                      grad_arena=grad_arena, img_idx=img_idx, arena_rows=arena_rows, grad_sink=grad_sink, list_tile=list_tile, caps=caps, prep_ws=prep_ws)


def train_view(params: Dict[str, Tensor], cam: Camera, grids: Sequence[Tensor], img_idx: int, sky: Tensor, target: Tensor,
               factors: Sequence[int] = FACTORS_3, tv_weight: float = 0.01, grid_grads=None, after_forward=None, **kw):
    """render_view + training_loss + backward of one view without an autograd graph (fused_view.train_view): same kernels, same
    gradients in the same places; returns dict(loss, rgb, depth, opacity, info)."""
    from .fused_view import train_view as _tv
    level_w = [0.5 * math.sqrt(g.shape[4] * g.shape[3] * g.shape[2]) for g in grids]  # modules.py:445
    return _tv(params, cam.viewmat, cam.K, cam.width, cam.height, grids, sky, factors, target, [tv_weight * w for w in level_w],
               img_idx=img_idx, grid_grads=grid_grads, after_forward=after_forward, cam_pos=cam.cam_pos, tile_cull=TILE_CULL, **kw)


def render_view_staged(params: Dict[str, Tensor], cam: Camera, grids: Sequence[Tensor], img_idx: int, sky: Tensor,
                       factors: Sequence[int] = FACTORS_3, sh_degree: int = 3, near_plane: float = 0.1, far_plane: float = 1e10,
                       radius_clip: float = 0.0, eps2d: float = 0.3):
    """One view's forward: returns dict(rgb, depth, opacity, rgb_gaussians, info).

    Same arithmetic as get_gaussians -> rasterization(...) -> split/clamp -> sky blend -> affine_transformation in
    the reference; the stages are called directly so that SH colours are only evaluated for Gaussians that survive
    projection (the reference evaluates all N before it knows which are visible)."""
    means = params["means"]
    W, H = cam.width, cam.height
    opac = torch.sigmoid(params["opacity_logits"])                         # vanilla.py:393
    scales = torch.exp(params["log_scales"])
    # get_quats normalises (vanilla.py:395) and so does the projection kernel internally; normalising twice is
    # the identity, so the raw quaternions go straight in and the normalisation's backward is part of the kernel
    quats = params["quats"]
    viewmats, Ks = cam.viewmat[None], cam.K[None]
    radii, means2d, depths, conics, _ = fully_fused_projection(means, quats, scales, viewmats, Ks, W, H, eps2d=eps2d,
                                                               near_plane=near_plane, far_plane=far_plane,
                                                               radius_clip=radius_clip)
    c2w_t = torch.linalg.inv(cam.viewmat)[:3, 3]
    viewdirs = means.detach() - c2w_t                                       # vanilla.py:385
    rgbs = spherical_harmonics(sh_degree, viewdirs, params["sh"], masks=radii[0] > 0)  # vanilla.py:388
    rgbs = torch.clamp(rgbs + 0.5, 0.0, 1.0)                                # vanilla.py:389
    colors = torch.cat([rgbs, depths[0][:, None]], dim=-1)[None]            # render_mode "RGB+ED"
    opac_c = opac[None]
    tw, th = math.ceil(W / TILE_SIZE), math.ceil(H / TILE_SIZE)
    tiles_per_gauss, _, flatten_ids, isect_offsets = isect_tiles(means2d, radii, depths, TILE_SIZE, tw, th,
                                                                 want_isect_ids=False, conics=conics if TILE_CULL else None,
                                                                 opacities=opac_c if TILE_CULL else None)
    renders, alphas = rasterize_to_pixels(means2d, conics, colors, opac_c, W, H, TILE_SIZE, isect_offsets, flatten_ids,
                                          absgrad=True)
    renders = renders[0]
    rgb_g = renders[..., :3]                                                 # base.py:414 (clamp is fused below)
    opacity = alphas[0]                                                      # [H,W,1]
    depth = renders[..., 3:4] / opacity.clamp(min=1e-10)                     # expected depth ("ED")
    grids_k = [g[img_idx:img_idx + 1] for g in grids]
    rgb = bilagrid_transform(rgb_g, grids_k, factors, alpha=opacity, sky=sky)  # clamp + sky blend + slice + affine
    info = {"means2d": means2d, "radii": radii, "depths": depths, "conics": conics, "width": W, "height": H,
            "tiles_per_gauss": tiles_per_gauss, "flatten_ids": flatten_ids, "isect_offsets": isect_offsets,
            "tile_size": TILE_SIZE, "n_cameras": 1, "n_isects": int(flatten_ids.numel())}
    return dict(rgb=rgb, depth=depth, opacity=opacity, rgb_gaussians=rgb_g, info=info)


def render_view_api(params: Dict[str, Tensor], cam: Camera, grids: Sequence[Tensor], img_idx: int, sky: Tensor,
                    factors: Sequence[int] = FACTORS_3, sh_degree: int = 3, near_plane: float = 0.1, far_plane: float = 1e10,
                    radius_clip: float = 0.0, eps2d: float = 0.3):
    """One view's forward through the reference's OWN call sequence, every gsplat / bilateral entry point replaced by this package's
    drop-in and nothing else changed: the Gaussian class's activations and dense SH pass (models/gaussians/vanilla.py:378-414:
    exp, sigmoid, normalised quaternions, ``spherical_harmonics`` over ALL Gaussians, clamp), ``rasterization(...)`` with the
    trainer's arguments (models/trainers/base.py:393-408), split / sky blend / colour transform (base.py:409-419,
    scene_graph.py:86-120,292-294; the transform through the module's fused ``transform``).  What bench.py reports as
    ``api_path_iters_per_sec``."""
    from .rendering import rasterization
    means = params["means"]
    W, H = cam.width, cam.height
    opac = torch.sigmoid(params["opacity_logits"])                         # vanilla.py:393
    scales = torch.exp(params["log_scales"])                               # vanilla.py:394
    quats = params["quats"] / params["quats"].norm(dim=-1, keepdim=True)   # vanilla.py:395 (get_quats)
    cam_pos = cam.cam_pos if cam.cam_pos is not None else torch.linalg.inv(cam.viewmat.detach())[:3, 3]
    viewdirs = means.detach() - cam_pos                                    # vanilla.py:384-385
    rgbs = spherical_harmonics(sh_degree, viewdirs, params["sh"])          # vanilla.py:388 (every Gaussian: nothing is culled yet)
    rgbs = torch.clamp(rgbs + 0.5, 0.0, 1.0)                               # vanilla.py:389
    renders, alphas, info = rasterization(means, quats, scales, opac, rgbs, cam.viewmat[None], cam.K[None], W, H, packed=False,
                                          absgrad=True, sparse_grad=False, rasterize_mode="classic", near_plane=near_plane,
                                          far_plane=far_plane, radius_clip=radius_clip, eps2d=eps2d, render_mode="RGB+ED")   # base.py:393-408
    rgb_g, depth, opacity = renders[0, ..., :3], renders[0, ..., 3:4], alphas[0]      # base.py:409-416
    grids_k = [g[img_idx:img_idx + 1] for g in grids]
    rgb = bilagrid_transform(rgb_g, grids_k, factors, alpha=opacity, sky=sky)          # clamp + sky blend + slice + affine
    return dict(rgb=rgb, depth=depth, opacity=opacity, rgb_gaussians=rgb_g, info=info)


class VanillaModel:
    """A Gaussian class with the attributes of the reference's ``VanillaGaussians`` (models/gaussians/vanilla.py:96-104,151-181:
    ``_means``, ``_scales`` (log), ``_quats`` (raw), ``_opacities`` [N,1] (logits), ``_features_dc`` [N,3], ``_features_rest``
    [N,K-1,3], ``sh_degree``, ``step``, ``ctrl_cfg.sh_degree_interval``) built from this harness's parameter dict -- what the a13
    mirrors and ``marshalling.install`` are exercised on where the reference's class (pytorch3d, omegaconf) can not be imported.
    ``get_gaussians`` is the eager mirror; ``marshalling.install(VanillaModel)`` swaps in the deferred one."""

    class _Ctrl:
        sh_degree_interval = 1000

    def __init__(self, params: Dict[str, Tensor], sh_degree: int = 3, step: int = 10 ** 6):
        leaf = lambda t: t.detach().clone().contiguous().requires_grad_(True)
        self._means, self._quats, self._scales = leaf(params["means"]), leaf(params["quats"]), leaf(params["log_scales"])
        self._opacities = leaf(params["opacity_logits"].reshape(-1, 1))
        self._features_dc, self._features_rest = leaf(params["sh"][:, 0, :]), leaf(params["sh"][:, 1:, :])
        self.sh_degree, self.step, self.ctrl_cfg = int(sh_degree), int(step), VanillaModel._Ctrl()

    def parameters(self):
        return [self._means, self._quats, self._scales, self._opacities, self._features_dc, self._features_rest]

    def get_gaussians(self, cam):
        from .marshalling import get_gaussians
        return get_gaussians(self, cam)


def reference_camera(cam: Camera):
    """The reference's ``dataclass_camera`` of a harness camera (cached on it: the trainer gets ``camtoworlds`` from its dataset,
    models/trainers/base.py:317-340; ``get_gaussians`` reads the camera centre only, vanilla.py:384)."""
    dc = getattr(cam, "_reference_camera", None)
    if dc is None:
        from .marshalling import dataclass_camera
        cam_pos = cam.cam_pos if cam.cam_pos is not None else torch.linalg.inv(cam.viewmat.detach())[:3, 3]
        c2w = torch.eye(4, device=cam.viewmat.device)
        c2w[:3, 3] = cam_pos.detach()
        dc = cam._reference_camera = dataclass_camera(camtoworlds=c2w, camtoworlds_gt=c2w, Ks=cam.K, H=cam.height, W=cam.width)
    return dc


def render_view_model(model, cam: Camera, grids: Sequence[Tensor], img_idx: int, sky: Tensor, factors: Sequence[int] = FACTORS_3,
                      near_plane: float = 0.1, far_plane: float = 1e10, radius_clip: float = 0.0, eps2d: float = 0.3):
    """One view's forward through the reference's call sequence for a scene of ONE Gaussian class, starting at the class's own
    ``get_gaussians`` (models/trainers/base.py:342-383 ``collect_gaussians``: every entry through ``torch.cat``; :385-419
    ``render_gaussians``: ``rasterization`` with the trainer's arguments, ``gs.opacities.squeeze()``; then split / sky blend / colour
    transform as ``render_view_api``).  With ``marshalling.install`` on the model's class the same lines reach the raw one-view node."""
    from .rendering import rasterization
    W, H = cam.width, cam.height
    gs = model.get_gaussians(reference_camera(cam))
    gs = {k: torch.cat([v], dim=0) for k, v in gs.items()}                                   # base.py:365-366
    renders, alphas, info = rasterization(means=gs["_means"], quats=gs["_quats"], scales=gs["_scales"], opacities=gs["_opacities"].squeeze(),
                                          colors=gs["_rgbs"], viewmats=cam.viewmat[None], Ks=cam.K[None], width=W, height=H, packed=False,
                                          absgrad=True, sparse_grad=False, rasterize_mode="classic", near_plane=near_plane,
                                          far_plane=far_plane, radius_clip=radius_clip, eps2d=eps2d, render_mode="RGB+ED")   # base.py:393-408
    renders = renders[0]                                                                      # base.py:409
    alphas = alphas[0].squeeze(-1)                                                            # base.py:410
    assert renders.shape[-1] == 4, "Must render rgb, depth and alpha"                         # base.py:413
    rgb_g, depth = torch.split(renders, [3, 1], dim=-1)                                       # base.py:414
    opacity = alphas[..., None]                                                               # base.py:417
    grids_k = [g[img_idx:img_idx + 1] for g in grids]
    rgb = bilagrid_transform(rgb_g, grids_k, factors, alpha=opacity, sky=sky)                 # (clamp(max=1), sky blend, transform)
    return dict(rgb=rgb, depth=depth, opacity=opacity, rgb_gaussians=rgb_g, info=info)


FUSED_LOSS = True   # one autograd node (losses.photometric_tv_loss) instead of ~30 framework kernels


def training_loss(out: Dict[str, Tensor], target: Tensor, grids: Sequence[Tensor], tv_weight: float = 0.01, grid_grads=None) -> Tensor:
    """L1 photometric + TV(grids) (trainers/base.py:518-565,590-594: losses.affine.w = 0.01 for the
    multi-scale config); every gradient path of the hot path is live."""
    level_w = [0.5 * math.sqrt(g.shape[4] * g.shape[3] * g.shape[2]) for g in grids]  # modules.py:445
    if FUSED_LOSS:
        from .losses import photometric_tv_loss
        return photometric_tv_loss(out["rgb"], target, grids, [tv_weight * w for w in level_w], grid_grads=grid_grads)
    assert grid_grads is None, "in-place grid gradients need the fused loss"
    loss = (out["rgb"] - target).abs().mean()
    for g, w in zip(grids, level_w):
        loss = loss + tv_weight * total_variation_loss(g, w)
    return loss


def densify_stats(info, width: int, height: int):
    """What trainers/base.py:279-297 reads after backward."""
    grads = info["means2d"].absgrad.clone()
    grads[..., 0] *= width / 2.0
    grads[..., 1] *= height / 2.0
    return grads, info["radii"]
