"""Pins of oracle/gs_oracle.py that do not need gsplat (PARITY UNPINNED, see its header):
known answers, independent scalar loop, finite differences, invariances.  CPU only."""
import math

import numpy as np
import pytest
import torch

from oracle import gs_oracle as G
from tests.util import make_scene

F64 = torch.float64


def test_sh_bases_match_closed_form():
    d = torch.randn(500, 3, dtype=F64)
    assert (G.sh_bases(3, d) - G.sh_bases_closed_form(d)).abs().max() < 1e-14
    # C0 is the one constant the reference itself pins (models/gaussians/basics.py:76-89)
    assert abs(float(G.sh_bases(0, d)[0, 0]) - 0.28209479177387814) < 1e-15
    # degree truncation uses a prefix of the same bases
    assert torch.equal(G.sh_bases(2, d), G.sh_bases(3, d)[:, :9])


def test_sh_orthonormal_on_sphere():
    # Monte-Carlo check that the 16 bases are orthonormal: pins signs/constants up to permutation
    g = torch.Generator().manual_seed(0)
    d = torch.randn(400000, 3, generator=g, dtype=F64)
    B = G.sh_bases(3, d)
    gram = (B.T @ B) / d.shape[0] * 4 * math.pi
    assert (gram - torch.eye(16, dtype=F64)).abs().max() < 2e-2


def test_projection_known_answer():
    # isotropic Gaussian on the optical axis: covariance (s*fx/z)^2 + eps2d, mean at principal point
    s, z, fx = 0.2, 4.0, 100.0
    means = torch.tensor([[0.0, 0.0, z]], dtype=F64)
    quats = torch.tensor([[0.3, -0.5, 0.1, 0.8]], dtype=F64)
    scales = torch.full((1, 3), s, dtype=F64)
    K = torch.tensor([[fx, 0, 32.0], [0, fx, 24.0], [0, 0, 1]], dtype=F64)
    radii, m2, dep, con, comp = G.project(means, quats, scales, torch.eye(4, dtype=F64), K, 64, 48, calc_compensations=True)
    var = (s * fx / z) ** 2 + 0.3
    assert torch.allclose(m2, torch.tensor([[32.0, 24.0]], dtype=F64))
    assert abs(float(dep) - z) < 1e-12
    assert torch.allclose(con, torch.tensor([[1 / var, 0.0, 1 / var]], dtype=F64), atol=1e-12)
    assert int(radii) == math.ceil(3 * math.sqrt(var))
    assert abs(float(comp) - ((s * fx / z) ** 2) / var) < 1e-12
    # behind the near plane / off screen -> culled, outputs zero
    r2, m22, _, _, _ = G.project(torch.tensor([[0.0, 0.0, 0.005], [50.0, 0, 4.0]], dtype=F64), quats.repeat(2, 1),
                                 scales.repeat(2, 1), torch.eye(4, dtype=F64), K, 64, 48)
    assert r2.tolist() == [0, 0] and float(m22.abs().max()) == 0.0


def test_single_gaussian_image_known_answer():
    # one isotropic Gaussian: alpha(p) = o * exp(-|p-mu|^2 / (2 var)) inside the 3-sigma tile box
    s, z, fx, o = 0.3, 5.0, 80.0, 0.8
    W, H = 48, 32
    means = torch.tensor([[0.0, 0.0, z]], dtype=F64)  # on the optical axis: the 2D footprint is exactly isotropic
    K = torch.tensor([[fx, 0, W / 2 + 1.3], [0, fx, H / 2 - 0.7], [0, 0, 1]], dtype=F64)
    col = torch.tensor([[0.2, 0.5, 0.9]], dtype=F64)
    r, a, meta = G.rasterization(means, torch.tensor([[1.0, 0, 0, 0]], dtype=F64), torch.full((1, 3), s, dtype=F64),
                                 torch.tensor([o], dtype=F64), col, torch.eye(4, dtype=F64)[None], K[None], W, H,
                                 render_mode="RGB+ED")
    var = (s * fx / z) ** 2 + 0.3
    mu = torch.tensor([W / 2 + 1.3, H / 2 - 0.7], dtype=F64)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=F64) + 0.5, torch.arange(W, dtype=F64) + 0.5, indexing="ij")
    alpha = o * torch.exp(-((xs - mu[0]) ** 2 + (ys - mu[1]) ** 2) / (2 * var))
    alpha = torch.where(alpha >= 1 / 255, alpha.clamp(max=0.999), torch.zeros_like(alpha))
    # tiles touched by the bounding square
    rad = float(meta["radii"][0, 0])
    tmask = torch.zeros(H, W, dtype=torch.bool)
    x0, x1 = int(math.floor((mu[0] - rad) / 16)), int(math.ceil((mu[0] + rad) / 16))
    y0, y1 = int(math.floor((mu[1] - rad) / 16)), int(math.ceil((mu[1] + rad) / 16))
    tmask[max(y0, 0) * 16:y1 * 16, max(x0, 0) * 16:x1 * 16] = True
    alpha = alpha * tmask
    assert (a[0, ..., 0] - alpha).abs().max() < 1e-12
    assert (r[0, ..., :3] - alpha[..., None] * col[0]).abs().max() < 1e-12
    assert (r[0, ..., 3][alpha > 0] - z).abs().max() < 1e-9  # expected depth of a single Gaussian is its depth


@pytest.mark.parametrize("seed", [0, 1])
def test_vectorised_blend_matches_scalar_loop(seed):
    sc = make_scene(150, 48, 40, seed=seed, dtype=F64)
    r, a, meta = G.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["colors"], sc["viewmats"],
                                 sc["Ks"], 48, 40, render_mode="RGB+D")
    col = torch.cat([sc["colors"], meta["depths"][0][:, None]], -1)
    r2, a2, l2 = G.rasterize_pixel_loop(meta["means2d"][0], meta["conics"][0], col, sc["opacities"], 48, 40, 16,
                                        meta["isect_offsets"][0], meta["flatten_ids"][0])
    assert (r[0] - r2.double()).abs().max() < 1e-5  # the loop stores float32
    assert (a[0] - a2.double()).abs().max() < 1e-6
    assert torch.equal(meta["last_ids"][0], l2)
    assert float(a.mean()) > 0.2  # the scene is not trivial


def test_isect_sorted_and_offsets():
    sc = make_scene(300, 64, 48, seed=3, dtype=torch.float32)
    _, _, meta = G.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["colors"], sc["viewmats"],
                                 sc["Ks"], 64, 48)
    ids = meta["isect_ids"][0]
    assert torch.all(ids[1:] >= ids[:-1])
    assert ids.numel() == int(meta["tiles_per_gauss"].sum())
    tid = ids >> 32
    offs = meta["isect_offsets"][0].reshape(-1).long()
    for t in range(offs.numel()):
        e = offs[t + 1] if t + 1 < offs.numel() else ids.numel()
        assert torch.all(tid[offs[t]:e] == t)
    # depth bits in the low word
    dep = meta["depths"][0][meta["flatten_ids"][0].long()]
    assert torch.equal((ids & 0xFFFFFFFF).to(torch.int32), dep.view(torch.int32))


def test_permutation_invariance():
    sc = make_scene(120, 48, 32, seed=5, dtype=F64)
    r, a, _ = G.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["colors"], sc["viewmats"], sc["Ks"], 48, 32)
    p = torch.randperm(120, generator=torch.Generator().manual_seed(1))
    r2, a2, _ = G.rasterization(sc["means"][p], sc["quats"][p], sc["scales"][p], sc["opacities"][p], sc["colors"][p],
                                sc["viewmats"], sc["Ks"], 48, 32)
    assert (r - r2).abs().max() < 1e-12 and (a - a2).abs().max() < 1e-12


def test_gradients_match_finite_differences():
    sc = make_scene(12, 32, 32, seed=7, dtype=F64, spread=0.5)
    params = {k: sc[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
    vm = sc["viewmats"].clone().requires_grad_(True)
    wt = torch.randn(1, 32, 32, 4, generator=torch.Generator().manual_seed(0), dtype=F64)
    wa = torch.randn(1, 32, 32, 1, generator=torch.Generator().manual_seed(1), dtype=F64)

    def loss_fn(means, quats, scales, opacities, colors, vmx):
        r, a, _ = G.rasterization(means, quats, scales, opacities, colors, vmx, sc["Ks"], 32, 32, render_mode="RGB+ED")
        return (r * wt).sum() + (a * wa).sum()

    loss = loss_fn(*params.values(), vm)
    loss.backward()
    eps = 1e-6
    rng = np.random.RandomState(0)
    for name, p in list(params.items()) + [("viewmats", vm)]:
        flat = p.detach().reshape(-1)
        for idx in rng.choice(flat.numel(), size=min(6, flat.numel()), replace=False):
            if name == "viewmats" and idx >= 12:
                continue  # bottom row is not used
            vals = []
            for sgn in (+1, -1):
                q = flat.clone()
                q[idx] += sgn * eps
                args = [q.reshape(p.shape) if n == name else v.detach() for n, v in params.items()]
                vmx = q.reshape(vm.shape) if name == "viewmats" else vm.detach()
                vals.append(float(loss_fn(*args, vmx)))
            fd = (vals[0] - vals[1]) / (2 * eps)
            an = float(p.grad.reshape(-1)[idx])
            assert abs(fd - an) <= 1e-4 * max(1.0, abs(fd)), (name, idx, fd, an)


def test_absgrad_probe_equals_brute_force_per_pixel_gradients():
    """G.absgrad_from_probe (what the GPU tests compare info["means2d"].absgrad with) == sum over pixels of the absolute value of that
    pixel's own gradient w.r.t. the projected mean, each obtained by a separate autograd pass (the definition the reference's
    densification relies on, /root/reference/project/models/trainers/base.py:280-297)."""
    N, W, H = 40, 32, 32
    sc = make_scene(N, W, H, seed=3, dtype=torch.float64)
    p = {k: sc[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
    probes = []
    r, a, meta = G.rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], sc["viewmats"], sc["Ks"], W, H,
                                 render_mode="RGB+ED", absgrad_probes=probes)
    wt = torch.randn(r.shape, generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    (r * wt).sum().backward()
    ab = G.absgrad_from_probe(probes[0], N)
    radii, m2, dep, con, _ = G.project(sc["means"], sc["quats"], sc["scales"], sc["viewmats"][0], sc["Ks"][0], W, H)
    col = torch.cat([sc["colors"], dep[:, None]], -1)
    tpg, iids, fids = G.isect_tiles(m2, radii, dep, 16, 2, 2)
    offs = G.isect_offset_encode(iids, 2, 2)
    m2l = m2.clone().requires_grad_(True)
    rr, aa, _ = G.rasterize_to_pixels(m2l, con, col, sc["opacities"], W, H, 16, offs, fids)
    rr = torch.cat([rr[..., :-1], rr[..., -1:] / aa.clamp(min=1e-10)], -1)
    ref = torch.zeros(N, 2, dtype=torch.float64)
    for i in range(0, H, 1):
        for j in range(0, W, 1):
            gr, = torch.autograd.grad((rr[i, j] * wt[0, i, j]).sum(), m2l, retain_graph=True)
            ref += gr.abs()
    assert float(ab.abs().max()) > 1.0
    assert float((ab - ref).abs().max()) < 1e-10 * float(ref.abs().max())
    # and it dominates the signed gradient, with equality only where every pixel pulls the same way
    assert bool((ab >= p["means"].grad.new_zeros(()).abs()).all())


def test_projection_and_degree_zero_colour_against_the_reference_trees_helpers():
    """The only anchors the reference tree itself offers for the rasterizer half (oracle/gen_golden_geometry.py): its pinhole projection
    (utils/geometry.py:39-57 after transform_points :7-21) fixes the oracle's world -> camera transform, means2d and depths for
    every point in front of the camera; SH2RGB (models/gaussians/basics.py:84-89) fixes the degree-0 colour the trainer forms
    (spherical_harmonics(0, .) + 0.5, vanilla.py:388-389).  The rest of gs_oracle stays parity-unpinned."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_geometry.npz"))
    t = lambda a: torch.from_numpy(z[a])
    N = z["points"].shape[0]
    viewmat = torch.linalg.inv(t("c2w"))
    quats = torch.tensor([1.0, 0, 0, 0], dtype=torch.float64).repeat(N, 1)
    scales = torch.full((N, 3), 0.05, dtype=torch.float64)
    W, H = 100000, 100000                                     # no image-bounds culling: compare every point in front of the camera
    K = t("K").clone(); K[0, 2] += W / 2; K[1, 2] += H / 2     # (principal point moved with it; undone below)
    radii, m2, depths, _, _ = G.project(t("points"), quats, scales, viewmat, K, W, H, near_plane=0.2)
    front = z["depth"] > 0.2
    assert front.sum() > 150 and (~front).sum() >= 0
    vis = (radii > 0).numpy()
    np.testing.assert_array_equal(vis, front)
    np.testing.assert_allclose(depths.numpy()[front], z["depth"][front], rtol=1e-12)
    np.testing.assert_allclose(depths.numpy()[front], z["cam_points"][front, 2], rtol=1e-12)
    uv = m2.numpy()[front] - np.array([W / 2, H / 2])
    exact = z["uv"][front] * ((z["depth"][front] + 1e-6) / z["depth"][front])[:, None]      # the reference divides by (depth + 1e-6)
    np.testing.assert_allclose(uv, exact, rtol=1e-9, atol=1e-7)
    dirs = torch.randn(N, 3, dtype=torch.float64)
    rgb = G.spherical_harmonics(0, dirs, t("sh_dc")[:, None, :]) + 0.5
    np.testing.assert_allclose(rgb.numpy(), z["rgb_from_sh"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(z["sh_from_rgb"], z["sh_dc"], rtol=1e-12, atol=1e-12)
