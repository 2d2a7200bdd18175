"""Cube-map sky on the device (csrc/envlight.hip through the C ABI) against the CPU oracle (oracle/cubemap_oracle.py, float32 path
for identical texel decisions, float64 for accuracy), plus the host mirrors: envlight.EnvLight and the nvdiffrast drop-in."""
import itertools
import os
import sys

import numpy as np
import pytest
import torch

from oracle import cubemap_oracle as CO

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def directions(n, seed):
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    special = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1), (-1, 1, -1),
               (0, 0, 0), (np.nan, 1, 0), (np.inf, 2, 1), (1e-30, 0, 0), (3, -3, 1)]
    d[:len(special)] = np.array(special, dtype=np.float32)
    # rays that graze edges and corners
    k = n // 4
    d[-k:, 0] = np.sign(d[-k:, 0]); d[-k:, 1] = np.sign(d[-k:, 1]) * (1 + rng.uniform(-1e-3, 1e-3, k).astype(np.float32))
    return d


@pytest.mark.parametrize("res,C", [(1, 3), (2, 3), (7, 3), (64, 3), (1024, 3), (16, 1), (16, 5)])
@pytest.mark.parametrize("use_rot", [False, True])
def test_forward_and_texture_gradient_equal_oracle(res, C, use_rot):
    from bilateral_driving_amd.envlight import cubemap_sample
    rng = np.random.default_rng(res * 10 + C)
    n = 40_000
    d = directions(n, seed=res + C)
    tex = rng.random((6, res, res, C)).astype(np.float32)
    rot = CO.TO_OPENGL.astype(np.float32) if use_rot else None
    t_tex = torch.from_numpy(tex).cuda().requires_grad_(True)
    out = cubemap_sample(t_tex, torch.from_numpy(d).cuda(), None if rot is None else torch.from_numpy(rot).cuda())
    ref32 = CO.cubemap_fwd(tex, d, rot, dtype=np.float32)
    got = out.detach().cpu().numpy()
    assert got.shape == (n, C)
    # same float32 decisions -> same taps; the weighted sums may differ in the last bits
    bad = np.abs(got - ref32) > 2e-6
    assert bad.mean() < 2e-4, bad.mean()                       # a floor() at an exact texel boundary may flip (weight ~ 0 or 1 texel)
    ref64 = CO.cubemap_fwd(tex.astype(np.float64), d.astype(np.float64), None if rot is None else rot.astype(np.float64))
    assert np.percentile(np.abs(got - ref64), 99.5) < 5e-4 * max(1, res / 64)
    v = rng.standard_normal((n, C)).astype(np.float32)
    v[::3] = 0                                                  # pixels without sky weight
    out.backward(torch.from_numpy(v).cuda())
    g = t_tex.grad.cpu().numpy()
    g_ref = CO.cubemap_bwd(tex.shape, d, v, rot, dtype=np.float32)
    denom = np.abs(g_ref).max() + 1e-12
    assert np.abs(g - g_ref).max() / denom < (1e-3 if res <= 64 else 2e-2)      # few texels collect thousands of atomics at low res
    # adjointness in float64 on the device result itself
    lhs = float((out.detach().double().cpu().numpy() * v).sum()); rhs = float((g.astype(np.float64) * tex).sum())
    assert abs(lhs - rhs) < 1e-3 * (abs(lhs) + 1)


@pytest.mark.parametrize("H,W,res", [(50, 70, 32), (16, 16, 4), (33, 129, 256), (270, 480, 1024)])
def test_image_shaped_directions_use_the_tiled_gradient_and_equal_oracle(H, W, res):
    """dirs [H,W,3] (the sky model's use): the backward pre-sums 16x16 pixel tiles in an LDS hash table; same gradient."""
    from bilateral_driving_amd.envlight import cubemap_sample
    rng = np.random.default_rng(H + W)
    jj, ii = np.meshgrid(np.arange(W) + 0.5, np.arange(H) + 0.5)
    f = 0.5 * W / np.tan(np.radians(35.0))
    d = np.stack([(jj - W / 2) / f, (ii - H / 2) / f, np.ones_like(jj)], -1).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d[0, 0] = np.nan; d[-1, -1] = 0
    tex = rng.random((6, res, res, 3)).astype(np.float32)
    v = (rng.standard_normal((H, W, 3)) * (rng.random((H, W, 1)) < 0.4)).astype(np.float32)
    rot = CO.TO_OPENGL.astype(np.float32)
    grads = []
    for shaped in (True, False):
        t_tex = torch.from_numpy(tex).cuda().requires_grad_(True)
        dd = torch.from_numpy(d).cuda()
        out = cubemap_sample(t_tex, dd if shaped else dd.reshape(-1, 3), torch.from_numpy(rot).cuda())
        out.backward(torch.from_numpy(v).cuda().reshape(out.shape))
        grads.append(t_tex.grad.cpu().numpy())
    g_ref = CO.cubemap_bwd(tex.shape, d, v, rot, dtype=np.float32)
    denom = np.abs(g_ref).max() + 1e-12
    assert np.abs(grads[0] - g_ref).max() / denom < 1e-4 and np.abs(grads[1] - g_ref).max() / denom < 1e-4
    assert np.abs(grads[0]).sum() > 0


def test_invalid_directions_and_constant_texture():
    from bilateral_driving_amd.envlight import cubemap_sample
    d = torch.tensor([[0.0, 0, 0], [float("nan"), 1, 0], [1.0, float("nan"), 0], [float("inf"), 3, -2], [1.0, 0, 0]], device="cuda")
    tex = torch.rand(6, 8, 8, 3, device="cuda")
    o = cubemap_sample(tex, d)
    assert torch.all(o[:3] == 0) and torch.allclose(o[3], o[4])
    dd = torch.randn(100_000, 3, device="cuda")
    assert torch.allclose(cubemap_sample(torch.full((6, 33, 33, 3), 0.25, device="cuda"), dd), torch.tensor(0.25, device="cuda"), atol=1e-6)


def test_continuous_across_edges_on_device():
    from bilateral_driving_amd.envlight import cubemap_sample
    res = 8
    tex = torch.rand(6, res, res, 3, device="cuda")
    t = torch.linspace(-0.85, 0.85, 60)
    worst = 0.0
    for a, b in itertools.permutations(range(3), 2):
        c = 3 - a - b
        for sa, sb in itertools.product((1, -1), repeat=2):
            da = torch.zeros(60, 3); db = torch.zeros(60, 3)
            da[:, a] = sa * (1 + 1e-6); da[:, b] = sb; da[:, c] = t
            db[:, a] = sa; db[:, b] = sb * (1 + 1e-6); db[:, c] = t
            worst = max(worst, float((cubemap_sample(tex, da.cuda()) - cubemap_sample(tex, db.cuda())).abs().max()))
    assert worst < 1e-4       # the two sides see the same two border texels with weights that differ by O(res * 1e-6)


def test_envlight_module_and_nvdiffrast_dropin():
    from bilateral_driving_amd.envlight import EnvLight, cubemap_sample
    sky = EnvLight("Sky", resolution=32)
    assert sky.base.shape == (6, 32, 32, 3) and float(sky.base.detach().min()) == 0.5 and list(sky.get_param_groups()) == ["Sky#all"]
    assert [k for k, _ in sky.state_dict().items()] == ["base"]            # loads the reference's checkpoint entry
    H, W = 24, 40
    dirs = torch.nn.functional.normalize(torch.randn(H, W, 3, device="cuda"), dim=-1)
    with torch.no_grad():
        sky.base.copy_(torch.rand_like(sky.base))
    out = sky({"viewdirs": dirs})
    assert out.shape == (H, W, 3)
    ref = CO.cubemap_fwd(sky.base.detach().cpu().numpy(), dirs.cpu().numpy(), CO.TO_OPENGL.astype(np.float32), dtype=np.float32)
    assert np.percentile(np.abs(out.detach().cpu().numpy() - ref), 99.5) < 1e-5
    out.square().sum().backward()
    assert sky.base.grad is not None and float(sky.base.grad.abs().sum()) > 0
    # the call the reference makes (models/modules.py:196-203) through the drop-in package
    sys.path.insert(0, os.path.join(ROOT, "bilateral_driving_amd", "dropin"))
    try:
        import nvdiffrast.torch as dr
        l = (dirs.reshape(-1, 3) @ sky.to_opengl.T).reshape(*dirs.shape).contiguous().reshape(1, 1, -1, 3)
        light = dr.texture(sky.base[None, ...], l, filter_mode="linear", boundary_mode="cube").view(H, W, -1)
        assert torch.allclose(light, out, atol=1e-5)
        with pytest.raises(NotImplementedError):
            dr.texture(sky.base[None, ...], l, filter_mode="linear", boundary_mode="wrap")
    finally:
        sys.path.pop(0)
        for m in [m for m in sys.modules if m == "nvdiffrast" or m.startswith("nvdiffrast.")]:
            del sys.modules[m]


def test_sky_feeds_the_fused_view_and_receives_its_gradient():
    """EnvLight -> sky input of the fused view node -> loss: the texture gradient arrives through v_sky."""
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd.envlight import EnvLight
    W, H, N = 192, 128, 2000
    cam = Hn.ring_cameras(W, H, yaws_deg=(0.0,), device="cuda")[0]
    p = {k: v.requires_grad_(True) for k, v in Hn.synthetic_scene(N, seed=1, device="cuda").items()}
    grids = [g.requires_grad_(True) for g in Hn.make_grids(1, device="cuda")]
    sky = EnvLight("Sky", resolution=16)
    jj, ii = torch.meshgrid(torch.arange(W, device="cuda") + 0.5, torch.arange(H, device="cuda") + 0.5, indexing="xy")
    Kinv = torch.linalg.inv(cam.K)
    dirs_cam = torch.stack([jj, ii, torch.ones_like(jj)], -1) @ Kinv.T
    c2w = torch.linalg.inv(cam.viewmat)
    dirs = torch.nn.functional.normalize(dirs_cam @ c2w[:3, :3].T, dim=-1)
    out = Hn.render_view(p, cam, grids, 0, sky({"viewdirs": dirs}))
    out["rgb"].mean().backward()
    g = sky.base.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
    assert float((g != 0).float().mean()) < 0.5           # one view sees a part of the cube only
