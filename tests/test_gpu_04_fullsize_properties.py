"""-m gpu: size-independent properties at the BENCHMARK's full size (2 M Gaussians, 1920x1080), where the CPU
oracle cannot run, plus edge cases (empty / fully culled inputs, multi-camera batches)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full():
    assert torch.cuda.is_available()
    from bilateral_driving_amd import harness as Hn
    import bilateral_driving_amd.gs_ops as ops
    dev = "cuda"
    N, W, H = 2_000_000, 1920, 1080
    cam = Hn.ring_cameras(W, H, device=dev)[0]
    p = Hn.synthetic_scene(N, seed=0, device=dev)
    opac = torch.sigmoid(p["opacity_logits"])
    scales = torch.exp(p["log_scales"])
    radii, m2, d, con, _ = ops.fully_fused_projection(p["means"], p["quats"], scales, cam.viewmat[None], cam.K[None], W, H, near_plane=0.1)
    return dict(Hn=Hn, ops=ops, N=N, W=W, H=H, cam=cam, p=p, opac=opac, radii=radii, m2=m2, d=d, con=con)


def test_fullsize_lists_are_ordered_and_consistent(full):
    ops, W, H = full["ops"], full["W"], full["H"]
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    for cull in (False, True):
        tpg, iids, fids, offs = ops.isect_tiles(full["m2"], full["radii"], full["d"], 16, tw, th,
                                                conics=full["con"] if cull else None, opacities=full["opac"][None] if cull else None)
        M = iids.numel()
        assert M == int(tpg.sum()) and M > 1_000_000
        assert bool((iids[1:] >= iids[:-1]).all())                      # (tile | depth) sorted
        tile = iids >> 32
        o = offs.reshape(-1).long()
        assert bool((o[1:] >= o[:-1]).all()) and int(o[0]) == 0
        cnt = torch.bincount(tile, minlength=tw * th)
        assert torch.equal(torch.cumsum(cnt, 0)[:-1], o[1:])            # offsets = lower bounds of every tile
        bits = full["d"].reshape(-1)[fids.long()].view(torch.int32).long()
        assert torch.equal(iids & 0xFFFFFFFF, bits)                     # low word = fp32 depth bits of the listed Gaussian
        assert bool((full["radii"].reshape(-1)[fids.long()] > 0).all())  # only visible Gaussians are listed
        assert int(torch.unique(fids).numel()) <= int((full["radii"] > 0).sum())


def test_fullsize_render_properties(full):
    ops, W, H, N = full["ops"], full["W"], full["H"], full["N"]
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    g = torch.Generator().manual_seed(0)
    c1 = torch.rand(1, N, 3, generator=g).cuda()
    c2 = torch.rand(1, N, 3, generator=g).cuda()
    op = full["opac"][None]
    _, _, fids_f, offs_f = ops.isect_tiles(full["m2"], full["radii"], full["d"], 16, tw, th, want_isect_ids=False)
    _, _, fids_c, offs_c = ops.isect_tiles(full["m2"], full["radii"], full["d"], 16, tw, th, want_isect_ids=False,
                                           conics=full["con"], opacities=op)
    assert fids_c.numel() < 0.6 * fids_f.numel()
    r1f, a1f = ops.rasterize_to_pixels(full["m2"], full["con"], c1, op, W, H, 16, offs_f, fids_f)
    r1, a1 = ops.rasterize_to_pixels(full["m2"], full["con"], c1, op, W, H, 16, offs_c, fids_c)
    assert torch.equal(r1, r1f) and torch.equal(a1, a1f)                 # exact tile culling is invisible, bit for bit
    r1b, _ = ops.rasterize_to_pixels(full["m2"], full["con"], c1, op, W, H, 16, offs_c, fids_c)
    assert torch.equal(r1, r1b)                                          # forward is deterministic
    r2, _ = ops.rasterize_to_pixels(full["m2"], full["con"], c2, op, W, H, 16, offs_c, fids_c)
    r12, a12 = ops.rasterize_to_pixels(full["m2"], full["con"], c1 + 2.0 * c2, op, W, H, 16, offs_c, fids_c)
    assert float((r12 - (r1 + 2.0 * r2)).abs().max()) < 2e-5             # linear in the colours
    assert torch.equal(a12, a1)                                          # alpha does not depend on colour
    assert float(a1.min()) >= 0.0 and float(a1.max()) <= 1.0
    ones = torch.ones(1, N, 1, device="cuda")
    r_one, a_one = ops.rasterize_to_pixels(full["m2"], full["con"], ones, op, W, H, 16, offs_c, fids_c)
    assert float((r_one[..., 0] - a_one[..., 0]).abs().max()) < 2e-5     # colour 1 everywhere -> rendered colour == alpha
    # backgrounds: out = colour + (1 - alpha) * bg
    bg = torch.tensor([[0.25, 0.5, 0.75]], device="cuda")
    rb, _ = ops.rasterize_to_pixels(full["m2"], full["con"], c1, op, W, H, 16, offs_c, fids_c, backgrounds=bg)
    assert float((rb - (r1 + (1 - a1) * bg[0])).abs().max()) < 1e-5


def test_fullsize_bilagrid_identity_and_training_step(full):
    Hn, W, H = full["Hn"], full["W"], full["H"]
    from bilateral_driving_amd.bilagrid import bilagrid_transform, total_variation_loss
    rgb = torch.rand(H, W, 3, device="cuda") * 1.2
    ident = [torch.tensor([1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0], device="cuda").reshape(1, 12, 1, 1, 1).repeat(1, 1, gl, gy, gx)
             for (gx, gy, gl) in Hn.LEVELS_3]
    out = bilagrid_transform(rgb, ident, Hn.FACTORS_3)
    assert float((out - rgb).abs().max()) < 1e-5                          # identity grids leave the image unchanged
    assert float(sum(total_variation_loss(g) for g in ident)) == 0.0
    # one full-size training step through the fused node: finite, every gradient path live
    p = {k: v.clone().requires_grad_(True) for k, v in full["p"].items()}
    grids = [g.requires_grad_(True) for g in Hn.make_grids(6, device="cuda")]
    sky, target = torch.rand(H, W, 3, device="cuda"), torch.rand(H, W, 3, device="cuda")
    o = Hn.render_view(p, full["cam"], grids, 0, sky)
    loss = Hn.training_loss(o, target, grids)
    loss.backward()
    assert math.isfinite(float(loss))
    for k, v in p.items():
        assert v.grad is not None and bool(torch.isfinite(v.grad).all()) and float(v.grad.abs().sum()) > 0, k
    vis = o["info"]["radii"][0] > 0
    assert float(p["sh"].grad[~vis].abs().max()) == 0.0                   # culled Gaussians get exactly zero gradient
    assert float(grids[0].grad[0].abs().sum()) > 0 and float(grids[2].grad[1:].abs().sum()) > 0  # slice + TV routes
    ag = o["info"]["means2d"].absgrad
    assert ag.shape == (1, full["N"], 2) and float(ag[0][~vis].abs().max()) == 0.0 and float(ag.sum()) > 0


def test_empty_and_fully_culled_inputs():
    import bilateral_driving_amd.rendering as R
    dev = "cuda"
    vm, K = torch.eye(4, device=dev)[None], torch.tensor([[50.0, 0, 32], [0, 50.0, 24], [0, 0, 1]], device=dev)[None]
    # all Gaussians behind the camera
    means = torch.randn(100, 3, device=dev) - torch.tensor([0.0, 0.0, 10.0], device=dev)
    means.requires_grad_(True)
    args = (torch.randn(100, 4, device=dev), torch.rand(100, 3, device=dev) * 0.1 + 0.01, torch.rand(100, device=dev),
            torch.rand(100, 3, device=dev))
    r, a, meta = R.rasterization(means, *args, vm, K, 64, 48, render_mode="RGB+ED", absgrad=True)
    assert int(meta["radii"].abs().sum()) == 0 and meta["flatten_ids"].numel() == 0
    assert float(r.abs().max()) == 0.0 and float(a.abs().max()) == 0.0
    (r.sum() + a.sum()).backward()
    assert float(means.grad.abs().max()) == 0.0
    # zero opacity Gaussians (invalid instance points, nodes/rigid.py:469) render nothing
    means2 = torch.randn(50, 3, device=dev) * 0.5 + torch.tensor([0.0, 0.0, 5.0], device=dev)
    r2, a2, _ = R.rasterization(means2, torch.randn(50, 4, device=dev), torch.rand(50, 3, device=dev) * 0.3 + 0.05,
                                torch.zeros(50, device=dev), torch.rand(50, 3, device=dev), vm, K, 64, 48)
    assert float(a2.abs().max()) == 0.0
    # 2-D Gaussians (one zero scale, gaussians/vanilla.py:132-135) are finite
    sc = torch.rand(50, 3, device=dev) * 0.3 + 0.05
    sc[:, 2] = 0.0
    r3, a3, _ = R.rasterization(means2, torch.randn(50, 4, device=dev), sc, torch.rand(50, device=dev), torch.rand(50, 3, device=dev),
                                vm, K, 64, 48)
    assert bool(torch.isfinite(r3).all()) and float(a3.max()) > 0


def test_multi_camera_batch_equals_loop():
    import bilateral_driving_amd.rendering as R
    from bilateral_driving_amd import harness as Hn
    dev = "cuda"
    W, H, N = 208, 120, 5000
    cams = Hn.ring_cameras(W, H, yaws_deg=(0.0, 40.0, -40.0), device=dev)
    p = Hn.synthetic_scene(N, seed=3, device=dev)
    p["means"] = p["means"] * torch.tensor([0.3, 0.3, 1.0], device=dev)
    base = dict(means=p["means"], quats=p["quats"], scales=torch.exp(p["log_scales"]), opacities=torch.sigmoid(p["opacity_logits"]),
                colors=torch.rand(N, 3, device=dev))
    vms, Ks = torch.stack([c.viewmat for c in cams]), torch.stack([c.K for c in cams])
    leaves = {k: v.clone().requires_grad_(True) for k, v in base.items()}
    r, a, meta = R.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"], vms, Ks,
                                 W, H, packed=False, absgrad=True, render_mode="RGB+ED")
    assert r.shape == (3, H, W, 4) and meta["radii"].shape == (3, N)
    wt = torch.randn(r.shape, device=dev)
    (r * wt).sum().backward()
    tot = {k: torch.zeros_like(v) for k, v in base.items()}
    for c in range(3):
        l2 = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        rc, ac, _ = R.rasterization(l2["means"], l2["quats"], l2["scales"], l2["opacities"], l2["colors"], vms[c:c + 1], Ks[c:c + 1],
                                    W, H, packed=False, render_mode="RGB+ED")
        assert torch.equal(rc[0], r[c]) and torch.equal(ac[0], a[c])
        (rc * wt[c:c + 1]).sum().backward()
        for k in tot:
            tot[k] += l2[k].grad
    for k in tot:
        assert float((leaves[k].grad - tot[k]).norm() / tot[k].norm()) < 1e-4, k


# ---- the other BASELINE.json configurations at their full size: size-independent properties through the fused view -----------
_CONFIGS = {
    # name: (Gaussians, rig yaws, W, H, grid levels, guidance factors)
    "c2_500k_1080p_single_scale": (500_000, (0.0,), 1920, 1080, ((16, 16, 8),), (1,)),
    "c3_2M_6cam_1600x900_3level": (2_000_000, None, 1600, 900, None, None),              # H = 900 is not a multiple of 16
    "c5_5M_5cam_1920x1280_4level": (5_000_000, "five", 1920, 1280, ((2, 2, 1), (4, 4, 2), (8, 8, 4), (16, 16, 8)), (8, 4, 4, 2)),
}


@pytest.mark.parametrize("name", sorted(_CONFIGS))
def test_other_baseline_configs_fullsize_properties(name):
    """configs[1], [2] and [4] of BASELINE.json at full size (the CPU oracle cannot run there): per view of the rig -- lists ordered
    and consistent with the offsets, the image equals the identity transform of the blended render when the grids are identities,
    linearity of the composite in the colours (SH dc shift), transmittance in [0, 1], every gradient path live and finite, culled
    Gaussians get exactly zero gradient, absgrad >= |grad|, accumulation over two views of the frame == sum of the views."""
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd.dist import FlatGradients, FrameExchange
    N, yaws, W, H, levels, factors = _CONFIGS[name]
    yaws = Hn.SIX_CAM_YAWS if yaws is None else (Hn.FIVE_CAM_YAWS if yaws == "five" else yaws)
    levels, factors = levels or Hn.LEVELS_3, factors or Hn.FACTORS_3
    dev = "cuda"
    cams = Hn.ring_cameras(W, H, yaws_deg=yaws, device=dev)
    base = Hn.synthetic_scene(N, seed=0, device=dev)
    g = torch.Generator().manual_seed(11)
    sky, target = torch.rand(H, W, 3, generator=g).to(dev), torch.rand(H, W, 3, generator=g).to(dev)
    # identity grids: the transform returns clamp(render) + sky * (1 - alpha)
    ident = [torch.tensor([1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0], device=dev).reshape(1, 12, 1, 1, 1).repeat(len(cams), 1, gl, gy, gx).contiguous()
             for (gx, gy, gl) in levels]
    with torch.no_grad():
        o = Hn.render_view(base, cams[0], ident, 0, sky, factors=factors)
        info = o["info"]
        M, nv = info["n_isects"], info["n_visible"]
        assert M > 100_000 and 0 < nv < N
        offs = info["isect_offsets"].reshape(-1).long()
        assert bool((offs[1:] >= offs[:-1]).all()) and int(offs[0]) == 0 and int(offs[-1]) <= M
        vis_ids = info["visible_ids"].long()
        assert bool((vis_ids[1:] > vis_ids[:-1]).all()) and torch.equal(vis_ids, (info["radii"][0] > 0).nonzero().squeeze(1))
        pos = info["flatten_ranks"].long()
        assert int(pos.min()) >= 0 and int(pos.max()) < nv
        alpha = o["opacity"]
        assert float(alpha.min()) >= 0.0 and float(alpha.max()) <= 1.0 + 1e-6
        blended = o["rgb_gaussians"] + sky * (1.0 - alpha)
        assert float((o["rgb"] - blended).abs().max()) < 2e-5
        assert float(alpha.mean()) > 0.2 and bool(torch.isfinite(o["depth"]).all())
    # gradients: two views of the frame accumulated through the flat buffer == the views' own gradients summed
    grids0 = Hn.make_grids(len(cams), levels=levels, device=dev)
    n_views = min(2, len(cams))
    ref = None
    for v in range(n_views):
        p = {k: t.clone().requires_grad_(True) for k, t in base.items()}
        grids = [x.clone().requires_grad_(True) for x in grids0]
        out = Hn.render_view(p, cams[v], grids, v, sky, factors=factors)
        Hn.training_loss(out, target, grids).backward()
        visible = out["info"]["radii"][0] > 0
        for k, t in p.items():
            assert t.grad is not None and bool(torch.isfinite(t.grad).all()) and float(t.grad.abs().sum()) > 0, k
        assert float(p["sh"].grad[~visible].abs().max()) == 0.0 and float(p["means"].grad[~visible].abs().max()) == 0.0
        m2 = out["info"]["means2d"]
        assert bool((m2.absgrad >= m2.grad.abs() - 1e-6 * m2.absgrad.abs().max()).all()) and float(m2.absgrad[0][~visible].abs().max()) == 0.0
        gflat = torch.cat([t.grad.reshape(-1) for t in list(p.values()) + grids])
        ref = gflat if ref is None else ref + gflat
        del out, p, grids
    p = {k: t.clone().requires_grad_(True) for k, t in base.items()}
    grids = [x.clone().requires_grad_(True) for x in grids0]
    flat = FlatGradients(list(p.values()) + grids, sparse_rows=True)
    fx = FrameExchange(flat, list(p.keys()) + [f"grid{i}" for i in range(len(grids))])
    fx.begin_frame()
    for v in range(n_views):
        out = Hn.render_view(p, cams[v], grids, v, sky, factors=factors, **fx.view_kwargs(v))
        fx.begin_view(out["info"])
        Hn.training_loss(out, target, grids).backward()
        fx.end_view()
    fx.end_frame()
    got = torch.cat([t.grad.reshape(-1) for t in list(p.values()) + grids])
    assert float((got - ref).norm() / ref.norm()) < 1e-4


_CONFIGS["lidar_1M_6cam_1080p_3level"] = (1_000_000, None, 1920, 1080, None, None)      # harness.lidar_scene: the small-splat end


@pytest.mark.parametrize("name", ["c3_2M_6cam_1600x900_3level", "c5_5M_5cam_1920x1280_4level", "lidar_1M_6cam_1080p_3level"])
def test_replayed_frame_at_full_size_equals_eager_frame(name):
    """What bench.py TIMES -- graph_view.FrameGraph.step(): device-side list counts, SH colours in the record pack, loss on the
    transform's launch, two streams, in-place gradient rows -- against the eager host-count frame on the same parameters, at
    BASELINE.json's configs[2] and [4] (bilateral_driving_amd/selfcheck.py: images bit-equal, per-view loss within 1e-5, the frame's
    flat gradient within 1e-4 norm-relative: atomics order only); two replays, the second after the first's row-wise clear."""
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd.graph_view import FrameGraph
    from bilateral_driving_amd.selfcheck import frame_against_eager
    N, yaws, W, H, levels, factors = _CONFIGS[name]
    yaws = Hn.SIX_CAM_YAWS if yaws is None else (Hn.FIVE_CAM_YAWS if yaws == "five" else yaws)
    levels, factors = levels or Hn.LEVELS_3, factors or Hn.FACTORS_3
    dev = "cuda"
    cams = Hn.ring_cameras(W, H, yaws_deg=yaws, device=dev)
    for c in cams:
        c.viewmat.requires_grad_(True)
    scene = Hn.lidar_scene(N, seed=0, device=dev) if name.startswith("lidar") else Hn.synthetic_scene(N, seed=0, device=dev)
    p = {k: t.requires_grad_(True) for k, t in scene.items()}
    grids = [g.requires_grad_(True) for g in Hn.make_grids(len(cams), levels=levels, device=dev)]
    gen = torch.Generator().manual_seed(13)
    skies = [torch.rand(H, W, 3, generator=gen).to(dev).requires_grad_(True) for _ in cams]
    targets = [torch.rand(H, W, 3, generator=gen).to(dev) for _ in cams]
    frame = FrameGraph(p, cams, grids, skies, targets, factors=factors)
    for rep in range(2):
        res = frame_against_eager(frame, p, cams, grids, skies, targets, factors)
        print(f"[fullsize frame {name} rep {rep}] {res}")
        assert res["ok"], res
        assert res["sky_grad_rel_err"] < 1e-5 and res["pose_grad_rel_err"] < 1e-3, res
    counts = frame.counts()
    assert all(M > 100_000 and 0 < nv < N for M, nv in counts), counts


def test_fullsize_coarse_lists_equal_16px_lists(full):
    """Benchmark size: the fused view through 64-px list tiles (its default) gives bit for bit the image of gsplat's 16-px lists and
    the same gradients; the 64-px lists are a fraction of the 16-px ones."""
    Hn, p0, cam, W, H = full["Hn"], full["p"], full["cam"], full["W"], full["H"]
    from bilateral_driving_amd.fused_view import fused_view
    grids0 = Hn.make_grids(1, device="cuda")
    g = torch.Generator().manual_seed(3)
    sky, target = torch.rand(H, W, 3, generator=g).cuda(), torch.rand(H, W, 3, generator=g).cuda()
    res = {}
    for lt in (16, 64):
        p = {k: v.detach().clone().requires_grad_(True) for k, v in p0.items()}
        grids = [x.clone().requires_grad_(True) for x in grids0]
        out = fused_view(p, cam.viewmat, cam.K, W, H, grids, sky, Hn.FACTORS_3, img_idx=0, cam_pos=cam.cam_pos, list_tile=lt)
        ((out["rgb"] - target).abs().mean() + 0.01 * out["depth"].mean()).backward()
        res[lt] = (out["rgb"].detach(), out["depth"].detach(), out["opacity"].detach(), {k: v.grad for k, v in p.items()},
                   [x.grad for x in grids], out["info"]["means2d"].absgrad, out["info"]["n_isects"])
        del out
    a, b = res[16], res[64]
    assert b[6] * 4 < a[6]                                              # several times fewer pairs to emit and sort
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for k in a[3]:
        assert float((a[3][k] - b[3][k]).norm()) <= 2e-5 * float(a[3][k].norm()), k          # atomics: summation order only
    for x, y in zip(a[4], b[4]):
        assert float((x - y).norm()) <= 2e-5 * float(x.norm())
    assert float((a[5] - b[5]).norm()) <= 2e-5 * float(a[5].norm())


def test_row_order_is_invisible_to_the_image_and_permutes_the_gradients():
    """densify.spatial_order: the same SET of Gaussians in another row order renders the same view (depth ties between DIFFERENT
    Gaussians in one tile are the only place where the order of the rows enters -- none in a random scene) and every gradient row
    moves with its Gaussian."""
    from bilateral_driving_amd import harness as Hn
    dev = "cuda"
    W, H, N = 960, 544, 200_000
    cam = Hn.ring_cameras(W, H, yaws_deg=(35.0,), device=dev)[0]
    cam.viewmat.requires_grad_(True)
    base = Hn.synthetic_scene(N, seed=11, device=dev)
    perm = Hn.spatial_order(base["means"])
    assert not torch.equal(perm, torch.arange(N, device=dev))
    gen = torch.Generator().manual_seed(3)
    sky = torch.rand(H, W, 3, generator=gen).to(dev)
    target = torch.rand(H, W, 3, generator=gen).to(dev)
    res = []
    for p in ({k: v.clone().requires_grad_(True) for k, v in base.items()}, Hn.reorder_params({k: v.requires_grad_(True) for k, v in base.items()}, perm)):
        grids = [g.requires_grad_(True) for g in Hn.make_grids(1, device=dev)]
        cam.viewmat.grad = None
        out = Hn.render_view(p, cam, grids, 0, sky)
        Hn.training_loss(out, target, grids).backward()
        res.append((out["rgb"].detach(), out["depth"].detach(), {k: v.grad for k, v in p.items()}, cam.viewmat.grad.clone(), out["info"]["n_visible"]))
    (rgb_a, d_a, g_a, vm_a, nv_a), (rgb_b, d_b, g_b, vm_b, nv_b) = res
    assert nv_a == nv_b > 10_000
    assert float((rgb_a - rgb_b).abs().max()) <= 1e-5 and float((d_a - d_b).abs().max()) <= 1e-4 * float(d_a.abs().max())
    for k in g_a:
        ref = g_a[k][perm]
        assert float((g_b[k] - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-9, k
    assert float((vm_a - vm_b).abs().max()) <= 1e-4 * float(vm_a.abs().max())
