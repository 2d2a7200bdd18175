"""-m gpu: the harness-level forms of the path (one fused autograd node per view, gradient arenas, row-wise buffers)
against the operator-by-operator form that tests/test_gpu_01_gs_parity.py checks against the oracle."""
import pytest
import torch

from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    import bilateral_driving_amd.gs_ops as ops
    from bilateral_driving_amd import _lib
    _lib.lib()  # fails loudly if libbds.so is missing
    return ops


def test_harness_view_matches_rasterization_api(ops):
    """harness.render_view (stage ops, SH after projection with the visibility mask, culling) ==
    the reference-shaped call sequence through rasterization() + separate torch post-processing."""
    import bilateral_driving_amd.rendering as R
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd.bilagrid import bilagrid_transform
    dev = "cuda"
    W, H, N = 320, 192, 4000
    cam = Hn.ring_cameras(W, H, yaws_deg=(0.0,), device=dev)[0]
    base = Hn.synthetic_scene(N, seed=1, device=dev)
    base["means"] = base["means"] * torch.tensor([0.3, 0.3, 1.0], device=dev)
    grids0 = Hn.make_grids(2, device=dev)
    sky = torch.rand(H, W, 3, device=dev)
    target = torch.rand(H, W, 3, device=dev)
    outs = []
    for mode in ("fused", "staged", "api"):
        p = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        grids = [g.clone().requires_grad_(True) for g in grids0]
        if mode in ("fused", "staged"):
            Hn.FUSED = True if mode == "fused" else "ops"      # "ops": the chain of individual operators
            try:
                out = Hn.render_view(p, cam, grids, 1, sky)
            finally:
                Hn.FUSED = True
            rgb, depth = out["rgb"], out["depth"]
            absg_holder = out["info"]["means2d"]
        else:
            dirs = p["means"].detach() - torch.linalg.inv(cam.viewmat)[:3, 3]
            col = torch.clamp(ops.spherical_harmonics(3, dirs, p["sh"]) + 0.5, 0.0, 1.0)
            rr, aa, info = R.rasterization(p["means"], p["quats"] / p["quats"].norm(dim=-1, keepdim=True), torch.exp(p["log_scales"]),
                                           torch.sigmoid(p["opacity_logits"]), col, cam.viewmat[None], cam.K[None], W, H,
                                           packed=False, absgrad=True, near_plane=0.1, render_mode="RGB+ED")
            rgb_g = torch.clamp(rr[0][..., :3], max=1.0)
            blended = rgb_g + sky * (1.0 - aa[0])
            rgb = bilagrid_transform(blended, [g[1:2] for g in grids], Hn.FACTORS_3)
            depth = rr[0][..., 3:4]
        loss = (rgb - target).abs().mean() + 0.1 * depth.mean() * 0.01
        loss.backward()
        if mode != "api":
            assert absg_holder.absgrad.shape == (1, N, 2) and float(absg_holder.absgrad.sum()) > 0
        outs.append((rgb.detach(), depth.detach(), {k: v.grad.clone() for k, v in p.items()}, [g.grad.clone() for g in grids]))
    for o in outs[:2]:
        assert rel_err(o[0], outs[2][0]) < 1e-5 and rel_err(o[1], outs[2][1]) < 1e-5
        for k in o[2]:
            a, b = o[2][k], outs[2][2][k]
            assert float((a - b).norm() / b.norm()) < 1e-4, k
        for a, b in zip(o[3], outs[2][3]):
            assert float((a - b).norm() / b.norm()) < 1e-4


def test_fused_view_writes_gradients_into_flat_buffer(ops):
    """Multi-GPU plumbing on one GPU: with grad_arena the backward kernels write the per-Gaussian gradients into
    the flat all-reduce buffer and autograd adopts those slices as .grad (no pack copy)."""
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd.dist import FlatGradients
    dev = "cuda"
    W, H, N = 256, 160, 3000
    cam = Hn.ring_cameras(W, H, yaws_deg=(0.0,), device=dev)[0]
    base = Hn.synthetic_scene(N, seed=2, device=dev)
    base["means"] = base["means"] * torch.tensor([0.3, 0.3, 1.0], device=dev)
    grids0 = Hn.make_grids(1, device=dev)
    sky, target = torch.rand(H, W, 3, device=dev), torch.rand(H, W, 3, device=dev)
    res = []
    for use_arena in (False, True):
        p = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        grids = [g.clone().requires_grad_(True) for g in grids0]
        flat = FlatGradients(list(p.values()) + grids)
        arena = flat.arena(list(p.keys())) if use_arena else None
        flat.zero()
        out = Hn.render_view(p, cam, grids, 0, sky, grad_arena=arena)
        Hn.training_loss(out, target, grids).backward()
        if use_arena:
            for k, v in p.items():
                assert v.grad.data_ptr() == arena[k].data_ptr(), k  # adopted in place
        res.append(flat.pack().clone())
    assert float((res[0] - res[1]).norm() / res[0].norm()) < 1e-4
    assert float(res[0].abs().sum()) > 0


def test_sparse_row_gradient_buffer_matches_dense_over_rotating_views(ops):
    """Opt-in FlatGradients(sparse_rows=True) + fused_view(arena_rows=1): the persistent buffer is cleared row-wise (only what the
    last step wrote) and the backward stores only the rows of the Gaussians the view sees.  Over steps that rotate through views
    with different visible sets every step's gradients equal the ones the dense form produces."""
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd.dist import FlatGradients
    dev = "cuda"
    W, H, N = 256, 160, 6000
    cams = Hn.ring_cameras(W, H, yaws_deg=(0.0, 120.0, 240.0, 60.0), device=dev)
    base = Hn.synthetic_scene(N, seed=3, device=dev)
    grids0 = Hn.make_grids(len(cams), device=dev)
    sky, target = torch.rand(H, W, 3, device=dev), torch.rand(H, W, 3, device=dev)
    res, seen = {}, []
    for sparse in (False, True):
        p = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        grids = [g.clone().requires_grad_(True) for g in grids0]
        flat = FlatGradients(list(p.values()) + grids, sparse_rows=sparse)
        arena = flat.arena(list(p.keys()))
        n_gauss = sum(x.numel() for x in p.values())
        out_steps = []
        for step in range(6):
            v = step % len(cams)
            flat.zero()
            assert flat.rows_clean == sparse
            if sparse:
                assert float(flat.flat[:n_gauss].abs().sum()) == 0.0          # really all-zero again
            out = Hn.render_view(p, cams[v], grids, v, sky, grad_arena=arena, arena_rows=1 if flat.rows_clean else 0)
            vis = out["info"]["radii"][0] > 0
            flat.mark_list(out["info"]["visible_ids"])
            if not sparse:
                seen.append(int(vis.sum()))
            Hn.training_loss(out, target, grids).backward()
            for k, t in p.items():
                assert t.grad.data_ptr() == arena[k].data_ptr(), k
            assert float(p["sh"].grad[~vis].abs().sum()) == 0.0 and float(p["means"].grad[~vis].abs().sum()) == 0.0
            out_steps.append(flat.pack().clone())
        res[sparse] = out_steps
    assert 0 < min(seen) and max(seen) < N and len(set(seen)) > 1      # the views cull different, proper subsets
    for a, b in zip(res[False], res[True]):
        assert float(a.abs().sum()) > 0
        assert float((a - b).norm() / a.norm()) < 1e-4      # the composite's atomics make two runs differ in the last bits


def test_views_of_a_frame_accumulate_into_the_arena(ops):
    """fused_view(arena_rows=2): the second and later views of a frame ADD their visible rows to the buffer that autograd already holds
    as .grad (one exchange per frame); the result equals the sum of the views' separately computed gradients."""
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd.dist import FlatGradients
    dev = "cuda"
    W, H, N = 256, 160, 6000
    cams = Hn.ring_cameras(W, H, yaws_deg=(0.0, 100.0, 200.0), device=dev)
    base = Hn.synthetic_scene(N, seed=4, device=dev)
    grids0 = Hn.make_grids(len(cams), device=dev)
    sky, target = torch.rand(H, W, 3, device=dev), torch.rand(H, W, 3, device=dev)
    # reference: dense gradients of every view, summed
    ref = None
    for v, cam in enumerate(cams):
        p = {k: t.clone().requires_grad_(True) for k, t in base.items()}
        grids = [g.clone().requires_grad_(True) for g in grids0]
        Hn.training_loss(Hn.render_view(p, cam, grids, v, sky), target, grids).backward()
        g = torch.cat([t.grad.reshape(-1) for t in list(p.values()) + grids])
        ref = g if ref is None else ref + g
    p = {k: t.clone().requires_grad_(True) for k, t in base.items()}
    grids = [g.clone().requires_grad_(True) for g in grids0]
    flat = FlatGradients(list(p.values()) + grids, sparse_rows=True)
    arena = flat.arena(list(p.keys()))
    for frame in range(2):          # twice: the second frame starts from the row-wise cleared buffer
        flat.zero()
        for v, cam in enumerate(cams):
            out = Hn.render_view(p, cam, grids, v, sky, grad_arena=arena, arena_rows=1 if v == 0 else 2)
            flat.mark_list(out["info"]["visible_ids"])
            Hn.training_loss(out, target, grids).backward()
        for k, t in p.items():
            assert t.grad.data_ptr() == arena[k].data_ptr(), k
        got = flat.pack()
        assert float((got - ref).norm() / ref.norm()) < 1e-4, frame


@pytest.mark.parametrize("N,W,H,seed,pull", [(1000, 256, 256, 0, 0.25), (3000, 320, 200, 1, 0.3), (400, 75, 50, 2, 0.2),
                                             (50_000, 640, 360, 3, 0.6)])   # (the last: ~1 min of float64 oracle on the host)
def test_fused_view_against_oracle_incl_pose_gradient_and_absgrad(ops, N, W, H, seed, pull):
    """The fused view (what bench.py times) against the float64 oracle: image 1e-4 rel, all gradients incl. the camera-pose
    gradient and the VALUE of info["means2d"].absgrad 1e-3-class; retain_grad() on info["means2d"] as trainers/base.py:429-430 does."""
    import __graft_entry__ as E
    E.smoke_check(verbose=True, N=N, W=W, H=H, seed=seed, pull=pull)


@pytest.mark.parametrize("views", [1, 3])
def test_replayed_frame_against_oracle_directly(ops, views):
    """graph_view.FrameGraph.step() -- the path bench.py times: device-side list counts, SH colours in the record pack, the loss on the
    colour transform's launch, hipGraph replay (one view: one graph each way on one stream; three: forwards on a second stream) --
    against gs_oracle + bilagrid_oracle DIRECTLY at configs[0]'s size: every image 1e-4, every parameter / grid / sky / pose gradient
    1e-3-class, the persistent screen-space gradient arrays against the oracle's absgrad probe."""
    import __graft_entry__ as E
    E.smoke_check(verbose=True, N=1000, W=256, H=256, seed=0, pull=0.25, views=views, graph=True)


def test_eager_multi_view_frame_against_oracle(ops):
    import __graft_entry__ as E
    E.smoke_check(verbose=True, N=1000, W=256, H=256, seed=4, pull=0.25, views=3, graph=False)


def test_rasterization_api_pose_gradient_and_fused_equivalence(ops):
    """rasterization() (the gsplat-shaped API) and the fused view give the same camera-pose gradient."""
    import bilateral_driving_amd.rendering as R
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd.bilagrid import bilagrid_transform
    dev = "cuda"
    W, H, N = 256, 160, 3000
    base = Hn.synthetic_scene(N, seed=5, device=dev)
    base["means"] = base["means"] * torch.tensor([0.3, 0.3, 1.0], device=dev)
    grids0 = Hn.make_grids(1, device=dev)
    sky, target = torch.rand(H, W, 3, device=dev), torch.rand(H, W, 3, device=dev)
    res = []
    for mode in ("fused", "api"):
        cam = Hn.ring_cameras(W, H, yaws_deg=(0.0,), device=dev)[0]
        cam.viewmat = cam.viewmat.clone().requires_grad_(True)
        p = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        grids = [g.clone().requires_grad_(True) for g in grids0]
        if mode == "fused":
            out = Hn.render_view(p, cam, grids, 0, sky)
            rgb = out["rgb"]
            assert float((out["rgb_gaussians"].max())) <= 1.0      # base.py:414 clamp(max=1.0)
            assert "original_rgb" in out                           # base.py:496-498: the colour entering the transform
            assert torch.equal(out["original_rgb"], out["rgb_gaussians"] + sky * (1.0 - out["opacity"].detach()))
        else:
            dirs = p["means"].detach() - torch.linalg.inv(cam.viewmat.detach())[:3, 3]
            col = torch.clamp(ops.spherical_harmonics(3, dirs, p["sh"]) + 0.5, 0.0, 1.0)
            rr, aa, info = R.rasterization(p["means"], p["quats"] / p["quats"].norm(dim=-1, keepdim=True), torch.exp(p["log_scales"]),
                                           torch.sigmoid(p["opacity_logits"]), col, cam.viewmat[None], cam.K[None], W, H,
                                           packed=False, absgrad=True, near_plane=0.1, render_mode="RGB+ED")
            rgb = bilagrid_transform(torch.clamp(rr[0][..., :3], max=1.0) + sky * (1.0 - aa[0]), [g[0:1] for g in grids], Hn.FACTORS_3)
        (rgb - target).abs().mean().backward()
        res.append(cam.viewmat.grad.clone())
    assert float(res[1].abs().max()) > 0
    assert float((res[0] - res[1]).norm() / res[1].norm()) < 1e-3


@pytest.mark.parametrize("drop_grid_grads", [False, True], ids=["in_place", "caller_dropped_the_grid_grads"])
def test_frame_loop_world1_grids_accumulate_in_place(ops, drop_grid_grads):
    """dist.FrameExchange at world size 1 (the bench's single-GPU loop): per-Gaussian rows through the arena modes, the grids'
    gradients (transform + TV term) ADDED in place to their .grad slices -- equals the sum of the views' separate gradients."""
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd.dist import FlatGradients, FrameExchange
    dev = "cuda"
    W, H, N = 256, 160, 5000
    cams = Hn.ring_cameras(W, H, yaws_deg=(0.0, 120.0, 240.0), device=dev)
    base = Hn.synthetic_scene(N, seed=6, device=dev)
    grids0 = Hn.make_grids(len(cams), device=dev)
    sky, target = torch.rand(H, W, 3, device=dev), torch.rand(H, W, 3, device=dev)
    ref = None
    for v, cam in enumerate(cams):
        p = {k: t.clone().requires_grad_(True) for k, t in base.items()}
        grids = [g.clone().requires_grad_(True) for g in grids0]
        Hn.training_loss(Hn.render_view(p, cam, grids, v, sky), target, grids).backward()
        g = torch.cat([t.grad.reshape(-1) for t in list(p.values()) + grids])
        ref = g if ref is None else ref + g
    p = {k: t.clone().requires_grad_(True) for k, t in base.items()}
    grids = [g.clone().requires_grad_(True) for g in grids0]
    flat = FlatGradients(list(p.values()) + grids, sparse_rows=True)
    fx = FrameExchange(flat, list(p.keys()) + [f"grid{i}" for i in range(len(grids))])
    assert not fx.active and fx.tail_grads() is not None and len(fx.tail_grads()) == len(grids)
    for frame in range(2):
        fx.begin_frame()
        if drop_grid_grads:          # e.g. an optimizer.zero_grad(set_to_none=True) after begin_frame: the in-place route must notice
            for g in grids:
                g.grad = None
        for v, cam in enumerate(cams):
            out = Hn.render_view(p, cam, grids, v, sky, **fx.view_kwargs(v))
            fx.begin_view(out["info"])
            Hn.training_loss(out, target, grids, grid_grads=None if drop_grid_grads else fx.tail_grads()).backward()
            fx.end_view()
        fx.end_frame()
        for g, a in zip(grids, fx.tail_grads()):
            assert g.grad is not None and (drop_grid_grads or g.grad.data_ptr() == a.data_ptr())
        got = torch.cat([t.grad.reshape(-1) for t in list(p.values()) + grids])
        assert float((got - ref).norm() / ref.norm()) < 1e-4, frame
        gg, rg = got[-sum(g.numel() for g in grids):], ref[-sum(g.numel() for g in grids):]
        assert float((gg - rg).norm() / rg.norm()) < 1e-4, frame


def test_train_view_direct_path_equals_autograd_path(ops):
    """harness.train_view (forward + loss + both backward passes without an autograd graph) leaves the same loss, images and
    gradients, in the same places, as render_view -> training_loss -> backward()."""
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd.dist import FlatGradients, FrameExchange
    dev = "cuda"
    W, H, N = 256, 160, 5000
    cams = Hn.ring_cameras(W, H, yaws_deg=(0.0, 120.0), device=dev)
    base = Hn.synthetic_scene(N, seed=8, device=dev)
    grids0 = Hn.make_grids(len(cams), device=dev)
    sky0, target = torch.rand(H, W, 3, device=dev), torch.rand(H, W, 3, device=dev)
    res = {}
    for mode in ("autograd", "direct", "direct_plain"):
        p = {k: t.clone().requires_grad_(True) for k, t in base.items()}
        grids = [g.clone().requires_grad_(True) for g in grids0]
        sky = sky0.clone().requires_grad_(True)
        for cam in cams:
            cam.viewmat.grad = None
            cam.viewmat.requires_grad_(True)
        losses, absg = [], []
        if mode == "direct_plain":            # no arena at all: gradients are adopted / added as autograd would
            for v, cam in enumerate(cams):
                out = Hn.train_view(p, cam, grids, v, sky, target)
                losses.append(float(out["loss"]))
                absg.append(out["info"]["means2d"].absgrad.clone())
        else:
            flat = FlatGradients(list(p.values()) + grids, sparse_rows=True)
            fx = FrameExchange(flat, list(p.keys()) + [f"grid{i}" for i in range(len(grids))])
            fx.begin_frame()
            for v, cam in enumerate(cams):
                if mode == "autograd":
                    out = Hn.render_view(p, cam, grids, v, sky, **fx.view_kwargs(v))
                    fx.begin_view(out["info"])
                    loss = Hn.training_loss(out, target, grids, grid_grads=fx.tail_grads())
                    loss.backward()
                else:
                    out = Hn.train_view(p, cam, grids, v, sky, target, grid_grads=fx.tail_grads(), after_forward=fx.begin_view,
                                        **fx.view_kwargs(v))
                    loss = out["loss"]
                    assert not out["rgb"].requires_grad and out["rgb"].grad_fn is None
                fx.end_view()
                losses.append(float(loss))
                absg.append(out["info"]["means2d"].absgrad.clone())
            fx.end_frame()
        res[mode] = (losses, torch.cat([t.grad.reshape(-1) for t in list(p.values()) + grids]).clone(), sky.grad.clone(),
                     [c.viewmat.grad.clone() for c in cams], absg, out["rgb"].detach().clone())
    a = res["autograd"]
    for mode in ("direct", "direct_plain"):
        b = res[mode]
        assert all(abs(x - y) <= 1e-6 * abs(x) for x, y in zip(a[0], b[0])), (a[0], b[0])   # same kernels; the L1 sum uses float atomics
        assert torch.equal(a[5], b[5])
        assert float((a[1] - b[1]).norm() / a[1].norm()) < 2e-5             # atomics: summation order only
        assert float((a[2] - b[2]).norm() / a[2].norm()) < 1e-6
        for x, y in zip(a[3], b[3]):
            assert float((x - y).norm() / x.norm()) < 1e-4
        for x, y in zip(a[4], b[4]):
            assert float((x - y).norm() / x.norm()) < 2e-5


def test_marshalled_scene_graph_route_equals_the_view_node():
    """the reference-named route -- per-class get_gaussians (SH colours at the active degree, activations) ->
    collect_gaussians (concatenation + class labels) -> render_gaussians (rasterization(), clamp, retain_grad; per-class re-render
    through render_fn(mask)) -- gives the image, depth, opacity and every parameter gradient of harness.render_view on the same
    Gaussians held as two classes."""
    import types
    from bilateral_driving_amd import harness as Hn, marshalling as M
    dev = "cuda"
    W, H, N = 320, 192, 5000
    cam = Hn.ring_cameras(W, H, yaws_deg=(0.0,), device=dev)[0]
    base = Hn.synthetic_scene(N, seed=4, device=dev)
    base["means"] = base["means"] * torch.tensor([0.3, 0.3, 1.0], device=dev)
    cut = 3200
    target = torch.rand(H, W, 3, device=dev)

    def as_class(sl):
        m = types.SimpleNamespace(sh_degree=3, step=5000, ctrl_cfg=types.SimpleNamespace(sh_degree_interval=1000))
        m._means = base["means"][sl].clone().requires_grad_(True)
        m._features_dc = base["sh"][sl, 0].clone().requires_grad_(True)
        m._features_rest = base["sh"][sl, 1:].clone().requires_grad_(True)
        m._opacities = base["opacity_logits"][sl, None].clone().requires_grad_(True)
        m._scales = base["log_scales"][sl].clone().requires_grad_(True)
        m._quats = base["quats"][sl].clone().requires_grad_(True)
        return m
    models = {"Background": as_class(slice(0, cut)), "RigidNodes": as_class(slice(cut, N))}
    classes = {"Background": 0, "RigidNodes": 1}
    c2w = torch.linalg.inv(cam.viewmat)
    camera = M.process_camera({"camera_to_world": c2w, "intrinsics": cam.K, "height": H, "width": W}, torch.tensor([0]))
    gs, labels = M.collect_gaussians(models, classes, camera)
    assert labels.shape == (N,) and int((labels == 1).sum()) == N - cut
    results, render_fn, info = M.render_gaussians(gs, camera, near_plane=0.1, far_plane=1e10, render_mode="RGB+ED", radius_clip=0.0)
    loss = (results["rgb_gaussians"] - target).abs().mean() + 0.001 * results["depth"].mean() + 0.01 * results["opacity"].mean()
    loss.backward()
    assert info["means2d"].grad is not None and info["means2d"].absgrad.shape == (1, N, 2)

    p = {k: v.clone().requires_grad_(True) for k, v in base.items()}
    out = Hn.render_view_staged(p, cam, Hn.make_grids(1, device=dev), 0, torch.zeros(H, W, 3, device=dev))
    rgb_g = torch.clamp(out["rgb_gaussians"], max=1.0)                       # base.py:414 (the staged chain clamps inside the grid op)
    loss2 = (rgb_g - target).abs().mean() + 0.001 * out["depth"].mean() + 0.01 * out["opacity"].mean()
    loss2.backward()
    torch.testing.assert_close(results["rgb_gaussians"], rgb_g, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(results["depth"], out["depth"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(results["opacity"], out["opacity"], rtol=1e-4, atol=2e-5)

    def rel(a, b):
        return float((a - b).norm() / b.norm().clamp_min(1e-20))
    cat = lambda name: torch.cat([getattr(models["Background"], name).grad, getattr(models["RigidNodes"], name).grad])
    assert rel(cat("_means"), p["means"].grad) < 1e-3
    assert rel(cat("_scales"), p["log_scales"].grad) < 1e-3
    assert rel(cat("_quats"), p["quats"].grad) < 1e-3
    assert rel(cat("_opacities")[:, 0], p["opacity_logits"].grad) < 1e-3
    assert rel(torch.cat([cat("_features_dc")[:, None], cat("_features_rest")], 1), p["sh"].grad) < 1e-3
    # the evaluation path's per-class render: the class mask multiplies the opacities (scene_graph.py:296-313)
    with torch.no_grad():
        rgb_bg, _, op_bg = render_fn((labels == 0).float())
        rgb_dyn, _, op_dyn = render_fn((labels == 1).float())
    assert float(op_bg.max()) <= 1.0 + 1e-5 and float(op_dyn.sum()) > 0
    assert float((op_bg + op_dyn - results["opacity"].detach()).min()) > -1e-4     # two partial scenes cover at least the joint one
