"""End-to-end regression on the device: sky cube -> fused view -> image loss + TV -> FusedAdam -> densification statistics ->
refinement_after, a short version of scripts/train_loop_demo.py.  The optimisation has to make progress (PSNR against targets
rendered from a ground-truth scene rises by several dB) and the densification has to change the set while the optimiser keeps working."""
import math
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_short_training_run_converges_and_densifies():
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd.bilagrid import total_variation_loss
    from bilateral_driving_amd.densify import refinement_after
    from bilateral_driving_amd.envlight import EnvLight
    from bilateral_driving_amd.losses import pixel_loss, ssim_loss
    from bilateral_driving_amd.optim import DensifyStats, FusedAdam
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    W, H, N_GT = 320, 192, 20_000
    cams = Hn.ring_cameras(W, H, device=dev)[:3]

    def view_dirs(cam):
        jj, ii = torch.meshgrid(torch.arange(W, device=dev) + 0.5, torch.arange(H, device=dev) + 0.5, indexing="xy")
        d = torch.stack([jj, ii, torch.ones_like(jj)], -1) @ torch.linalg.inv(cam.K).T @ torch.linalg.inv(cam.viewmat)[:3, :3].T
        return torch.nn.functional.normalize(d, dim=-1).contiguous()
    dirs = [view_dirs(c) for c in cams]
    gt = Hn.synthetic_scene(N_GT, seed=1, device=dev)
    gt_sky = EnvLight("Sky", resolution=16)
    with torch.no_grad():
        gt_sky.base.copy_(torch.rand_like(gt_sky.base))
    gt_grids = Hn.make_grids(len(cams), seed=3, device=dev)
    targets, sky_masks, lidar = [], [], []
    with torch.no_grad():
        for v, cam in enumerate(cams):
            out = Hn.render_view(gt, cam, gt_grids, v, gt_sky({"viewdirs": dirs[v]}))
            targets.append(out["rgb"].clone())
            sky_masks.append((out["opacity"].squeeze(-1) < 0.5).float())
            lidar.append(out["depth"].squeeze(-1) * ((torch.rand(H, W, device=dev) < 0.3) & (out["opacity"].squeeze(-1) > 0.9)))
    sel = torch.randperm(N_GT, device=dev)[: N_GT // 2]
    m = types.SimpleNamespace(
        class_prefix="Background#", scene_scale=30.0, num_train_images=len(cams), step=0, xys_grad_norm=None, vis_counts=None, max_2Dsize=None,
        ctrl_cfg=types.SimpleNamespace(warmup_steps=10, reset_alpha_interval=3000, refine_interval=50, n_split_samples=2, reset_alpha_value=0.01,
                                       densify_grad_thresh=0.0002, densify_size_thresh=0.002, cull_alpha_thresh=0.005, cull_scale_thresh=0.5,
                                       cull_screen_size=0.15, split_screen_size=0.05, stop_screen_size_at=4000, stop_split_at=15000))
    P = torch.nn.Parameter
    m._means, m._quats, m._scales = P(gt["means"][sel] + 0.05 * torch.randn(len(sel), 3, device=dev)), P(gt["quats"][sel].clone()), P(gt["log_scales"][sel] + 0.2)
    m._features_dc, m._features_rest = P(torch.zeros(len(sel), 3, device=dev)), P(torch.zeros(len(sel), 15, 3, device=dev))
    m._opacities = P(torch.full((len(sel), 1), -1.0, device=dev))
    sky = EnvLight("Sky", resolution=16)
    grids = [g.requires_grad_(True) for g in Hn.make_grids(len(cams), seed=0, device=dev)]
    names = dict(_means=("xyz", 1.6e-3), _features_dc=("sh_dc", 2.5e-3), _features_rest=("sh_rest", 1.25e-4), _opacities=("opacity", 5e-2),
                 _scales=("scaling", 5e-3), _quats=("rotation", 1e-3))
    groups = [{"params": [getattr(m, a)], "name": m.class_prefix + n, "lr": lr, "eps": 1e-15, "weight_decay": 0} for a, (n, lr) in names.items()]
    groups += [{"params": [g], "name": f"Affine#grid{i}", "lr": 2e-3, "eps": 1e-15, "weight_decay": 0} for i, g in enumerate(grids)]
    groups += [{"params": [sky.base], "name": "Sky#all", "lr": 1e-2, "eps": 1e-15, "weight_decay": 0}]
    opt = FusedAdam(groups, lr=0.0, eps=1e-15)
    tvw = [0.01 * 0.5 * math.sqrt(g.shape[4] * g.shape[3] * g.shape[2]) for g in grids]

    def params():
        return dict(means=m._means, quats=m._quats, log_scales=m._scales, opacity_logits=m._opacities.squeeze(-1),
                    sh=torch.cat([m._features_dc[:, None, :], m._features_rest], dim=1))

    def psnr():
        with torch.no_grad():
            mse = sum(float(((Hn.render_view(params(), c, grids, v, sky({"viewdirs": dirs[v]}))["rgb"] - targets[v]) ** 2).mean())
                      for v, c in enumerate(cams))
        return -10 * math.log10(mse / len(cams))

    p0, n0, stats, sizes = psnr(), m._means.shape[0], None, []
    for step in range(1, 201):
        m.step = step
        v = step % len(cams)
        opt.zero_grad(set_to_none=True)
        out = Hn.render_view(params(), cams[v], grids, v, sky({"viewdirs": dirs[v]}))
        loss = pixel_loss(out["rgb"], out["opacity"], out["depth"], targets[v], sky_masks[v], lidar[v]).sum() + 0.2 * ssim_loss(out["rgb"], targets[v])
        for g, w in zip(grids, tvw):
            loss = loss + total_variation_loss(g, w)
        loss.backward()
        assert torch.isfinite(loss)
        opt.step()
        stats = stats or DensifyStats(m._means.shape[0], dev)
        stats.update(out["info"])
        if step % m.ctrl_cfg.refine_interval == 0:
            m.xys_grad_norm, m.vis_counts, m.max_2Dsize = stats.xys_grad_norm, stats.vis_counts, stats.max_2Dsize
            refinement_after(m, step, opt, verbose=False)
            stats = None
            sizes.append(m._means.shape[0])
            for a in names:   # optimiser state follows the new set
                st = opt.state[getattr(m, a)]
                assert st["exp_avg"].shape == getattr(m, a).shape and st["exp_avg_sq"].shape == getattr(m, a).shape
    p1 = psnr()
    print(f"[train_loop] PSNR {p0:.2f} -> {p1:.2f} dB, Gaussians {n0} -> {sizes}")
    assert p1 > p0 + 4.0, (p0, p1)
    assert len(set(sizes + [n0])) > 1, sizes                     # densification changed the number of Gaussians
    assert all(torch.isfinite(getattr(m, a)).all() for a in names) and torch.isfinite(sky.base).all()


def test_replayed_training_loop_with_random_views_converges():
    """The reference's loop shape (tools/train.py:250-283: a random image per step) on ONE captured view: graph_view.FrameGraph(
    dynamic=True) replays it with another camera / target / image index every step, FusedAdam(consume_grads=True) clears every
    gradient as it consumes it (FrameGraph(clear_grads=False): no clearing pass), step() returns the frame's validity.  The
    optimisation has to make progress and every gradient buffer has to be zero behind each optimizer step."""
    import math
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd.graph_view import FrameGraph
    from bilateral_driving_amd.optim import FusedAdam
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    W, H, N_GT, n_img = 320, 192, 20_000, 6
    cams = Hn.ring_cameras(W, H, device=dev)
    gt = Hn.synthetic_scene(N_GT, seed=1, device=dev)
    gt_grids = Hn.make_grids(n_img, seed=3, device=dev)
    sky = torch.rand(H, W, 3, device=dev)
    with torch.no_grad():
        targets = [Hn.render_view(gt, c, gt_grids, v, sky)["rgb"].clone() for v, c in enumerate(cams)]
    sel = torch.randperm(N_GT, device=dev)[: N_GT // 2]
    p = {"means": gt["means"][sel] + 0.05 * torch.randn(len(sel), 3, device=dev), "quats": gt["quats"][sel].clone(),
         "log_scales": gt["log_scales"][sel] + 0.2, "opacity_logits": torch.full((len(sel),), -1.0, device=dev),
         "sh": torch.zeros(len(sel), 16, 3, device=dev)}
    p = {k: v.contiguous().requires_grad_(True) for k, v in p.items()}
    grids = [g.requires_grad_(True) for g in Hn.make_grids(n_img, seed=0, device=dev)]
    lrs = dict(means=1.6e-3, quats=1e-3, log_scales=5e-3, opacity_logits=5e-2, sh=2.5e-3)
    groups = [{"params": [p[k]], "lr": lr, "eps": 1e-15} for k, lr in lrs.items()] + [{"params": [g], "lr": 2e-3, "eps": 1e-15} for g in grids]
    opt = FusedAdam(groups, lr=0.0, eps=1e-15, consume_grads=True)
    frame = FrameGraph(p, [cams[0]], grids, [sky], [targets[0]], img_indices=[0], dynamic=True, calib_cams=cams, clear_grads=False)

    def psnr():
        with torch.no_grad():
            mse = sum(float(((Hn.render_view(p, c, grids, v, sky)["rgb"] - targets[v]) ** 2).mean()) for v, c in enumerate(cams))
        return -10 * math.log10(mse / len(cams))

    p0, n_cap, n_valid = psnr(), frame.n_captures, 0
    for step in range(150):
        v = int(torch.randint(0, len(cams), (1,)))
        frame.set_view(0, cams[v], targets[v], sky, v)
        if frame.step():
            n_valid += 1
            assert all(t.grad.data_ptr() == frame.arena[k].data_ptr() for k, t in p.items())
            opt.step()
            if step % 50 == 0:
                assert float(frame.flat.flat.abs().max()) == 0.0      # consumed AND cleared
    p1 = psnr()
    print(f"[replayed loop] PSNR {p0:.2f} -> {p1:.2f} dB over {n_valid} valid steps, {frame.n_captures - n_cap} re-captures")
    assert n_valid >= 140 and p1 > p0 + 3.0, (p0, p1, n_valid)
    assert all(torch.isfinite(t).all() for t in p.values())


def test_replayed_loop_with_the_deferred_row_optimizer_follows_the_dense_one():
    """The reference's cadence -- ONE view, then one optimizer step over every parameter (tools/train.py:252-283 ->
    models/trainers/base.py:502-516) -- with optim.DeferredRowAdam: the SH rows step through the view's visible-id list, the missed
    zero-gradient steps are replayed inside the next forward that sees the row (FrameGraph(row_catchup=...)).  Bit-equality of the
    update itself is tests/test_gpu_10's; here the whole loop: the same 60 steps with the dense FusedAdam and with the deferred one end
    in the same parameters up to what two dense runs differ by (the float atomics of the composite backward), rows NO view has named
    are behind until flush(), and every gradient is cleared behind each step."""
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd.graph_view import FrameGraph
    from bilateral_driving_amd.optim import DeferredRowAdam, FusedAdam
    dev = torch.device("cuda", 0)
    W, H, N, n_img, steps = 320, 192, 12_000, 6, 60
    cams = Hn.ring_cameras(W, H, device=dev)
    base = Hn.synthetic_scene(N, seed=2, device=dev)
    grids0 = Hn.make_grids(n_img, seed=0, device=dev)
    gen = torch.Generator().manual_seed(4)
    sky = torch.rand(H, W, 3, generator=gen).to(dev)
    targets = [torch.rand(H, W, 3, generator=gen).to(dev) for _ in cams]
    picks = torch.randint(0, len(cams), (steps,), generator=gen).tolist()
    lrs = dict(means=1.6e-4, quats=1e-3, log_scales=5e-3, opacity_logits=5e-2)

    def run(deferred):
        p = {k: v.clone().contiguous().requires_grad_(True) for k, v in base.items()}
        grids = [g.clone().requires_grad_(True) for g in grids0]
        groups = [{"params": [p[k]], "lr": lr, "eps": 1e-15} for k, lr in lrs.items()] + [{"params": [g], "lr": 2e-3, "eps": 1e-15} for g in grids]
        if deferred:
            groups.append({"params": [p["sh"]], "lr": 2.5e-3, "lr_b": 1.25e-4, "col_split": 3, "deferred_rows": True, "eps": 1e-15})
            opt = DeferredRowAdam(groups, lr=0.0, eps=1e-15, consume_grads=True)
        else:    # the same rates through the dense pass: dc and rest columns as two passes is not expressible -> one rate for both runs
            groups.append({"params": [p["sh"]], "lr": 2.5e-3, "eps": 1e-15})
            opt = FusedAdam(groups, lr=0.0, eps=1e-15, consume_grads=True)
        if deferred:
            opt.param_groups[-1]["lr_b"] = 2.5e-3
        frame = FrameGraph(p, [cams[0]], grids, [sky], [targets[0]], img_indices=[0], dynamic=True, calib_cams=cams, clear_grads=False,
                           row_catchup=opt.catchup if deferred else None)
        behind = None
        for it, v in enumerate(picks):
            frame.set_view(0, cams[v], targets[v], sky, v)
            assert frame.step() is True
            if deferred:
                opt.step(lists=frame.row_lists())
            else:
                opt.step()
            if it % 20 == 0:
                assert float(frame.flat.flat.abs().max()) == 0.0          # consumed AND cleared (the SH rows through the list)
        if deferred:
            behind = int((opt.state[p["sh"]]["last_step"] < steps).sum())
            opt.flush()
            assert int((opt.state[p["sh"]]["last_step"] != steps).sum()) == 0
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in p.items()}, behind, frame.n_captures

    a, _, _ = run(False)
    a2, _, _ = run(False)
    b, behind, caps = run(True)
    assert behind > 0 and caps == 1
    for k in a:
        noise = float((a[k] - a2[k]).norm() / a[k].norm())
        diff = float((a[k] - b[k]).norm() / a[k].norm())
        print(f"[deferred loop] {k}: dense vs dense {noise:.2e}, dense vs deferred {diff:.2e}")
        assert diff <= max(10.0 * noise, 1e-5), (k, diff, noise)


def test_deferred_row_adam_inside_the_replayed_loop_is_the_dense_adam_on_the_same_gradients():
    """The integration, bit for bit: a replayed one-view loop with ``optim.DeferredRowAdam`` (catch-up inside the captured forward,
    step over ``frame.row_lists()``); every step's SH gradient buffer is recorded and fed to a dense ``FusedAdam`` next to it.
    (1) the gradient rows outside the view's list are exact zeros -- what makes the deferral legal; (2) when a view's forward has
    run, every row it lists is current (its step word = the optimizer's step) and holds the dense run's value -- the pack read what
    the reference's dense optimizer would have left there; (3) after a flush the parameter and both moments equal the dense run's."""
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd.graph_view import FrameGraph
    from bilateral_driving_amd.optim import DeferredRowAdam, FusedAdam
    dev = torch.device("cuda", 0)
    W, H, N, steps = 320, 192, 9_000, 40
    cams = Hn.ring_cameras(W, H, device=dev)
    base = Hn.synthetic_scene(N, seed=6, device=dev)
    gen = torch.Generator().manual_seed(8)
    sky = torch.rand(H, W, 3, generator=gen).to(dev)
    targets = [torch.rand(H, W, 3, generator=gen).to(dev) for _ in cams]
    picks = torch.randint(0, len(cams), (steps,), generator=gen).tolist()
    p = {k: v.clone().contiguous().requires_grad_(True) for k, v in base.items()}
    grids = [g.requires_grad_(True) for g in Hn.make_grids(len(cams), seed=0, device=dev)]
    lrs = dict(means=1.6e-4, quats=1e-3, log_scales=5e-3, opacity_logits=5e-2)
    groups = [{"params": [p[k]], "lr": lr, "eps": 1e-15} for k, lr in lrs.items()] + [{"params": [g], "lr": 2e-3, "eps": 1e-15} for g in grids]
    groups.append({"params": [p["sh"]], "lr": 2.5e-3, "lr_b": 1.25e-4, "col_split": 3, "deferred_rows": True, "eps": 1e-15})
    opt = DeferredRowAdam(groups, lr=0.0, eps=1e-15, consume_grads=True)
    # the dense twin of the SH parameter: dc and rest columns as two tensors with their own rates
    dc = base["sh"][:, :1].clone().contiguous().requires_grad_(True)
    rest = base["sh"][:, 1:].clone().contiguous().requires_grad_(True)
    twin = FusedAdam([{"params": [dc], "lr": 2.5e-3, "eps": 1e-15}, {"params": [rest], "lr": 1.25e-4, "eps": 1e-15}], lr=0.0, eps=1e-15)
    frame = FrameGraph(p, [cams[0]], grids, [sky], [targets[0]], img_indices=[0], dynamic=True, calib_cams=cams, clear_grads=False,
                       row_catchup=opt.catchup)
    last = opt.state[p["sh"]]["last_step"]
    seen = torch.zeros(N, dtype=torch.bool, device=dev)
    for t, v in enumerate(picks):
        frame.set_view(0, cams[v], targets[v], sky, v)
        assert frame.step() is True
        (cap, n_dev, ids_ptr), = frame.row_lists()
        n_vis = frame.counts()[0][1]
        ws = frame.prep_ws[0]
        ids = ws[frame._ids_off:frame._ids_off + 4 * cap].view(torch.int32)[:n_vis].long()
        assert 0 < n_vis < N and ids_ptr == ws.data_ptr() + frame._ids_off
        seen[ids] = True
        # (2) the rows the pack has just read: current, and bit-equal to the dense run's
        assert bool((last[ids] == t).all())
        sh_now = p["sh"].detach()
        assert torch.equal(sh_now[ids, :1], dc.detach()[ids]) and torch.equal(sh_now[ids, 1:], rest.detach()[ids])
        # (1) the gradient outside the list is exactly zero
        g = p["sh"].grad.clone()
        outside = torch.ones(N, dtype=torch.bool, device=dev)
        outside[ids] = False
        assert float(g[outside].abs().max()) == 0.0 and float(g[ids].abs().max()) > 0.0
        dc.grad, rest.grad = g[:, :1].contiguous(), g[:, 1:].contiguous()
        twin.step()
        opt.step(lists=frame.row_lists())
        assert float(p["sh"].grad.abs().max()) == 0.0                      # consumed and cleared through the list
    behind = int((last < steps).sum())
    assert behind > 0 and int((~seen).sum()) > 0                            # rows no view listed: never touched so far ...
    assert torch.equal(p["sh"].detach()[~seen], base["sh"][~seen])
    opt.flush()                                                             # ... until the flush replays their zero-gradient steps
    sh_end = p["sh"].detach()
    assert torch.equal(sh_end[:, :1], dc.detach()) and torch.equal(sh_end[:, 1:], rest.detach())
    st = opt.state[p["sh"]]
    assert torch.equal(st["exp_avg"][:, :1], twin.state[dc]["exp_avg"]) and torch.equal(st["exp_avg"][:, 1:], twin.state[rest]["exp_avg"])
    assert torch.equal(st["exp_avg_sq"][:, :1], twin.state[dc]["exp_avg_sq"]) and torch.equal(st["exp_avg_sq"][:, 1:], twin.state[rest]["exp_avg_sq"])
    assert frame.n_captures == 1
