"""-m gpu: the device-count form of the view (no host read-back: list capacities from the host, counts on the device) and its
hipGraph replay (graph_view.FrameGraph) against the host-count frame loop that tests/test_gpu_03_harness.py ties to the oracle.
Same kernels, same order: lists bit-equal, images bit-equal, gradients equal up to the atomics' order."""
import pytest
import torch

from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def same_sh_kernel(monkeypatch):
    """The device-count form evaluates the SH colours in the record pack; the host-count form defaults to the dense pass (hidden behind
    its host wait).  Bit-equality between the two forms is asserted with both on the pack's arithmetic."""
    from bilateral_driving_amd import fused_view as FV
    monkeypatch.setattr(FV, "SH_IN_PACK", FV.SH_IN_PACK_DEV)


@pytest.fixture(scope="module")
def mods():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from bilateral_driving_amd import _lib
    _lib.lib()  # fails loudly if libbds.so is missing
    from bilateral_driving_amd import fused_view as FV
    from bilateral_driving_amd import graph_view as GV
    from bilateral_driving_amd import harness as Hn
    return FV, GV, Hn


def _scene(Hn, N, W, H, yaws, seed, dev="cuda"):
    cams = Hn.ring_cameras(W, H, yaws_deg=yaws, device=dev)
    for c in cams:
        c.viewmat.requires_grad_(True)
    p = Hn.synthetic_scene(N, seed=seed, device=dev)
    p["means"] = p["means"] * torch.tensor([0.4, 0.4, 1.0], device=dev)
    p = {k: v.requires_grad_(True) for k, v in p.items()}
    grids = [g.requires_grad_(True) for g in Hn.make_grids(len(cams), device=dev)]
    gen = torch.Generator().manual_seed(11 + seed)
    skies = [torch.rand(H, W, 3, generator=gen).to(dev).requires_grad_(True) for _ in cams]
    targets = [torch.rand(H, W, 3, generator=gen).to(dev) for _ in cams]
    return cams, p, grids, skies, targets


def _eager_frame(Hn, cams, p, grids, skies, targets):
    """The host-count frame loop (what bench.py ran before the graphs): dense per-parameter gradients summed by autograd."""
    for t in list(p.values()) + grids + skies + [c.viewmat for c in cams]:
        t.grad = None
    outs = []
    for v, cam in enumerate(cams):
        out = Hn.render_view(p, cam, grids, v, skies[v])
        loss = Hn.training_loss(out, targets[v], grids)
        loss.backward()
        outs.append((out["rgb"].detach().clone(), out["depth"].detach().clone(), float(loss), out["info"]["n_isects"], out["info"]["n_visible"],
                     out["info"]["means2d"].absgrad.clone()))
    grads = {k: t.grad.clone() for k, t in p.items()}
    grads.update({f"grid{i}": g.grad.clone() for i, g in enumerate(grids)})
    return outs, grads, [s.grad.clone() for s in skies], [c.viewmat.grad.clone() for c in cams]


def test_device_count_view_equals_host_count_view(mods):
    """fused_view(caps=...) == fused_view(): images bit-equal, lists equal over the counted range, gradients to atomics noise."""
    FV, GV, Hn = mods
    W, H, N = 320, 192, 5000
    cams, p, grids, skies, targets = _scene(Hn, N, W, H, (0.0,), 1)
    cam = cams[0]
    ref = Hn.render_view(p, cam, grids, 0, skies[0])
    M, nv = ref["info"]["n_isects"], ref["info"]["n_visible"]
    assert M > 0 and nv > 0
    Hn.training_loss(ref, targets[0], grids).backward()
    g_ref = {k: t.grad.clone() for k, t in p.items()}
    vm_ref, sky_ref = cam.viewmat.grad.clone(), skies[0].grad.clone()
    absg_ref = ref["info"]["means2d"].absgrad.clone()
    for t in list(p.values()) + grids + skies + [cam.viewmat]:
        t.grad = None
    caps = FV.ListCapacity(int(M * 1.5) + 100, int(nv * 1.5) + 100)
    out = Hn.render_view(p, cam, grids, 0, skies[0], caps=caps)
    Hn.training_loss(out, targets[0], grids).backward()
    torch.cuda.synchronize()
    assert caps.observed() == (M, nv) and not caps.overflowed()
    assert torch.equal(out["rgb"], ref["rgb"]) and torch.equal(out["depth"], ref["depth"]) and torch.equal(out["opacity"], ref["opacity"])
    assert torch.equal(out["info"]["flatten_ranks"][:M], ref["info"]["flatten_ranks"])
    assert torch.equal(out["info"]["visible_ids"][:nv], ref["info"]["visible_ids"])
    assert torch.equal(out["info"]["isect_offsets"], ref["info"]["isect_offsets"])
    for k, t in p.items():
        assert rel_err(t.grad, g_ref[k]) < 2e-5, k
    assert rel_err(cam.viewmat.grad, vm_ref) < 1e-4 and rel_err(skies[0].grad, sky_ref) < 1e-6
    assert rel_err(out["info"]["means2d"].absgrad, absg_ref) < 2e-5


def test_device_count_overflow_renders_nothing_and_is_flagged(mods):
    """A list that outgrows its capacity: effective counts zero (empty image, zero gradients, nothing out of bounds), overflow word set."""
    FV, GV, Hn = mods
    W, H, N = 256, 160, 4000
    cams, p, grids, skies, targets = _scene(Hn, N, W, H, (0.0,), 2)
    cam = cams[0]
    ref = Hn.render_view(p, cam, grids, 0, skies[0])
    M, nv = ref["info"]["n_isects"], ref["info"]["n_visible"]
    for m_cap, nv_cap in ((max(M // 2, 1), nv + 10), (M + 10, max(nv // 2, 1))):
        caps = FV.ListCapacity(m_cap, nv_cap)
        for t in list(p.values()) + grids:
            t.grad = None
        out = Hn.render_view(p, cam, grids, 0, skies[0], caps=caps)
        Hn.training_loss(out, targets[0], grids).backward()
        torch.cuda.synchronize()
        # (a visible-count overflow leaves M uncounted: the launches behind the compaction are sized by the capacity)
        assert caps.overflowed() and caps.observed()[1] == nv and caps.observed()[0] == (M if nv_cap >= nv else 0)
        assert caps.wanted()[0] > caps.observed()[0] - 1 and caps.wanted()[1] == nv and (nv_cap >= nv or caps.wanted()[0] > m_cap)
        assert float(out["opacity"].abs().max()) == 0.0
        for k, t in p.items():
            assert t.grad is None or float(t.grad.abs().max()) == 0.0, k


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("n_views", [1, 3])
def test_frame_graph_equals_eager_frame(mods, n_views, overlap):
    """FrameGraph.step() (begin graph + one hipGraph per view) == the eager host-count frame, over several replays, and the flat
    gradient buffer holds exactly one frame's gradients each time (row-wise clear by the begin graph)."""
    FV, GV, Hn = mods
    W, H, N = 320, 192, 6000
    yaws = (0.0, 120.0, 240.0)[:n_views]
    cams, p, grids, skies, targets = _scene(Hn, N, W, H, yaws, 3)
    outs, g_ref, sky_ref, vm_ref = _eager_frame(Hn, cams, p, grids, skies, targets)
    frame = GV.FrameGraph(p, cams, grids, skies, targets, overlap=bool(overlap))
    for rep in range(3):
        assert frame.step() is True      # (waits for the frame, returns its validity)
        for v, vg in enumerate(frame.views):
            assert torch.equal(vg.rgb, outs[v][0]) and torch.equal(vg.depth, outs[v][1]), (rep, v)
            assert abs(float(vg.loss) - outs[v][2]) < 1e-6 * max(1.0, abs(outs[v][2]))
            assert frame.counts()[v] == (outs[v][3], outs[v][4])
            assert rel_err(vg.v_sky, sky_ref[v]) < 1e-6 and rel_err(vg.v_viewmat, vm_ref[v]) < 1e-4
            # the view's persistent screen-space gradient arrays (info["means2d"].absgrad of the eager view): rows of this visit only
            assert rel_err(frame.g2d[v][1], outs[v][5][0]) < 2e-4, (rep, v)
            assert torch.equal(frame.g2d[v][1] != 0, outs[v][5][0] != 0)
        for k, t in p.items():
            assert t.grad.data_ptr() == frame.arena[k].data_ptr()
            assert rel_err(t.grad, g_ref[k]) < 2e-4, (rep, k)   # (float atomics in the compositor backward: the order of the sums varies)
        for i, g in enumerate(grids):
            assert rel_err(g.grad, g_ref[f"grid{i}"]) < 3e-5, (rep, i)


def test_frame_graph_follows_parameter_updates_and_grows_on_overflow(mods):
    """The graphs read the parameters in place: after an in-place update the replay equals a fresh eager frame; when the update makes
    a list outgrow its capacity, valid() returns False, re-captures, and the repeated frame is right."""
    FV, GV, Hn = mods
    W, H, N = 256, 160, 5000
    cams, p, grids, skies, targets = _scene(Hn, N, W, H, (0.0, 180.0), 4)
    frame = GV.FrameGraph(p, cams, grids, skies, targets)
    assert frame.step()
    # tight capacities (what a scene that grew since the calibration looks like), captured again
    for v, (M, nv) in enumerate(frame.counts()):
        frame.caps[v] = FV.ListCapacity(M + 16, nv + 16)
    frame.capture()
    assert frame.step(wait=False) is None
    for vg in frame.views:
        vg.done.synchronize()
    assert frame._check_counts()
    with torch.no_grad():
        p["log_scales"].add_(0.7)      # every splat 2x larger: more (tile, Gaussian) pairs than the lists hold
        p["opacity_logits"].add_(0.5)
    n_cap = frame.n_captures
    assert frame.step() is False and frame.n_captures == n_cap + 1      # overflow: captured again with larger lists, frame to be dropped
    assert frame.step() is True
    got = {k: t.grad.clone() for k, t in p.items()}
    rgb = [vg.rgb.clone() for vg in frame.views]
    absg = [g[1].clone() for g in frame.g2d]
    outs, g_ref, _, _ = _eager_frame(Hn, cams, p, grids, skies, targets)
    for v in range(len(cams)):
        assert torch.equal(rgb[v], outs[v][0])
        assert rel_err(absg[v], outs[v][5][0]) < 3e-5      # (stale rows of the visits before the update cleared by their lists)
    for k in got:
        assert rel_err(got[k], g_ref[k]) < 3e-5, k


def test_loss_on_the_transform_launch_equals_the_separate_loss_launch(mods, monkeypatch):
    """train_view with the L1 + TV loss folded into the colour transform's full-resolution kernel (bds_bilagrid_ms_ed_train_fwd) ==
    train_view with the loss as its own launch (bds_l1_tv_train): image bit-equal, loss to fp32 summation order, v_rgb-driven
    gradients to atomics noise, TV gradient in the grids' slices."""
    FV, GV, Hn = mods
    W, H, N = 322, 190, 5000       # (W * H not a multiple of the workgroup: the pixel tail)
    cams, p, grids, skies, targets = _scene(Hn, N, W, H, (0.0,), 6)
    res = {}
    for fused in (False, True):
        monkeypatch.setattr(FV, "_LOSS_IN_TRANSFORM", fused)
        for t in list(p.values()) + grids + skies + [cams[0].viewmat]:
            t.grad = None
        gg = [torch.zeros_like(g) for g in grids]
        out = Hn.train_view(p, cams[0], grids, 0, skies[0], targets[0], grid_grads=gg)
        torch.cuda.synchronize()
        res[fused] = (out["rgb"].clone(), float(out["loss"]), {k: t.grad.clone() for k, t in p.items()}, [g.clone() for g in gg],
                      skies[0].grad.clone())
    a, b = res[False], res[True]
    assert torch.equal(a[0], b[0])
    assert abs(a[1] - b[1]) <= 2e-6 * abs(a[1])
    for k in a[2]:
        assert rel_err(b[2][k], a[2][k]) < 2e-4, k
    for x, y in zip(a[3], b[3]):
        assert float(x.abs().max()) > 0 and rel_err(y, x) < 3e-5
    assert rel_err(b[4], a[4]) < 1e-6


def test_sh_in_the_record_pack_equals_the_dense_sh_pass(mods, monkeypatch):
    """The record pack that evaluates the SH colours itself (LDS-staged coefficient rows, visible Gaussians only: the device-count form's
    default) against the dense SH pass + plain pack: images and gradients to fp32 rounding of the colour sums."""
    FV, GV, Hn = mods
    W, H, N = 320, 192, 7000
    cams, p, grids, skies, targets = _scene(Hn, N, W, H, (0.0,), 8)
    res = {}
    for in_pack in (False, True):
        monkeypatch.setattr(FV, "SH_IN_PACK", in_pack)
        for t in list(p.values()) + grids + skies + [cams[0].viewmat]:
            t.grad = None
        out = Hn.render_view(p, cams[0], grids, 0, skies[0])
        Hn.training_loss(out, targets[0], grids).backward()
        res[in_pack] = (out["rgb"].detach().clone(), {k: t.grad.clone() for k, t in p.items()})
    assert float((res[True][0] - res[False][0]).abs().max()) < 2e-6
    for k in res[False][1]:
        assert rel_err(res[True][1][k], res[False][1][k]) < 2e-4, k     # (float atomics in the compositor backward: order of the sums)


def test_reprovisioning_is_deferred_until_the_valid_frame_was_consumed(mods):
    """A count within 8 % of its capacity: the frame stays valid, its static outputs (sky / pose gradients, images) stay those of the
    replay -- the larger lists are captured at the START of the next step(), not inside valid()."""
    FV, GV, Hn = mods
    W, H, N = 256, 160, 5000
    cams, p, grids, skies, targets = _scene(Hn, N, W, H, (0.0, 180.0), 5)
    frame = GV.FrameGraph(p, cams, grids, skies, targets)
    assert frame.step()
    sky_ref, vm_ref = [vg.v_sky.clone() for vg in frame.views], [vg.v_viewmat.clone() for vg in frame.views]
    rgb_ref = [vg.rgb.clone() for vg in frame.views]
    for v, (M, nv) in enumerate(frame.counts()):
        frame.caps[v] = FV.ListCapacity(int(M * 1.02) + 1, int(nv * 1.02) + 1)     # > 92 % full
    frame.capture()
    n_cap = frame.n_captures
    assert frame.step() is True and frame._reprovision and frame.n_captures == n_cap      # nothing captured yet
    for v, vg in enumerate(frame.views):
        assert skies[v].grad is vg.v_sky and torch.equal(vg.v_sky, sky_ref[v]) and rel_err(vg.v_viewmat, vm_ref[v]) < 1e-5
        assert torch.equal(vg.rgb, rgb_ref[v])
    assert frame.step() is True and frame.n_captures == n_cap + 1 and not frame._reprovision
    for v, vg in enumerate(frame.views):
        assert torch.equal(vg.rgb, rgb_ref[v]) and torch.equal(vg.v_sky, sky_ref[v])


def test_one_captured_view_replays_over_random_cameras_and_images(mods):
    """FrameGraph(dynamic=True): ONE captured view, 20 replays with a random camera / target / sky / image index each (what the
    reference's loop feeds a step, tools/train.py:250-283) == the eager view of the same inputs, every time: image bit-equal, loss,
    per-Gaussian / grid / sky / pose gradients.  The image's grids are picked on the device (bds_bilagrid_select)."""
    FV, GV, Hn = mods
    W, H, N, n_img = 256, 160, 5000, 7
    dev = "cuda"
    gen = torch.Generator().manual_seed(21)
    yaws = [float(y) for y in (torch.rand(12, generator=gen) * 360.0)]
    pool = Hn.ring_cameras(W, H, yaws_deg=yaws, device=dev)
    for i, c in enumerate(pool):      # each camera its own centre and intrinsics: everything a slot holds changes between replays
        vm = c.viewmat.clone()
        vm[:3, 3] += (torch.rand(3, generator=gen) - 0.5).to(dev)
        c.viewmat = vm
        c.K = c.K.clone()
        c.K[0, 0] *= 1.0 + 0.02 * i
        c.K[1, 1] *= 1.0 + 0.02 * i
        c.cam_pos = torch.linalg.inv(vm)[:3, 3].contiguous()
    p = Hn.synthetic_scene(N, seed=9, device=dev)
    p["means"] = p["means"] * torch.tensor([0.4, 0.4, 1.0], device=dev)
    p = {k: v.requires_grad_(True) for k, v in p.items()}
    grids = [g.requires_grad_(True) for g in Hn.make_grids(n_img, device=dev)]
    sky0 = torch.rand(H, W, 3, generator=gen).to(dev).requires_grad_(True)
    slot_cam = Hn.Camera(pool[0].viewmat.clone().requires_grad_(True), pool[0].K, W, H, pool[0].cam_pos)
    frame = GV.FrameGraph(p, [slot_cam], grids, [sky0], [torch.rand(H, W, 3, generator=gen).to(dev)], img_indices=[0], dynamic=True,
                          calib_cams=pool)
    n_cap = frame.n_captures
    for rep in range(20):
        k = int(torch.randint(0, len(pool), (1,), generator=gen))
        img = int(torch.randint(0, n_img, (1,), generator=gen))
        target = torch.rand(H, W, 3, generator=gen).to(dev)
        sky = torch.rand(H, W, 3, generator=gen).to(dev)
        frame.set_view(0, pool[k], target, sky, img if rep % 2 else torch.tensor([img], device=dev))
        assert frame.step() is True
        vg = frame.views[0]
        got = {n: t.grad.clone() for n, t in p.items()}
        got_g = [g.grad.clone() for g in grids]
        got_rgb, got_loss, got_sky, got_vm = vg.rgb.clone(), float(vg.loss), vg.v_sky.clone(), vg.v_viewmat.clone()
        # the eager view of the same inputs (host-count path, autograd)
        cam = Hn.Camera(pool[k].viewmat.clone().requires_grad_(True), pool[k].K, W, H, pool[k].cam_pos)
        sky_e = sky.clone().requires_grad_(True)
        p_e = {n: t.detach().clone().requires_grad_(True) for n, t in p.items()}
        g_e = [g.detach().clone().requires_grad_(True) for g in grids]
        out = Hn.render_view(p_e, cam, g_e, img, sky_e)
        loss = Hn.training_loss(out, target, g_e)
        loss.backward()
        assert torch.equal(got_rgb, out["rgb"].detach()), rep
        assert abs(got_loss - float(loss)) < 1e-6 * max(1.0, abs(float(loss)))
        for n in p:
            assert rel_err(got[n], p_e[n].grad) < 2e-4, (rep, n)
        for i, g in enumerate(g_e):
            assert rel_err(got_g[i], g.grad) < 3e-5, (rep, i)
            others = [j for j in range(n_img) if j != img]
            # (only the TV term reaches the other images' grids)
            assert rel_err(got_g[i][others], g.grad[others]) < 3e-5
        assert rel_err(got_sky, sky_e.grad) < 1e-6 and rel_err(got_vm, cam.viewmat.grad) < 1e-4
    assert frame.n_captures == n_cap      # one capture served them all


def test_rebind_makes_the_same_frame_over_other_parameter_tensors(mods):
    """Densification re-allocates every per-Gaussian tensor with another length; ``FrameGraph.rebind(params)`` is the frame again over
    the new tensors (gradient buffer, per-view buffers, lists, graphs of the new size): images and gradients equal a frame built from
    scratch over them, for a fixed frame and for replayable slots."""
    FV, GV, Hn = mods
    FrameGraph = GV.FrameGraph
    for dynamic in (False, True):
        cams, p, grids, skies, targets = _scene(Hn, 9000, 256, 160, (0.0, 120.0), 21)
        frame = FrameGraph(p, cams, grids, skies, targets, dynamic=dynamic, calib_cams=cams if dynamic else None)
        assert frame.step() is True
        keep = torch.arange(0, 9000, 3, device="cuda")                        # "densification": another set of another length
        q = {k: torch.cat([v.detach()[keep], v.detach()[:700] * 1.01]).contiguous().requires_grad_(True) for k, v in p.items()}
        frame.rebind(q)
        assert frame.N == q["means"].shape[0] != 9000 and frame.flat.flat.numel() >= frame.N * 59
        assert frame.step() is True
        fresh = FrameGraph({k: v.detach().clone().requires_grad_(True) for k, v in q.items()}, cams, [g.detach().clone().requires_grad_(True) for g in grids],
                           [s.detach().clone().requires_grad_(True) for s in skies], targets, dynamic=dynamic, calib_cams=cams if dynamic else None)
        assert fresh.step() is True
        for a, b in zip(frame.views, fresh.views):
            assert torch.equal(a.rgb, b.rgb)
        for k in q:
            ga, gb = q[k].grad, fresh.params[k].grad
            assert float((ga - gb).norm() / gb.norm()) < 1e-5, k
