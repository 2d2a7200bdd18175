"""-m gpu: the reference's call sequence with ``marshalling.install`` (deferred activations -> the raw one-view node,
``rendering._RasterizeRawView``) against the same sequence with the eager mirror of ``get_gaussians`` (vanilla.py:378-414: dense
activations + dense SH, ``rendering._RasterizeView``), which tests/test_gpu_03 / 06 tie to the oracle.  Same image (the projection
normalises the raw quaternion itself, the colours come from the same SH routine: equal to rounding), gradients of the six raw
parameters to 1e-4 of their largest entry, the ``info["means2d"]`` contract of trainers/base.py:279-297,429-430, the NaN check."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from bilateral_driving_amd import _lib
    _lib.lib()
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd import marshalling as M
    return Hn, M


def _setup(Hn, N=40000, W=640, H=384, seed=5, dev="cuda"):
    cam = Hn.ring_cameras(W, H, yaws_deg=(20.0,), device=dev)[0]
    cam.viewmat.requires_grad_(True)
    p = Hn.synthetic_scene(N, seed=seed, device=dev)
    grids = [g.requires_grad_(True) for g in Hn.make_grids(1, device=dev)]
    gen = torch.Generator().manual_seed(seed)
    sky = torch.rand(H, W, 3, generator=gen).to(dev)
    target = torch.rand(H, W, 3, generator=gen).to(dev)
    return cam, p, grids, sky, target


def _run(Hn, model, cam, grids, sky, target, retain=False):
    for t in model.parameters() + grids + [cam.viewmat]:
        t.grad = None
    out = Hn.render_view_model(model, cam, grids, 0, sky)
    if retain:
        out["info"]["means2d"].retain_grad()
    loss = Hn.training_loss(out, target, grids) + 0.1 * out["depth"].mean() + 0.1 * out["opacity"].mean()
    loss.backward()
    grads = [t.grad.detach().clone() for t in model.parameters()] + [cam.viewmat.grad.detach().clone()] + [g.grad.detach().clone() for g in grids]
    return out, float(loss), grads


@pytest.mark.parametrize("step", [10 ** 6, 2500])       # all three SH bands active; two of them (vanilla.py:387)
def test_installed_sequence_equals_the_eager_mirror(mods, step):
    Hn, M = mods
    cam, p, grids, sky, target = _setup(Hn)
    model = Hn.VanillaModel(p, step=step)
    out_e, loss_e, g_e = _run(Hn, model, cam, grids, sky, target, retain=True)
    absgrad_e = out_e["info"]["means2d"].absgrad.clone()
    grad_e = out_e["info"]["means2d"].grad.clone()
    M.install(Hn.VanillaModel)
    try:
        out_l, loss_l, g_l = _run(Hn, model, cam, grids, sky, target, retain=True)
        assert out_l["info"]["means2d"].absgrad is not None
    finally:
        M.uninstall(Hn.VanillaModel)
    assert (out_l["info"]["radii"] > 0).sum() > 1000
    assert torch.equal(out_l["info"]["radii"], out_e["info"]["radii"])
    for k in ("rgb", "depth", "opacity"):
        err = float((out_l[k] - out_e[k]).abs().max())
        assert err <= 2e-5 * max(1.0, float(out_e[k].abs().max())), (k, err)
    assert abs(loss_l - loss_e) <= 1e-5 * max(1.0, abs(loss_e))
    names = ["means", "quats", "scales", "opacities", "features_dc", "features_rest", "viewmat", "grid0", "grid1", "grid2"]
    for n, a, b in zip(names, g_l, g_e):
        assert a.shape == b.shape, n
        ref = float(b.abs().max())
        assert float((a - b).abs().max()) <= 1e-4 * ref + 1e-9, (n, float((a - b).abs().max()), ref)
    for a, b, n in ((out_l["info"]["means2d"].absgrad, absgrad_e, "absgrad"), (out_l["info"]["means2d"].grad, grad_e, "grad")):
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-12, n


def test_the_raw_node_is_what_runs_and_partial_use_falls_back(mods):
    Hn, M = mods
    from bilateral_driving_amd import rendering as R
    cam, p, grids, sky, target = _setup(Hn, N=8000, W=256, H=192)
    model = Hn.VanillaModel(p)
    seen = []
    raw_apply, one_apply = R._RasterizeRawView.apply, R._RasterizeView.apply
    M.install(Hn.VanillaModel)
    try:
        R._RasterizeRawView.apply = staticmethod(lambda *a: (seen.append("raw"), raw_apply(*a))[1])
        R._RasterizeView.apply = staticmethod(lambda *a: (seen.append("one"), one_apply(*a))[1])
        with torch.no_grad():
            out = Hn.render_view_model(model, cam, grids, 0, sky)
            assert seen == ["raw"]
            # an opacity mask (the evaluation renders, scene_graph.py:296-313): the placeholders materialise, the activated-input node runs
            from bilateral_driving_amd.marshalling import dataclass_camera
            c2w = torch.eye(4, device="cuda")
            gs = model.get_gaussians(dataclass_camera(camtoworlds=c2w, camtoworlds_gt=c2w, Ks=cam.K, H=cam.height, W=cam.width))
            mask = torch.ones(8000, device="cuda")
            r2, a2, _ = R.rasterization(gs["_means"], gs["_quats"], gs["_scales"], gs["_opacities"].squeeze() * mask, gs["_rgbs"], cam.viewmat[None],
                                        cam.K[None], cam.width, cam.height, render_mode="RGB+ED", absgrad=True, packed=False)
            assert seen == ["raw", "one"]
    finally:
        R._RasterizeRawView.apply, R._RasterizeView.apply = raw_apply, one_apply
        M.uninstall(Hn.VanillaModel)
    assert float((r2[0, ..., :3] - out["rgb_gaussians"]).abs().max()) <= 2e-5


def test_nan_in_a_parameter_raises_as_the_reference_does(mods):
    Hn, M = mods
    cam, p, grids, sky, target = _setup(Hn, N=8000, W=256, H=192)
    model = Hn.VanillaModel(p)
    M.install(Hn.VanillaModel)
    try:
        with torch.no_grad():
            Hn.render_view_model(model, cam, grids, 0, sky)                       # clean: no raise
            model._features_rest[4321, 7, 1] = float("nan")
            with pytest.raises(ValueError, match="features_rest"):
                Hn.render_view_model(model, cam, grids, 0, sky)
            model._features_rest[4321, 7, 1] = 0.0
            model._scales[17, 2] = float("inf")
            with pytest.raises(ValueError, match="scales"):
                Hn.render_view_model(model, cam, grids, 0, sky)
            model._scales[17, 2] = 0.0
            Hn.render_view_model(model, cam, grids, 0, sky)
    finally:
        M.uninstall(Hn.VanillaModel)


def test_nonfinite_flags_sees_every_element(mods):
    """bds_nonfinite_flags over tensors of awkward sizes / alignments: a NaN or an Inf in the first, the last or any middle element
    sets the tensor's bit and only that bit."""
    import ctypes
    from bilateral_driving_amd import _lib as L
    lib = L.lib()
    flag = torch.zeros(1, device="cuda", dtype=torch.int32)
    a, b, c = (torch.rand(n, device="cuda") - 0.5 for n in (3 * 70001 + 7, 16, 4096 * 5 + 16))
    ts = [a[1:70001 * 3 - 1], b[3:5], c[7:7 + 4096 * 5], a[:0]]       # 4-byte aligned starts, ragged tails, one empty tensor

    def run():
        ptrs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        cnts = (ctypes.c_int64 * len(ts))(*[t.numel() for t in ts])
        L.check(lib.bds_nonfinite_flags(len(ts), ptrs, cnts, flag.data_ptr(), None, L.stream()), "bds_nonfinite_flags")
        return int(flag.item())

    assert run() == 0
    for t_i, idx, val in ((0, 0, float("nan")), (0, ts[0].numel() - 1, float("inf")), (0, 12345, -float("inf")), (1, 1, float("nan")),
                          (2, 4096 * 5 - 1, float("nan")), (2, 1023 * 4 + 2, float("inf"))):
        keep = float(ts[t_i][idx])
        ts[t_i][idx] = val
        assert run() == 1 << t_i, (t_i, idx, val)
        ts[t_i][idx] = keep
    assert run() == 0
    big = torch.finfo(torch.float32).max
    ts[0][5] = big                                                               # the largest finite value is finite
    assert run() == 0


def test_nonfinite_kinds_follow_the_activations(mods):
    """bds_nonfinite_flags_kinds: the raw one-view node checks the raw parameters, so the bit has to say what the reference's check of
    the ACTIVATED tensors would (vanilla.py:393-395,407-412): exp of a log-scale >= 88.72284 is Inf (and of -Inf is 0: fine), a
    zero quaternion normalises to NaN, sigmoid maps +-Inf logits to 0 / 1 (fine) -- each against torch's own activation."""
    import ctypes
    from bilateral_driving_amd import _lib as L
    lib = L.lib()
    flag = torch.zeros(1, device="cuda", dtype=torch.int32)
    g = torch.Generator().manual_seed(0)
    ls = (torch.rand(30011, generator=g) * 8 - 6).cuda()
    q = torch.randn(5001, 4, generator=g).cuda()
    lo = (torch.randn(7003, generator=g) * 3).cuda()
    ts, kinds = [ls, q, lo], (1, 2, 3)
    acts = [torch.exp, lambda t: t / t.norm(dim=-1, keepdim=True), torch.sigmoid]

    def run():
        ptrs = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
        cnts = (ctypes.c_int64 * 3)(*[t.numel() for t in ts])
        L.check(lib.bds_nonfinite_flags_kinds(3, ptrs, cnts, (ctypes.c_int * 3)(*kinds), flag.data_ptr(), None, L.stream()), "kinds")
        got = int(flag.item())
        want = sum((1 << i) for i, (t, f) in enumerate(zip(ts, acts)) if not bool(torch.isfinite(f(t)).all()))
        assert got == want, (got, want)
        return got

    assert run() == 0
    cases = [(0, 17, 88.72284, 1), (0, 17, 88.7228, 0), (0, 30010, float("inf"), 1), (0, 3, -float("inf"), 0), (0, 9, float("nan"), 1),
             (0, 11, 200.0, 1), (2, 5, float("inf"), 0), (2, 5, -float("inf"), 0), (2, 7002, float("nan"), 4)]
    for t_i, idx, val, want in cases:
        keep = float(ts[t_i].view(-1)[idx])
        ts[t_i].view(-1)[idx] = val
        assert run() == want, (t_i, idx, val)
        ts[t_i].view(-1)[idx] = keep
    for row, vals, want in ((123, (0.0, 0.0, 0.0, 0.0), 2), (123, (0.0, -0.0, 0.0, 0.0), 2), (123, (0.0, 0.0, 1e-30, 0.0), 2), (123, (0.0, 0.0, 1e-10, 0.0), 0),
                            (5000, (float("inf"), 1.0, 0.0, 0.0), 2), (0, (float("nan"), 1.0, 0.0, 0.0), 2)):
        keep = q[row].clone()
        q[row] = torch.tensor(vals, device="cuda")
        assert run() == want, (row, vals)
        q[row] = keep
    assert run() == 0


def test_raw_node_raises_on_what_the_activations_make_nonfinite(mods):
    Hn, M = mods
    cam, p, grids, sky, target = _setup(Hn, N=8000, W=256, H=192)
    model = Hn.VanillaModel(p)
    M.install(Hn.VanillaModel)
    try:
        with torch.no_grad():
            model._scales[17, 2] = 90.0                          # finite, but exp(90) is not
            with pytest.raises(ValueError, match="scales"):
                Hn.render_view_model(model, cam, grids, 0, sky)
            model._scales[17, 2] = -float("inf")                 # exp(-Inf) = 0: the reference does not raise
            model._opacities.view(-1)[5] = float("inf")          # sigmoid(Inf) = 1: neither
            Hn.render_view_model(model, cam, grids, 0, sky)
            model._scales[17, 2] = 0.0
            model._opacities.view(-1)[5] = 0.0
            model._quats[99] = 0.0                               # 0 / 0
            with pytest.raises(ValueError, match="quats"):
                Hn.render_view_model(model, cam, grids, 0, sky)
    finally:
        M.uninstall(Hn.VanillaModel)
