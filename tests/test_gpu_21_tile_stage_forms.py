"""-m gpu: the launch forms of the device-count tile stage (include/bds.h option 0: launches behind the compaction sized by the
visible-entry capacity instead of N) against each other and against the host-count form that tests/test_gpu_01 / test_gpu_08 tie to
the oracle's stable (tile | depth) sort, at sizes whose tile key is one 9-bit / 10-bit digit.  Integer work: lists, offsets and
counts are compared bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def same_sh_kernel(monkeypatch):
    from bilateral_driving_amd import fused_view as FV
    monkeypatch.setattr(FV, "SH_IN_PACK", FV.SH_IN_PACK_DEV)


@pytest.fixture(scope="module")
def mods():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from bilateral_driving_amd import _lib
    _lib.lib()
    from bilateral_driving_amd import fused_view as FV
    from bilateral_driving_amd import graph_view as GV
    from bilateral_driving_amd import harness as Hn
    return _lib, FV, GV, Hn


@pytest.fixture()
def options(mods):
    """Restores the tile-stage options after a test."""
    L = mods[0]
    keep = {w: int(L.lib().bds_get_option(w)) for w in (L.OPT_CAP_LAUNCH,)}
    yield L
    for w, v in keep.items():
        L.set_option(w, v)


def _scene(Hn, N, W, H, seed, dev="cuda", spread=0.5):
    cam = Hn.ring_cameras(W, H, yaws_deg=(0.0,), device=dev)[0]
    cam.viewmat.requires_grad_(True)
    p = Hn.synthetic_scene(N, seed=seed, device=dev)
    p["means"] = p["means"] * torch.tensor([spread, spread, 1.0], device=dev)
    p = {k: v.requires_grad_(True) for k, v in p.items()}
    grids = [g.requires_grad_(True) for g in Hn.make_grids(1, device=dev)]
    gen = torch.Generator().manual_seed(21 + seed)
    sky = torch.rand(H, W, 3, generator=gen).to(dev).requires_grad_(True)
    target = torch.rand(H, W, 3, generator=gen).to(dev)
    return cam, p, grids, sky, target


def _lists(out, M, nv):
    i = out["info"]
    return i["flatten_ranks"][:M].clone(), i["visible_ids"][:nv].clone(), i["isect_offsets"].clone(), i["tiles_per_gauss"].clone()


# (W, H): 64-px list tiles -> a 9-bit tile key (20 x 13 = 260 lists) and a 10-bit one (30 x 20 = 600): the two instantiations
@pytest.mark.parametrize("W,H,N", [(1280, 832, 60000), (1920, 1280, 40000), (1920, 1080, 150000)])
def test_launch_forms_give_identical_lists(mods, options, W, H, N):
    L, FV, GV, Hn = mods
    cam, p, grids, sky, target = _scene(Hn, N, W, H, 3)
    with torch.no_grad():
        ref = Hn.render_view(p, cam, grids, 0, sky)               # host-count form
        M, nv = ref["info"]["n_isects"], ref["info"]["n_visible"]
        assert M > 1000 and nv > 1000
        want = _lists(ref, M, nv)
        seen = []
        for cap_launch, groups in ((0, 0), (1, 0)):
            L.set_option(L.OPT_CAP_LAUNCH, cap_launch)
            for rep in range(2):     # (twice: the tables are re-armed by the projection's prepare launch)
                caps = FV.ListCapacity(int(M * 1.3) + 100, int(nv * 1.3) + 100)
                out = Hn.render_view(p, cam, grids, 0, sky, caps=caps)
                torch.cuda.synchronize()
                assert caps.observed() == (M, nv) and not caps.overflowed(), (cap_launch, groups, caps.observed(), (M, nv), int(caps.counts.np[2]))
                got = _lists(out, M, nv)
                for a, b, name in zip(got, want, ("flatten_ranks", "visible_ids", "isect_offsets", "tiles_per_gauss")):
                    assert torch.equal(a, b), (cap_launch, groups, rep, name)
                assert torch.equal(out["rgb"], ref["rgb"]) and torch.equal(out["depth"], ref["depth"]), (cap_launch, groups, rep)
            seen.append((cap_launch, groups))
        assert len(seen) == 2


@pytest.mark.parametrize("device_counts", [False, True])
def test_row_form_of_the_projection_outputs_changes_nothing(mods, monkeypatch, device_counts):
    """include/bds.h "ROW FORM": means2d / depths / conics / opacities as the columns of one [N,8] block (recognised by the addresses)
    against five separate arrays -- lists, counts, images and gradients bit for bit (the same values from other addresses)."""
    L, FV, GV, Hn = mods
    res = {}
    for rows in (False, True):
        monkeypatch.setattr(FV, "_PROJ_ROWS", rows)
        cam, p, grids, sky, target = _scene(Hn, 90000, 1280, 832, 5)
        caps = None
        if device_counts:
            with torch.no_grad():
                ref = Hn.render_view(p, cam, grids, 0, sky)
            caps = FV.ListCapacity(int(ref["info"]["n_isects"] * 1.3) + 100, int(ref["info"]["n_visible"] * 1.3) + 100)
        out = Hn.render_view(p, cam, grids, 0, sky, caps=caps)
        assert out["info"]["means2d"].is_contiguous() != rows          # (the row form really is what ran)
        Hn.training_loss(out, target, grids).backward()
        torch.cuda.synchronize()
        M, nv = (caps.observed() if device_counts else (out["info"]["n_isects"], out["info"]["n_visible"]))
        res[rows] = (_lists(out, M, nv), out["rgb"].detach().clone(), out["depth"].detach().clone(), out["info"]["means2d"].detach().clone(),
                     [p[k].grad.clone() for k in sorted(p)], cam.viewmat.grad.clone())
    a, b = res[False], res[True]
    for x, y, name in zip(a[0], b[0], ("flatten_ranks", "visible_ids", "isect_offsets", "tiles_per_gauss")):
        assert torch.equal(x, y), name
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    for x, y, k in zip(a[4], b[4], sorted(res[True][4] and ["log_scales", "means", "opacity_logits", "quats", "sh"])):
        assert (x - y).abs().max() <= 1e-6 * max(1.0, float(y.abs().max())), k       # (atomics in the compositor's backward: order only)
    assert (a[5] - b[5]).abs().max() <= 1e-5 * max(1.0, float(b[5].abs().max()))


@pytest.mark.parametrize("ordered", [True, False], ids=["spatial_order", "given_order"])
@pytest.mark.parametrize("device_counts", [False, True])
def test_block_bounds_skip_blocks_without_changing_anything(mods, ordered, device_counts):
    """bds_gaussian_block_bounds + bds_project_view_{,prepare_}fwd_blocks: a projection that skips the 256-row blocks it can not see
    leaves the same radii, lists, images and gradients as the one that reads every row -- with the rows in spatial order (where
    most blocks go) and in the generator's order (where hardly any does), for cameras all round the rig."""
    L, FV, GV, Hn = mods
    dev = "cuda"
    W, H, N = 960, 544, 150_000
    p = Hn.synthetic_scene(N, seed=2, device=dev)
    if ordered:
        p = Hn.reorder_params(p, Hn.spatial_order(p["means"]))
    p = {k: v.requires_grad_(True) for k, v in p.items()}
    bounds = torch.zeros((N + 255) // 256, 8, device=dev)
    L.check(L.lib().bds_gaussian_block_bounds(N, L.ptr(p["means"]), L.ptr(p["log_scales"]), L.ptr(bounds), L.stream()), "bds_gaussian_block_bounds")
    lo, hi = bounds[:, 0:3], bounds[:, 4:7]
    blk = torch.arange(N, device=dev) // 256
    assert bool((p["means"].detach() >= lo[blk]).all()) and bool((p["means"].detach() <= hi[blk]).all())
    assert torch.allclose(bounds[:, 3], torch.zeros_like(bounds[:, 3]).scatter_reduce(0, blk, p["log_scales"].detach().exp().amax(1), "amax", include_self=False))
    grids = [g.requires_grad_(True) for g in Hn.make_grids(1, device=dev)]
    gen = torch.Generator().manual_seed(4)
    sky, target = torch.rand(H, W, 3, generator=gen).to(dev), torch.rand(H, W, 3, generator=gen).to(dev)
    skipped = 0
    for cam in Hn.ring_cameras(W, H, yaws_deg=(0.0, 100.0, -150.0), device=dev):
        outs = []
        for b in (None, bounds):
            for t in list(p.values()) + grids:
                t.grad = None
            caps = None
            if device_counts:
                with torch.no_grad():
                    ref = Hn.render_view(p, cam, grids, 0, sky)
                caps = FV.ListCapacity(int(ref["info"]["n_isects"] * 1.3) + 100, int(ref["info"]["n_visible"] * 1.3) + 100)
            out = Hn.train_view(p, cam, grids, 0, sky, target, caps=caps, block_bounds=b)
            torch.cuda.synchronize()
            M, nv = (caps.observed() if device_counts else (out["info"]["n_isects"], out["info"]["n_visible"]))
            outs.append((out["radii"].clone(), _lists(out, M, nv), out["rgb"].clone(), out["info"]["means2d"].detach().clone(),
                         {k: v.grad.clone() for k, v in p.items()}))
        (ra, la, ia, ma, ga), (rb, lb, ib, mb, gb) = outs
        assert torch.equal(ra, rb) and int((ra > 0).sum()) > 1000
        for x, y, name in zip(la, lb, ("flatten_ranks", "visible_ids", "isect_offsets", "tiles_per_gauss")):
            assert torch.equal(x, y), name
        assert torch.equal(ia, ib)
        vis = (ra.reshape(-1) > 0)
        assert torch.equal(ma.reshape(-1, 2)[vis], mb.reshape(-1, 2)[vis]) and float(mb.reshape(-1, 2)[~vis].abs().max()) == 0.0
        for k in ga:
            assert (ga[k] - gb[k]).abs().max() <= 1e-6 * max(1.0, float(ga[k].abs().max())), k
        # how many blocks the bound rejects for this camera: every row of a rejected block is culled
        per_block = torch.zeros(bounds.shape[0], device=dev).scatter_add_(0, blk, vis.float())
        skipped += int((per_block == 0).sum())
    if ordered:
        assert skipped > 1.5 * bounds.shape[0]      # (of 3 x the blocks: most of what the cameras do not see goes block-wise)
