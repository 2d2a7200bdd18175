"""-m gpu: the launch forms of the device-count tile stage (include/bds.h options 0 and 5) against each other and against the
host-count form that tests/test_gpu_01 / test_gpu_08 tie to the oracle's stable (tile | depth) sort: launches sized by the
visible-entry capacity instead of N, and the whole stage as ONE persistent launch (bds_isect_lists_dev: a fixed set of workgroups
walks the stage's phases separated by a device-wide barrier).  Integer work: lists, offsets and counts are compared bit for bit."""
import pytest
import torch

from tests.util import rel_err

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
    keep = {w: int(L.lib().bds_get_option(w)) for w in (L.OPT_CAP_LAUNCH, L.OPT_TILE_PERSIST)}
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
        for cap_launch, groups in ((0, 0), (1, 0), (1, 8), (1, 64), (1, 128), (0, 256)):
            L.set_option(L.OPT_CAP_LAUNCH, cap_launch)
            L.set_option(L.OPT_TILE_PERSIST, groups)
            for rep in range(2):     # (twice: the barrier words and the tables are re-armed by the projection's prepare launch)
                caps = FV.ListCapacity(int(M * 1.3) + 100, int(nv * 1.3) + 100)
                out = Hn.render_view(p, cam, grids, 0, sky, caps=caps)
                torch.cuda.synchronize()
                assert caps.observed() == (M, nv) and not caps.overflowed(), (cap_launch, groups, caps.observed(), (M, nv), int(caps.counts.np[2]))
                got = _lists(out, M, nv)
                for a, b, name in zip(got, want, ("flatten_ranks", "visible_ids", "isect_offsets", "tiles_per_gauss")):
                    assert torch.equal(a, b), (cap_launch, groups, rep, name)
                assert torch.equal(out["rgb"], ref["rgb"]) and torch.equal(out["depth"], ref["depth"]), (cap_launch, groups, rep)
            seen.append((cap_launch, groups))
        assert len(seen) == 6


def test_persistent_launch_overflow_renders_nothing_and_is_flagged(mods, options):
    L, FV, GV, Hn = mods
    W, H, N = 1280, 832, 30000
    cam, p, grids, sky, target = _scene(Hn, N, W, H, 5)
    with torch.no_grad():
        ref = Hn.render_view(p, cam, grids, 0, sky)
    M, nv = ref["info"]["n_isects"], ref["info"]["n_visible"]
    L.set_option(L.OPT_TILE_PERSIST, 32)
    for m_cap, nv_cap in ((max(M // 2, 1), nv + 10), (M + 10, max(nv // 2, 1))):
        caps = FV.ListCapacity(m_cap, nv_cap)
        for t in list(p.values()) + grids:
            t.grad = None
        out = Hn.render_view(p, cam, grids, 0, sky, caps=caps)
        Hn.training_loss(out, target, grids).backward()
        torch.cuda.synchronize()
        assert caps.overflowed() and int(caps.counts.np[2]) == 1 and caps.observed()[1] == nv
        assert caps.observed()[0] == (M if nv_cap >= nv else 0)
        assert float(out["opacity"].abs().max()) == 0.0
        for k, t in p.items():
            assert t.grad is None or float(t.grad.abs().max()) == 0.0, k


def test_persistent_launch_in_the_replayed_frame(mods, options):
    """FrameGraph with the persistent tile stage == FrameGraph with the 13-launch stage: images bit-equal, gradients to atomics noise,
    over several replays on two streams (the launch holds its waves next to the other stream's kernels)."""
    L, FV, GV, Hn = mods
    W, H, N = 1280, 832, 50000
    cams = Hn.ring_cameras(W, H, yaws_deg=(0.0, 120.0, 240.0), device="cuda")
    for c in cams:
        c.viewmat.requires_grad_(True)
    p = Hn.synthetic_scene(N, seed=9, device="cuda")
    p["means"] = p["means"] * torch.tensor([0.5, 0.5, 1.0], device="cuda")
    p = {k: v.requires_grad_(True) for k, v in p.items()}
    grids = [g.requires_grad_(True) for g in Hn.make_grids(3, device="cuda")]
    gen = torch.Generator().manual_seed(4)
    skies = [torch.rand(H, W, 3, generator=gen).cuda().requires_grad_(True) for _ in cams]
    targets = [torch.rand(H, W, 3, generator=gen).cuda() for _ in cams]

    def run(groups):
        L.set_option(L.OPT_TILE_PERSIST, groups)
        frame = GV.FrameGraph(p, cams, grids, skies, targets)
        res = None
        for _ in range(3):
            assert frame.step() is True
            torch.cuda.synchronize()
            cur = ([v.rgb.clone() for v in frame.views], frame.flat.flat.clone(), [c.viewmat.grad.clone() for c in cams], frame.counts())
            if res is not None:
                assert all(torch.equal(a, b) for a, b in zip(cur[0], res[0])) and cur[3] == res[3]
            res = cur
        del frame
        return res

    a = run(0)
    b = run(96)
    assert a[3] == b[3]
    for x, y in zip(a[0], b[0]):
        assert torch.equal(x, y)
    assert rel_err(b[1], a[1]) < 2e-5
    for x, y in zip(a[2], b[2]):
        assert rel_err(y, x) < 1e-4
