"""-m gpu: long tiles composited strip by strip (include/bds.h bds_rasterize_fwd_dev / _bwd_dev ``split_len``): a tile whose list
holds at least split_len entries (the first split_cap of them) is taken by FOUR waves, one 16 x 4 strip each, the candidates filtered per strip -- against the
one-wave-per-tile form that tests/test_gpu_01 / test_gpu_03 tie to the oracle.  Same pixels in the same order: images and last-id maps
bit-identical, gradients equal to the order of their atomics."""
import pytest
import torch

from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def same_sh_kernel(monkeypatch):
    from bilateral_driving_amd import fused_view as FV
    monkeypatch.setattr(FV, "SH_IN_PACK", FV.SH_IN_PACK_DEV)


def _scene(N, W, H, seed, lidar=False, dev="cuda"):
    from bilateral_driving_amd import harness as Hn
    cam = Hn.ring_cameras(W, H, yaws_deg=(0.0,), device=dev)[0]
    cam.viewmat.requires_grad_(True)
    if lidar:
        p = Hn.lidar_scene(N, seed=seed, device=dev)
    else:
        p = Hn.synthetic_scene(N, seed=seed, device=dev)
        p["means"] = p["means"] * torch.tensor([0.5, 0.5, 1.0], device=dev)
    p = {k: v.requires_grad_(True) for k, v in p.items()}
    grids = [g.requires_grad_(True) for g in Hn.make_grids(1, device=dev)]
    gen = torch.Generator().manual_seed(31 + seed)
    sky = torch.rand(H, W, 3, generator=gen).to(dev).requires_grad_(True)
    target = torch.rand(H, W, 3, generator=gen).to(dev)
    return Hn, cam, p, grids, sky, target


def _run(Hn, FV, cam, p, grids, sky, target, M, nv):
    for t in list(p.values()) + grids + [sky, cam.viewmat]:
        t.grad = None
    caps = FV.ListCapacity(int(M * 1.3) + 100, int(nv * 1.3) + 100)
    out = Hn.render_view(p, cam, grids, 0, sky, caps=caps)
    Hn.training_loss(out, target, grids).backward()
    torch.cuda.synchronize()
    assert caps.observed() == (M, nv) and not caps.overflowed()
    return (out["rgb"].detach().clone(), out["depth"].detach().clone(), out["opacity"].detach().clone(),
            {k: t.grad.clone() for k, t in p.items()}, [g.grad.clone() for g in grids], sky.grad.clone(), cam.viewmat.grad.clone(),
            out["info"]["means2d"].absgrad.clone())


@pytest.mark.parametrize("N,W,H,lidar", [(30000, 640, 384, False), (200000, 1280, 704, True), (777, 250, 130, False)])
def test_strip_split_equals_one_wave_per_tile(monkeypatch, N, W, H, lidar):
    """split_len = 1 (EVERY non-empty tile strip by strip), a mixed threshold, and off: the device-count view end to end."""
    from bilateral_driving_amd import _lib
    _lib.lib()
    from bilateral_driving_amd import fused_view as FV
    Hn, cam, p, grids, sky, target = _scene(N, W, H, 2, lidar)
    with torch.no_grad():
        ref = Hn.render_view(p, cam, grids, 0, sky)
    M, nv = ref["info"]["n_isects"], ref["info"]["n_visible"]
    offs = ref["info"]["isect_offsets"].reshape(-1).long()
    lens = torch.diff(offs, append=offs.new_tensor([M]))
    mixed = max(int(lens.float().quantile(0.7)), 2)          # ~30 % of the list tiles above the threshold
    monkeypatch.setenv("BDS_SPLIT_LEN", "0")
    base = _run(Hn, FV, cam, p, grids, sky, target, M, nv)
    assert torch.equal(base[0], ref["rgb"])
    # (cap 0: every tile; 7: most long tiles overflow the list.  pool: int32 words per long tile of the capacity for the tiles' REFINED
    #  lists -- 4096: every long tile's strips walk its own candidates; 0: off, the strips walk the list-tile list; 24: the pool runs out
    #  after a few tiles, the rest fall back -- all of them the same pixels)
    for split, cap, pool in ((1, 0, 4096), (mixed, 0, 4096), (mixed, 0, 0), (mixed, 0, 24), (mixed, 7, 4096), (int(lens.max()) + 1, 0, 4096)):
        monkeypatch.setenv("BDS_SPLIT_LEN", str(split))
        monkeypatch.setenv("BDS_SPLIT_CAP", str(cap))
        monkeypatch.setattr(FV, "SPLIT_POOL_PER_TILE", pool)
        got = _run(Hn, FV, cam, p, grids, sky, target, M, nv)
        assert torch.equal(got[0], base[0]) and torch.equal(got[1], base[1]) and torch.equal(got[2], base[2]), split
        for k in base[3]:
            if lidar and k == "quats":      # isotropic splats: the rotation's gradient is cancellation noise (|g| ~ 1e-7 of the others)
                assert float(got[3][k].norm()) < 1e-3 * float(base[3]["means"].norm())
                continue
            assert rel_err(got[3][k], base[3][k]) < 1e-4, (split, k, rel_err(got[3][k], base[3][k]))    # (four times the atomics per pair)
        for a, b in zip(got[4], base[4]):
            assert rel_err(a, b) < 2e-5, split
        assert rel_err(got[5], base[5]) < 1e-6 and rel_err(got[6], base[6]) < 2e-4 and rel_err(got[7], base[7]) < 1e-4, split


def test_frame_graph_picks_the_split_for_a_view_with_a_few_very_long_lists(monkeypatch):
    """graph_slots.split_len_for: off on the ring scene, on for a camera that looks down a lidar-initialised street; the replayed frame
    with the split equals the eager frame (bilateral_driving_amd/selfcheck.py)."""
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd import graph_slots as GS
    from bilateral_driving_amd.graph_view import FrameGraph
    from bilateral_driving_amd.selfcheck import frame_against_eager
    monkeypatch.setattr(GS, "SPLIT_LIST_LEN", 1500)
    dev = "cuda"
    W, H, N = 1280, 704, 300_000
    cams = Hn.ring_cameras(W, H, yaws_deg=(0.0, 110.0), device=dev)
    for c in cams:
        c.viewmat.requires_grad_(True)
    p = {k: t.requires_grad_(True) for k, t in Hn.lidar_scene(N, seed=0, device=dev).items()}
    grids = [g.requires_grad_(True) for g in Hn.make_grids(len(cams), device=dev)]
    gen = torch.Generator().manual_seed(5)
    skies = [torch.rand(H, W, 3, generator=gen).to(dev).requires_grad_(True) for _ in cams]
    targets = [torch.rand(H, W, 3, generator=gen).to(dev) for _ in cams]
    frame = FrameGraph(p, cams, grids, skies, targets)
    assert frame.split_len[0] == 1500 and frame.split_cap[0] > 64, (frame.split_len, frame.split_cap)        # the camera looking down the street
    res = frame_against_eager(frame, p, cams, grids, skies, targets, Hn.FACTORS_3)
    assert res["ok"], res
    q = {k: t.requires_grad_(True) for k, t in Hn.synthetic_scene(50_000, seed=0, device=dev).items()}
    ring = FrameGraph(q, cams, grids, skies, targets)
    assert ring.split_len == [0, 0], ring.split_len
