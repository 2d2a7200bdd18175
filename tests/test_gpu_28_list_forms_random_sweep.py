"""-m gpu: the fused view's list forms against each other on the random scenes of tests/test_gpu_25: list tiles of 16 / 32 / 64 /
128 px (the compositor filters a larger list tile's candidates per 16-px tile), long tiles composited strip by strip from a random
threshold on (``split_len``), with a capped long-tile list (``split_cap``) and refined candidate lists from a pool that is ample,
tiny (runs out after a few tiles: the rest fall back to the list-tile list) or absent -- include/bds.h bds_rasterize_fwd_dev /
_bwd_dev.  Every form composites the same (pixel, Gaussian) pairs in the same order: images BIT-identical, gradients to the
re-association of their atomics.  tests/test_gpu_08 / test_gpu_22 hold the same on the benchmark and lidar scenes.

BDS_SWEEP_CASES (default 8) scenes x 4 forms."""
import os

import pytest
import torch

from tests.test_gpu_25_gs_random_sweep import random_scene
from tests.util import rel_err

pytestmark = pytest.mark.gpu
N_CASES = int(os.environ.get("BDS_SWEEP_CASES", "8"))
C0 = 0.28209479177387814


def _run(Hn, FV, cam, p, grids, sky, target, M_cap, nv_cap, kw, list_tile):
    for t in list(p.values()) + grids + [sky, cam.viewmat]:
        t.grad = None
    caps = FV.ListCapacity(M_cap, nv_cap)
    out = Hn.render_view(p, cam, grids, 0, sky, caps=caps, list_tile=list_tile, **kw)
    Hn.training_loss(out, target, grids).backward()
    torch.cuda.synchronize()
    assert not caps.overflowed()
    return (out["rgb"].detach().clone(), out["depth"].detach().clone(), out["opacity"].detach().clone(),
            {k: t.grad.clone() for k, t in p.items()}, [g.grad.clone() for g in grids], sky.grad.clone(), cam.viewmat.grad.clone(),
            out["info"]["means2d"].absgrad.clone(), caps.observed())


@pytest.mark.parametrize("seed", list(range(N_CASES)))
def test_list_forms_composite_the_same_pixels(monkeypatch, seed):
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from bilateral_driving_amd import _lib, fused_view as FV, harness as Hn
    _lib.lib()
    monkeypatch.setattr(FV, "SH_IN_PACK", FV.SH_IN_PACK_DEV)      # (the host-count view below through the same SH kernel as the others)
    sc, W, H, _, kw, _ = random_scene(300 + seed)
    spheres = kw.pop("anisotropy") == 1.0
    g = torch.Generator().manual_seed(900 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    N = sc["means"].shape[0]
    sh = torch.empty(N, 16, 3)
    sh[:, 0] = (sc["colors"] - 0.5) / C0
    sh[:, 1:] = torch.randn(N, 15, 3, generator=g) * 0.1
    p = dict(means=sc["means"], log_scales=torch.log(sc["scales"]), quats=sc["quats"],
             opacity_logits=torch.logit(sc["opacities"].double().clamp(2e-9, 1 - 2e-9)).float(), sh=sh)
    p = {k: v.cuda().contiguous().requires_grad_(True) for k, v in p.items()}
    vm = sc["viewmats"][0].cuda().requires_grad_(True)
    cam = Hn.Camera(vm, sc["Ks"][0].cuda(), W, H, torch.linalg.inv(vm.detach())[:3, 3].contiguous())
    grids = [x.requires_grad_(True) for x in Hn.make_grids(1, seed=seed, device="cuda")]
    sky = torch.rand(H, W, 3, generator=g).cuda().requires_grad_(True)
    target = torch.rand(H, W, 3, generator=g).cuda()
    with torch.no_grad():
        ref = Hn.render_view(p, cam, grids, 0, sky, list_tile=16, **kw)
    M16, nv = ref["info"]["n_isects"], ref["info"]["n_visible"]
    M_cap, nv_cap = int(M16 * 1.3) + 100, int(nv * 1.3) + 100        # (larger list tiles list fewer pairs)
    monkeypatch.setenv("BDS_SPLIT_LEN", "0")
    monkeypatch.setenv("BDS_SPLIT_CAP", "0")
    base = _run(Hn, FV, cam, p, grids, sky, target, M_cap, nv_cap, kw, 16)
    assert torch.equal(base[0], ref["rgb"]) and base[8] == (M16, nv)
    for form in range(4):
        lt = (16, 32, 64, 128)[ri(0, 3)]
        with torch.no_grad():
            info = Hn.render_view(p, cam, grids, 0, sky, list_tile=lt, **kw)["info"]
        offs = info["isect_offsets"].reshape(-1).long()
        lens = torch.diff(offs, append=offs.new_tensor([info["n_isects"]]))
        split = (1, max(int(lens.float().quantile(0.5)), 1), max(int(lens.float().quantile(0.9)), 1), int(lens.max()) + 1)[ri(0, 3)] if lens.numel() else 1
        cap, pool = (0, 0, 3)[ri(0, 2)], (4096, 4096, 0, 24, 300)[ri(0, 4)]
        monkeypatch.setenv("BDS_SPLIT_LEN", str(split))
        monkeypatch.setenv("BDS_SPLIT_CAP", str(cap))
        monkeypatch.setattr(FV, "SPLIT_POOL_PER_TILE", pool)
        got = _run(Hn, FV, cam, p, grids, sky, target, M_cap, nv_cap, kw, lt)
        tag = dict(seed=seed, list_tile=lt, split_len=split, split_cap=cap, pool=pool, N=N, W=W, H=H, visible=nv, pairs=M16)
        assert torch.equal(got[0], base[0]) and torch.equal(got[1], base[1]) and torch.equal(got[2], base[2]), tag
        if nv == 0:
            continue
        for k in base[3]:
            if k == "quats" and spheres:
                continue
            if float(base[3][k].abs().max()) == 0.0:
                assert float(got[3][k].abs().max()) == 0.0, (k, tag)
                continue
            assert float((got[3][k] - base[3][k]).norm() / base[3][k].norm()) < 1e-4, (k, tag)
        for a, b in zip(got[4], base[4]):
            assert rel_err(a, b) < 5e-5, tag
        assert rel_err(got[5], base[5]) < 1e-5, tag
        if float(base[6].abs().max()) > 0:
            assert float((got[6] - base[6]).norm() / base[6].norm()) < 1e-3, tag
        if float(base[7].abs().max()) > 0:
            assert float((got[7] - base[7]).norm() / base[7].norm()) < 1e-4, tag
