"""-m gpu: coarse lists (include/bds.h): depth-ordered lists built for 32 / 64 / 128-px tiles and filtered per 16 x 16 compositing
tile give the image and the gradients of gsplat's 16-px lists -- through the rasterization() API (tile_size argument), through the
fused view (list_tile) and against the CPU oracle."""
import numpy as np
import pytest
import torch

from tests.util import make_scene, rel_err

pytestmark = pytest.mark.gpu


def _run_api(sc, W, H, tile_size, mode="RGB+ED", absgrad=True, backgrounds=None):
    import bilateral_driving_amd.rendering as R
    leaves = {k: sc[k].clone().cuda().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
    rr, aa, meta = R.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"],
                                   sc["viewmats"].cuda(), sc["Ks"].cuda(), W, H, packed=False, absgrad=absgrad, render_mode=mode,
                                   tile_size=tile_size, backgrounds=backgrounds)
    g = torch.Generator(device="cpu").manual_seed(7)
    w1, w2 = torch.randn(rr.shape, generator=g).cuda(), torch.randn(aa.shape, generator=g).cuda()
    ((rr * w1).sum() + (aa * w2).sum()).backward()
    grads = {k: v.grad.clone() for k, v in leaves.items()}
    if absgrad:
        grads["absgrad"] = meta["means2d"].absgrad.clone()
    return rr.detach(), aa.detach(), grads, meta


@pytest.mark.parametrize("seed,N,W,H,C", [(0, 1500, 200, 130, 1), (1, 800, 75, 50, 2)])
@pytest.mark.parametrize("tile_size", [32, 64, 128])
def test_rasterization_api_tile_size_equals_oracle_with_that_tile_size(seed, N, W, H, C, tile_size):
    """gsplat semantics of the tile_size argument: a splat is clipped to its bounding square at THAT tile granularity."""
    from oracle import gs_oracle as O
    sc = make_scene(N, W, H, seed=seed)
    if C > 1:   # a second camera, shifted sideways
        vm2 = sc["viewmats"][0].clone()
        vm2[0, 3] += 0.4
        sc["viewmats"] = torch.stack([sc["viewmats"][0], vm2])
        sc["Ks"] = sc["Ks"].expand(2, 3, 3).contiguous()
    bg = torch.rand(C, 3)
    rc, ac, gc, mc = _run_api(sc, W, H, tile_size, backgrounds=bg.cuda(), absgrad=False)
    r16, a16, _, m16 = _run_api(sc, W, H, 16, backgrounds=bg.cuda(), absgrad=False)
    assert mc["tile_size"] == tile_size and mc["flatten_ids"].numel() < m16["flatten_ids"].numel()
    assert mc["isect_offsets"].shape[1:] == (-(-H // tile_size), -(-W // tile_size))
    leaves = {k: sc[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
    rr, aa, _ = O.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"], sc["viewmats"],
                                sc["Ks"], W, H, render_mode="RGB+ED", tile_size=tile_size, backgrounds=torch.cat([bg, torch.zeros(C, 1)], -1))
    g = torch.Generator(device="cpu").manual_seed(7)
    w1, w2 = torch.randn(rr.shape, generator=g), torch.randn(aa.shape, generator=g)
    ((rr * w1).sum() + (aa * w2).sum()).backward()
    assert rel_err(rc.cpu()[..., :3], rr.detach()[..., :3]) < 1e-5 and rel_err(ac.cpu(), aa.detach()) < 1e-5
    cov = (aa.detach() > 1e-3).squeeze(-1)
    assert rel_err(rc.cpu()[..., 3][cov], rr.detach()[..., 3][cov]) < 1e-4
    for k, v in leaves.items():
        assert rel_err(gc[k].cpu(), v.grad) < 2e-4, (k, rel_err(gc[k].cpu(), v.grad))
    # larger list tiles clip the splats later (at their own tile granularity): the image moves a little
    assert rel_err(rc[..., :3], r16[..., :3]) < 0.05 and rel_err(ac, a16) < 0.05


@pytest.mark.parametrize("W,H,N", [(320, 192, 4000), (200, 130, 1500)])
def test_fused_view_list_tile_64_equals_16(W, H, N):
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd.fused_view import fused_view, render_classes
    dev = "cuda"
    cam = Hn.ring_cameras(W, H, yaws_deg=(0.0,), device=dev)[0]
    base = Hn.synthetic_scene(N, seed=3, device=dev)
    base["means"] = base["means"] * torch.tensor([0.3, 0.3, 1.0], device=dev)
    grids0 = Hn.make_grids(2, device=dev)
    sky, target = torch.rand(H, W, 3, device=dev), torch.rand(H, W, 3, device=dev)
    res = {}
    for lt in (16, 64):
        p = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        grids = [g.clone().requires_grad_(True) for g in grids0]
        vm = cam.viewmat.clone().requires_grad_(True)
        out = fused_view(p, vm, cam.K, W, H, grids, sky, Hn.FACTORS_3, img_idx=1, list_tile=lt)
        assert out["info"]["tile_size"] == lt
        ((out["rgb"] - target).abs().mean() + 0.01 * out["depth"].mean() + 0.1 * out["opacity"].mean()).backward()
        res[lt] = (out["rgb"].detach(), out["depth"].detach(), out["opacity"].detach(), {k: v.grad for k, v in p.items()},
                   [g.grad for g in grids], vm.grad, out["info"]["means2d"].absgrad, out["info"]["n_isects"])
    a, b = res[16], res[64]
    assert b[7] < a[7]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for k in a[3]:
        assert rel_err(b[3][k], a[3][k]) < 2e-5, k
    for x, y in zip(a[4], b[4]):
        assert rel_err(y, x) < 2e-5
    assert rel_err(b[5], a[5]) < 1e-4 and rel_err(b[6], a[6]) < 2e-5
    # the evaluation re-renders go through the same front
    m = {"half": torch.arange(N, device=dev) % 2 == 0}
    o16 = render_classes(base, cam.viewmat, cam.K, W, H, m, list_tile=16)
    o64 = render_classes(base, cam.viewmat, cam.K, W, H, m, list_tile=64)
    for k in o16:
        assert torch.equal(o16[k], o64[k]), k


@pytest.mark.parametrize("list_tile", [16, 64])
def test_schedule_keys_left_by_the_forward_compositor(list_tile):
    """bds_rasterize_fwd_dev(tile_order=...) + bds_rasterize_bwd_schedule_sort == bds_rasterize_bwd_schedule (keys re-derived from
    last_ids by a launch of its own): same per-tile keys, same schedule."""
    import math
    from bilateral_driving_amd import _lib as L
    from bilateral_driving_amd import gs_ops as ops
    from bilateral_driving_amd import harness as Hn
    dev = torch.device("cuda")
    W, H, N = 330, 200, 6000
    cam = Hn.ring_cameras(W, H, yaws_deg=(0.0,), device=dev)[0]
    p = Hn.synthetic_scene(N, seed=5, device=dev)
    lib, st = L.lib(), L.stream()
    with torch.no_grad():
        radii, m2, dep, con, _ = ops.fully_fused_projection(p["means"], p["quats"], torch.exp(p["log_scales"]), cam.viewmat[None], cam.K[None],
                                                            W, H, near_plane=0.1)
        op = torch.sigmoid(p["opacity_logits"])[None].contiguous()
        ltw, lth = math.ceil(W / list_tile), math.ceil(H / list_tile)
        tw, th = math.ceil(W / 16), math.ceil(H / 16)
        _, _, fids, offs = ops.isect_tiles(m2, radii, dep, list_tile, ltw, lth, want_isect_ids=False, conics=con, opacities=op)
        M = fids.numel()
        col = torch.rand(1, N, 4, device=dev)
        rec = torch.empty(N, L.SPLAT_RECORD_FLOATS, device=dev)
        L.check(lib.bds_splat_pack(N, 4, None, L.ptr(m2), L.ptr(con), L.ptr(col), L.ptr(op), L.ptr(radii), L.ptr(rec), st), "pack")
        render, alphas = torch.empty(1, H, W, 4, device=dev), torch.empty(1, H, W, 1, device=dev)
        last = torch.zeros(1, H, W, dtype=torch.int32, device=dev)
        m_dev = torch.tensor([M], dtype=torch.int64, device=dev)
        n_ints = int(lib.bds_rasterize_schedule_ints(1, tw, th))
        total, NB = tw * th, 32          # (bins per XCD range: csrc/rasterize.hip kSchedLogBins)

        def forward(order):
            L.check(lib.bds_rasterize_fwd_dev(1, N, M + 100, m_dev.data_ptr(), 4, L.ptr(rec), None, W, H, 16, list_tile, tw, th, L.ptr(offs),
                                              L.ptr(fids), L.ptr(render), L.ptr(alphas), None, L.ptr(last), L.ptr(order), 0, 0, 0, st), "fwd")
            L.check(lib.bds_rasterize_bwd_schedule_sort(1, tw, th, L.ptr(order), st), "sort")

        # sorted form (option 8 = 0): the waves leave their keys, one launch sorts them
        L.set_option(L.OPT_SCHED_BINS, 0)
        try:
            order = torch.full((n_ints,), -7, dtype=torch.int32, device=dev)
            forward(order)
        finally:
            L.set_option(L.OPT_SCHED_BINS, 1)
        ref = ops.bwd_schedule(1, W, H, list_tile, offs, last)
        # binned form (default): the waves drop their tile into the bin of its length; no launch (the sort call is a no-op).  The header
        # is cleared by the record pack in the product; here by hand
        binned = torch.full((n_ints,), -7, dtype=torch.int32, device=dev)
        binned[:1 + 8 * NB] = 0
        forward(binned)
        torch.cuda.synchronize()
    assert int(order[0]) == 1 and int(ref[0]) == 1 and int(order[1:1 + 2 * total].min()) >= 0
    assert torch.equal(order[1 + total:1 + 2 * total], ref[1 + total:1 + 2 * total]) and int(ref[1 + total:1 + 2 * total].max()) > 0        # keys
    assert torch.equal(order[1:1 + total], ref[1:1 + total])                                  # schedule
    work = ref[1 + total:1 + 2 * total].cpu().long()
    b = binned.cpu().long()
    assert int(b[0]) == 0
    counts = b[1:1 + 8 * NB].reshape(8, NB)
    stride = total // 8 + 1
    lists = b[1 + 8 * NB:1 + 8 * NB + 8 * NB * stride].reshape(8, NB, stride)
    per, rem = divmod(total, 8)
    first = 0
    for x in range(8):
        cnt = per + (1 if x < rem else 0)
        assert int(counts[x].sum()) == cnt
        seen, floor_prev = [], None
        for k in range(NB):
            items = lists[x, k, :int(counts[x, k])]
            seen.append(items)
            if items.numel():
                w = work[items]
                if k < NB - 1:
                    assert int(w.min()) > 0
                    if floor_prev is not None:
                        assert int(w.max()) <= floor_prev                 # bins are ordered longest first
                    floor_prev = int(w.min())
                else:
                    assert int(w.max()) == 0                              # the last bin: nothing to do
        seen = torch.cat(seen).sort().values
        assert torch.equal(seen, torch.arange(first, first + cnt))        # every tile of the XCD's range exactly once
        first += cnt
