"""-m gpu: the tile stage (``isect_tiles``: counts, (tile | depth) keys, sorted ids, offsets) against the oracle BIT FOR BIT on the
random scenes of tests/test_gpu_25 -- 1 to 3 cameras, list tiles of 16 ... 128 px, with the size-regime hooks forced either way
(two-launch short radix passes | generic passes; packed entries | (key, id) pairs), Gaussians behind the camera / off screen /
image-filling, images narrower than a tile, nothing visible -- and the culled lists: a subset of the pairs in the same order, equal
across the regimes.  (gsplat's isect_tiles + isect_offset_encode as called inside rasterization(), trainers/base.py:393-408.)

BDS_SWEEP_CASES (default 10) cases."""
import os

import pytest
import torch

from oracle import gs_oracle as G
from tests.test_gpu_25_gs_random_sweep import _cams, random_scene

pytestmark = pytest.mark.gpu
N_CASES = int(os.environ.get("BDS_SWEEP_CASES", "10"))


@pytest.mark.parametrize("seed", list(range(N_CASES)))
def test_isect_random_scene(seed):
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    import bilateral_driving_amd.gs_ops as ops
    from bilateral_driving_amd import _lib as L
    L.lib()
    sc, W, H, _, kw, _ = random_scene(800 + seed)
    kw.pop("anisotropy")
    vms, Ks, _, g = _cams(sc, seed)
    C, N = vms.shape[0], sc["means"].shape[0]
    ts = (16, 16, 32, 64, 128)[int(torch.randint(0, 5, (1,), generator=g))]
    short, packed = int(torch.randint(0, 2, (1,), generator=g)), int(torch.randint(0, 2, (1,), generator=g))
    radii, m2, d, con, _ = ops.fully_fused_projection(sc["means"].cuda(), sc["quats"].cuda(), sc["scales"].cuda(), vms.cuda(), Ks.cuda(), W, H,
                                                       near_plane=kw["near_plane"], radius_clip=kw["radius_clip"])
    tw, th = (W + ts - 1) // ts, (H + ts - 1) // ts
    op = sc["opacities"].cuda()[None].expand(C, N).contiguous()
    tag = dict(seed=seed, N=N, W=W, H=H, C=C, tile=ts, short=short, packed=packed, visible=int((radii > 0).sum()))
    try:
        ref_cull = ops.isect_tiles(m2, radii, d, ts, tw, th, want_isect_ids=False, conics=con, opacities=op)      # default regime
        L.set_option(L.OPT_SHORT_SORT, short); L.set_option(L.OPT_PACKED, packed)
        tpg, iids, fids, offs = ops.isect_tiles(m2, radii, d, ts, tw, th)
        got_cull = ops.isect_tiles(m2, radii, d, ts, tw, th, want_isect_ids=False, conics=con, opacities=op)
    finally:
        L.set_option(L.OPT_SHORT_SORT, 1); L.set_option(L.OPT_PACKED, 1)
    keys, vals, cnts = [], [], []
    for c in range(C):
        t, k, v = G.isect_tiles(m2[c].cpu(), radii[c].cpu(), d[c].cpu(), ts, tw, th)
        cnts.append(t)
        keys.append(k + (c * tw * th << 32))
        vals.append(v.long() + c * N)
    keys, vals = torch.cat(keys), torch.cat(vals)
    assert torch.equal(tpg.cpu(), torch.stack(cnts)), tag
    assert iids.numel() == keys.numel() and torch.equal(iids.cpu(), keys) and torch.equal(fids.cpu().long(), vals), tag
    ref_off = torch.searchsorted((keys >> 32).contiguous(), torch.arange(C * tw * th)).to(torch.int32).reshape(C, th, tw)
    assert torch.equal(offs.cpu(), ref_off), tag
    # the culled lists: equal across the regimes; per tile a subsequence of the full list
    for a, b in zip(got_cull, ref_cull):
        assert (a is None and b is None) or torch.equal(a, b), tag
    tpg_c, _, fids_c, offs_c = got_cull
    assert fids_c.numel() <= fids.numel() and bool((tpg_c <= tpg).all()), tag
    full, cul = fids.cpu().tolist(), fids_c.cpu().tolist()
    o_f, o_c = offs.reshape(-1).cpu().tolist() + [len(full)], offs_c.reshape(-1).cpu().tolist() + [len(cul)]
    for tile in range(0, C * tw * th, max(1, C * tw * th // 60)):
        a, b = full[o_f[tile]:o_f[tile + 1]], cul[o_c[tile]:o_c[tile + 1]]
        it = iter(a)
        assert all(x in it for x in b), (tag, tile)         # b is a subsequence of a (same order)
