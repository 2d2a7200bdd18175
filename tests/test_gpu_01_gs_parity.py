"""-m gpu: HIP kernels (through the C-ABI / ctypes binding) against the oracle, op by op and
end to end.  The oracle for this half is PARITY-UNPINNED against gsplat 1.3.0 (see
oracle/gs_oracle.py); tolerances are north_star's: RGB 1e-4 rel, gradients 1e-3 rel."""
import numpy as np
import pytest
import torch

from oracle import gs_oracle as G
from tests.util import grad_errors, make_scene, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    import bilateral_driving_amd.gs_ops as ops
    from bilateral_driving_amd import _lib
    _lib.lib()  # fails loudly if libbds.so is missing
    return ops


def dev(t):
    return t.cuda() if torch.is_tensor(t) else t


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
@pytest.mark.parametrize("n,K", [(1, 16), (255, 16), (1000, 16), (70001, 16), (513, 9), (300, 4)])
def test_sh_fwd_bwd(ops, deg, n, K):
    if (deg + 1) ** 2 > K:
        pytest.skip("K too small for this degree")
    g = torch.Generator().manual_seed(n + deg)
    dirs = torch.randn(n, 3, generator=g) * 2
    coeffs = torch.randn(n, K, 3, generator=g)
    masks = torch.rand(n, generator=g) > 0.2
    v = torch.randn(n, 3, generator=g)
    for m in (None, masks):
        d_ref = dirs.double().requires_grad_(True)
        c_ref = coeffs.double().requires_grad_(True)
        ref = G.spherical_harmonics(deg, d_ref, c_ref, None if m is None else m.double())
        (ref * v.double()).sum().backward()
        d_g = dirs.cuda().requires_grad_(True)
        c_g = coeffs.cuda().requires_grad_(True)
        out = ops.spherical_harmonics(deg, d_g, c_g, None if m is None else m.cuda())
        (out * v.cuda()).sum().backward()
        assert rel_err(out.cpu(), ref.detach()) < 1e-5
        assert rel_err(c_g.grad.cpu(), c_ref.grad) < 1e-5
        if deg > 0:
            assert rel_err(d_g.grad.cpu(), d_ref.grad) < 1e-4


@pytest.mark.parametrize("seed,N,W,H", [(0, 1000, 256, 256), (1, 5000, 640, 360), (2, 257, 100, 37)])
def test_projection_fwd_bwd(ops, seed, N, W, H):
    sc = make_scene(N, W, H, seed=seed, spread=1.5)
    ref_in = {k: sc[k].double().requires_grad_(True) for k in ("means", "quats", "scales")}
    vm = sc["viewmats"].double().requires_grad_(True)
    radii_r, m2_r, d_r, c_r, _ = G.project(ref_in["means"], ref_in["quats"], ref_in["scales"], vm[0], sc["Ks"][0].double(), W, H)
    gpu_in = {k: sc[k].cuda().requires_grad_(True) for k in ("means", "quats", "scales")}
    vm_g = sc["viewmats"].cuda().requires_grad_(True)
    radii, m2, d, c, comp = ops.fully_fused_projection(gpu_in["means"], gpu_in["quats"], gpu_in["scales"], vm_g, sc["Ks"].cuda(), W, H)
    assert comp is None
    same = radii[0].cpu() == radii_r
    print(f"[projection] seed {seed}: radii decisions equal to the fp64 oracle's for {float(same.float().mean()):.5f} of the Gaussians")
    # measured on MI355X: all three cases 1.00000 (every radius / cull decision equals the fp64 oracle's); one Gaussian in a thousand
    # on a ceil() boundary is what fp32 against fp64 may legitimately flip
    assert same.float().mean() >= 0.999
    vis = (radii_r > 0) & same
    assert int(vis.sum()) > N // 10
    assert rel_err(m2[0].cpu()[vis], m2_r[vis]) < 1e-5
    assert rel_err(d[0].cpu()[vis], d_r[vis]) < 1e-6
    assert ((c[0].cpu()[vis].double() - c_r[vis]).abs() / c_r[vis].abs().clamp(min=1e-3)).max() < 5e-4
    cull = (radii[0].cpu() == 0)
    assert float(m2[0].cpu()[cull].abs().max()) == 0.0 and float(c[0].cpu()[cull].abs().max()) == 0.0
    g = torch.Generator().manual_seed(seed)
    w2, wd, wc = torch.randn(N, 2, generator=g), torch.randn(N, generator=g), torch.randn(N, 3, generator=g)
    m = same.double()  # ignore the handful of Gaussians whose cull decision differs between fp32 and fp64
    ((m2_r * w2.double() * m[:, None]).sum() + (d_r * wd.double() * m).sum() + (c_r * wc.double() * m[:, None]).sum()).backward()
    mg = same.cuda().float()
    ((m2[0] * w2.cuda() * mg[:, None]).sum() + (d[0] * wd.cuda() * mg).sum() + (c[0] * wc.cuda() * mg[:, None]).sum()).backward()
    # the yardstick: the same oracle in float32 on the same loss (a random gradient on the conics of wide, thin splats is dominated by a
    # few ill-conditioned rows: plain fp32 autograd itself is at 1.2e-4 / 1.3e-4 on quats / scales of the 5000-Gaussian case)
    in32 = {k: sc[k].detach().clone().float().requires_grad_(True) for k in ("means", "quats", "scales")}
    _, m2_32, d_32, c_32, _ = G.project(in32["means"], in32["quats"], in32["scales"], sc["viewmats"][0].float(), sc["Ks"][0].float(), W, H)
    ((m2_32 * w2 * m.float()[:, None]).sum() + (d_32 * wd * m.float()).sum() + (c_32 * wc * m.float()[:, None]).sum()).backward()
    for k in ("means", "quats", "scales"):
        got, ref = gpu_in[k].grad.cpu().double(), ref_in[k].grad
        e, e32 = float((got - ref).norm() / ref.norm()), float((in32[k].grad.double() - ref).norm() / ref.norm())
        print(f"[projection] seed {seed} {k}: gradient norm-rel error {e:.2e} (fp32 oracle {e32:.2e})")
        assert e < max(1e-4, 2.0 * e32), (k, e, e32)
    got, ref = vm_g.grad.cpu().double()[0, :3], vm.grad[0, :3]
    assert float((got - ref).norm() / ref.norm()) < 1e-4


def test_projection_multi_camera(ops):
    sc = make_scene(800, 128, 96, seed=5)
    vm2 = sc["viewmats"].clone()
    vm2[0, :3, 3] += torch.tensor([0.3, 0.1, -0.2])
    vms = torch.cat([sc["viewmats"], vm2]).cuda().requires_grad_(True)
    Ks = sc["Ks"].repeat(2, 1, 1).cuda()
    leaves = {k: sc[k].cuda().requires_grad_(True) for k in ("means", "quats", "scales")}
    radii, m2, d, c, _ = ops.fully_fused_projection(leaves["means"], leaves["quats"], leaves["scales"], vms, Ks, 128, 96)
    (m2.sum() + d.sum() * 0.5 + (c ** 2).sum()).backward()
    tot = {k: v.grad.clone() for k, v in leaves.items()}
    gv = vms.grad.clone()
    for cam in range(2):
        l2 = {k: sc[k].cuda().requires_grad_(True) for k in ("means", "quats", "scales")}
        v1 = vms.detach()[cam:cam + 1].clone().requires_grad_(True)
        r1, a, b, cc, _ = ops.fully_fused_projection(l2["means"], l2["quats"], l2["scales"], v1, Ks[cam:cam + 1], 128, 96)
        assert torch.equal(r1[0], radii[cam]) and torch.equal(a[0], m2[cam])
        (a.sum() + b.sum() * 0.5 + (cc ** 2).sum()).backward()
        for k in tot:
            tot[k] -= l2[k].grad
        assert rel_err(gv[cam], v1.grad[0]) < 1e-4
    for k in tot:
        assert float(tot[k].abs().max()) < 1e-3 * float(leaves[k].grad.abs().max())


@pytest.mark.parametrize("seed,N,W,H,C", [(0, 2000, 256, 256, 1), (1, 20000, 640, 368, 1), (2, 3000, 200, 120, 3), (3, 10, 64, 64, 1),
                                          (4, 150000, 1920, 1080, 1), (5, 60000, 1600, 900, 1)])
def test_isect_bit_exact(ops, seed, N, W, H, C):
    sc = make_scene(N, W, H, seed=seed, spread=1.3)
    vms = sc["viewmats"].repeat(C, 1, 1)
    for c in range(C):
        vms[c, 0, 3] += 0.2 * c
    radii, m2, d, con, _ = ops.fully_fused_projection(sc["means"].cuda(), sc["quats"].cuda(), sc["scales"].cuda(), vms.cuda(),
                                                       sc["Ks"].repeat(C, 1, 1).cuda(), W, H)
    tw, th = (W + 15) // 16, (H + 15) // 16
    tpg, iids, fids, offs = ops.isect_tiles(m2, radii, d, 16, tw, th)
    # oracle on the SAME projected values (bit-exact integer contract), camera by camera
    keys, vals, cnts = [], [], []
    for c in range(C):
        t, k, v = G.isect_tiles(m2[c].cpu(), radii[c].cpu(), d[c].cpu(), 16, tw, th)
        cnts.append(t)
        keys.append(k + (c * tw * th << 32))
        vals.append(v.long() + c * N)
    keys, vals = torch.cat(keys), torch.cat(vals)
    assert torch.equal(tpg.cpu(), torch.stack(cnts))
    assert iids.numel() == keys.numel()
    assert torch.equal(iids.cpu(), keys)
    assert torch.equal(fids.cpu().long(), vals)
    ref_off = torch.searchsorted((keys >> 32).contiguous(), torch.arange(C * tw * th)).to(torch.int32).reshape(C, th, tw)
    assert torch.equal(offs.cpu(), ref_off)


@pytest.mark.parametrize("seed,N,W,H,ts", [(0, 30000, 1920, 1080, 64), (1, 5000, 640, 368, 32), (2, 20000, 1920, 1080, 128), (3, 4000, 333, 211, 48)])
def test_isect_bit_exact_other_tile_sizes(ops, seed, N, W, H, ts):
    """Lists for tiles of 32 ... 128 px (the coarse lists of the fused view; gsplat's tile_size argument), with and without culling."""
    sc = make_scene(N, W, H, seed=seed, spread=1.3)
    radii, m2, d, con, _ = ops.fully_fused_projection(sc["means"].cuda(), sc["quats"].cuda(), sc["scales"].cuda(), sc["viewmats"].cuda(),
                                                       sc["Ks"].cuda(), W, H)
    tw, th = (W + ts - 1) // ts, (H + ts - 1) // ts
    tpg, iids, fids, offs = ops.isect_tiles(m2, radii, d, ts, tw, th)
    t, k, v = G.isect_tiles(m2[0].cpu(), radii[0].cpu(), d[0].cpu(), ts, tw, th)
    assert torch.equal(tpg.cpu()[0], t) and torch.equal(iids.cpu(), k) and torch.equal(fids.cpu().long(), v.long())
    ref_off = torch.searchsorted((k >> 32).contiguous(), torch.arange(tw * th)).to(torch.int32).reshape(1, th, tw)
    assert torch.equal(offs.cpu(), ref_off)
    # without the 64-bit keys the offsets come out of the (single, wide) tile pass itself
    tpg2, _, fids2, offs2 = ops.isect_tiles(m2, radii, d, ts, tw, th, want_isect_ids=False)
    assert torch.equal(tpg2, tpg) and torch.equal(fids2, fids) and torch.equal(offs2.cpu(), ref_off)
    # culled lists: a subset of the pairs, same order
    op = sc["opacities"].cuda()[None].contiguous()
    tpg_c, _, fids_c, offs_c = ops.isect_tiles(m2, radii, d, ts, tw, th, want_isect_ids=False, conics=con, opacities=op)
    assert fids_c.numel() <= fids.numel() and bool((tpg_c <= tpg).all())
    full, cul = fids.cpu().tolist(), fids_c.cpu().tolist()
    o_f, o_c = offs.reshape(-1).cpu().tolist() + [len(full)], offs_c.reshape(-1).cpu().tolist() + [len(cul)]
    for tile in range(0, tw * th, max(1, tw * th // 40)):
        a, b = full[o_f[tile]:o_f[tile + 1]], cul[o_c[tile]:o_c[tile + 1]]
        it = iter(a)
        assert all(x in it for x in b), tile      # b is a subsequence of a


def test_isect_tiles_one_call_and_capacity(ops):
    """bds_isect_tiles == prepare + build when the caller's buffers are large enough; BDS_ECAPACITY (and M) otherwise."""
    import ctypes as C
    from bilateral_driving_amd import _lib as L
    N, W, H = 6000, 400, 240
    sc = make_scene(N, W, H, seed=5)
    radii, m2, d, con, _ = ops.fully_fused_projection(sc["means"].cuda(), sc["quats"].cuda(), sc["scales"].cuda(), sc["viewmats"].cuda(),
                                                      sc["Ks"].cuda(), W, H)
    op = sc["opacities"].cuda()[None].contiguous()
    tw, th = (W + 15) // 16, (H + 15) // 16
    tpg_ref, iids_ref, fids_ref, offs_ref = ops.isect_tiles(m2, radii, d, 16, tw, th, conics=con, opacities=op)
    M = fids_ref.numel()
    lib = L.lib()
    ws_bytes = lib.bds_isect_prepare_workspace_bytes(1, N)
    for cap, expect_ok in ((M, True), (M + 1000, True), (M - 1, False), (0, False)):
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
        ws2_bytes = lib.bds_isect_build_workspace_bytes(1, N, cap)
        ws2 = torch.empty(max(ws2_bytes, 16), dtype=torch.uint8, device="cuda")
        fids = torch.full((max(cap, 1),), -7, dtype=torch.int32, device="cuda")
        iids = torch.full((max(cap, 1),), -7, dtype=torch.int64, device="cuda")
        offs = torch.full((1, th, tw), -7, dtype=torch.int32, device="cuda")
        tpg = torch.empty(1, N, dtype=torch.int32, device="cuda")
        m, nv = C.c_int64(-1), C.c_int64(-1)
        rc = lib.bds_isect_tiles(1, N, L.ptr(m2.detach()), L.ptr(radii), L.ptr(d.detach()), L.ptr(con.detach()), L.ptr(op), 16, tw, th, L.ptr(tpg),
                                 L.ptr(ws), ws_bytes, L.ptr(ws2), ws2_bytes, cap, L.ptr(iids), L.ptr(fids), L.ptr(offs), C.byref(m),
                                 C.byref(nv), L.stream())
        torch.cuda.synchronize()
        assert m.value == M and nv.value == int((radii > 0).sum())
        assert torch.equal(tpg, tpg_ref)
        if expect_ok:
            assert rc == 0
            assert torch.equal(fids[:M], fids_ref) and torch.equal(iids[:M], iids_ref) and torch.equal(offs, offs_ref)
            assert bool((fids[M:] == -7).all())
        else:
            assert rc == L.ECAPACITY
            assert bool((fids == -7).all()) and bool((offs == -7).all())   # nothing built
            # the prepared workspace is intact: the two-call continuation gives the reference lists
            ws2b = torch.empty(max(lib.bds_isect_build_workspace_bytes(1, N, M), 16), dtype=torch.uint8, device="cuda")
            fids2 = torch.empty(M, dtype=torch.int32, device="cuda")
            vis_ids = torch.empty(nv.value, dtype=torch.int32, device="cuda")
            L.check(lib.bds_isect_build(1, N, M, -1 if cap == 0 else nv.value, L.ptr(m2.detach()), L.ptr(radii), L.ptr(d.detach()), L.ptr(con.detach()), L.ptr(op), 16, tw, th,
                                        L.ptr(ws), ws_bytes, L.ptr(ws2b), ws2b.numel(), None, L.ptr(fids2), L.ptr(offs), None, 0, L.stream()), "build")
            assert torch.equal(fids2, fids_ref) and torch.equal(offs, offs_ref)
            # ascending list of the visible entries (available in both modes)
            L.check(lib.bds_isect_build(1, N, M, nv.value, L.ptr(m2.detach()), L.ptr(radii), L.ptr(d.detach()), L.ptr(con.detach()), L.ptr(op), 16, tw, th,
                                        L.ptr(ws), ws_bytes, L.ptr(ws2b), ws2b.numel(), None, L.ptr(fids2), L.ptr(offs), L.ptr(vis_ids), 0,
                                        L.stream()), "build + visible ids")
            assert torch.equal(vis_ids.long(), (radii[0] > 0).nonzero().squeeze(1)) and torch.equal(fids2, fids_ref)
    # COMPACT lists (what the fused view composites from): list values = positions in the ascending visible list; same lists
    for packed in (1, 0):
        L.set_option(L.OPT_PACKED, packed)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
        m, nv = C.c_int64(-1), C.c_int64(-1)
        L.check(lib.bds_isect_prepare(1, N, L.ptr(m2.detach()), L.ptr(radii), L.ptr(d.detach()), L.ptr(con.detach()), L.ptr(op), 16, tw, th, L.ptr(tpg),
                                      L.ptr(ws), ws_bytes, C.byref(m), C.byref(nv), 1, L.stream()), "prepare compact")
        assert m.value == M and torch.equal(tpg, tpg_ref)
        ws2b = torch.empty(max(lib.bds_isect_build_workspace_bytes(1, N, M), 16), dtype=torch.uint8, device="cuda")
        pos, offs = torch.empty(M, dtype=torch.int32, device="cuda"), torch.empty(1, th, tw, dtype=torch.int32, device="cuda")
        vis_ids = torch.empty(nv.value, dtype=torch.int32, device="cuda")
        L.check(lib.bds_isect_build(1, N, M, nv.value, L.ptr(m2.detach()), L.ptr(radii), L.ptr(d.detach()), L.ptr(con.detach()), L.ptr(op), 16, tw, th,
                                    L.ptr(ws), ws_bytes, L.ptr(ws2b), ws2b.numel(), None, L.ptr(pos), L.ptr(offs), L.ptr(vis_ids), 1, L.stream()),
                "build compact")
        assert torch.equal(vis_ids.long(), (radii[0] > 0).nonzero().squeeze(1))
        assert int(pos.min()) >= 0 and int(pos.max()) < nv.value
        assert torch.equal(vis_ids[pos.long()], fids_ref) and torch.equal(offs, offs_ref)
    L.set_option(L.OPT_PACKED, 1)


@pytest.mark.parametrize("short,packed", [(0, 1), (1, 0), (0, 0)], ids=["generic_sort", "pair_lists", "generic_sort+pair_lists"])
def test_isect_large_input_fallback_paths(ops, short, packed):
    """The tile stage has two size regimes per step: up to 8.4 M (camera, Gaussian) entries the depth order uses the two-launch radix
    passes, beyond it the generic histogram / scan / scatter passes; packed 32-bit entries while the visible count fits the rank bits,
    (tile key, id) pairs beyond.  The test hooks force the large-input regime on small inputs: same lists as the oracle, and the
    culled lists equal the default regime's."""
    from bilateral_driving_amd import _lib as L
    try:
        for seed, N, W, H in ((0, 9000, 320, 200), (1, 70000, 640, 368)):
            sc = make_scene(N, W, H, seed=seed)
            radii, m2, d, con, _ = ops.fully_fused_projection(sc["means"].cuda(), sc["quats"].cuda(), sc["scales"].cuda(),
                                                              sc["viewmats"].cuda(), sc["Ks"].cuda(), W, H)
            tw, th = (W + 15) // 16, (H + 15) // 16
            op = sc["opacities"].cuda()[None].contiguous()
            ref = ops.isect_tiles(m2, radii, d, 16, tw, th, conics=con, opacities=op)       # default regime
            L.set_option(L.OPT_SHORT_SORT, short); L.set_option(L.OPT_PACKED, packed)
            tpg, iids, fids, offs = ops.isect_tiles(m2, radii, d, 16, tw, th)
            got = ops.isect_tiles(m2, radii, d, 16, tw, th, conics=con, opacities=op)
            L.set_option(L.OPT_SHORT_SORT, 1); L.set_option(L.OPT_PACKED, 1)
            tpg_o, iids_o, fids_o = G.isect_tiles(m2[0].cpu(), radii[0].cpu(), d[0].cpu(), 16, tw, th)
            assert torch.equal(tpg[0].cpu(), tpg_o) and torch.equal(iids.cpu(), iids_o) and torch.equal(fids.cpu(), fids_o)
            for a, b in zip(got, ref):
                assert torch.equal(a, b)
    finally:
        L.set_option(L.OPT_SHORT_SORT, 1)
        L.set_option(L.OPT_PACKED, 1)


def test_meta_isect_ids_lazy_equals_kernel(ops):
    import bilateral_driving_amd.rendering as R
    sc = make_scene(3000, 200, 120, seed=9)
    args = [sc[k].cuda() for k in ("means", "quats", "scales", "opacities", "colors")]
    _, _, meta = R.rasterization(*args, sc["viewmats"].cuda(), sc["Ks"].cuda(), 200, 120)
    assert dict.__getitem__(meta, "isect_ids") is None
    lazy = meta["isect_ids"]
    _, iids, fids, offs = ops.isect_tiles(meta["means2d"], meta["radii"], meta["depths"], 16, meta["tile_width"], meta["tile_height"],
                                          conics=meta["conics"], opacities=meta["opacities"])  # rasterization() culls by default
    assert torch.equal(lazy, iids) and torch.equal(fids, meta["flatten_ids"]) and torch.equal(offs, meta["isect_offsets"])


def test_isect_empty(ops):
    z = torch.zeros(1, 50, device="cuda")
    tpg, iids, fids, offs = ops.isect_tiles(torch.zeros(1, 50, 2, device="cuda"), torch.zeros(1, 50, dtype=torch.int32, device="cuda"), z, 16, 4, 3)
    assert iids.numel() == 0 and fids.numel() == 0 and int(offs.abs().sum()) == 0 and int(tpg.sum()) == 0


_ORACLE_CACHE = {}
_GRAD_LOG = []     # per (case, parameter): the HIP path's gradient errors next to the fp32 oracle's, both against the fp64 oracle


def teardown_module(module):
    """Leave the measured gradient errors where scripts/gpu_round.sh collects them (gpurun_out/ -> profiles/)."""
    import json
    import os
    if _GRAD_LOG:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        try:
            os.makedirs(d, exist_ok=True)
            json.dump(_GRAD_LOG, open(os.path.join(d, "gs_gradient_errors.json"), "w"), indent=1)
        except OSError:
            pass


def _oracle_render(key, sc, W, H, mode, bg, seed):
    """float64 oracle forward + backward of the fixed test loss, computed once per scene and shared by the
    kernel-variant parametrisations (it is the slow part of these tests)."""
    if key in _ORACLE_CACHE:
        return _ORACLE_CACHE[key]
    ref_in = {k: sc[k].double().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
    probes = []
    r_ref, a_ref, m_ref = G.rasterization(ref_in["means"], ref_in["quats"], ref_in["scales"], ref_in["opacities"], ref_in["colors"],
                                          sc["viewmats"].double(), sc["Ks"].double(), W, H, render_mode=mode,
                                          backgrounds=None if bg is None else bg.double(), return_unstable=True, absgrad_probes=probes)
    stable = ~m_ref["unstable"][0]
    g = torch.Generator().manual_seed(seed)
    wt = torch.randn(r_ref.shape, generator=g) * stable[None, ..., None]
    wa = torch.randn(a_ref.shape, generator=g) * stable[None, ..., None]
    ((r_ref * wt.double()).sum() + (a_ref * wa.double()).sum()).backward()
    # the SAME oracle in float32 on the same loss: what plain fp32 arithmetic of the reference's formulation loses against fp64 -- the
    # yardstick the HIP path's element-wise gradient error is held against below (norm-relative alone hides single elements)
    in32 = {k: sc[k].detach().clone().float().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
    r32, a32, m32 = G.rasterization(in32["means"], in32["quats"], in32["scales"], in32["opacities"], in32["colors"], sc["viewmats"].float(),
                                    sc["Ks"].float(), W, H, render_mode=mode, backgrounds=None if bg is None else bg.float())
    ((r32 * wt).sum() + (a32 * wa).sum()).backward()
    out = dict(r=r_ref.detach(), a=a_ref.detach(), radii=m_ref["radii"], stable=stable, wt=wt, wa=wa,
               absgrad=G.absgrad_from_probe(probes[0], sc["means"].shape[0]), means2d=m_ref["means2d"].detach(),
               grads={k: (None if v.grad is None else v.grad.clone()) for k, v in ref_in.items()},
               grads32={k: (None if v.grad is None else v.grad.clone()) for k, v in in32.items()},
               radii32_equal=bool(torch.equal(m32["radii"], m_ref["radii"])))
    _ORACLE_CACHE[key] = out
    return out


@pytest.mark.parametrize("seed,N,W,H,mode", [(0, 1000, 256, 256, "RGB+ED"), (1, 4000, 320, 200, "RGB"), (2, 600, 75, 50, "RGB+ED"),
                                              (3, 3000, 128, 128, "ED"), (4, 300, 64, 64, "RGB+D")])
def test_rasterization_end_to_end(ops, seed, N, W, H, mode):
    sc = make_scene(N, W, H, seed=seed)
    bg = torch.rand(1, 3, generator=torch.Generator().manual_seed(99)) if mode == "RGB" else None
    import bilateral_driving_amd.rendering as R
    ref = _oracle_render((seed, N, W, H, mode), sc, W, H, mode, bg, seed)
    gpu_in = {k: sc[k].cuda().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
    r, a, meta = R.rasterization(gpu_in["means"], gpu_in["quats"], gpu_in["scales"], gpu_in["opacities"], gpu_in["colors"],
                                 sc["viewmats"].cuda(), sc["Ks"].cuda(), W, H, packed=False, absgrad=True, render_mode=mode,
                                 backgrounds=None if bg is None else bg.cuda())
    stable, r_ref, a_ref = ref["stable"], ref["r"], ref["a"]
    assert stable.float().mean() > 0.995
    # a Gaussian whose fp32 cull/radius decision differs from fp64 would change whole tiles: require none here
    assert torch.equal(meta["radii"].cpu(), ref["radii"]), "pick another seed: fp32/fp64 radius decisions differ"
    rc, ac = r[0].cpu().double(), a[0].cpu().double()
    err = (rc - r_ref[0]).abs() / r_ref[0].abs().clamp(min=1.0)
    assert float(err[stable].max()) < 1e-4, float(err[stable].max())
    # ... and TRULY relative wherever the value is not tiny (the bound above is absolute for values below 1)
    sizable = stable[..., None] & (r_ref[0].abs() > 1e-2)
    relerr = ((rc - r_ref[0]).abs() / r_ref[0].abs().clamp(min=1e-2))[sizable]
    assert float(relerr.max()) < 2e-4, float(relerr.max())   # measured: <= 1.2e-4 (an absolute 2e-6 on a value of 0.015)
    assert float((ac - a_ref[0]).abs()[stable].max()) < 1e-4
    assert float(a_ref.mean()) > 0.3
    # gradients: loss restricted to stable pixels
    ((r * ref["wt"].cuda()).sum() + (a * ref["wa"].cuda()).sum()).backward()
    for k, gref in ref["grads"].items():
        if gref is None:  # e.g. colours in depth-only modes
            assert gpu_in[k].grad is None or float(gpu_in[k].grad.abs().max()) == 0.0
            continue
        rel, elem, elem99 = grad_errors(gpu_in[k].grad, gref)
        rel32, elem32, elem99_32 = grad_errors(ref["grads32"][k], gref) if ref["radii32_equal"] else (float("nan"),) * 3
        _GRAD_LOG.append(dict(case=f"seed{seed}_N{N}_{W}x{H}_{mode}", param=k, hip_norm_rel=rel, hip_elem_worst=elem, hip_elem_p99=elem99,
                              oracle_fp32_norm_rel=rel32, oracle_fp32_elem_worst=elem32, oracle_fp32_elem_p99=elem99_32))
        assert rel < 1e-3, (k, rel)
        # element-wise, next to the norm bound (entries above 1e-3 of the largest one), held against the SAME oracle run in float32 (what
        # plain fp32 arithmetic of the reference's formulation loses against fp64): the HIP path's worst element and 99th percentile
        # stay within 2x the fp32 oracle's (+ a floor for cases where that is ~1e-5) and under north_star's 1e-3 / 3e-4 outright.
        # Measured (profiles/r09_gs_gradient_errors.json): worst 9.3e-4 (fp32 oracle: 1.2e-3), 99th percentile <= 1.4e-4 (1.1e-4).
        # Before the backward started from the exact final transmittance (bds.h ``t_final``; it used 1 - alpha as gsplat does) the
        # dense cases sat at 4.0e-2 / 3.8e-3 -- 40x the fp32 oracle: not "fp32 cancellation", the rounding of alpha = 1 - T.
        assert elem < 2e-3 and elem99 < 3e-4, (k, elem, elem99)
        if ref["radii32_equal"]:
            assert elem <= 2.0 * elem32 + 1e-4 and elem99 <= 2.0 * elem99_32 + 2e-5, (k, elem, elem32, elem99, elem99_32)
    # absgrad (trainers/base.py:280-297 -> gaussians/vanilla.py:163-191 drive split / duplicate with it): the VALUE against the
    # oracle's sum over pixels of |dL/dmean2d through that pixel|, on the same tensor object the caller holds
    assert hasattr(meta["means2d"], "absgrad") and meta["means2d"].absgrad.shape == meta["means2d"].shape
    ag, ag_ref = meta["means2d"].absgrad[0].cpu().double(), ref["absgrad"]
    assert float(ag_ref.abs().max()) > 0
    assert float((ag - ag_ref).norm() / ag_ref.norm()) < 1e-3, float((ag - ag_ref).norm() / ag_ref.norm())
    big = ag_ref.abs() > 1e-3 * ag_ref.abs().max()       # element-wise where the value is not noise-sized
    assert float(((ag - ag_ref).abs() / ag_ref.abs().clamp(min=1e-30))[big].max()) < 5e-3
    assert meta["means2d"].grad is None  # not retained unless asked


def test_absgrad_and_retain_grad_contract(ops):
    """trainers/base.py:422-430 retain_grad() on meta['means2d']; :280-297 reads .absgrad / .grad."""
    import bilateral_driving_amd.rendering as R
    sc = make_scene(800, 128, 96, seed=11)
    p = {k: sc[k].cuda().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
    W = torch.tensor(128, device="cuda")
    Ht = torch.tensor(96, device="cuda")  # 0-d GPU tensors, as the reference passes them
    r, a, meta = R.rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], sc["viewmats"].cuda(),
                                 sc["Ks"].cuda(), W, Ht, packed=False, absgrad=True, sparse_grad=False,
                                 rasterize_mode="classic", render_mode="RGB+ED", near_plane=0.1, far_plane=1e10, radius_clip=0.0)
    assert r.shape == (1, 96, 128, 4) and a.shape == (1, 96, 128, 1)
    for k in ("means2d", "radii", "width", "height", "depths", "conics", "isect_offsets", "flatten_ids", "tile_size", "n_cameras"):
        assert k in meta
    meta["means2d"].retain_grad()
    (r.sum() + a.sum()).backward()
    g, ag = meta["means2d"].grad, meta["means2d"].absgrad
    assert g.shape == ag.shape == (1, 800, 2)
    assert bool((ag >= g.abs() - 1e-6 * ag.abs().max()).all())
    assert float(ag.abs().sum()) > 0
    # opacity mask path (base.py:397): masked-out Gaussians contribute nothing
    mask = (torch.arange(800, device="cuda") % 2 == 0)
    r2, a2, _ = R.rasterization(p["means"], p["quats"], p["scales"], p["opacities"] * mask, p["colors"], sc["viewmats"].cuda(),
                                sc["Ks"].cuda(), 128, 96, packed=False, render_mode="RGB+ED")
    r3, a3, _ = R.rasterization(p["means"][mask], p["quats"][mask], p["scales"][mask], p["opacities"][mask], p["colors"][mask],
                                sc["viewmats"].cuda(), sc["Ks"].cuda(), 128, 96, packed=False, render_mode="RGB+ED")
    assert rel_err(r2.cpu(), r3.cpu()) < 1e-5 and rel_err(a2.cpu(), a3.cpu()) < 1e-5


def test_permutation_and_determinism(ops):
    import bilateral_driving_amd.rendering as R
    sc = make_scene(3000, 192, 160, seed=21)
    args = [sc[k].cuda() for k in ("means", "quats", "scales", "opacities", "colors")]
    r1, a1, _ = R.rasterization(*args, sc["viewmats"].cuda(), sc["Ks"].cuda(), 192, 160, render_mode="RGB+ED")
    r1b, a1b, _ = R.rasterization(*args, sc["viewmats"].cuda(), sc["Ks"].cuda(), 192, 160, render_mode="RGB+ED")
    assert torch.equal(r1, r1b) and torch.equal(a1, a1b)  # forward is deterministic
    perm = torch.randperm(3000, device="cuda")
    r2, a2, _ = R.rasterization(*[x[perm] for x in args], sc["viewmats"].cuda(), sc["Ks"].cuda(), 192, 160, render_mode="RGB+ED")
    assert rel_err(r2.cpu(), r1.cpu()) < 1e-6  # order only changes tie-breaking of exactly equal depths


@pytest.mark.parametrize("seed,N,W,H", [(0, 3000, 256, 192), (1, 20000, 640, 368), (2, 400, 96, 64)])
def test_exact_tile_culling_is_invisible(ops, seed, N, W, H):
    """Dropping (tile, Gaussian) pairs in which no pixel reaches alpha >= 1/255 changes neither image nor
    gradients; the culled lists are sub-lists of gsplat's bounding-square lists."""
    sc = make_scene(N, W, H, seed=seed)
    g = torch.Generator().manual_seed(seed)
    res = {}
    for cull in (False, True):
        p = {k: sc[k].cuda().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
        radii, m2, d, con, _ = ops.fully_fused_projection(p["means"], p["quats"], p["scales"], sc["viewmats"].cuda(), sc["Ks"].cuda(), W, H)
        col = torch.cat([p["colors"], d[0][:, None]], -1)[None]
        op = p["opacities"][None]
        tw, th = (W + 15) // 16, (H + 15) // 16
        tpg, iids, fids, offs = ops.isect_tiles(m2, radii, d, 16, tw, th, conics=con if cull else None, opacities=op if cull else None)
        r, a = ops.rasterize_to_pixels(m2, con, col, op, W, H, 16, offs, fids, absgrad=True)
        wt = torch.randn(r.shape, generator=torch.Generator().manual_seed(seed)).cuda()
        ((r * wt).sum() + a.sum()).backward()
        res[cull] = dict(r=r.detach(), a=a.detach(), tpg=tpg, iids=iids, grads={k: v.grad.clone() for k, v in p.items()},
                         m2=m2.detach(), con=con.detach(), op=op.detach(), fids=fids, offs=offs)
    full, cul = res[False], res[True]
    assert torch.equal(full["r"], cul["r"]) and torch.equal(full["a"], cul["a"])  # forward is bit-identical
    assert bool((cul["tpg"] <= full["tpg"]).all())
    assert cul["iids"].numel() < 0.8 * full["iids"].numel()  # the lists really shrink
    # culled (key, Gaussian) pairs are a subset of the full pairs
    pf = torch.stack([full["iids"], full["fids"].long()], 1)
    pc = torch.stack([cul["iids"], cul["fids"].long()], 1)
    _, counts = torch.unique(torch.cat([pf, pc]), dim=0, return_counts=True)
    assert int(counts.max()) == 2 and int((counts == 2).sum()) == pc.shape[0]
    for k in full["grads"]:
        ref, got = full["grads"][k], cul["grads"][k]
        assert float((got - ref).norm() / ref.norm().clamp(min=1e-20)) < 2e-4, k  # atomics summation order differs
    # brute force (float64, CPU): every dropped pair has max alpha over its tile's pixel centres < 1/255
    tw = (W + 15) // 16
    fk = full["iids"].cpu()
    enc = lambda d: d["iids"].cpu() * (1 << 20) % (1 << 62) + d["fids"].cpu().long()  # (key, id) -> one int64 (collision-free enough)
    keep = torch.isin(enc(full), enc(cul))
    dropped = (~keep).nonzero()[:, 0]
    sel = dropped[torch.randperm(dropped.numel(), generator=g)[:3000]]
    gid = full["fids"].cpu().long()[sel]
    tile = (fk[sel] >> 32)
    m2, con, op = full["m2"][0].cpu().double(), full["con"][0].cpu().double(), full["op"][0].cpu().double()
    ys, xs = torch.meshgrid(torch.arange(16, dtype=torch.float64) + 0.5, torch.arange(16, dtype=torch.float64) + 0.5, indexing="ij")
    px = (tile % tw).double()[:, None, None] * 16 + xs
    py = (tile // tw).double()[:, None, None] * 16 + ys
    dx, dy = m2[gid, 0][:, None, None] - px, m2[gid, 1][:, None, None] - py
    sig = 0.5 * (con[gid, 0][:, None, None] * dx * dx + con[gid, 2][:, None, None] * dy * dy) + con[gid, 1][:, None, None] * dx * dy
    alpha = op[gid][:, None, None] * torch.exp(-sig)
    assert float(alpha.reshape(len(sel), -1).max(dim=1).values.max()) < 1.0 / 255.0


@pytest.mark.parametrize("seed,N,W,H,C", [(0, 4000, 320, 200, 1), (1, 8000, 480, 272, 1), (2, 1500, 100, 52, 3), (3, 30, 40, 24, 1)])
def test_backward_schedule_is_a_permutation_and_invisible(ops, seed, N, W, H, C):
    """bds_rasterize_bwd_schedule: every XCD range of tiles is permuted longest-first; gradients do not depend on it.
    Both launch orders are compared with the float64 oracle run on the SAME projected inputs and lists (two fp32 runs differ from
    each other by atomics summation noise, which says nothing about either)."""
    from bilateral_driving_amd import _lib as L
    sc = make_scene(N, W, H, seed=seed)
    if C > 1:  # extra cameras: the same pose shifted sideways
        vms = sc["viewmats"].repeat(C, 1, 1)
        vms[:, 0, 3] += torch.arange(C, dtype=vms.dtype) * 0.3
        sc["viewmats"], sc["Ks"] = vms, sc["Ks"].repeat(C, 1, 1)
    tw, th = (W + 15) // 16, (H + 15) // 16
    with torch.no_grad():
        radii, m2, d, con, _ = ops.fully_fused_projection(sc["means"].cuda(), sc["quats"].cuda(), sc["scales"].cuda(), sc["viewmats"].cuda(),
                                                          sc["Ks"].cuda(), W, H)
    Cn = m2.shape[0]
    col0 = sc["colors"].cuda()[None].expand(Cn, -1, -1).contiguous()
    op0 = sc["opacities"].cuda()[None].expand(Cn, -1).contiguous()
    _, _, fids, offs = ops.isect_tiles(m2, radii, d, 16, tw, th, conics=con, opacities=op0)
    # oracle (float64, CPU), camera by camera, on the kernels' own inputs and lists
    leaves64 = {k: v.detach().cpu().double().requires_grad_(True) for k, v in dict(m2=m2, con=con, col=col0, op=op0).items()}
    offs_c, fids_c = offs.cpu().long(), fids.cpu().long()
    M = fids_c.numel()
    gen = torch.Generator().manual_seed(seed)
    wts, loss64 = [], 0.0
    for c in range(Cn):
        lo = int(offs_c[c].reshape(-1)[0])
        hi = int(offs_c[c + 1].reshape(-1)[0]) if c + 1 < Cn else M
        r64, a64, _, unstable = G.rasterize_to_pixels(leaves64["m2"][c], leaves64["con"][c], leaves64["col"][c], leaves64["op"][c], W, H, 16,
                                                      (offs_c[c] - lo).to(torch.int32), fids_c[lo:hi] - c * N, return_unstable=True)
        wt = torch.randn(r64.shape, generator=gen) * (~unstable)[..., None]
        wa = (~unstable)[..., None].float()
        wts.append((wt, wa))
        loss64 = loss64 + (r64 * wt.double()).sum() + (a64 * wa.double()).sum()
    loss64.backward()
    wt_g = torch.stack([w[0] for w in wts]).cuda()
    wa_g = torch.stack([w[1] for w in wts]).cuda()
    for sched in (False, True):
        ops.set_bwd_schedule(sched)
        lv = {k: v.detach().clone().requires_grad_(True) for k, v in dict(m2=m2, con=con, col=col0, op=op0).items()}
        r, a = ops.rasterize_to_pixels(lv["m2"], lv["con"], lv["col"], lv["op"], W, H, 16, offs, fids, absgrad=True)
        ((r * wt_g).sum() + (a * wa_g).sum()).backward()
        for k in lv:
            ref, got = leaves64[k].grad, lv[k].grad.cpu().double()
            assert float(ref.abs().max()) > 0
            assert float((got - ref).norm() / ref.norm()) < 1e-3, (k, sched)     # north_star's gradient tolerance
    ops.set_bwd_schedule(True)
    col, op = col0, op0
    # the schedule itself
    last = torch.zeros(Cn, H, W, dtype=torch.int32, device="cuda")
    rr, aa = torch.empty(Cn, H, W, 3, device="cuda"), torch.empty(Cn, H, W, 1, device="cuda")
    M = fids.numel()
    rec = torch.empty(Cn * N, L.SPLAT_RECORD_FLOATS, device="cuda")
    L.check(L.lib().bds_splat_pack(Cn * N, 3, None, L.ptr(m2.detach()), L.ptr(con.detach()), L.ptr(col.detach()), L.ptr(op.detach()), None, L.ptr(rec),
                                   L.stream()), "pack")
    L.check(L.lib().bds_rasterize_fwd(Cn, Cn * N, M, 3, L.ptr(rec), None, W, H, 16, 16, tw, th, L.ptr(offs), L.ptr(fids), L.ptr(rr), L.ptr(aa),
                                      None, L.ptr(last), L.stream()), "fwd")
    order = ops.bwd_schedule(Cn, W, H, 16, offs, last)
    total = Cn * tw * th
    assert int(order[0]) == 1                         # [tag = sorted | order | work] (csrc/rasterize.hip: pick_item)
    o = order[1:1 + total].cpu().long()
    work = order[1 + total:1 + 2 * total].cpu().long()
    assert torch.equal(o.sort().values, torch.arange(total))
    Hp, Wp = th * 16, tw * 16
    lid = torch.zeros(Cn, Hp, Wp, dtype=torch.long); lid[:, :H, :W] = last.cpu().long()
    tmax = lid.reshape(Cn, th, 16, tw, 16).permute(0, 1, 3, 2, 4).reshape(total, 256).max(dim=1).values
    assert torch.equal(work, (tmax - offs.reshape(-1).cpu().long() + 1).clamp(min=0))
    per, rem = divmod(total, 8)
    first = 0
    for x in range(8):
        cnt = per + (1 if x < rem else 0)
        seg = o[first:first + cnt]
        assert bool(((seg >= first) & (seg < first + cnt)).all())          # tiles stay inside their XCD range
        w = work[seg]
        shift = 0
        while cnt and (int(work[first:first + cnt].max()) >> shift) >= 1024:
            shift += 1
        assert bool(((w >> shift)[:-1] >= (w >> shift)[1:]).all())           # longest first (bucket granularity)
        first += cnt


@pytest.mark.parametrize("mode,radius_clip", [("RGB+ED", 0.0), ("RGB", 4.0)])
def test_one_view_node_equals_operator_chain(ops, mode, radius_clip, monkeypatch):
    """rasterization() for one camera runs as ONE autograd node over compact lists (rendering._RasterizeView) -- the reference's two
    call patterns (trainers/base.py:393-408: RGB+ED; :811-826: RGB, radius_clip = 4) -- and must equal the chain of individual
    operators (BDS_API_FUSED=0): images, every input gradient incl. the camera pose, .absgrad and the retained .grad of
    meta["means2d"], and the lazily materialised gsplat lists."""
    import bilateral_driving_amd.rendering as R
    sc = make_scene(3000, 320, 200, seed=5)
    res = {}
    for fused in (False, True):
        monkeypatch.setattr(R, "_ONE_VIEW_NODE", fused)
        p = {k: sc[k].cuda().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
        vm = sc["viewmats"].cuda().requires_grad_(True)
        r, a, meta = R.rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, sc["Ks"].cuda(), 320, 200,
                                     packed=False, absgrad=True, render_mode=mode, radius_clip=radius_clip, near_plane=0.1)
        meta["means2d"].retain_grad()
        g = torch.Generator().manual_seed(3)
        wt, wa = torch.randn(r.shape, generator=g).cuda(), torch.randn(a.shape, generator=g).cuda()
        ((r * wt).sum() + (a * wa).sum() + 0.1 * (meta["means2d"] ** 2).sum() * 1e-4).backward()     # incl. a term ON meta["means2d"]
        res[fused] = (r.detach(), a.detach(), {k: v.grad.clone() for k, v in p.items()}, vm.grad.clone(), meta["means2d"].absgrad.clone(),
                      meta["means2d"].grad.clone(), meta["radii"].clone(), meta["flatten_ids"].clone(), meta["isect_offsets"].clone())
    a_, b_ = res[False], res[True]
    assert torch.equal(a_[6], b_[6]) and torch.equal(a_[7], b_[7]) and torch.equal(a_[8], b_[8])
    assert float((a_[0] - b_[0]).abs().max()) < 2e-6 and float((a_[1] - b_[1]).abs().max()) < 2e-6
    for k in a_[2]:
        assert rel_err(b_[2][k], a_[2][k]) < 2e-4, k       # (float atomics in the compositor backward: the order of the sums varies)
    assert rel_err(b_[3], a_[3]) < 2e-4 and rel_err(b_[4], a_[4]) < 2e-4
    # the retained gradient of meta["means2d"]: the compositor's part + the term sent into it
    assert rel_err(b_[5], a_[5]) < 2e-4
