"""-m gpu: the one-view / RGB+ED / asynchronous entry points of the fused step against the general forms they fold
together (which are themselves checked against the oracle in test_gpu_gs_parity.py / test_gpu_bilagrid_parity.py)."""
import ctypes as C
import math

import pytest
import torch

from tests.util import make_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    import bilateral_driving_amd.gs_ops as ops
    from bilateral_driving_amd import _lib as L
    L.lib()
    return ops, L


def _scene(N, W, H, seed):
    sc = make_scene(N, W, H, seed=seed)
    d = {k: v.cuda() for k, v in sc.items()}
    d["log_scales"] = d["scales"].log()
    d["logits"] = torch.logit(d["opacities"].clamp(1e-4, 1 - 1e-4))
    return d


@pytest.mark.parametrize("seed,N,W,H", [(0, 3000, 320, 200), (1, 257, 96, 64)])
def test_project_view_equals_general_form_with_activations(env, seed, N, W, H):
    ops, L = env
    lib, st = L.lib(), L.stream()
    s = _scene(N, W, H, seed)
    vm, K = s["viewmats"][0].contiguous(), s["Ks"][0].contiguous()
    scales, opac = torch.empty(N, 3, device="cuda"), torch.empty(N, device="cuda")
    radii = torch.empty(1, N, dtype=torch.int32, device="cuda")
    m2, dep, con = torch.empty(1, N, 2, device="cuda"), torch.empty(1, N, device="cuda"), torch.empty(1, N, 3, device="cuda")
    L.check(lib.bds_project_view_fwd(N, L.ptr(s["means"]), L.ptr(s["quats"]), L.ptr(s["log_scales"]), L.ptr(s["logits"]), L.ptr(vm), L.ptr(K),
                                     W, H, 0.3, 0.01, 1e10, 0.0, L.ptr(scales), L.ptr(opac), L.ptr(radii), L.ptr(m2), L.ptr(dep), L.ptr(con), st), "fwd")
    ref_scales, ref_opac = torch.exp(s["log_scales"]), torch.sigmoid(s["logits"])
    assert torch.allclose(scales, ref_scales, rtol=2e-6, atol=0) and torch.allclose(opac, ref_opac, rtol=2e-6, atol=1e-7)
    r2, rm2, rdep, rcon, _ = ops.fully_fused_projection(s["means"], s["quats"], scales, vm[None], K[None], W, H)
    assert torch.equal(radii, r2) and torch.equal(m2, rm2) and torch.equal(dep, rdep) and torch.equal(con, rcon)
    # backward (list-driven over the visible entries, gradient records as input): raw-parameter gradients == general backward
    # chained with the activation derivatives; rows of culled Gaussians are not touched
    g = torch.Generator().manual_seed(seed)
    vis = (radii[0] > 0)
    ids = vis.nonzero().squeeze(1).to(torch.int32)
    ids = ids[torch.randperm(ids.numel(), generator=g).cuda()].contiguous()      # any order (the fused view hands a depth order)
    n = ids.numel()
    v_rec = torch.randn(n, 16, generator=g).cuda()
    il = ids.long()
    v_m2, v_dep = torch.zeros(1, N, 2, device="cuda"), torch.zeros(1, N, device="cuda")
    v_con, v_op = torch.zeros(1, N, 3, device="cuda"), torch.zeros(N, device="cuda")
    v_m2[0, il] = v_rec[:, 7:9]; v_dep[0, il] = v_rec[:, 3]; v_con[0, il] = v_rec[:, 4:7]; v_op[il] = v_rec[:, 11]
    ref = [torch.empty(N, k, device="cuda") for k in (3, 4, 3)]
    ref_vm = torch.empty(1, 4, 4, device="cuda")
    L.check(lib.bds_project_bwd(1, N, L.ptr(s["means"]), L.ptr(s["quats"]), L.ptr(scales), L.ptr(vm), L.ptr(K), W, H, 0.3, L.ptr(radii),
                                L.ptr(con), None, L.ptr(v_m2), L.ptr(v_dep), L.ptr(v_con), None, L.ptr(ref[0]), L.ptr(ref[1]), L.ptr(ref[2]),
                                L.ptr(ref_vm), st), "ref bwd")
    for acc in (0, 1):
        out = [torch.full((N, k), 7.0, device="cuda") for k in (3, 4, 3)] + [torch.full((N,), 7.0, device="cuda")]
        slots = torch.zeros(L.POSE_GRAD_SLOTS, 4, 4, device="cuda")                # camera-pose gradient partials (trainers/base.py:328-329,399)
        g2d, ag2d = torch.full((N, 2), 7.0, device="cuda"), torch.full((N, 2), 7.0, device="cuda")
        L.check(lib.bds_project_view_bwd_list(n, L.ptr(ids), L.ptr(s["means"]), L.ptr(s["quats"]), L.ptr(scales), L.ptr(opac), L.ptr(vm), L.ptr(K),
                                              W, H, 0.3, L.ptr(v_rec), L.ptr(out[0]), L.ptr(out[1]), L.ptr(out[2]), L.ptr(out[3]), L.ptr(slots),
                                              L.ptr(g2d), L.ptr(ag2d), None, acc, st), "bwd list")
        base = 7.0 * acc
        # (two separately compiled kernels: fused multiply-adds differ in the last bits, and the projection vjp cancels)
        for a, b in ((out[0], ref[0]), (out[1], ref[1]), (out[2], ref[2] * scales)):
            assert float((a[il] - base - b[il]).abs().max()) <= 1e-3 * float(b.abs().max()) + 1e-5 * base
            assert float((a[il] - base - b[il]).norm()) <= 1e-4 * float(b.norm()) + 1e-4 * base
            assert bool((a[~vis] == 7.0).all())                                   # culled rows untouched
        assert torch.allclose(out[3][il] - base, (v_op * opac * (1 - opac))[il], rtol=1e-5, atol=1e-6 + 1e-6 * base)
        assert bool((out[3][~vis] == 7.0).all())
        assert torch.equal(g2d[il], v_rec[:, 7:9]) and torch.equal(ag2d[il], v_rec[:, 9:11]) and bool((g2d[~vis] == 7.0).all())
        v_vm = slots.sum(0)
        assert float(ref_vm[0, :3].abs().max()) > 0 and float(v_vm[3].abs().max()) == 0.0
        assert float((v_vm - ref_vm[0]).norm()) <= 1e-4 * float(ref_vm.norm())      # reduction order differs
    # without a pose gradient buffer nothing else changes
    out2 = [torch.zeros(N, k, device="cuda") for k in (3, 4, 3)] + [torch.zeros(N, device="cuda")]
    out3 = [torch.zeros(N, k, device="cuda") for k in (3, 4, 3)] + [torch.zeros(N, device="cuda")]
    slots = torch.zeros(L.POSE_GRAD_SLOTS, 4, 4, device="cuda")
    for o, sl in ((out2, None), (out3, slots)):
        L.check(lib.bds_project_view_bwd_list(n, L.ptr(ids), L.ptr(s["means"]), L.ptr(s["quats"]), L.ptr(scales), L.ptr(opac), L.ptr(vm), L.ptr(K),
                                              W, H, 0.3, L.ptr(v_rec), L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[2]), L.ptr(o[3]), L.ptr(sl), None, None,
                                              None, 0, st), "bwd list (pose on / off)")
    for a, b in zip(out2, out3):   # two template instantiations: fused multiply-adds differ in the last bits and the vjp cancels
        assert float((a - b).norm()) <= 1e-4 * float(b.norm())


@pytest.mark.parametrize("deg", [0, 3])
def test_sh_view_equals_general_form_with_glue(env, deg):
    ops, L = env
    lib, st = L.lib(), L.stream()
    N, W, H = 5000, 320, 200
    s = _scene(N, W, H, 2)
    radii, _, dep, _, _ = ops.fully_fused_projection(s["means"], s["quats"], s["scales"], s["viewmats"], s["Ks"], W, H)
    g = torch.Generator().manual_seed(3)
    sh = (torch.randn(N, 16, 3, generator=g) * 0.6).cuda()
    cam_pos = torch.linalg.inv(s["viewmats"][0])[:3, 3].contiguous()
    sh_rgb, colors = torch.empty(N, 3, device="cuda"), torch.empty(N, 4, device="cuda")
    L.check(lib.bds_sh_view_fwd(N, 16, deg, L.ptr(s["means"]), L.ptr(cam_pos), L.ptr(sh), L.ptr(radii), L.ptr(dep), L.ptr(sh_rgb),
                                L.ptr(colors), st), "fwd")
    vis = radii[0] > 0
    dirs = s["means"] - cam_pos
    ref = ops.spherical_harmonics(deg, dirs, sh, masks=vis)
    assert torch.allclose(sh_rgb, ref, rtol=1e-5, atol=2e-6)   # separately compiled kernels: last-bit fma differences
    assert torch.equal(colors[:, :3], (sh_rgb + 0.5).clamp(0, 1)) and torch.equal(colors[:, 3], dep[0])
    assert bool(((ref[vis] + 0.5 < 0) | (ref[vis] + 0.5 > 1)).any())   # the clamp is exercised
    # backward, list-driven: rows of the visible Gaussians from the colour part of their gradient records
    ids = vis.nonzero().squeeze(1).to(torch.int32)
    ids = ids[torch.randperm(ids.numel(), generator=g).cuda()].contiguous()
    n, il = ids.numel(), ids.long()
    v_rec = torch.randn(n, 16, generator=g).cuda()
    v_col = torch.zeros(N, 3, device="cuda")
    v_col[il] = v_rec[:, :3]
    x = sh_rgb + 0.5
    v_rgb = (v_col * ((x >= 0) & (x <= 1))).contiguous()
    ref_v = torch.empty(N, 16, 3, device="cuda")
    mask8 = vis.to(torch.uint8)
    L.check(lib.bds_sh_bwd(N, 16, deg, L.ptr(dirs.contiguous()), L.ptr(sh), L.ptr(mask8), L.ptr(v_rgb), L.ptr(ref_v), None, st), "ref bwd")
    for acc in (0, 1):
        v_sh = torch.full((N, 16, 3), 7.0, device="cuda")
        L.check(lib.bds_sh_view_bwd_list(n, L.ptr(ids), 16, deg, L.ptr(s["means"]), L.ptr(cam_pos), L.ptr(sh_rgb), 0, L.ptr(v_rec), L.ptr(v_sh),
                                         None, acc, st), "bwd list")
        assert torch.allclose(v_sh[il] - 7.0 * acc, ref_v[il], rtol=1e-5, atol=2e-6 + 1e-5 * acc)
        assert bool((v_sh[~vis] == 7.0).all())                                    # culled rows untouched
    # the same colours evaluated by the record pack (visible entries only, in list order) + the backward fed from its compact sh_rgb
    m2 = torch.rand(N, 2, device="cuda") * 100.0
    con = torch.rand(N, 3, device="cuda") + 0.1
    opac = torch.rand(N, device="cuda")
    rec = torch.full((n, L.SPLAT_RECORD_FLOATS), -1.0, device="cuda")
    sh_rgb_c = torch.empty(n, 3, device="cuda")
    L.check(lib.bds_splat_pack_sh(n, L.ptr(ids), 16, deg, L.ptr(s["means"]), L.ptr(cam_pos), L.ptr(sh), L.ptr(m2), L.ptr(con), L.ptr(dep),
                                  L.ptr(opac), L.ptr(radii), L.ptr(rec), L.ptr(sh_rgb_c), st), "pack sh")
    ref_rec = torch.empty(n, L.SPLAT_RECORD_FLOATS, device="cuda")
    L.check(lib.bds_splat_pack(n, 4, L.ptr(ids), L.ptr(m2), L.ptr(con), L.ptr(colors), L.ptr(opac), L.ptr(radii), L.ptr(ref_rec), st), "pack")
    assert torch.equal(rec.view(torch.int32)[:, 11], ref_rec.view(torch.int32)[:, 11])        # radius slot
    assert torch.equal(rec[:, :6], ref_rec[:, :6]) and torch.equal(rec[:, 9], ref_rec[:, 9])     # geometry, opacity, depth channel
    assert torch.allclose(rec[:, 6:9], ref_rec[:, 6:9], rtol=1e-6, atol=1e-6)                     # colour (separately compiled SH sums)
    assert torch.allclose(sh_rgb_c, sh_rgb[il], rtol=1e-6, atol=1e-6)
    v_sh2 = torch.zeros(N, 16, 3, device="cuda")
    L.check(lib.bds_sh_view_bwd_list(n, L.ptr(ids), 16, deg, L.ptr(s["means"]), L.ptr(cam_pos), L.ptr(sh_rgb_c), 1, L.ptr(v_rec), L.ptr(v_sh2),
                                     None, 0, st), "bwd list by rank")
    assert torch.allclose(v_sh2[il], ref_v[il], rtol=1e-5, atol=2e-6)
    # row-wise clear
    bufs = [torch.full((N, k), 3.0, device="cuda") for k in (3, 4, 3)] + [torch.full((N,), 3.0, device="cuda"), torch.full((N, 16, 3), 3.0, device="cuda")]
    L.check(lib.bds_view_grads_clear_list(n, L.ptr(ids), 16, *[L.ptr(b) for b in bufs], st), "clear list")
    for b in bufs:
        assert float(b[il].abs().max()) == 0.0 and bool((b[~vis] == 3.0).all())


@pytest.mark.parametrize("H,W,use_sky", [(57, 91, True), (120, 200, False)])
def test_bilagrid_ed_form_equals_rgb_form_plus_depth_normalise(env, H, W, use_sky):
    ops, L = env
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd.bilagrid import _levels_struct
    lib, st = L.lib(), L.stream()
    g = torch.Generator().manual_seed(5)
    render = (torch.rand(H, W, 4, generator=g) * 1.3).cuda()
    alpha = torch.rand(H, W, generator=g).cuda()
    alpha[0, :5] = 0.0   # alpha below the 1e-10 clamp
    sky = torch.rand(H, W, 3, generator=g).cuda() if use_sky else None
    grids = [x[1:2].contiguous().cuda() for x in Hn.make_grids(3, seed=4)]
    factors = Hn.FACTORS_3
    lv = _levels_struct(grids, None, factors)
    wsb = lib.bds_bilagrid_ms_workspace_bytes(3, lv, H, W)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    rgb, depth = torch.empty(H, W, 3, device="cuda"), torch.empty(H, W, device="cuda")
    L.check(lib.bds_bilagrid_ms_ed_fwd(3, lv, H, W, L.ptr(render), L.ptr(alpha), L.ptr(sky), L.ptr(ws), wsb, L.ptr(rgb), L.ptr(depth), st), "ed fwd")
    rgb3 = render[..., :3].contiguous()
    ws2 = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    ref = torch.empty(H, W, 3, device="cuda")
    L.check(lib.bds_bilagrid_ms_fwd(3, lv, H, W, L.ptr(rgb3), L.ptr(alpha), L.ptr(sky), L.ptr(ws2), wsb, L.ptr(ref), None, st), "fwd")
    assert torch.equal(rgb, ref)
    assert torch.equal(depth, render[..., 3] / alpha.clamp(min=1e-10))
    # backward
    v_out = torch.randn(H, W, 3, generator=g).cuda()
    v_depth, v_opac = torch.randn(H, W, generator=g).cuda(), torch.randn(H, W, generator=g).cuda()
    vg = [torch.zeros_like(x) for x in grids]
    lvb = _levels_struct(grids, vg, factors)
    v_render, v_alpha = torch.empty(H, W, 4, device="cuda"), torch.empty(H, W, device="cuda")
    v_sky = torch.empty(H, W, 3, device="cuda") if use_sky else None
    L.check(lib.bds_bilagrid_ms_ed_bwd(3, lvb, H, W, L.ptr(render), L.ptr(alpha), L.ptr(sky), L.ptr(ws), wsb, L.ptr(v_out), L.ptr(v_depth),
                                       L.ptr(v_opac), L.ptr(v_render), L.ptr(v_alpha), L.ptr(v_sky), st), "ed bwd")
    vg2 = [torch.zeros_like(x) for x in grids]
    lvb2 = _levels_struct(grids, vg2, factors)
    r_rgb, r_alpha = torch.empty(H, W, 3, device="cuda"), torch.zeros(H, W, device="cuda")
    r_sky = torch.empty(H, W, 3, device="cuda") if use_sky else None
    L.check(lib.bds_bilagrid_ms_bwd(3, lvb2, H, W, L.ptr(rgb3), L.ptr(alpha), L.ptr(sky), L.ptr(ws2), wsb, L.ptr(v_out), L.ptr(r_rgb),
                                    L.ptr(r_alpha) if use_sky else None, L.ptr(r_sky), st), "bwd")
    ac = alpha.clamp(min=1e-10)
    assert torch.equal(v_render[..., :3], r_rgb)
    assert torch.equal(v_render[..., 3], v_depth / ac)
    exp_alpha = (r_alpha if use_sky else 0.0) + v_opac - torch.where(alpha >= 1e-10, render[..., 3] * v_depth / (ac * ac), torch.zeros_like(ac))
    assert torch.allclose(v_alpha, exp_alpha, rtol=1e-5, atol=1e-6)
    if use_sky:
        assert torch.equal(v_sky, r_sky)
    for a, b in zip(vg, vg2):   # LDS float atomics inside a workgroup: summation order is not fixed
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())


def test_isect_prepare_async_reports_the_same_counts(env):
    ops, L = env
    lib, st = L.lib(), L.stream()
    N, W, H = 8000, 400, 240
    s = _scene(N, W, H, 6)
    radii, m2, d, con, _ = ops.fully_fused_projection(s["means"], s["quats"], s["scales"], s["viewmats"], s["Ks"], W, H)
    op = s["opacities"][None].contiguous()
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    tpg_ref, _, fids_ref, offs_ref = ops.isect_tiles(m2, radii, d, 16, tw, th, want_isect_ids=False, conics=con, opacities=op)
    wsb = lib.bds_isect_prepare_workspace_bytes(1, N)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    counts = torch.full((2,), -1, dtype=torch.int64).pin_memory()
    ev = torch.cuda.Event()
    ev.record()
    tpg = torch.empty(1, N, dtype=torch.int32, device="cuda")
    L.check(lib.bds_isect_prepare_async(1, N, L.ptr(m2), L.ptr(radii), L.ptr(d), L.ptr(con), L.ptr(op), 16, tw, th, L.ptr(tpg), L.ptr(ws), wsb,
                                        counts.data_ptr(), ev.cuda_event, 0, st), "prepare_async")
    ev.synchronize()
    M, nv = int(counts[0]), int(counts[1])
    assert M == fids_ref.numel() and nv == int((radii > 0).sum()) and torch.equal(tpg, tpg_ref)
    ws2b = lib.bds_isect_build_workspace_bytes(1, N, M)
    ws2 = torch.empty(max(ws2b, 16), dtype=torch.uint8, device="cuda")
    fids, offs = torch.empty(M, dtype=torch.int32, device="cuda"), torch.empty(1, th, tw, dtype=torch.int32, device="cuda")
    L.check(lib.bds_isect_build(1, N, M, nv, L.ptr(m2), L.ptr(radii), L.ptr(d), L.ptr(con), L.ptr(op), 16, tw, th, L.ptr(ws), wsb, L.ptr(ws2),
                                ws2b, None, L.ptr(fids), L.ptr(offs), None, 0, st), "build")
    assert torch.equal(fids, fids_ref) and torch.equal(offs, offs_ref)


def test_fused_view_with_nothing_on_screen(env):
    """Every Gaussian behind the camera: M = 0 through the whole fused step (image = sky through the transform, all
    parameter gradients exactly zero, grid gradients live through the sky)."""
    ops, L = env
    from bilateral_driving_amd import harness as Hn
    from bilateral_driving_amd.fused_view import fused_view
    W, H, N = 96, 64, 500
    cam = Hn.ring_cameras(W, H, device="cuda")[0]
    p = Hn.synthetic_scene(N, seed=1, device="cuda")
    with torch.no_grad():
        # move everything behind the first camera (it looks along +z of its own frame)
        R, t = cam.viewmat[:3, :3], cam.viewmat[:3, 3]
        pc = p["means"] @ R.T + t
        pc[:, 2] = -pc[:, 2].abs() - 1.0
        p["means"].copy_((pc - t) @ R)
    for v in p.values():
        v.requires_grad_(True)
    grids = [g.cuda().requires_grad_(True) for g in Hn.make_grids(6, seed=2)]
    sky = torch.rand(H, W, 3, device="cuda")
    for rep in range(2):   # the second call goes through the provisioned-capacity path
        out = fused_view(p, cam.viewmat, cam.K, W, H, grids, sky, Hn.FACTORS_3, img_idx=0, cam_pos=cam.cam_pos)
        assert out["info"]["n_isects"] == 0 and out["info"]["flatten_ids"].numel() == 0 and int((out["info"]["radii"] > 0).sum()) == 0
        assert float(out["opacity"].abs().max()) == 0.0 and float(out["depth"].abs().max()) == 0.0
        (out["rgb"] * torch.rand_like(out["rgb"])).sum().backward()
        for k, v in p.items():
            assert v.grad is not None and float(v.grad.abs().max()) == 0.0, k
        assert any(float(g.grad.abs().max()) > 0 for g in grids)
        for t_ in list(p.values()) + grids:
            t_.grad = None


@pytest.mark.parametrize("W,H,levels,factors", [(320, 192, None, None), (256, 160, [(8, 8, 4), (16, 16, 8)], [2, 1])])
def test_epilogue_deferred_to_the_compositor_equals_the_three_launch_backward(env, W, H, levels, factors):
    """The colour transform's backward with its last stage (guidance route, clamp / sky blend / expected-depth backward) formed per
    pixel inside the compositor's backward (bds_bilagrid_ms_ed_bwd_deferred + bds_rasterize_bwd_ms: the one-stream default of the
    fused view) == the three-launch form (bit 2 of bds_set_option(7, ..)): every gradient, with a loss that also reaches the depth
    and the opacity image (v_depth / v_alpha_in) and a learnable sky.  A pyramid with a full-resolution level included."""
    ops, L = env
    from bilateral_driving_amd import harness as Hn
    N = 5000
    cam = Hn.ring_cameras(W, H, yaws_deg=(30.0,), device="cuda")[0]
    cam.viewmat.requires_grad_(True)
    base = Hn.synthetic_scene(N, seed=12, device="cuda")
    base["means"] = base["means"] * torch.tensor([0.4, 0.4, 1.0], device="cuda")
    lv = Hn.LEVELS_3 if levels is None else levels
    fs = Hn.FACTORS_3 if factors is None else factors
    grids0 = Hn.make_grids(2, levels=lv, device="cuda")
    g = torch.Generator().manual_seed(4)
    sky0 = torch.rand(H, W, 3, generator=g).cuda()
    wr, wd, wo = torch.randn(H, W, 3, generator=g).cuda(), torch.randn(H, W, 1, generator=g).cuda(), torch.randn(H, W, 1, generator=g).cuda()
    res = {}
    for mode in (3 | 4, 3):     # three launches; deferred
        L.set_option(L.OPT_CELLS, mode)
        try:
            assert bool(L.lib().bds_get_option(L.OPT_CELLS) & 4) == (mode == 7)
            p = {k: v.clone().requires_grad_(True) for k, v in base.items()}
            grids = [x.clone().requires_grad_(True) for x in grids0]
            sky = sky0.clone().requires_grad_(True)
            cam.viewmat.grad = None
            out = Hn.render_view(p, cam, grids, 1, sky, factors=fs)
            ((out["rgb"] * wr).sum() + 0.05 * (out["depth"] * wd).sum() + (out["opacity"] * wo).sum()).backward()
            res[mode] = ({k: v.grad.clone() for k, v in p.items()}, [x.grad.clone() for x in grids], sky.grad.clone(), cam.viewmat.grad.clone(),
                         out["info"]["means2d"].absgrad.clone(), out["rgb"].detach().clone())
        finally:
            L.set_option(L.OPT_CELLS, 3)
    a, b = res[7], res[3]
    assert torch.equal(a[5], b[5])
    for k in a[0]:
        assert float((a[0][k] - b[0][k]).norm() / a[0][k].norm()) < 2e-4, k      # (float atomics in the compositor backward)
    for x, y in zip(a[1], b[1]):
        assert float((x - y).norm() / x.norm()) < 3e-5
    assert torch.allclose(a[2], b[2], rtol=1e-5, atol=1e-7)                        # sky gradient: the same per-pixel arithmetic
    assert float((a[3] - b[3]).norm() / a[3].norm()) < 2e-4 and float((a[4] - b[4]).norm() / a[4].norm()) < 2e-4
