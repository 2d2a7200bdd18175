"""The kernels' __host__ __device__ per-element formulas (csrc/gs_math.h, csrc/bilagrid_math.h),
run on the CPU through tests/hostmath_shim.hip, against the oracle.  CPU only: this de-risks the
formulas before GPU time is spent; the product never calls the shim."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import bilagrid_oracle as BO
from oracle import gs_oracle as G
from tests.util import fptr, hostmath, make_scene, rel_err

F64 = torch.float64


@pytest.fixture(scope="module")
def hm():
    return hostmath()


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_bases_and_vjp(hm, deg):
    g = torch.Generator().manual_seed(deg)
    n = 300
    dirs = torch.randn(n, 3, generator=g, dtype=F64) * 3
    out = np.zeros((n, 16), np.float32)
    d32 = dirs.float().numpy().copy()
    hm.hm_sh_bases(n, deg, fptr(d32), fptr(out))
    nb = (deg + 1) ** 2
    ref = G.sh_bases(deg, dirs)
    assert np.abs(out[:, :nb] - ref.numpy()).max() < 2e-6
    # vjp w.r.t. the (un-normalised) direction
    gk = torch.randn(n, 16, generator=g, dtype=F64)
    gk[:, nb:] = 0
    dd = dirs.clone().requires_grad_(True)
    if deg > 0:
        (G.sh_bases(deg, dd) * gk[:, :nb]).sum().backward()
    else:
        dd.grad = torch.zeros_like(dd)  # degree 0 does not depend on the direction
    vd = np.zeros((n, 3), np.float32)
    g32 = gk.float().numpy().copy()
    hm.hm_sh_bases_vjp(n, deg, fptr(d32), fptr(g32), fptr(vd))
    assert np.abs(vd - dd.grad.numpy()).max() < 1e-5 * max(1.0, float(dd.grad.abs().max()))


@pytest.mark.parametrize("seed,W,H", [(0, 64, 48), (1, 200, 120), (2, 33, 57)])
def test_projection_fwd_bwd(hm, seed, W, H):
    sc = make_scene(400, W, H, seed=seed, dtype=F64, spread=1.6)  # spread>1: many off-screen / clamped-FOV cases
    n = 400
    vm, K = sc["viewmats"][0], sc["Ks"][0]
    leaves = {k: sc[k].clone().requires_grad_(True) for k in ("means", "quats", "scales")}
    vmr = vm.clone().requires_grad_(True)
    radii, m2, dep, con, comp = G.project(leaves["means"], leaves["quats"], leaves["scales"], vmr, K, W, H,
                                          calc_compensations=True)
    a32 = {k: sc[k].float().numpy().copy() for k in ("means", "quats", "scales")}
    vm32, K32 = vm.float().numpy().copy(), K.float().numpy().copy()
    o_r = np.zeros(n, np.int32); o_m = np.zeros((n, 2), np.float32); o_d = np.zeros(n, np.float32)
    o_c = np.zeros((n, 3), np.float32); o_p = np.zeros(n, np.float32)
    hm.hm_project_fwd(n, fptr(a32["means"]), fptr(a32["quats"]), fptr(a32["scales"]), fptr(vm32), fptr(K32), W, H,
                      C.c_float(0.3), C.c_float(0.01), C.c_float(1e10), C.c_float(0.0), fptr(o_r), fptr(o_m), fptr(o_d),
                      fptr(o_c), fptr(o_p))
    same = torch.from_numpy(o_r) == radii
    # fp32 vs fp64 may disagree on a ceil()/cull decision for a handful of Gaussians
    assert same.float().mean() > 0.98
    vis = (radii > 0) & same
    assert int(vis.sum()) > 50 and int((radii == 0).sum()) > 20
    assert rel_err(torch.from_numpy(o_m)[vis], m2[vis]) < 1e-5
    assert rel_err(torch.from_numpy(o_d)[vis], dep[vis]) < 1e-6
    assert ((torch.from_numpy(o_c)[vis].double() - con[vis]).abs() / con[vis].abs().clamp(min=1e-3)).max() < 2e-4
    assert rel_err(torch.from_numpy(o_p)[vis], comp[vis]) < 1e-4
    # backward
    g = torch.Generator().manual_seed(10 + seed)
    v_m2 = torch.randn(n, 2, generator=g, dtype=F64)
    v_d = torch.randn(n, generator=g, dtype=F64)
    v_c = torch.randn(n, 3, generator=g, dtype=F64)
    ((m2 * v_m2).sum() + (dep * v_d).sum() + (con * v_c).sum()).backward()
    rr = radii.numpy().astype(np.int32).copy()
    vm2, vd, vc = v_m2.float().numpy().copy(), v_d.float().numpy().copy(), v_c.float().numpy().copy()
    g_m = np.zeros((n, 3), np.float32); g_q = np.zeros((n, 4), np.float32); g_s = np.zeros((n, 3), np.float32)
    g_R = np.zeros(9, np.float32); g_t = np.zeros(3, np.float32)
    hm.hm_project_bwd(n, fptr(a32["means"]), fptr(a32["quats"]), fptr(a32["scales"]), fptr(vm32), fptr(K32), W, H,
                      C.c_float(0.3), fptr(rr), fptr(vm2), fptr(vd), fptr(vc), fptr(g_m), fptr(g_q), fptr(g_s), fptr(g_R),
                      fptr(g_t))
    for got, ref in ((g_m, leaves["means"].grad), (g_q, leaves["quats"].grad), (g_s, leaves["scales"].grad)):
        got = torch.from_numpy(got).double()
        # per-Gaussian relative error against the row norm (fp32 kernel maths vs fp64 autograd)
        err = (got - ref).norm(dim=-1) / ref.norm(dim=-1).clamp(min=1e-6 * float(ref.norm(dim=-1).max()))
        assert float(err.max()) < 2e-3, float(err.max())
        assert float((got - ref).norm() / ref.norm()) < 1e-4
    gR = vmr.grad[:3, :3].reshape(-1)
    gt = vmr.grad[:3, 3]
    assert float((torch.from_numpy(g_R).double() - gR).norm() / gR.norm()) < 1e-4
    assert float((torch.from_numpy(g_t).double() - gt).norm() / gt.norm()) < 1e-4


def test_tile_rect_matches_oracle(hm):
    sc = make_scene(500, 200, 120, seed=4, dtype=torch.float32, spread=1.5)
    radii, m2, dep, con, _ = G.project(sc["means"], sc["quats"], sc["scales"], sc["viewmats"][0], sc["Ks"][0], 200, 120)
    tw, th = 13, 8
    x0, y0, x1, y1 = G.tile_rect(m2, radii, 16, tw, th)
    rect = np.zeros((500, 4), np.int32)
    hm.hm_tile_rect(500, fptr(m2.numpy().copy()), fptr(radii.numpy().copy()), 16, tw, th, fptr(rect))
    vis = (radii > 0).numpy()
    ref = torch.stack([x0, y0, x1, y1], -1).numpy()
    assert np.array_equal(rect[vis], ref[vis])


@pytest.mark.parametrize("out_size,in_size", [(4, 17), (5, 23), (17, 4), (23, 5), (57, 14), (64, 16), (16, 64), (7, 7),
                                               (1080, 270), (270, 1080), (1, 3)])
def test_resample_taps_match_oracle(hm, out_size, in_size):
    i0 = np.zeros(out_size, np.int32); i1 = np.zeros(out_size, np.int32); w1 = np.zeros(out_size, np.float32)
    hm.hm_resample_taps(out_size, in_size, fptr(i0), fptr(i1), fptr(w1))
    r0, r1, rw = BO._src_index(out_size, in_size, torch.float32, "cpu")
    assert np.array_equal(i0, r0.numpy()) and np.array_equal(i1, r1.numpy())
    assert np.array_equal(w1, rw.numpy())  # bit-exact float32


@pytest.mark.parametrize("n", [1, 2, 3, 5, 14, 23, 270, 480, 1080, 1920])
def test_linspace_is_torch_linspace(hm, n):
    out = np.zeros(n, np.float32)
    hm.hm_linspace01(n, fptr(out))
    ref = torch.linspace(0, 1.0, n).numpy()
    # torch's CPU kernel mixes fused (vector body) and unfused (scalar tail) arithmetic: allow 1 ulp
    assert np.abs(out - ref).max() <= 6e-8
    assert (out != ref).mean() <= 0.1


@pytest.mark.parametrize("gs", [(2, 2, 1), (4, 4, 2), (8, 8, 4), (16, 16, 8), (5, 7, 3)])
def test_slice_matches_oracle(hm, gs):
    gx, gy, gl = gs
    g = torch.Generator().manual_seed(gx)
    P = 1000
    grid = torch.randn(12, gl, gy, gx, generator=g, dtype=F64)
    xy = torch.rand(P, 2, generator=g, dtype=F64)
    xy[:20] = torch.randint(0, 2, (20, 2), generator=g).double()  # exact corners
    rgb = torch.rand(P, 3, generator=g, dtype=F64) * 1.3 - 0.1
    rgb_r = rgb.clone().requires_grad_(True)
    aff = BO.slice_grid(grid, xy[:, 0], xy[:, 1], BO.rgb2gray(rgb_r))
    out = np.zeros((P, 12), np.float32); dg = np.zeros((P, 12), np.float32)
    hm.hm_slice(P, fptr(grid.float().numpy().copy()), gx, gy, gl, fptr(xy.float().numpy().copy()),
                fptr(rgb.float().numpy().copy()), fptr(out), fptr(dg))
    assert np.abs(out - aff.detach().numpy()).max() < 2e-5
    # d(aff_c)/d(gray) through autograd, channel by channel for a subset
    for ch in (0, 5, 11):
        gr, = torch.autograd.grad(aff[:, ch].sum(), rgb_r, retain_graph=True)
        dgray = gr[:, 0] / 0.299
        assert np.abs(dg[:, ch] - dgray.numpy()).max() < 2e-4 * max(1.0, float(dgray.abs().max()))


def test_tile_culling_is_conservative_and_tight(hm):
    """Brute force over pixel centres (float64): every tile of gsplat's bounding square that holds a pixel
    with alpha >= 1/255 is kept by the row-span culling; few kept tiles are useless."""
    W, H, ts = 320, 208, 16
    tw, th = W // ts, H // ts
    sc = make_scene(600, W, H, seed=12, dtype=torch.float32, spread=1.2)
    radii, m2, dep, con, _ = G.project(sc["means"], sc["quats"], sc["scales"], sc["viewmats"][0], sc["Ks"][0], W, H)
    op = sc["opacities"].clone()
    op[:20] = 0.003  # below 1/255: never visible
    op[20:40] = 1.0
    ys, xs = torch.meshgrid(torch.arange(H, dtype=F64) + 0.5, torch.arange(W, dtype=F64) + 0.5, indexing="ij")
    x0, y0, x1, y1 = G.tile_rect(m2, radii, ts, tw, th)
    kept_tot = needed_tot = box_tot = 0
    for g in torch.nonzero(radii > 0)[:, 0].tolist():
        a, b, c = [float(v) for v in con[g]]
        dx, dy = float(m2[g, 0]) - xs, float(m2[g, 1]) - ys
        alpha = float(op[g]) * torch.exp(-(0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy))
        hit = (alpha >= 1.0 / 255.0).reshape(th, ts, tw, ts).any(3).any(1)
        box = torch.zeros(th, tw, dtype=torch.bool)
        box[y0[g]:y1[g], x0[g]:x1[g]] = True
        needed = hit & box
        mask = np.zeros(tw * th, np.uint8)
        n = hm.hm_culled_tiles(C.c_float(float(m2[g, 0])), C.c_float(float(m2[g, 1])), int(radii[g]), C.c_float(a), C.c_float(b),
                               C.c_float(c), C.c_float(float(op[g])), ts, tw, th, fptr(mask))
        kept = torch.from_numpy(mask.reshape(th, tw).astype(bool))
        assert n == int(kept.sum())
        assert bool((kept | ~needed).all()), f"gaussian {g}: a needed tile was culled"
        assert bool((box | ~kept).all()), "kept tiles must lie inside gsplat's bounding square"
        kept_tot += int(kept.sum()); needed_tot += int(needed.sum()); box_tot += int(box.sum())
    assert needed_tot > 500
    assert kept_tot <= 1.25 * needed_tot + 50, (kept_tot, needed_tot)  # tight
    assert kept_tot < 0.7 * box_tot, (kept_tot, box_tot)               # and a real reduction vs the bounding squares


@pytest.mark.parametrize("n,g", [(1920, 16), (1080, 16), (1600, 8), (301, 7), (33, 2), (64, 1), (17, 16), (2, 16), (1, 5), (4096, 16)])
def test_axis_cell_is_slice_cell_with_the_last_node_folded_into_the_last_cell(hm, n, g):
    """csrc/bilagrid_math.h::axis_cell (the cell-aligned tiles of the fused image transform): every index of a linspace axis lies in one
    of the g - 1 cells, the cells are contiguous index ranges in ascending order, and the two node weights equal slice_cell's."""
    i0 = np.zeros(n, np.int32); f = np.zeros(n, np.float32); x0 = np.zeros(n, np.int32); x1 = np.zeros(n, np.int32); fx = np.zeros(n, np.float32)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    hm.hm_axis_cells(n, g, ip(i0), fptr(f), ip(x0), ip(x1), fptr(fx))
    assert i0.min() >= 0 and i0.max() <= max(g - 2, 0)
    assert np.all(np.diff(i0) >= 0)                                   # cells are contiguous runs
    if n >= 4 * g and g > 1:
        assert len(np.unique(i0)) == g - 1                            # ... and all of them occur
    # node weights: w[node] over the g nodes, per index
    wa = np.zeros((n, g), np.float64); wb = np.zeros((n, g), np.float64)
    idx = np.arange(n)
    np.add.at(wa, (idx, i0), 1.0 - f.astype(np.float64)); np.add.at(wa, (idx, np.minimum(i0 + 1, g - 1)), f.astype(np.float64))
    np.add.at(wb, (idx, x0), 1.0 - fx.astype(np.float64)); np.add.at(wb, (idx, x1), fx.astype(np.float64))
    np.testing.assert_array_equal(wa, wb)
    if g > 1 and n > 1:
        assert i0[-1] == g - 2 and f[-1] == 1.0 and x0[-1] == g - 1 and fx[-1] == 0.0     # the folded case really occurs: the last index


def test_block_bound_never_rejects_a_box_that_holds_a_visible_gaussian(hm):
    """csrc/gs_math.h box_may_be_visible (the bound behind bds_project_view_*_fwd_blocks) against project_one itself: whenever a box is
    rejected, NO Gaussian with its centre inside the box and scales up to the box's maximum comes out of the projection with a
    radius -- for boxes in front of, beside, behind and around the camera, near-plane straddlers and large splats included.  And the
    bound does reject: most boxes well outside the frustum go."""
    import ctypes
    rng = np.random.default_rng(0)
    W, H = 640, 360
    f = 0.5 * W / np.tan(np.radians(35.0))
    K = np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], np.float32)
    hm.hm_box_may_be_visible.restype = ctypes.c_int
    n_rejected = n_boxes = n_visible_in_kept = 0
    for trial in range(400):
        yaw, pitch = rng.uniform(-np.pi, np.pi), rng.uniform(-0.3, 0.3)
        cy_, sy_, cp_, sp_ = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
        R = (np.array([[1, 0, 0], [0, cp_, -sp_], [0, sp_, cp_]]) @ np.array([[cy_, 0, sy_], [0, 1, 0], [-sy_, 0, cy_]])).astype(np.float32)
        t = rng.uniform(-3, 3, 3).astype(np.float32)
        vm = np.eye(4, dtype=np.float32); vm[:3, :3] = R; vm[:3, 3] = t
        centre = rng.uniform(-40, 40, 3).astype(np.float32)
        if trial % 5 == 0:
            centre = (R.T @ (np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(-0.5, 1.0)], np.float32) - t)).astype(np.float32)  # around the camera
        half = np.exp(rng.uniform(np.log(0.05), np.log(8.0), 3)).astype(np.float32)
        lo, hi = centre - half, centre + half
        smax = float(np.exp(rng.uniform(np.log(0.005), np.log(3.0))))
        n = 600
        means = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
        means[:8] = np.array([[lo[0] if k & 1 == 0 else hi[0], lo[1] if k & 2 == 0 else hi[1], lo[2] if k & 4 == 0 else hi[2]] for k in range(8)], np.float32)
        scales = (smax * np.exp(rng.uniform(np.log(0.01), 0.0, (n, 3)))).astype(np.float32)
        scales[::7] = smax
        quats = rng.standard_normal((n, 4)).astype(np.float32)
        radii = np.zeros(n, np.int32); m2 = np.zeros((n, 2), np.float32); d = np.zeros(n, np.float32); c = np.zeros((n, 3), np.float32); cp = np.zeros(n, np.float32)
        hm.hm_project_fwd(n, fptr(means), fptr(quats), fptr(scales), fptr(vm), fptr(K), W, H, ctypes.c_float(0.3), ctypes.c_float(0.1),
                          ctypes.c_float(1e10), ctypes.c_float(0.0), fptr(radii), fptr(m2), fptr(d), fptr(c), fptr(cp))
        keep = hm.hm_box_may_be_visible(fptr(lo.astype(np.float32)), fptr(hi.astype(np.float32)), ctypes.c_float(np.float32(smax)), fptr(vm), fptr(K), W, H,
                                        ctypes.c_float(0.3), ctypes.c_float(0.1), ctypes.c_float(1e10))
        n_boxes += 1
        if not keep:
            n_rejected += 1
            assert int((radii > 0).sum()) == 0, (trial, int((radii > 0).sum()), lo, hi, smax)
        else:
            n_visible_in_kept += int((radii > 0).any())
    assert n_rejected > 0.45 * n_boxes and n_visible_in_kept > 0.1 * n_boxes, (n_rejected, n_visible_in_kept, n_boxes)
