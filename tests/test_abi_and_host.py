"""CPU-only checks of the boundary: libbds.so builds for gfx950, loads, and exports every symbol that
include/bds.h declares; the product refuses to run without a GPU (no CPU fallback); host-side logic."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from bilateral_driving_amd import build
    return build.build()


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "bds.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bds_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(libpath):
    names = _declared_symbols()
    assert len(names) >= 19
    h = ctypes.CDLL(libpath)
    for n in names:
        assert hasattr(h, n), f"{n} declared in include/bds.h but not exported by libbds.so"
    h.bds_abi_version.restype = ctypes.c_int
    from bilateral_driving_amd import _lib as _L
    assert h.bds_abi_version() == _L.ABI_VERSION == 2
    h.bds_strerror.restype = ctypes.c_char_p
    assert b"workspace" in h.bds_strerror(-2)


def test_binding_table_matches_header(libpath):
    from bilateral_driving_amd import _lib
    assert sorted(_lib.EXPORTS) == _declared_symbols()
    _lib.lib()  # sets argtypes for every symbol; raises if one is missing


def test_library_is_gfx950_only(libpath):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", libpath], capture_output=True, text=True).stdout
    blob = open(libpath, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_80", b"sm_90"):
        assert other not in blob, other


def test_argument_validation_without_gpu(libpath):
    """EINVAL paths return before any launch, so they can be exercised on a CPU-only box."""
    from bilateral_driving_amd import _lib
    h = _lib.lib()
    assert h.bds_sh_fwd(10, 16, 7, None, None, None, None, None) == -1        # degree > 3
    assert h.bds_sh_fwd(0, 16, 3, None, None, None, None, None) == 0          # empty input is fine
    assert h.bds_rasterize_fwd(1, 10, 0, 5, None, None, 64, 64, 16, 16, 4, 4, None, None, None, None, None, None, None) == -1    # CH = 5
    assert h.bds_rasterize_fwd(1, 10, 0, 3, None, None, 64, 64, 8, 8, 8, 8, None, None, None, None, None, None, None) == -1     # tile size 8
    assert h.bds_rasterize_fwd(1, 10, 0, 3, None, None, 64, 64, 16, 24, 4, 4, None, None, None, None, None, None, None) == -1   # list tile not a multiple of 16
    assert h.bds_splat_pack(0, 4, None, None, None, None, None, None, None, None) == 0 and h.bds_splat_pack(5, 2, None, None, None, None, None, None, None, None, None) == -1
    assert h.bds_sh_view_bwd_list(0, None, 16, 3, None, None, None, 0, None, None, None, 0, None) == 0
    assert h.bds_sh_view_bwd_list(4, None, 16, 3, None, None, None, 0, None, None, None, 0, None) == -1               # null list
    assert h.bds_splat_pack_sh(0, None, 16, 3, None, None, None, None, None, None, None, None, None, None, None) == 0
    assert h.bds_splat_pack_sh(4, None, 16, 3, None, None, None, None, None, None, None, None, None, None, None, None) == -1   # null list
    assert h.bds_splat_pack_sh(4, None, 15, 3, None, None, None, None, None, None, None, None, None, None, None, None) == -1   # K < 16 bases
    assert h.bds_isect_prepare_workspace_bytes(1, 1000) > 5 * 4000
    assert h.bds_isect_build_workspace_bytes(1, 1000, 50000) > 3 * 4 * 50000
    lv = (_lib.BdsLevel * 1)()
    lv[0].gx, lv[0].gy, lv[0].gl, lv[0].factor, lv[0].n_avg = 8, 8, 4, 2, 1
    assert h.bds_bilagrid_ms_workspace_bytes(1, lv, 64, 64) >= 2 * 32 * 32 * 48  # low-res maps + their gradients
    assert h.bds_bilagrid_ms_workspace_bytes(0, lv, 64, 64) == 0


def test_no_cpu_fallback():
    from bilateral_driving_amd import _lib
    import bilateral_driving_amd.gs_ops as ops
    import bilateral_driving_amd.bilagrid as B
    with pytest.raises(_lib.BdsError):
        ops.spherical_harmonics(3, torch.randn(4, 3), torch.randn(4, 16, 3))
    with pytest.raises(_lib.BdsError):
        ops.fully_fused_projection(torch.randn(4, 3), torch.randn(4, 4), torch.rand(4, 3), torch.eye(4)[None], torch.eye(3)[None], 8, 8)
    with pytest.raises(_lib.BdsError):
        B.bilagrid_transform(torch.rand(8, 8, 3), [torch.zeros(12, 1, 2, 2)], [1])
    with pytest.raises(_lib.BdsError):
        B.total_variation_loss(torch.zeros(1, 12, 2, 2, 2))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "bilateral_driving_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(dp, f)


def test_dropin_import_surfaces():
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "bilateral_driving_amd", "dropin") + os.pathsep + ROOT)
    code = ("from gsplat.rendering import rasterization\n"
            "from gsplat.cuda_legacy._wrapper import num_sh_bases\n"
            "from gsplat.cuda_legacy._torch_impl import quat_to_rotmat\n"
            "from gsplat.cuda._wrapper import spherical_harmonics\n"
            "from bilateral.lib_bilagrid import BilateralGrid, color_correct, slice, total_variation_loss, NeuralBilateralGrid, slice_feature\n"
            "import torch\n"
            "assert num_sh_bases(3) == 16\n"
            "R = quat_to_rotmat(torch.tensor([[2.0, 0, 0, 0]]))\n"
            "assert torch.allclose(R[0], torch.eye(3))\n"
            "g = BilateralGrid(3, 4, 5, 2)\n"
            "assert g.grids.shape == (3, 12, 2, 5, 4) and list(g.state_dict()) == ['grids', 'rgb2gray_weight']\n"
            "assert float(g.grids[1, 0].min()) == 1.0 and float(g.grids[1, 1].abs().max()) == 0.0 and float(g.grids[2, 5].min()) == 1.0\n"
            "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr


def test_quat_to_rotmat_matches_oracle():
    sys.path.insert(0, os.path.join(ROOT, "bilateral_driving_amd", "dropin"))
    try:
        from gsplat.cuda_legacy._torch_impl import quat_to_rotmat
    finally:
        sys.path.pop(0)
    from oracle import gs_oracle as G
    q = torch.randn(50, 4, dtype=torch.float64)
    assert (quat_to_rotmat(q) - G.quat_to_rotmat(q)).abs().max() < 1e-14


def test_saved_input_tensor_identity_for_absgrad():
    """The .absgrad contract relies on autograd handing back the SAME tensor object that was passed in."""
    class F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            ctx.save_for_backward(x)
            return x * 2

        @staticmethod
        def backward(ctx, g):
            (x,) = ctx.saved_tensors
            x.absgrad = g.abs()
            return g * 2
    a = torch.randn(5, requires_grad=True)
    mid = a * 1.0
    meta = {"means2d": mid}
    F.apply(mid).sum().backward()
    assert hasattr(meta["means2d"], "absgrad")


def test_scene_generator_shapes():
    from bilateral_driving_amd import harness as Hn
    p = Hn.synthetic_scene(1000, seed=0)
    assert p["means"].shape == (1000, 3) and p["sh"].shape == (1000, 16, 3) and p["quats"].shape == (1000, 4)
    r = p["means"][:, :2].norm(dim=-1)
    assert float(r.min()) >= 2.0 - 1e-4 and float(r.max()) <= 80.0 + 1e-3
    cams = Hn.ring_cameras(1920, 1080)
    assert len(cams) == 6
    for c in cams:
        R = c.viewmat[:3, :3]
        assert torch.allclose(R @ R.T, torch.eye(3), atol=1e-6) and abs(float(torch.linalg.det(R)) - 1) < 1e-6
    # camera 0 looks down +x: a point on +x projects to the principal point
    pc = cams[0].viewmat[:3, :3] @ torch.tensor([10.0, 0, 0]) + cams[0].viewmat[:3, 3]
    assert torch.allclose(pc, torch.tensor([0.0, 0.0, 10.0]), atol=1e-6)
    g = Hn.make_grids(4)
    assert [tuple(x.shape) for x in g] == [(4, 12, 1, 2, 2), (4, 12, 2, 4, 4), (4, 12, 4, 8, 8)]


@pytest.mark.skipif(not os.path.isdir("/root/reference/project/models"), reason="the reference tree is only mounted in the build container")
def test_reference_modules_import_against_the_dropin_packages():
    """The UNMODIFIED reference modules that sit on the hot path import with this repo's `gsplat` / `bilateral` packages in
    front (third-party packages the path never touches are stubbed): the import surface of SURVEY.md 8b resolves here."""
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1",
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "bilateral_driving_amd", "dropin"), ROOT, "/root/reference/project"]))
    code = ("import sys, types\n"
            "def stub(n, **a):\n"
            "    m = types.ModuleType(n); m.__dict__.update(a); sys.modules[n] = m\n"
            "stub('tensorly', set_backend=lambda *_: None)\n"
            "stub('pytorch3d'); stub('pytorch3d.ops', knn_points=None)\n"
            "stub('pytorch3d.transforms', matrix_to_quaternion=None, quaternion_to_matrix=None)\n"
            "stub('omegaconf', OmegaConf=type('OmegaConf', (), {}))\n"
            "import models.gaussians.basics as B, models.gaussians.vanilla as V, models.modules as M\n"
            "import bilateral_driving_amd.rendering as R, bilateral_driving_amd.gs_ops as O, bilateral_driving_amd.bilagrid as G\n"
            "assert B.rasterization is R.rasterization and B.spherical_harmonics is O.spherical_harmonics\n"
            "assert M.BilateralGrid is G.BilateralGrid and M.slice is G.slice and M.total_variation_loss is G.total_variation_loss\n"
            "m = M.MultiScaleBilateralAffineTransform.__init__\n"
            "import bilateral_driving_amd.envlight as E\n"
            "assert M.dr.texture.__module__ == 'nvdiffrast.torch' and M.dr.cubemap_sample is E.cubemap_sample\n"
            "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_neural_modules_refuse_cpu_tensors():
    """No CPU fallback behind the module mirrors either: the feature slice underneath raises."""
    import torch
    from bilateral_driving_amd import _lib as L
    from bilateral_driving_amd.modules import NeuralBilateralAffineTransform
    mod = NeuralBilateralAffineTransform("Affine", 2, 4, 4, 2, feature_dim=8, hidden_dim=16, device="cpu")
    with pytest.raises(L.BdsError):
        mod(torch.rand(5, 6, 3), {"img_idx": 0})


def test_sky_and_colour_correct_refuse_cpu_tensors():
    import torch
    from bilateral_driving_amd import _lib as L
    from bilateral_driving_amd.colorcorrect import color_correct
    from bilateral_driving_amd.envlight import EnvLight, cubemap_sample
    with pytest.raises(L.BdsError):
        color_correct(torch.rand(4, 4, 3), torch.rand(4, 4, 3))
    with pytest.raises(L.BdsError):
        cubemap_sample(torch.rand(6, 4, 4, 3), torch.rand(5, 3))
    sky = EnvLight("Sky", resolution=4, device="cpu")
    assert list(sky.state_dict()) == ["base"] and sky.base.shape == (6, 4, 4, 3)
    with pytest.raises(L.BdsError):
        sky({"viewdirs": torch.rand(3, 5, 3)})
    with pytest.raises(ValueError):
        color_correct(torch.rand(4, 4, 3), torch.rand(4, 4, 4))


def test_dropin_lib_bilagrid_exports_the_product_functions():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "bilateral_driving_amd", "dropin"), ROOT]))
    code = ("import bilateral.lib_bilagrid as LB, bilateral_driving_amd.colorcorrect as C, bilateral_driving_amd.bilagrid as G\n"
            "assert LB.color_correct is C.color_correct and LB.slice_feature is G.slice_feature and LB.NeuralBilateralGrid is G.NeuralBilateralGrid\n"
            "import nvdiffrast.torch as dr, bilateral_driving_amd.envlight as E\n"
            "assert dr.cubemap_sample is E.cubemap_sample\n"
            "try:\n    LB.slice4d()\nexcept NotImplementedError:\n    print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_meta_dict_materialises_its_lazy_lists_through_every_accessor(monkeypatch):
    """rendering._Meta (the one-view fast path of rasterization()): gsplat's per-tile lists are built on first access -- through [],
    .get(), .items(), .values(), dict(meta) and ** alike (round-4 advisor finding: only [] did)."""
    import torch
    from bilateral_driving_amd import rendering as R
    calls = []

    def fake_isect_tiles(means2d, radii, depths, tile_size, tw, th, want_isect_ids=False, conics=None, opacities=None):
        calls.append(1)
        return torch.tensor([1]), None, torch.tensor([7, 8]), torch.tensor([0])
    monkeypatch.setattr(R, "isect_tiles", fake_isect_tiles)

    def fresh():
        return R._Meta({"means2d": 0, "radii": 0, "depths": torch.zeros(1, 9), "conics": 0, "opacities": torch.zeros(1, 9), "tile_size": 16,
                        "tile_width": 1, "tile_height": 1, "tiles_per_gauss": None, "isect_ids": None, "flatten_ids": None,
                        "isect_offsets": None, "_cull": False})
    for read in (lambda m: m["flatten_ids"], lambda m: m.get("flatten_ids"), lambda m: dict(m.items())["flatten_ids"],
                 lambda m: dict(m)["flatten_ids"], lambda m: (lambda **kw: kw["flatten_ids"])(**m),
                 lambda m: list(m.values())[list(m.keys()).index("flatten_ids")]):
        got = read(fresh())
        assert got is not None and got.tolist() == [7, 8]
    assert fresh().get("no_such_key", 5) == 5 and len(calls) >= 6
