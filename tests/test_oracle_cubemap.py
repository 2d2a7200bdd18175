"""Cube-map sky oracle (oracle/cubemap_oracle.py; parity UNPINNED -- nvdiffrast is absent): its own pins.
Known answers on the face axes, the OpenGL orientation of every face, constant textures, continuity across all 12 edges,
adjointness of the texture gradient, the reference's to_opengl frame change."""
import itertools

import numpy as np
import pytest

from oracle import cubemap_oracle as CO

AXES = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]


def test_face_axes_hit_face_centres():
    rng = np.random.default_rng(0)
    for res in (1, 2, 7, 8):
        tex = rng.random((6, res, res, 3))
        for f, ax in enumerate(AXES):
            o = CO.cubemap_fwd(tex, np.array([ax], float))[0]
            if res % 2:
                c = tex[f, res // 2, res // 2]
            else:
                c = tex[f, res // 2 - 1:res // 2 + 1, res // 2 - 1:res // 2 + 1].mean(axis=(0, 1))
            np.testing.assert_allclose(o, c, atol=1e-14)


def test_opengl_orientation_of_every_face():
    """(sc, tc) per face as in the OpenGL specification, table 'Selection of cube map images': +x (-z,-y)  -x (+z,-y)
    +y (+x,+z)  -y (+x,-z)  +z (+x,-y)  -z (-x,-y).  A texture that stores its own (u, v) at every texel reads them back."""
    res = 64
    c = (np.arange(res) + 0.5) / res
    tex = np.zeros((6, res, res, 2))
    tex[..., 0] = c[None, None, :]          # u along columns
    tex[..., 1] = c[None, :, None]          # v along rows
    rng = np.random.default_rng(1)
    a, b = rng.uniform(-0.9, 0.9, 500), rng.uniform(-0.9, 0.9, 500)
    one = np.ones_like(a)
    cases = {0: (np.stack([one, b, a], -1), -a, -b), 1: (np.stack([-one, b, a], -1), a, -b),
             2: (np.stack([a, one, b], -1), a, b), 3: (np.stack([a, -one, b], -1), a, -b),
             4: (np.stack([a, b, one], -1), a, -b), 5: (np.stack([a, b, -one], -1), -a, -b)}
    for f, (d, sc, tc) in cases.items():
        face, u, v = CO.cube_face(d)
        assert np.all(face == f)
        np.testing.assert_allclose(u, sc / 2 + 0.5, atol=1e-14)
        np.testing.assert_allclose(v, tc / 2 + 0.5, atol=1e-14)
        np.testing.assert_allclose(CO.cubemap_fwd(tex, d * rng.uniform(0.1, 9, (500, 1))), np.stack([u, v], -1), atol=1e-12)


def test_constant_texture_and_partition_of_unity():
    rng = np.random.default_rng(2)
    d = rng.standard_normal((5000, 3))
    for res in (1, 3, 16):
        np.testing.assert_allclose(CO.cubemap_fwd(np.full((6, res, res, 3), 0.7), d), 0.7, atol=1e-14)
        t, w = CO.taps(d, res)
        np.testing.assert_allclose(w.sum(-1), 1.0, atol=1e-14)
        assert t.min() >= 0 and t.max() < 6 * res * res and w.min() >= 0


def test_continuous_across_all_twelve_edges():
    rng = np.random.default_rng(3)
    res = 8
    tex = rng.random((6, res, res, 3))
    t = np.linspace(-0.85, 0.85, 60)         # the 8 corners have no fourth texel: excluded (see the oracle's header)
    worst = 0.0
    for a, b in itertools.permutations(range(3), 2):
        c = 3 - a - b
        for sa, sb in itertools.product((1, -1), repeat=2):
            da = np.zeros((60, 3)); db = np.zeros((60, 3))
            da[:, a] = sa * (1 + 1e-9); da[:, b] = sb; da[:, c] = t
            db[:, a] = sa; db[:, b] = sb * (1 + 1e-9); db[:, c] = t
            fa, fb = CO.cube_face(da)[0], CO.cube_face(db)[0]
            assert np.all(fa != fb)
            worst = max(worst, np.abs(CO.cubemap_fwd(tex, da) - CO.cubemap_fwd(tex, db)).max())
    assert worst < 1e-7


def test_texture_gradient_is_the_adjoint():
    rng = np.random.default_rng(4)
    res = 5
    tex = rng.random((6, res, res, 3))
    d = rng.standard_normal((3000, 3))
    v = rng.standard_normal((3000, 3))
    g = CO.cubemap_bwd(tex.shape, d, v)
    np.testing.assert_allclose((CO.cubemap_fwd(tex, d) * v).sum(), (g * tex).sum(), rtol=1e-12)


def test_frame_change_and_invalid_directions():
    rng = np.random.default_rng(5)
    tex = rng.random((6, 4, 4, 3))
    d = rng.standard_normal((100, 3))
    np.testing.assert_allclose(CO.cubemap_fwd(tex, d, rot=CO.TO_OPENGL), CO.cubemap_fwd(tex, d @ CO.TO_OPENGL.T), atol=1e-14)
    # camera looking along +z of the dataset frame (x right, y down, z forward) -> OpenGL (x, z, -y): "forward" hits face +y
    assert CO.cube_face(np.array([[0.0, 0.0, 1.0]]) @ CO.TO_OPENGL.T)[0][0] == 2
    bad = np.array([[0.0, 0.0, 0.0], [np.nan, 1.0, 0.0], [1.0, np.nan, 0.0]])
    np.testing.assert_array_equal(CO.cubemap_fwd(tex, bad), 0.0)
    np.testing.assert_allclose(CO.cubemap_fwd(tex, np.array([[np.inf, 3.0, -2.0]])), CO.cubemap_fwd(tex, np.array([[1.0, 0.0, 0.0]])))


def test_float32_follows_float64():
    rng = np.random.default_rng(6)
    tex = rng.random((6, 32, 32, 3)).astype(np.float32)
    d = rng.standard_normal((20000, 3)).astype(np.float32)
    a = CO.cubemap_fwd(tex, d, dtype=np.float32); b = CO.cubemap_fwd(tex, d, dtype=np.float64)
    assert np.percentile(np.abs(a - b), 99.9) < 1e-4      # a texel-boundary decision may flip for a handful (weights ~ 0 there)
