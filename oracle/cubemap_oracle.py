"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement (numpy) of the cube-map lookup behind the reference's sky
model -- EnvLight.forward, /root/reference/project/models/modules.py:176-211:
    l = viewdirs @ to_opengl^T (:196);  light = dr.texture(base[None], l, filter_mode='linear', boundary_mode='cube') (:202)

**PARITY UNPINNED**: `dr` is nvdiffrast (models/modules.py:10), an external CUDA-only package that is neither vendored in
/root/reference nor installed here, and the reference pins no version of it (README.md:83).  This file restates the public
convention it implements (the OpenGL cube map: face order +x -x +y -y +z -z, (sc, tc) per face, bilinear over texel centres) and is
pinned only by this repo's own tests (known answers on the face axes, constant textures, continuity across edges, adjointness).
Equality with nvdiffrast itself -- in particular its treatment of the 8 cube corners -- could not be checked.
Not imported by the product."""
from __future__ import annotations

import numpy as np

TO_OPENGL = np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0]], dtype=np.float64)       # modules.py:189


def cube_face(d, dtype=np.float64):
    """d [n,3] -> (face [n] (-1 invalid), u [n], v [n]); u, v in [0,1]."""
    d = d.astype(dtype)
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    ax, ay, az = np.abs(x), np.abs(y), np.abs(z)
    isz = az > np.maximum(ax, ay)
    isy = ~isz & (ay > ax)
    isx = ~isz & ~isy
    c = np.where(isz, z, np.where(isy, y, x))
    sc = np.where(isx, z, x)
    tc = np.where(isy, z, y)
    idx = np.where(isz, 4, np.where(isy, 2, 0)) + (c < 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        m = (dtype(0.5) / np.abs(c)).astype(dtype)
        m0 = np.where((idx == 0) | (idx == 5), -m, m)
        m1 = np.where(idx == 2, m, -m)
        u = (sc * m0 + dtype(0.5)).astype(dtype)
        v = (tc * m1 + dtype(0.5)).astype(dtype)
    bad = ~(np.isfinite(u) & np.isfinite(v))
    u = np.clip(np.where(bad, 0, u), 0, 1).astype(dtype)
    v = np.clip(np.where(bad, 0, v), 0, 1).astype(dtype)
    return np.where(bad, -1, idx), u, v


def cube_dir(idx, sc, tc):
    one = np.ones_like(sc)
    x = np.select([idx == 0, idx == 1, idx == 5], [one, -one, -sc], sc)
    y = np.select([idx == 2, idx == 3], [one, -one], -tc)
    z = np.select([idx == 0, idx == 1, idx == 2, idx == 3, idx == 4], [-sc, sc, tc, -tc, one], -one)
    return np.stack([x, y, z], axis=-1)


def cube_texel(idx, iu, iv, res, dtype=np.float64):
    """Linear texel index; texels outside the face come from the face their extended centre lands on."""
    out_u, out_v = (iu < 0) | (iu >= res), (iv < 0) | (iv >= res)
    iv = np.where(out_u & out_v, np.clip(iv, 0, res - 1), iv)      # cube corner: no fourth texel; take the u-neighbour's corner texel
    out = out_u | out_v
    sc = ((2 * iu + 1).astype(dtype) / dtype(res) - 1).astype(dtype)
    tc = ((2 * iv + 1).astype(dtype) / dtype(res) - 1).astype(dtype)
    f2, u2, v2 = cube_face(cube_dir(idx, sc, tc), dtype)
    iu2 = np.clip(np.floor(u2 * dtype(res)).astype(np.int64), 0, res - 1)
    iv2 = np.clip(np.floor(v2 * dtype(res)).astype(np.int64), 0, res - 1)
    f = np.where(out, f2, idx); a = np.where(out, iu2, iu); b = np.where(out, iv2, iv)
    return (f.astype(np.int64) * res + b) * res + a


def taps(dirs, res, rot=None, dtype=np.float64):
    d = dirs.reshape(-1, 3).astype(dtype)
    if rot is not None:
        r = np.asarray(rot, dtype=dtype)
        with np.errstate(invalid="ignore"):
            d = np.stack([(d[:, 0] * r[i, 0] + d[:, 1] * r[i, 1]) + d[:, 2] * r[i, 2] for i in range(3)], axis=-1).astype(dtype)
    idx, u, v = cube_face(d, dtype)
    valid = idx >= 0
    idx = np.where(valid, idx, 0)
    ut, vt = (u * dtype(res) - dtype(0.5)).astype(dtype), (v * dtype(res) - dtype(0.5)).astype(dtype)
    fu0, fv0 = np.floor(ut), np.floor(vt)
    iu0, iv0 = fu0.astype(np.int64), fv0.astype(np.int64)
    fu, fv = (ut - fu0).astype(dtype), (vt - fv0).astype(dtype)
    t = np.stack([cube_texel(idx, iu0, iv0, res, dtype), cube_texel(idx, iu0 + 1, iv0, res, dtype),
                  cube_texel(idx, iu0, iv0 + 1, res, dtype), cube_texel(idx, iu0 + 1, iv0 + 1, res, dtype)], axis=-1)
    w = np.stack([(1 - fu) * (1 - fv), fu * (1 - fv), (1 - fu) * fv, fu * fv], axis=-1).astype(dtype)
    w = w * valid[:, None]
    return t, w


def cubemap_fwd(tex, dirs, rot=None, dtype=np.float64):
    """tex [6,res,res,C], dirs [...,3] -> [...,C]."""
    res, C = tex.shape[1], tex.shape[-1]
    t, w = taps(dirs, res, rot, dtype)
    flat = tex.reshape(-1, C).astype(dtype)
    out = np.zeros((t.shape[0], C), dtype)
    for k in range(4):
        out = out + w[:, k:k + 1] * flat[t[:, k]]
    return out.reshape(dirs.shape[:-1] + (C,))


def cubemap_bwd(tex_shape, dirs, v_out, rot=None, dtype=np.float64):
    """Adjoint of cubemap_fwd with respect to the texture."""
    res, C = tex_shape[1], tex_shape[-1]
    t, w = taps(dirs, res, rot, dtype)
    g = np.zeros((6 * res * res, C), dtype)
    vo = v_out.reshape(-1, C).astype(dtype)
    for k in range(4):
        np.add.at(g, t[:, k], w[:, k:k + 1] * vo)
    return g.reshape(tex_shape)
