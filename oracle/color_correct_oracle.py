"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement (numpy, float64 normal equations) of the colour-correct
post-process -- bilateral/lib_bilagrid.py:56-120 `color_correct`, used by the evaluation path at
models/video_utils_color_correction.py:201.  Not imported by the product.

PINNED: tests/test_oracle_color_correct.py checks it against tests/golden/color_correct_*.npz, outputs of the reference's own function
(oracle/gen_golden_color_correct.py).  It is written the way the HIP path computes (masked normal equations per channel instead of a
least-squares solve of the tall [pixels x 10] system), so passing the goldens also validates that formulation."""
from __future__ import annotations

import numpy as np


def expand(x):
    r, g, b = x[:, 0], x[:, 1], x[:, 2]
    return np.stack([r * r, r * g, r * b, g * g, g * b, b * b, r, g, b, np.ones_like(r)], axis=-1)      # :98-104


def color_correct(img, ref, num_iters=5, eps=0.5 / 255, dtype=np.float64):
    shape = img.shape
    cur = img.reshape(-1, 3).astype(dtype)
    rf = ref.reshape(-1, 3).astype(dtype)
    unclipped = lambda z: (z >= dtype(eps)) & (z <= dtype(1) - dtype(eps))
    mask0 = unclipped(cur)
    for _ in range(num_iters):
        a = expand(cur)
        warp = np.zeros((10, 3))
        for c in range(3):
            m = mask0[:, c] & unclipped(cur[:, c]) & unclipped(rf[:, c])                                   # :110
            am = a[m].astype(np.float64)
            G, h = am.T @ am, am.T @ rf[m, c].astype(np.float64)
            warp[:, c] = np.linalg.pinv(G, hermitian=True) @ h
        assert np.all(np.isfinite(warp))
        cur = np.clip(a @ warp.astype(dtype), 0, 1).astype(dtype)                                          # :117
    return cur.reshape(shape)
