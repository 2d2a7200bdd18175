#!/usr/bin/env python3
"""Golden vectors for the colour-correct post-process by IMPORTING THE REFERENCE: bilateral/lib_bilagrid.py color_correct (:56-120),
called as the evaluation path does (models/video_utils_color_correction.py:201: color_correct(rgb, pixels, 10)) and with defaults.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_color_correct.py        (build container only; tensorly stubbed)

img = a smooth colour transform of ref plus noise, with clipped regions on both sides (the masks matter).  The float64 run is the
accurate answer (the reference's float32 run solves a [pixels x 10] least-squares problem by QR in single precision and is itself only
good to ~1e-4); both are stored."""
import os
import sys
import types

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    sys.dont_write_bytecode = True
    m = types.ModuleType("tensorly"); m.set_backend = lambda *_: None
    sys.modules["tensorly"] = m
    sys.path.insert(0, "/root/reference/project")
    import bilateral.lib_bilagrid as LB
    return LB


def main():
    LB = import_reference()
    for name, H, W, iters, seed in (("a", 48, 64, 5, 0), ("b", 40, 56, 10, 1)):
        g = torch.Generator().manual_seed(seed)
        ref = torch.rand(H, W, 3, generator=g, dtype=torch.float64) * 1.1 - 0.05
        ref = ref.clamp(0, 1)
        A = torch.eye(3, dtype=torch.float64) + 0.15 * torch.randn(3, 3, generator=g, dtype=torch.float64)
        img = (ref @ A.T + 0.08 * ref ** 2 - 0.03 + 0.01 * torch.randn(H, W, 3, generator=g, dtype=torch.float64)).clamp(0, 1)
        out64 = LB.color_correct(img, ref, iters)
        out32 = LB.color_correct(img.float(), ref.float(), iters)
        np.savez_compressed(os.path.join(OUT, f"color_correct_{name}.npz"), img=img.numpy(), ref=ref.numpy(), num_iters=np.array(iters),
                            out_f64=out64.numpy(), out_f32=out32.numpy())
        print(name, float((out64 - ref).abs().mean()), float((img - ref).abs().mean()), float((out64 - out32.double()).abs().max()))


if __name__ == "__main__":
    main()
