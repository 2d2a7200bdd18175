"""Register budget of the hot kernels, read from hipcc's own resource report (-Rpass-analysis=kernel-resource-usage) with the product's
build flags: no kernel may spill to scratch, and the kernels whose speed depends on how many waves fit a SIMD keep the occupancy they
were tuned at.  Cross-compiles for gfx950, needs no GPU."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

from bilateral_driving_amd import build as B

FILES = ["rasterize.hip", "tiles.hip", "bilagrid.hip", "bilagrid_cells.hip", "bilagrid_tile.hip", "sh.hip", "project.hip", "mlp_head.hip", "loss.hip", "refine.hip"]
# scratch allowed (bytes / lane): a shape no shipped config uses
SCRATCH_OK = {"neural_image_fwd_kernelILi32ELi8ELi0ELi0E": 16}
# kernel name prefix (mangled, after the length digits) -> minimum waves / SIMD
MIN_OCCUPANCY = {
    "rasterize_fwd_wave_kernelILi4ELb1ELb1E": 7,        # the benchmark's forward compositor (RGB+ED, coarse lists)
    "rasterize_bwd_wave_kernelILi4ELb1ELb1ELb0E": 5,    # ... and its backward (absgrad)
    "rasterize_bwd_epi_kernelILb1ELb1E": 4,             # ... with the colour transform's deferred epilogue (one-stream frames)
    "ms_apply_fwd_kernelILi3E": 6,
    "ms_apply_bwd_x_kernelILi3E": 5,
    "ms_lowres_bwd_kernelILb1ELi4E": 3,
    "ms_tile_fwd_kernelILi3ELb1E": 4,                   # pyramid forward, one pass (headline / c3)
    "ms_tile_fwd_kernelILi4ELb1E": 4,                   # ... c5
    "cell_fwd_kernelILb1ELb1E": 5,                      # single-scale transform in one launch (c2)
    "cell_bwd_kernelILb1E": 4,
    "cell_bwd_kernelILb0E": 4,                          # low-resolution stage of a pyramid's backward
    "mlp_head_fwd_kernelILi24E": 2,
    "neural_image_fwd_kernelILi24ELi8ELi0ELi0E": 2,
    "neural_image_bwd_kernelILi24ELi8ELi0ELi0E": 1,     # 256 VGPRs + the weight / slot gradient accumulators in AGPRs: one wave per SIMD by design
}


def report(src):
    cmd = [B._hipcc(), f"--offload-arch={B.ARCH}", *B.FLAGS, *B.EXTRA_FLAGS.get(src, []), "-Rpass-analysis=kernel-resource-usage", "-c",
           os.path.join(B.CSRC, src), "-o", os.devnull]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    out, cur = {}, None
    for line in r.stdout.splitlines():
        m = re.search(r"Function Name: _ZN3bds\d+(\S+)", line)
        if m:
            cur = m.group(1)
            out[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            out[cur][m.group(1).strip()] = int(m.group(2))
    return out


@pytest.fixture(scope="module")
def kernels():
    with ThreadPoolExecutor(max_workers=8) as ex:
        reps = list(ex.map(report, FILES))
    merged = {}
    for r in reps:
        merged.update(r)
    assert len(merged) > 100
    return merged


def test_no_kernel_spills_to_scratch(kernels):
    bad = {}
    for name, res in kernels.items():
        allowed = max([v for k, v in SCRATCH_OK.items() if name.startswith(k)], default=0)
        if res.get("ScratchSize", 0) > allowed:
            bad[name] = res["ScratchSize"]
    assert not bad, bad


def test_tuned_kernels_keep_their_occupancy(kernels):
    for prefix, occ in MIN_OCCUPANCY.items():
        hits = [(n, r) for n, r in kernels.items() if n.startswith(prefix)]
        assert hits, f"kernel {prefix} not found in the report"
        for n, r in hits:
            assert r["Occupancy"] >= occ, (n, r)
