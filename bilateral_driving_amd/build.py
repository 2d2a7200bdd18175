"""Build libbds.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m bilateral_driving_amd.build [--force]

The library is a plain C-ABI shared object (include/bds.h); it links only against the HIP
runtime (libamdhip64.so.7), which resolves to the copy PyTorch-ROCm has already loaded.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbds.so")
SOURCES = ["api.hip", "sh.hip", "project.hip", "tiles.hip", "rasterize.hip", "bilagrid.hip", "bilagrid_cells.hip", "bilagrid_tile.hip", "loss.hip", "optim.hip", "refine.hip", "envlight.hip", "colorcorrect.hip", "mlp_head.hip", "exchange.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=fast-honor-pragmas", "-Wall", "-Wno-unused-function"]
# per-file additions.  bilagrid.hip: the SLP vectoriser pairs fp32 operations into v_pk_* instructions, which issue at half rate on
# gfx950 (scripts/ubench/valu_rate.hip) and need register shuffles (v_mov) to line their operands up: net loss in kernels that
# are bound by instruction issue (static counts: DESIGN.md section 4)
EXTRA_FLAGS = {"bilagrid.hip": ["-fno-slp-vectorize"], "bilagrid_cells.hip": ["-fno-slp-vectorize"], "bilagrid_tile.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 to build libbds.so for gfx950)")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "bds.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, defines=(), out: str = LIB, subdir: str = "build") -> str:
    """``defines`` / ``out`` / ``subdir``: an A/B variant of the library next to the product (e.g. ``defines=("BDS_PRIO=0",)``,
    ``out=".../libbds_prio0.so"``; ``BDS_LIB=<path>`` makes ``_lib`` load it) -- measurement sessions only."""
    if out == LIB and not force and not _stale():
        return LIB
    hipcc = _hipcc()
    objs = []
    bdir = os.path.join(HERE, subdir)
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        cmd = [hipcc, f"--offload-arch={ARCH}", *FLAGS, *[f"-D{d}" for d in defines], *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        log, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{log.decode()}")
        if verbose and log:
            print(log.decode())
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", out + ".tmp"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode()}")
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    if "--variant" in sys.argv:     # python -m bilateral_driving_amd.build --variant prio0 BDS_PRIO=0
        i = sys.argv.index("--variant")
        name, defs = sys.argv[i + 1], sys.argv[i + 2:]
        print(build(force=True, verbose=False, defines=defs, out=os.path.join(HERE, f"libbds_{name}.so"), subdir=f"build_{name}"))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
