"""gsplat.cuda_legacy._wrapper.num_sh_bases (models/gaussians/basics.py:13; used at vanilla.py:73,94)."""


def num_sh_bases(degree: int) -> int:
    if degree < 0 or degree > 4:
        raise AssertionError(f"Invalid SH degree {degree}")
    return (degree + 1) ** 2
