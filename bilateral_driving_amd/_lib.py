"""ctypes binding of libbds.so (include/bds.h).

There is no CPU fallback: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BDS_LIB") or os.path.join(_HERE, "libbds.so")    # (BDS_LIB: an A/B variant built by build.py --variant)
ABI_VERSION = 2

_lock = threading.Lock()
_lib = None

_f = C.c_void_p  # every device pointer is passed as an integer address
_i, _i64, _sz, _fl = C.c_int, C.c_int64, C.c_size_t, C.c_float


class BdsLevel(C.Structure):
    """bds_bilagrid_level_t"""

    _fields_ = [("grid", C.c_void_p), ("v_grid", C.c_void_p), ("gx", C.c_int32), ("gy", C.c_int32), ("gl", C.c_int32),
                ("factor", C.c_int32), ("n_avg", C.c_int32)]


class BdsFeatLevel(C.Structure):
    """bds_feat_level"""

    _fields_ = [("grid", C.c_void_p), ("v_grid", C.c_void_p), ("gx", C.c_int32), ("gy", C.c_int32), ("gl", C.c_int32), ("nch", C.c_int32)]


_SIGS = {
    "bds_abi_version": (C.c_int, []),
    "bds_strerror": (C.c_char_p, [_i]),
    "bds_set_option": (_i, [_i, _i]),
    "bds_get_option": (_i, [_i]),
    "bds_timer_create": (C.c_void_p, []),
    "bds_timer_destroy": (_i, [_f]),
    "bds_timer_mark": (_i, [_f, _f]),
    "bds_timer_elapsed_ms": (C.c_float, [_f, _f]),
    "bds_last_hip_error": (_i, []),
    "bds_sh_fwd": (_i, [_i64, _i, _i, _f, _f, _f, _f, _f]),
    "bds_sh_bwd": (_i, [_i64, _i, _i, _f, _f, _f, _f, _f, _f, _f]),
    "bds_project_fwd": (_i, [_i, _i64, _f, _f, _f, _f, _f, _i, _i, _fl, _fl, _fl, _fl, _f, _f, _f, _f, _f, _f]),
    "bds_project_bwd": (_i, [_i, _i64, _f, _f, _f, _f, _f, _i, _i, _fl, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f]),
    "bds_isect_prepare_workspace_bytes": (_sz, [_i, _i64]),
    "bds_isect_visible_ids_offset": (_sz, [_i, _i64]),
    "bds_isect_build_workspace_bytes": (_sz, [_i, _i64, _i64]),
    "bds_isect_prepare": (_i, [_i, _i64, _f, _f, _f, _f, _f, _i, _i, _i, _f, _f, _sz, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _i, _f]),
    "bds_isect_build": (_i, [_i, _i64, _i64, _i64, _f, _f, _f, _f, _f, _i, _i, _i, _f, _sz, _f, _sz, _f, _f, _f, _f, _i, _f]),
    "bds_isect_prepare_async": (_i, [_i, _i64, _f, _f, _f, _f, _f, _i, _i, _i, _f, _f, _sz, _f, _f, _i, _f]),
    "bds_isect_tiles": (_i, [_i, _i64, _f, _f, _f, _f, _f, _i, _i, _i, _f, _f, _sz, _f, _sz, _i64, _f, _f, _f, C.POINTER(C.c_int64),
                             C.POINTER(C.c_int64), _f]),
    "bds_splat_pack_rgbd": (_i, [_i64, _f, _f, _f, _f, _f, _f, _f, _f, _f]),
    "bds_expected_depth_fwd": (_i, [_i64, _f, _f, _f, _f]),
    "bds_expected_depth_bwd": (_i, [_i64, _i, _i, _f, _f, _f, _f, _f, _f, _f]),
    "bds_expected_depth_split_fwd": (_i, [_i64, _i, _f, _f, _f, _f, _f]),
    "bds_expected_depth_split_bwd": (_i, [_i64, _i, _f, _f, _f, _f, _f, _f, _f, _f]),
    "bds_splat_pack": (_i, [_i64, _i, _f, _f, _f, _f, _f, _f, _f, _f]),
    "bds_rasterize_fwd": (_i, [_i, _i64, _i64, _i, _f, _f, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f]),
    "bds_rasterize_bwd": (_i, [_i, _i64, _i64, _i, _f, _f, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _i, _f, _f]),
    "bds_rasterize_kernel_name": (_i, [_i, _i, _i, _i, C.c_char_p, _i]),
    "bds_rasterize_bwd_schedule": (_i, [_i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f]),
    "bds_project_view_fwd": (_i, [_i64, _f, _f, _f, _f, _f, _f, _i, _i, _fl, _fl, _fl, _fl, _f, _f, _f, _f, _f, _f, _f]),
    "bds_project_view_prepare_fwd": (_i, [_i64, _f, _f, _f, _f, _f, _f, _i, _i, _fl, _fl, _fl, _fl, _f, _f, _f, _f, _f, _f, _f, _f, _sz, _f]),
    "bds_project_view_prepare_fwd_blocks": (_i, [_i64, _f, _f, _f, _f, _f, _f, _i, _i, _fl, _fl, _fl, _fl, _f, _f, _f, _f, _f, _f, _f, _f, _sz, _f, _f]),
    "bds_project_view_fwd_blocks": (_i, [_i64, _f, _f, _f, _f, _f, _f, _i, _i, _fl, _fl, _fl, _fl, _f, _f, _f, _f, _f, _f, _f, _f]),
    "bds_gaussian_block_bounds": (_i, [_i64, _f, _f, _f, _f]),
    "bds_sh_view_fwd": (_i, [_i64, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f]),
    "bds_sh_view_bwd_list": (_i, [_i64, _f, _i, _i, _f, _f, _f, _i, _f, _f, _f, _i, _f]),
    "bds_splat_pack_sh": (_i, [_i64, _f, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f]),
    "bds_splat_pack_sh_split": (_i, [_i64, _f, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f]),
    "bds_sh_view_bwd_list_split": (_i, [_i64, _f, _i, _i, _f, _f, _f, _i, _f, _f, _f, _i, _f]),
    "bds_nonfinite_flags": (_i, [_i, _f, _f, _f, _f, _f]),
    "bds_nonfinite_flags_kinds": (_i, [_i, _f, _f, _f, _f, _f, _f]),
    "bds_project_bwd_list": (_i, [_i64, _f, _f, _f, _f, _f, _f, _f, _i, _i, _fl, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f]),
    "bds_project_view_bwd_list": (_i, [_i64, _f, _f, _f, _f, _f, _f, _f, _i, _i, _fl, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _f]),
    "bds_view_grads_clear_list": (_i, [_i64, _f, _i, _f, _f, _f, _f, _f, _f]),
    "bds_view_grads_add_list": (_i, [_i64, _f, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f]),
    "bds_isect_counts_offset": (_sz, [_i]),
    "bds_isect_prepare_dev": (_i, [_i, _i64, _f, _f, _f, _f, _f, _i, _i, _i, _f, _f, _sz, _i64, _i64, _f, _i, _f]),
    "bds_isect_build_dev": (_i, [_i, _i64, _i64, _i64, _f, _f, _f, _f, _f, _i, _i, _i, _f, _sz, _f, _sz, _f, _f, _i, _f]),
    "bds_splat_pack_sh_dev": (_i, [_i64, _f, _f, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i64, _f, _f]),
    "bds_splat_pack_dev": (_i, [_i64, _f, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i64, _f, _f]),
    "bds_rasterize_schedule_ints": (_i64, [_i, _i, _i]),
    "bds_rasterize_fwd_dev": (_i, [_i, _i64, _i64, _f, _i, _f, _f, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i64, _f]),
    "bds_rasterize_split_pool_ints": (_i64, [_i, _i, _i, _i, _i64, _i64]),
    "bds_rasterize_bwd_schedule_sort": (_i, [_i, _i, _i, _f, _f]),
    "bds_rasterize_bwd_dev": (_i, [_i, _i64, _i64, _f, _i, _f, _f, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _i, _f, _i, _i, _i64, _f]),
    "bds_sh_view_bwd_list_dev": (_i, [_i64, _f, _f, _i, _i, _f, _f, _f, _i, _f, _f, _f, _i, _f]),
    "bds_project_view_bwd_list_dev": (_i, [_i64, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _fl, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _f]),
    "bds_view_grads_clear_list_dev": (_i, [_i64, _f, _f, _i, _f, _f, _f, _f, _f, _f, _f, _f]),
    "bds_union_slots_workspace_bytes": (_sz, [_i64]),
    "bds_union_slots": (_i, [_i64, _f, _i64, _i, _f, _f, _f, _f, _f, _f, _f, _f, _sz, _f, _f, _f]),
    "bds_bilagrid_slice_fwd": (_i, [_i64, _f, _i, _i, _i, _f, _f, _f, _f]),
    "bds_bilagrid_slice_bwd": (_i, [_i64, _f, _i, _i, _i, _f, _f, _f, _f, _f, _f]),
    "bds_bilagrid_ms_workspace_bytes": (_sz, [_i, C.POINTER(BdsLevel), _i, _i]),
    "bds_bilagrid_select": (_i, [_i, C.POINTER(BdsLevel), _f, C.POINTER(C.c_void_p), _f, _f]),
    "bds_bilagrid_select_bwd": (_i, [_i, C.POINTER(BdsLevel), _f, C.POINTER(C.c_void_p), _f, _f]),
    "bds_bilagrid_kernel_names": (_i, [_i, C.POINTER(BdsLevel), _i, _i, _i, _i, C.c_char_p, _i]),
    "bds_bilagrid_ms_fwd": (_i, [_i, C.POINTER(BdsLevel), _i, _i, _f, _f, _f, _f, _sz, _f, C.POINTER(C.c_void_p), _f]),
    "bds_bilagrid_ms_bwd": (_i, [_i, C.POINTER(BdsLevel), _i, _i, _f, _f, _f, _f, _sz, _f, _f, _f, _f, _f]),
    "bds_bilagrid_ms_ed_train_fwd": (_i, [_i, C.POINTER(BdsLevel), _i, _i, _f, _f, _f, _f, _sz, _f, _f, _f, _i, C.POINTER(BdsLevel),
                                          C.POINTER(C.c_float), _fl, _f, _i, _f, _f]),
    "bds_bilagrid_ms_ed_fwd": (_i, [_i, C.POINTER(BdsLevel), _i, _i, _f, _f, _f, _f, _sz, _f, _f, _f]),
    "bds_bilagrid_ms_ed_bwd_deferrable": (_i, [_i, C.POINTER(BdsLevel), _i, _i]),
    "bds_bilagrid_ms_ed_bwd_deferred": (_i, [_i, C.POINTER(BdsLevel), _i, _i, _f, _f, _f, _f, _sz, _f, _f, _f]),
    "bds_rasterize_bwd_ms": (_i, [_i64, _i64, _f, _f, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _i, _f, _i, C.POINTER(BdsLevel), _f, _sz,
                                  _f, _f, _f, _f, _f, _f, _f]),
    "bds_bilagrid_ms_ed_bwd": (_i, [_i, C.POINTER(BdsLevel), _i, _i, _f, _f, _f, _f, _sz, _f, _f, _f, _f, _f, _f, _f]),
    "bds_l1_tv_train": (_i, [_i64, _f, _f, _i, C.POINTER(BdsLevel), C.POINTER(C.c_float), _fl, _f, _i, _f, _f]),
    "bds_l1_mean_fwd": (_i, [_i64, _f, _f, _f, _f]),
    "bds_l1_mean_bwd": (_i, [_i64, _f, _f, _f, _f, _f]),
    "bds_ssim_workspace_bytes": (_sz, [_i, _i, _i]),
    "bds_ssim_fwd": (_i, [_i, _i, _i, _f, _f, _f, _f, _sz, _f]),
    "bds_ssim_bwd": (_i, [_i, _i, _i, _f, _f, _f, _sz, _f, _f, _f]),
    "bds_pixel_loss_fwd": (_i, [_i64, _f, _f, _f, _f, _f, _f, _f, _fl, _fl, _fl, _i, _fl, _f, _f, _f]),
    "bds_pixel_loss_bwd": (_i, [_i64, _f, _f, _f, _f, _f, _f, _f, _fl, _fl, _fl, _i, _fl, _f, _f, _f, _f, _f, _f]),
    "bds_reg_loss_fwd": (_i, [_i, _i, _f, _f, _f, _f, _f, _f, _fl, _f, _f, _f]),
    "bds_reg_loss_bwd": (_i, [_i, _i, _f, _f, _f, _f, _f, _f, _fl, _f, _f, _f, _f, _f, _f]),
    "bds_densify_stats": (_i, [_i64, _f, _f, _i, _i, _i, _i, _i, _f, _f, _f, _f]),
    "bds_refine_plan_temp_bytes": (_sz, [_i64]),
    "bds_refine_plan": (_i, [_i64, _f, _f, _f, _f, _f, _f, _i, _fl, _fl, _i, _fl, _i, _fl, _i, _fl, _i, _fl, _f, _f, _f, _f, _sz, _f]),
    "bds_refine_geometry": (_i, [_i64, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f]),
    "bds_refine_rows": (_i, [_i64, _i, _i, _f, _f, _f, _f, _f, _i, _f]),
    "bds_refine_out_of_bound": (_i, [_i64, _f, _f, _i64, _f, _f, _f]),
    "bds_bilagrid_slice_feat_image_ok": (_i, [_i, _i, _i, _i]),
    "bds_bilagrid_slice_feat_image_fwd": (_i, [_i, _i, _i, _f, _i, _i, _i, _f, _f, _f]),
    "bds_bilagrid_slice_feat_image_bwd": (_i, [_i, _i, _i, _f, _i, _i, _i, _f, _f, _f, _f, _f]),
    "bds_neural_image_ok": (_i, [_i, _i, _i, C.POINTER(BdsFeatLevel), _i]),
    "bds_neural_image_bwd_temp_bytes": (_sz, [_i]),
    "bds_neural_image_fwd": (_i, [_i, _i, _i, C.POINTER(BdsFeatLevel), _i, _f, _f, _f, _f, _i, _f, _f]),
    "bds_neural_image_bwd": (_i, [_i, _i, _i, C.POINTER(BdsFeatLevel), _i, _f, _f, _f, _f, _i, _f, _f, _f, _f, _f, _i, _f, _sz, _f]),
    "bds_mlp_head_bwd_temp_bytes": (_sz, [_i64, _i]),
    "bds_mlp_head_fwd": (_i, [_i64, _i, _i, _f, _f, _f, _f, _f, _i, _f, _f, _f]),
    "bds_mlp_head_bwd": (_i, [_i64, _i, _i, _f, _f, _f, _f, _f, _i, _f, _f, _f, _f, _f, _f, _f, _i, _f, _sz, _f]),
    "bds_opacity_reset": (_i, [_i64, _f, _fl, _f, _f, _f]),
    "bds_cubemap_fwd": (_i, [_i64, _i, _i, _f, _f, _f, _f, _f]),
    "bds_cubemap_bwd": (_i, [_i64, _i, _i, _i, _f, _f, _f, _f, _f]),
    "bds_color_correct_step": (_i, [_i64, _f, _f, _f, _fl, _f, _f, _f, _f]),
    "bds_adam_step": (_i, [_i64, _f, _f, _f, _f, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _i64, _f]),
    "bds_adam_step_consume": (_i, [_i64, _f, _f, _f, _f, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _i64, _f]),
    "bds_adam_step_rows": (_i, [_i64, _i, _i64, _f, _f, _f, _f, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _i64, _i, _f]),
    "bds_adam_step_rowblock": (_i, [_i64, _f, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _f]),
    "bds_adam_step_multi": (_i, [_i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _f]),
    "bds_adam_rows_advance": (_i, [_i64, _f, _f, _i64, _i, _i, _f, _f, _i, _f, _f, _f, _f, _i64, _i, _f, _i, C.c_double, C.c_double,
                                   C.c_double, C.c_double, C.c_double, C.c_double, _f]),
    "bds_bilagrid_slice_feat_fwd": (_i, [_i64, _i, _f, _i, _i, _i, _f, _f, _f, _f]),
    "bds_bilagrid_slice_feat_bwd": (_i, [_i64, _i, _f, _i, _i, _i, _f, _f, _f, _f, _f, _f]),
    "bds_grid_tv_fwd": (_i, [_i64, _i, _i, _i, _i, _f, _fl, _f, _f]),
    "bds_grid_tv_bwd": (_i, [_i64, _i, _i, _i, _i, _f, _fl, _f, _f, _f]),
    "bds_bilagrid_tv_ms_fwd": (_i, [_i, C.POINTER(BdsLevel), C.POINTER(C.c_float), _f, _f]),
    "bds_bilagrid_tv_ms_bwd": (_i, [_i, C.POINTER(BdsLevel), C.POINTER(C.c_float), _f, _f]),
    "bds_bilagrid_tv_fwd": (_i, [_i64, _i, _i, _i, _f, _fl, _f, _f]),
    "bds_bilagrid_tv_bwd": (_i, [_i64, _i, _i, _i, _f, _fl, _f, _f, _f]),
}

EXPORTS = tuple(_SIGS)


BDS_OK, BDS_EINVAL, BDS_EWORKSPACE, BDS_ELAUNCH, BDS_ECAPACITY = 0, -1, -2, -3, -4      # include/bds.h


class BdsError(RuntimeError):
    pass


def lib():
    """Load libbds.so once.  Raises (never falls back) when it is missing or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise BdsError(
                f"{LIB_PATH} is missing: build it with `python -m bilateral_driving_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
            )
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if handle.bds_abi_version() != ABI_VERSION:
            raise BdsError(f"libbds.so ABI {handle.bds_abi_version()} != expected {ABI_VERSION}")
        _lib = handle
    return _lib


SPLAT_RECORD_FLOATS, GRAD_RECORD_FLOATS, POSE_GRAD_SLOTS = 12, 16, 64
OPT_DEBUG = 3          # profiling only: ablation mask
OPT_SCHED_BINS = 8     # device-count compositor: 1 [default] = the forward's waves bin the backward's schedule, 0 = a sort launch (include/bds.h)
OPT_CELLS = 7          # bilateral transform, bit mask [default 3]: 1 = cell-aligned kernels, 2 = one-pass pyramid forward, 0 = general kernels (include/bds.h)
LOSS_SLOTS, LOSS_SLOT_STRIDE = 64, 64      # slotted loss accumulators (include/bds.h BDS_LOSS_SLOT_STRIDE)
OPT_SHORT_SORT, OPT_PACKED = 4, 6   # test hooks: force the large-input fallback paths of the tile stage (include/bds.h)
ECAPACITY = -4
OPT_CAP_LAUNCH = 0     # device-count tile stage: 1 [default] = launches sized by the visible-entry capacity, 0 = by all N


def rasterize_kernel_name(backward, CH: int = 4, absgrad: bool = True, list_tile_size: int = 64) -> str:
    """``backward``: False / True, or 2 = the backward that also runs the colour transform's deferred epilogue."""
    buf = C.create_string_buffer(128)
    check(lib().bds_rasterize_kernel_name(int(backward), CH, int(absgrad), list_tile_size, buf, 128), "bds_rasterize_kernel_name")
    return buf.value.decode()


def bilagrid_kernel_names(levels, H: int, W: int, backward: bool, train: bool = False) -> list:
    """Kernel names (as rocprofv3 prints them) of the bilateral transform's launches for a BdsLevel array."""
    buf = C.create_string_buffer(512)
    check(lib().bds_bilagrid_kernel_names(len(levels), levels, H, W, int(backward), int(train), buf, 512), "bds_bilagrid_kernel_names")
    return buf.value.decode().split(",")


def set_option(which: int, value: int) -> None:
    check(lib().bds_set_option(which, value), "bds_set_option")


def check(code: int, what: str) -> None:
    if code != 0:
        raise BdsError(f"{what} failed: {lib().bds_strerror(code).decode()} (code {code})")


def ptr(t):
    """Device address of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "libbds.so takes contiguous tensors"
    return t.data_ptr()


def stream(device=None):
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu(*tensors):
    """Every tensor lives on ONE GPU, and it is the current device: the kernels are enqueued on the current device's current
    stream (``stream()``), so a tensor of another GPU would be touched from the wrong device's queue."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise BdsError(
                "bilateral_driving_amd runs on MI355X only (tensor on %s); there is no CPU path "
                "in the product -- the CPU restatement lives in oracle/ and is test infrastructure" % t.device
            )
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise BdsError(f"tensors on different GPUs in one call ({dev} and {t.device})")
    if dev is not None and dev.index != torch.cuda.current_device():
        raise BdsError(f"tensors live on {dev} but the current device is cuda:{torch.cuda.current_device()}: "
                       "call torch.cuda.set_device (one process per GPU) before using bilateral_driving_amd")


# ----------------------------------------------------------------------------------------------
# optional per-call device timing (bench.py): HIP events on the stream the kernels are launched on
# ----------------------------------------------------------------------------------------------
import contextlib

TIMERS = None  # dict name -> list[(start_event, end_event)] when enabled


def enable_timers(on: bool = True, only=None) -> None:
    """HIP-event timing of the operators wrapped in ``timed(name)``.  ``only``: an iterable of names to time (every event
    pair is two more packets on the stream; a timed benchmark region keeps them to the one kernel it reports)."""
    global TIMERS, TIMERS_ONLY
    TIMERS = {} if on else None
    TIMERS_ONLY = None if only is None else frozenset(only)


TIMERS_ONLY = None


@contextlib.contextmanager
def timed(name: str):
    if TIMERS is None or (TIMERS_ONLY is not None and name not in TIMERS_ONLY):
        yield
        return
    if torch.cuda.is_current_stream_capturing():
        # inside a hipGraph capture: marks that become event-record nodes (re-recorded by every replay; read with replay_timers)
        s, e = _Mark(), _Mark()
        if lib().bds_timer_mark(s.h, stream()) != 0:
            raise BdsError(f"bds_timer_mark failed under capture (HIP error {lib().bds_last_hip_error()})")
        try:
            yield
        finally:
            check(lib().bds_timer_mark(e.h, stream()), "bds_timer_mark")
            GRAPH_MARKS.setdefault(name, []).append((s, e))
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(torch.cuda.current_stream())
    try:
        yield
    finally:
        e.record(torch.cuda.current_stream())
        TIMERS.setdefault(name, []).append((s, e))


class _Mark:
    """One libbds timing mark (a hipEvent that is recorded as a graph node under capture)."""

    def __init__(self):
        self.h = lib().bds_timer_create()
        if not self.h:
            raise BdsError("bds_timer_create failed")

    def __del__(self):
        try:
            if self.h and _lib is not None:
                _lib.bds_timer_destroy(self.h)
        except Exception:
            pass


GRAPH_MARKS = {}   # name -> [(start mark, stop mark)] captured inside hipGraphs while timers were enabled


def graph_mark_samples(name: str, marks=None):
    """Milliseconds between every captured mark pair of ``name`` as last recorded by a replay (call after a synchronise; pairs of
    graphs that have not been replayed yet give negative values and are dropped).  ``marks``: the dict a capture collected
    (``graph_view.FrameGraph.marks``), default: everything captured so far."""
    out = []
    for s, e in (GRAPH_MARKS if marks is None else marks).get(name, []):
        ms = float(lib().bds_timer_elapsed_ms(s.h, e.h))
        if ms >= 0.0:
            out.append(ms)
    return out


def timer_summary() -> dict:
    """name -> (calls, mean ms); call after torch.cuda.synchronize()."""
    out = {}
    for k, evs in (TIMERS or {}).items():
        ms = [s.elapsed_time(e) for s, e in evs]
        out[k] = (len(ms), sum(ms) / max(len(ms), 1))
    return out
