"""ctypes binding of libvings_hip.so (the C ABI declared in include/vings_hip.h).

There is no CPU fallback: if the shared object is missing or a call fails the
caller gets an exception.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# VINGS_HIP_LIB points at an alternative build of the same library (kernel A/B experiments)
LIB_PATH = os.environ.get("VINGS_HIP_LIB") or os.path.join(_HERE, "libvings_hip.so")

_c = ctypes
_P = _c.c_void_p

_SIGNATURES = {
    "vings_last_error": (_c.c_char_p, []),
    "vings_abi_version": (_c.c_int, []),
    "vings_surfel_project": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_float] + [_P] * 14 +
                             [_c.POINTER(_c.c_int64), _c.POINTER(_c.c_uint32), _P]),
    "vings_surfel_project_launch": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_float] + [_P] * 15),
    "vings_surfel_project_launch_activate": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_float] + [_P] * 19),
    "vings_surfel_project_enqueue_activate": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_float] + [_P] * 20),
    "vings_surfel_project_backward_finish_guarded": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_float] +
                                                     [_P] * 14 + [_c.c_int64] + [_P] * 7 + [_c.c_uint32, _P]),
    "vings_sparse_adam_multi_guarded": (_c.c_int, [_c.c_int] + [_P] * 7 + [_c.c_float] * 3 +
                                        [_c.c_int, _P, _c.c_uint32, _P]),
    "vings_surfel_set_tight_binning": (_c.c_int, [_c.c_int]),
    "vings_surfel_set_tight_binning_thread": (_c.c_int, [_c.c_int]),
    "vings_surfel_project_wait": (_c.c_int, [_c.POINTER(_c.c_int64), _c.POINTER(_c.c_uint32)]),
    "vings_surfel_bin_low_bits": (_c.c_int, [_c.c_uint32, _c.c_uint32]),
    "vings_surfel_bin_speculative": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_int64, _c.c_int] +
                                     [_P] * 9 + [_c.c_size_t, _P]),
    "vings_surfel_bin_workspace_bytes": (_c.c_size_t, [_c.c_int64]),
    "vings_surfel_bin": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_int64, _c.c_uint32,
                                    _c.c_uint32] + [_P] * 8 + [_c.c_size_t, _P]),
    "vings_surfel_blend_forward": (_c.c_int, [_c.c_int, _c.c_int] + [_P] * 10),
    "vings_surfel_blend_forward_lean": (_c.c_int, [_c.c_int, _c.c_int] + [_P] * 10),
    "vings_surfel_blend_backward_accumulate_lean": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int] + [_P] * 11),
    "vings_surfel_tile_order": (_c.c_int, [_c.c_int, _c.c_int] + [_P] * 3),
    "vings_surfel_blend_forward_ordered": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int] + [_P] * 11),
    "vings_surfel_blend_backward_accumulate_ordered": (_c.c_int, [_c.c_int] * 4 + [_P] * 12),
    "vings_surfel_blend_backward_accumulate_stable": (_c.c_int, [_c.c_int] * 4 + [_P] * 13),
    "vings_surfel_blend_backward_auto": (_c.c_int, [_c.c_int] * 3 + [_P] * 13),
    "vings_surfel_blend_backward": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int] + [_P] * 11),
    "vings_surfel_blend_backward_accumulate": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int] + [_P] * 11),
    "vings_surfel_project_backward": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_float] +
                                      [_P] * 15),
    "vings_sparse_adam": (_c.c_int, [_P] * 5 + [_c.c_float] * 4 + [_c.c_int, _c.c_int, _P]),
    "vings_sparse_adam_multi": (_c.c_int, [_c.c_int] + [_P] * 7 + [_c.c_float] * 3 +
                                [_c.c_int, _P]),
    "vings_splat_activate": (_c.c_int, [_c.c_int] + [_P] * 7),
    "vings_surfel_project_backward_finish": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_float] +
                                             [_P] * 14 + [_c.c_int64, _c.c_int] + [_P] * 10),
    "vings_splat_grad_finish": (_c.c_int, [_c.c_int] + [_P] * 15 + [_c.c_int64, _c.c_int] + [_P] * 7),
    "vings_surfel_blend_importance": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int] + [_P] * 5 +
                                      [_c.c_int, _P, _P, _P]),
    "vings_surfel_blend_importance_ordered": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int] + [_P] * 5 +
                                              [_c.c_int, _P, _P, _P, _P]),
    "vings_splat_compact_workspace_bytes": (_c.c_size_t, [_c.c_int]),
    "vings_splat_compact_plan": (_c.c_int, [_c.c_int, _P, _P, _P, _c.c_size_t,
                                            _c.POINTER(_c.c_int64), _P]),
    "vings_splat_compact_apply": (_c.c_int, [_c.c_int, _P, _c.c_int, _P, _P, _P, _P]),
    "vings_splat_compact_static_max_rows": (_c.c_int, []),
    "vings_splat_compact_static_prepare": (_c.c_int, []),
    "vings_splat_compact_plan_static": (_c.c_int, [_c.c_int, _P, _P, _P, _P]),
    "vings_knn3_mean_dist2": (_c.c_int, [_c.c_int, _P, _P, _P]),
    "vings_keyframe_backproject": (_c.c_int, [_c.c_int] * 3 + [_P] * 3 + [_c.POINTER(_c.c_float)] +
                                   [_c.c_float] * 4 + [_P] * 3),
    "vings_depth_refresh": (_c.c_int, [_c.c_int, _c.c_int] + [_P] * 5),
    "vings_splat_top3_update": (_c.c_int, [_c.c_int, _P, _c.c_int, _P, _P, _P]),
    "vings_splat_top3_update_gated": (_c.c_int, [_c.c_int, _P, _c.c_int, _P, _P, _P, _c.c_float, _P]),
    "vings_splat_top3_merge": (_c.c_int, [_c.c_int, _c.c_int] + [_P] * 5),
    "vings_splat_reanchor": (_c.c_int, [_c.c_int, _P, _P, _P] + [_c.c_int] * 4 + [_P] * 5),
    "vings_splat_pack_exchange": (_c.c_int, [_c.c_int] + [_P] * 7 + [_c.c_int64] + [_P] * 3),
    "vings_splat_unpack_exchange": (_c.c_int, [_c.c_int] + [_P] * 10),
    "vings_splat_pack_columns": (_c.c_int, []),
    "vings_splat_merge_records": (_c.c_int, [_c.c_int] + [_P] * 5),
    "vings_splat_exchange_columns": (_c.c_int, []),
    "vings_splat_touched": (_c.c_int, [_c.c_int, _P, _P, _P]),
    "vings_splat_compact_plan_device": (_c.c_int, [_c.c_int, _P, _P, _P, _c.c_size_t, _P, _P]),
    "vings_surfel_project_backward_pack": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_float] +
                                           [_P] * 10 + [_c.c_int64] + [_P] * 5 + [_c.c_int, _P]),
    "vings_splat_exchange_step": (_c.c_int, [_c.c_int, _c.c_int] + [_P] * 6 + [_c.c_float] * 3 +
                                  [_P] * 4),
    "vings_depth_covariance": (_c.c_int, [_c.c_int] * 4 + [_P] * 8),
    "vings_upsample_maps": (_c.c_int, [_c.c_int] * 3 + [_P, _P, _c.c_int, _P, _P, _P]),
    "vings_package_workspace_bytes": (_c.c_size_t, [_c.c_int]),
    "vings_package_keyframes": (_c.c_int, [_c.c_int] * 3 + [_P] * 4 + [_c.c_float] * 2 + [_P] * 5 +
                                [_c.c_size_t, _P]),
    "vings_mapping_loss_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.c_int]),
    "vings_mapping_loss": (_c.c_int, [_c.c_int, _c.c_int] + [_P] * 5 +
                           [_c.POINTER(_c.c_float)] + [_c.c_float] * 4 +
                           [_c.POINTER(_c.c_float)] + [_P] * 4 + [_c.c_size_t, _P]),
    "vings_mapping_loss_gt_cache_bytes": (_c.c_size_t, [_c.c_int, _c.c_int]),
    "vings_mapping_loss_cached": (_c.c_int, [_c.c_int, _c.c_int] + [_P] * 5 +
                                  [_c.POINTER(_c.c_float)] + [_c.c_float] * 4 +
                                  [_c.POINTER(_c.c_float)] + [_P] * 4 + [_c.c_size_t, _P, _c.c_int, _P]),
}

_lib = None


class VingsError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VingsError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C vings_mono_amd/csrc` (hipcc, gfx950)")
        handle = ctypes.CDLL(LIB_PATH)
        missing = [name for name in _SIGNATURES if not hasattr(handle, name)]
        if missing:
            # a stale build: fail here, by name, not later with an AttributeError
            raise VingsError(f"{LIB_PATH} does not export {', '.join(sorted(missing))}: "
                             "rebuild it (`make -C vings_mono_amd/csrc`)")
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def bind(name, restype, argtypes):
    """Signature of an entry point that a module other than this one binds (dba.py)."""
    fn = getattr(lib(), name, None)
    if fn is None:
        raise VingsError(f"{LIB_PATH} does not export {name}: rebuild it "
                         "(`make -C vings_mono_amd/csrc`)")
    fn.restype = restype
    fn.argtypes = argtypes
    return fn


def check(rc, what):
    if rc != 0:
        msg = lib().vings_last_error().decode("utf-8", "replace")
        raise VingsError(f"{what} failed with status {rc}: {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL). The tensor must be contiguous."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor handed to the C ABI must be contiguous"
    return t.data_ptr()


def stream_ptr(device=None):
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu_tensor(t, name, dtype=None):
    if not t.is_cuda:
        raise VingsError(f"{name} must live on the GPU (got {t.device}); there is no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise VingsError(f"{name} must be {dtype} (got {t.dtype})")
