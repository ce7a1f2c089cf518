"""ctypes binding of libsmrf_hip.so (the C ABI declared in include/smrf_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, an exception is
raised.  Build the library with ``python -m neilpy_amd.build`` (or ``__graft_entry__.build()``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NEILPY_AMD_LIB: another build of the same library (developer A/B runs of bench.py on one box; never a CPU path)
LIB_PATH = os.environ.get("NEILPY_AMD_LIB") or os.path.join(_HERE, "_lib", "libsmrf_hip.so")

IMPL_AUTO, IMPL_RING, IMPL_DIRECT = 0, 1, 2
ROUTE_TWO_PASS, ROUTE_FUSED, ROUTE_DIRECT, ROUTE_COPY, ROUTE_CHAIN = 0, 1, 2, 3, 4
RING_MAX_RADIUS = 64


class SmrfHipError(RuntimeError):
    pass


_p = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_d = C.c_double
_sz = C.c_size_t

# name -> (restype, argtypes); mirrors include/smrf_hip.h one to one
SIGNATURES = {
    "smrf_abi_version": (_i, []),
    "smrf_last_error": (C.c_char_p, []),
    "smrf_device_count": (_i, []),
    "smrf_switches_reload": (None, []),
    "smrf_disk_filter_f32": (_i, [_p, _p, _i, _i, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "smrf_disk_filter_f64": (_i, [_p, _p, _i, _i, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "smrf_pf_dilate_flag_f32": (_i, [_p, _p, _p, _p, _p, _d, _i, _i, _i, _i64, _i, _i, _i, _i, _i, _i, _i, _p]),
    "smrf_pf_dilate_flag_f64": (_i, [_p, _p, _p, _p, _p, _d, _i, _i, _i, _i64, _i, _i, _i, _i, _i, _i, _i, _p]),
    "smrf_fused_open_supported": (_i, [_i, _i]),
    "smrf_pf_open_flag_f32": (_i, [_p, _p, _p, _p, _d, _i, _i, _i, _i64, _i, _i, _i, _i, _i, _p]),
    "smrf_pf_open_flag_f64": (_i, [_p, _p, _p, _p, _d, _i, _i, _i, _i64, _i, _i, _i, _i, _i, _p]),
    "smrf_pf_chain_length": (_i, [_i, _p, _i, _i64]),
    "smrf_pf_chain_flag_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i64, _i, _i, _i, _i, _p]),
    "smrf_pf_chain_flag_f64": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i64, _i, _i, _i, _i, _p]),
    "smrf_progressive_filter_workspace_bytes": (_sz, [_i, _i, _i]),
    "smrf_progressive_filter_f32": (_i, [_p, _i, _i, _p, _p, _i, _p, _p, _p, _sz, _i, _i, _p]),
    "smrf_progressive_filter_f64": (_i, [_p, _i, _i, _p, _p, _i, _p, _p, _p, _sz, _i, _i, _p]),
    "smrf_progressive_filter_timed_f32": (_i, [_p, _i, _i, _p, _p, _i, _p, _p, _p, _sz, _i, _i, _p, _p, _p]),
    "smrf_progressive_filter_timed_f64": (_i, [_p, _i, _i, _p, _p, _i, _p, _p, _p, _sz, _i, _i, _p, _p, _p]),
    "smrf_count_nan_f32": (_i, [_p, _i64, C.POINTER(_i64), _p]),
    "smrf_count_nan_f64": (_i, [_p, _i64, C.POINTER(_i64), _p]),
    "smrf_points_extent_f64": (_i, [_p, _p, _i64, C.POINTER(_d), _p, _sz, _p]),
    "smrf_las_decode_xyz_f64": (_i, [_p, _i64, _i, C.POINTER(_d), _p, _p, _p, _p]),
    "smrf_affine_apply_f64": (_i, [_p, _p, _i64, C.POINTER(_d), _p, _p, _p]),
    "smrf_points_band_count_f64": (_i, [_p, _p, _i64, C.POINTER(_d), _i, _i, _p, _p]),
    "smrf_points_band_pack_f64": (_i, [_p, _p, _p, _i64, C.POINTER(_d), _i, _i, _p, _p, _p, _p, _p]),
    "smrf_grid_clear_u64": (_i, [_p, _i64, _p]),
    "smrf_grid_bin_f64": (_i, [_p, _p, _p, _i64, C.POINTER(_d), C.POINTER(_d), _p, _i, _i, _i, _i, _i, _p, _p]),
    "smrf_grid_finalize_f64": (_i, [_p, _p, _p, _i64, _i, _p]),
    "smrf_springs_workspace_bytes": (_sz, [_i, _i]),
    "smrf_springs_lsqr_f64": (_i, [_p, _i, _i, _d, _d, _d, _i64, C.POINTER(_i), C.POINTER(_i64),
                                   C.POINTER(_i64), _p, _sz, _p]),
    "smrf_fda_workspace_bytes": (_sz, [_i, _i]),
    "smrf_fda_lsqr_f64": (_i, [_p, _i, _i, _d, _d, _d, _i64, C.POINTER(_i), C.POINTER(_i64),
                               C.POINTER(_i64), _p, _sz, _p]),
    "smrf_fda_apply_f64": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "smrf_springs_band_workspace_bytes": (_sz, [_i, _i]),
    "smrf_springs_band_layout": (_i, [_i, _i, C.POINTER(_i64)]),
    "smrf_springs_band_begin": (_i, [_i, _i, _d, _d, _d, _i64, _p, _sz, _p]),
    "smrf_springs_band_phase": (_i, [_i, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "smrf_springs_band_status": (_i, [_p, _i, _i, C.POINTER(_i), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i), _p]),
    "smrf_gradient_slope_f64": (_i, [_p, _p, _i, _i, _d, _p]),
    "smrf_pssm_f64": (_i, [_p, _p, _p, _p, _i, _i, _d, _d, _p]),
    "smrf_spline_solve_f64": (_i, [_p, _i, _i, _p, _p, _p]),
    "smrf_spline_solve_ws_f64": (_i, [_p, _p, _i, _i, _p, _p, _p]),
    "smrf_spline_eval_f64": (_i, [_p, _i, _i, _p, _p, _p, _p, _i64, _p, _p]),
    "smrf_classify_points_f64": (_i, [_p, _p, _p, _i64, _d, _d, _p, _p]),
    "smrf_negate_f64": (_i, [_p, _p, _i64, _p]),
    "smrf_mask_apply_f64": (_i, [_p, _p, _p, _p, _p, _i64, _p]),
}

_lib = None


def load():
    """dlopen the library once and attach the prototypes; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SmrfHipError(
            "libsmrf_hip.so is not built (%s missing). Run `python -m neilpy_amd.build`; "
            "neilpy_amd has no CPU fallback." % LIB_PATH)
    # torch first: its wheel carries its own libamdhip64 (same SONAME as /opt/rocm's, other file name).  Loaded after
    # this library it would be a SECOND HIP runtime in the process, and the one that initialises last sees no device
    # (`python __graft_entry__.py smoke`: build() loads the library before anything imports torch).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    override = bool(os.environ.get("NEILPY_AMD_LIB"))
    for name, (res, args) in SIGNATURES.items():
        if override and not hasattr(lib, name):
            continue                 # an older build under A/B may lack newer entry points; the product library may not
        fn = getattr(lib, name)      # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    if lib.smrf_abi_version() != 1:
        raise SmrfHipError("libsmrf_hip ABI version %d, expected 1" % lib.smrf_abi_version())
    _lib = lib
    return lib


def reload_switches():
    """have the library read its SMRF_* environment switches again (it reads them once, at load): tests and A/B tools
    that change them inside a process"""
    lib = load()
    if hasattr(lib, "smrf_switches_reload"):      # an older build under NEILPY_AMD_LIB reads its environment per call
        lib.smrf_switches_reload()


def check(rc):
    if rc != 0:
        msg = load().smrf_last_error().decode("utf-8", "replace")
        raise SmrfHipError("libsmrf_hip error %d: %s" % (rc, msg))


def require_gpu():
    import torch
    if not torch.cuda.is_available() or load().smrf_device_count() < 1:
        raise SmrfHipError("no HIP device visible: neilpy_amd runs on MI355X only (no CPU fallback)")
