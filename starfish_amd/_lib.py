"""
ctypes binding of ``libstarfish_amd.so`` (include/starfish_amd.h).

There is deliberately NO CPU fallback: every numerical entry point of this package goes through the
HIP kernels, and :func:`require_gpu` raises when the shared library or a gfx950 device is missing.
PyTorch is used only to own device memory / streams (``tensor.data_ptr()``).
"""

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SF_LIB_PATH: development hook for tools/ -- load another build of the SAME library (the `make TUNING=1` variant with
# the environment tuning switches compiled in).  The release library itself reads no environment variable.
LIB_PATH = os.path.join(_HERE, "libstarfish_amd.so")
if os.environ.get("SF_LIB_PATH"):
    import warnings

    LIB_PATH = os.environ["SF_LIB_PATH"]
    warnings.warn(f"starfish_amd: SF_LIB_PATH overrides the packaged library: loading {LIB_PATH}", RuntimeWarning)

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class OrderDesc(C.Structure):
    _fields_ = [
        ("n", C.c_int32),
        ("nf", C.c_int32),
        ("m", C.c_int32),
        ("n_grid", C.c_int32),
        ("M", C.c_int32),
        ("reserved", C.c_int32),
        ("wave", c_double_p),
        ("flux", c_double_p),
        ("sigma", c_double_p),
        ("min_dv_wave", c_double_p),
        ("bulk_fluxes", c_double_p),
        ("grid_points", c_double_p),
        ("variances", c_double_p),
        ("lengthscales", c_double_p),
        ("v11", c_double_p),
        ("w_hat", c_double_p),
        ("linv", c_double_p),
        ("alpha", c_double_p),
    ]


class ModelDesc(C.Structure):
    _fields_ = [
        ("has_vsini", C.c_int32),
        ("has_vz", C.c_int32),
        ("has_log_scale", C.c_int32),
        ("has_global", C.c_int32),
        ("n_local", C.c_int32),
        ("n_cheb", C.c_int32),
        ("use_sigma_w", C.c_int32),
        ("has_av", C.c_int32),
    ]


class Segment(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("d_params", C.c_void_p), ("B", C.c_int32), ("reserved", C.c_int32)]


# name -> (restype, argtypes); kept in one table so the CPU test-suite can check that every symbol
# declared in include/starfish_amd.h is exported.
_VP = C.c_void_p
SIGNATURES = {
    "sf_version": (C.c_char_p, []),
    "sf_last_error": (C.c_char_p, []),
    "sf_device_count": (C.c_int, []),
    "sf_global_cov": (C.c_int, [_VP, C.c_int, C.c_double, C.c_double, _VP, _VP]),
    "sf_local_cov": (C.c_int, [_VP, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, _VP, _VP]),
    "sf_rotational_broaden": (
        C.c_int,
        [_VP, C.c_int, C.c_int, C.c_double, C.c_double, _VP, _VP, C.c_size_t, _VP],
    ),
    "sf_instrumental_broaden": (
        C.c_int,
        [_VP, C.c_int, C.c_int, C.c_double, C.c_double, _VP, _VP, C.c_size_t, _VP],
    ),
    "sf_fft_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "sf_resample": (
        C.c_int,
        [c_double_p, C.c_int, _VP, C.c_int, _VP, C.c_int, _VP, _VP, C.c_size_t, _VP],
    ),
    "sf_resample_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "sf_chebyshev_correct": (
        C.c_int,
        [_VP, C.c_int, C.c_double, _VP, C.c_int, c_double_p, C.c_int, _VP, _VP],
    ),
    "sf_extinct_ccm89": (C.c_int, [_VP, C.c_int, _VP, C.c_int, C.c_double, C.c_double, _VP, _VP]),
    "sf_extinct": (C.c_int, [_VP, C.c_int, _VP, C.c_int, C.c_double, C.c_double, C.c_int, _VP, _VP]),
    "sf_emulator_v11_build": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, _VP, _VP, _VP, C.c_int, C.c_int, _VP]),
    "sf_potrf_batch": (
        C.c_int,
        [_VP, C.c_int, C.c_int, C.c_int64, C.c_int, _VP, _VP, C.c_size_t, _VP],
    ),
    "sf_potrf_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "sf_logdet_sqmah_batch": (
        C.c_int,
        [_VP, C.c_int, C.c_int, C.c_int64, C.c_int, _VP, C.c_int, _VP, C.c_size_t, _VP, _VP, _VP],
    ),
    "sf_ctx_create": (_VP, [C.POINTER(OrderDesc), C.c_int, c_int_p]),
    "sf_ctx_destroy": (None, [_VP]),
    "sf_ctx_npad": (C.c_int, [_VP]),
    "sf_ctx_lda": (C.c_int, [_VP]),
    "sf_param_stride": (C.c_int, [_VP, C.POINTER(ModelDesc)]),
    "sf_workspace_bytes": (C.c_size_t, [_VP, C.POINTER(ModelDesc), C.c_int]),
    "sf_emulator_query_batch": (
        C.c_int,
        [_VP, C.POINTER(ModelDesc), C.c_int, _VP, _VP, _VP, _VP, _VP, C.c_size_t, _VP],
    ),
    "sf_transform_batch": (
        C.c_int,
        [_VP, C.POINTER(ModelDesc), C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_size_t, _VP],
    ),
    "sf_forward_batch": (
        C.c_int,
        [_VP, C.POINTER(ModelDesc), C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, C.c_size_t, _VP],
    ),
    "sf_cov_fill_batch": (
        C.c_int,
        [_VP, C.POINTER(ModelDesc), C.c_int, _VP, _VP, C.c_int, C.c_int64, C.c_int, C.c_int, _VP, _VP, C.c_size_t, _VP],
    ),
    "sf_emulator_joint_batch": (
        C.c_int,
        [_VP, C.POINTER(ModelDesc), C.c_int, _VP, _VP, _VP, _VP, _VP, C.c_size_t, _VP],
    ),
    "sf_loglike_batch": (
        C.c_int,
        [_VP, C.POINTER(ModelDesc), C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_size_t, _VP],
    ),
    "sf_multi_workspace_bytes": (C.c_size_t, [C.POINTER(Segment), C.c_int, C.POINTER(ModelDesc)]),
    "sf_loglike_multi_batch": (
        C.c_int,
        [C.POINTER(Segment), C.c_int, C.POINTER(ModelDesc), _VP, _VP, _VP, _VP, _VP, _VP, C.c_size_t, _VP],
    ),
    "sf_banded_max_halfwidth": (C.c_int, [_VP]),
    "sf_banded_window_halfwidth": (C.c_int, [_VP]),
    "sf_banded_workspace_bytes": (C.c_size_t, [_VP, C.POINTER(ModelDesc), C.c_int, C.c_int]),
    "sf_loglike_banded_batch": (
        C.c_int,
        [_VP, C.POINTER(ModelDesc), C.c_int, _VP, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_size_t, _VP],
    ),
    "sf_band_logdet_gram_batch": (
        C.c_int,
        [_VP, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, _VP, C.c_int, C.c_int, C.c_int64, _VP, _VP, _VP, _VP],
    ),
    "sf_debug_clock_probe": (C.c_int, [_VP, C.c_longlong, _VP]),
    "sf_debug_stream_write": (C.c_int, [_VP, C.c_size_t, C.c_double, _VP]),
    "sf_debug_cholesky_sequence": (C.c_int, [C.c_int]),
    "sf_persistent_potrf": (C.c_int, [C.c_int]),
    "sf_persistent_potrf_status": (C.c_int, [C.POINTER(C.c_longlong)]),
    "sf_profile_enable": (C.c_int, [C.c_int]),
    "sf_profile_read": (C.c_int, [c_double_p, c_double_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
}

_lib = None


class StarfishAMDError(RuntimeError):
    """The HIP library is missing, no MI355X is visible, or a C-ABI call failed."""


def load():
    """Load the shared library (no GPU needed for loading / symbol checks)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise StarfishAMDError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C starfish_amd/csrc`.  There is no CPU fallback."
        )
    # torch ships its own libamdhip64.so.7 (same SONAME as /opt/rocm's): load torch FIRST so that the
    # process holds exactly one HIP runtime, shared by torch's allocator/streams and these kernels.
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if not hasattr(lib, name):
            # development hook only: an OLDER build of the library in a same-box A/B lacks the newest entry points.  It has
            # to be asked for (SF_ALLOW_OLD_LIB=1 next to SF_LIB_PATH) and every missing name is reported: a stale or
            # mismatched library must not load silently and fail later inside an error path.
            if os.environ.get("SF_LIB_PATH") and os.environ.get("SF_ALLOW_OLD_LIB") == "1":
                import warnings

                warnings.warn(f"starfish_amd: {LIB_PATH} lacks {name} (SF_ALLOW_OLD_LIB=1: skipped)", RuntimeWarning)
                continue
            raise StarfishAMDError(f"{LIB_PATH} does not export {name}: stale or mismatched build of the library "
                                   "(rebuild with `make -C starfish_amd/csrc`)")
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def require_gpu():
    """Return the library, raising loudly when the HIP path cannot run."""
    lib = load()
    if lib.sf_device_count() <= 0:
        raise StarfishAMDError(
            "no HIP device visible: starfish_amd computes only on MI355X (gfx950); "
            "there is no CPU fallback"
        )
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().sf_last_error().decode(errors="replace")
        raise StarfishAMDError(f"{what} failed (code {rc}): {msg}")


def as_double_p(arr):
    return arr.ctypes.data_as(c_double_p)
