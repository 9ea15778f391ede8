"""ctypes binding of libsupnerf_hip.so (C ABI in include/supnerf_hip.h).

There is no CPU fallback: if the library is missing or an entry point fails, the caller gets an
exception.  The library is built in-tree by ``build.py`` (hipcc, gfx950)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsupnerf_hip.so")

Z_SHARED, Z_PER_OBJECT, Z_PER_RAY, Z_BOX = 0, 1, 2, 3
WHITE_BKGD, METRIC_Z, BOX_DETACH = 1, 2, 4
FP32, BF16X3 = 0, 1
ERRORS = {-1: "SNR_E_ARG", -2: "SNR_E_SHAPE", -3: "SNR_E_WORKSPACE", -4: "SNR_E_LAUNCH", -5: "SNR_E_UNSUPPORTED"}


class SnrError(RuntimeError):
    pass


class RenderArgs(C.Structure):
    _fields_ = [("rays_o", C.c_void_p), ("rays_d", C.c_void_p), ("t_vals", C.c_void_p), ("xyz_div", C.c_void_p),
                ("z_scale", C.c_void_p), ("latent", C.c_void_p), ("packed", C.c_void_p),
                ("frame", C.c_float * 9), ("xyz_mul", C.c_float), ("z_mode", C.c_int32), ("flags", C.c_int32),
                ("n_rays", C.c_int64), ("rays_per_obj", C.c_int64), ("n_samples", C.c_int32),
                ("shape_blocks", C.c_int32), ("texture_blocks", C.c_int32), ("precision", C.c_int32),
                ("latent_bias", C.c_void_p), ("box_half", C.c_void_p), ("rng_seed", C.c_uint64), ("rng_offset", C.c_uint64),
                ("rng_threads", C.c_uint64)]


_lib = None


def header_abi_version() -> int:
    """SNR_ABI_VERSION as include/supnerf_hip.h declares it: the single source of truth the loader, build() and the tests
    compare the library against."""
    import re
    with open(os.path.join(_HERE, "..", "include", "supnerf_hip.h")) as f:
        m = re.search(r"#define\s+SNR_ABI_VERSION\s+(\d+)", f.read())
    if m is None:
        raise SnrError("include/supnerf_hip.h does not define SNR_ABI_VERSION")
    return int(m.group(1))

_P = C.c_void_p
_SIGS = {
    "snr_abi_version": (C.c_int, []),
    "snr_last_hip_error": (C.c_char_p, []),
    "snr_packed_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "snr_pack_weights": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, C.c_int, _P, _P]),
    "snr_mask_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "snr_precision_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64]),
    "snr_decoder_fwd": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int64, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, _P]),
    "snr_decoder_bwd_ws_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int, C.c_int]),
    "snr_decoder_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int64, C.c_int, C.c_int, _P, _P, _P, _P, _P,
                                  C.c_size_t, C.c_int, _P]),
    "snr_render_fwd": (C.c_int, [C.POINTER(RenderArgs), _P, _P, _P, _P, _P, _P, _P]),
    "snr_render_bwd_ws_bytes": (C.c_size_t, [C.POINTER(RenderArgs)]),
    "snr_render_bwd": (C.c_int, [C.POINTER(RenderArgs), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "snr_scene_composite_fwd": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "snr_composite_fwd": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, _P, _P, _P, _P]),
    "snr_composite_bwd": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, _P, _P, _P, _P, _P, _P, _P]),
    "snr_encode_fwd": (C.c_int, [C.POINTER(RenderArgs), _P, _P, _P, _P, _P, _P, _P]),
    "snr_pe_points": (C.c_int, [_P, _P, C.c_int64, _P, _P]),
    "snr_loss_tail_fwd": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int64, C.c_float, _P, _P]),
    "snr_loss_tail_bwd": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int64, C.c_float, _P, _P, _P, _P]),
    "snr_weight_grad_ws_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "snr_weight_grad": (C.c_int, [_P, C.c_int64, C.c_int, _P, C.c_int64, C.c_int, C.c_int64, _P, C.c_int64, _P, C.c_int, _P, C.c_size_t, _P]),
    "snr_pose_rays_fwd": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int64, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "snr_pose_rays_bwd": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_int, _P, _P, _P, _P, _P, _P]),
    "snr_cam_rays_fwd": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int64, C.c_int, _P, _P, _P, _P]),
    "snr_cam_rays_bwd": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, _P, _P, _P]),
    "snr_metric_row": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P, _P, C.c_int64, C.c_int, _P, _P, _P]),
    "snr_adamw_step": (C.c_int, [C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(C.c_float),
                                 C.c_int, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "snr_latent_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_int, _P, _P, _P]),
    "snr_latent_bwd": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, C.c_int, _P, _P, _P]),
    "snr_adamw_table_step": (C.c_int, [_P, C.c_int, C.c_int64, C.POINTER(C.c_float), C.c_int, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
}


def exported_symbols():
    """Every entry point include/supnerf_hip.h declares."""
    return list(_SIGS.keys())


def lib():
    """The loaded library (loads on first use; raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SnrError(f"{LIB_PATH} is missing: run `python sup-nerf_amd/build.py` (hipcc, gfx950). "
                           "There is no CPU fallback for the product path.")
        # torch (if imported) has already loaded its libamdhip64.so.7, which this library binds to by soname
        l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)       # AttributeError if the header and the library disagree
            fn.restype, fn.argtypes = res, args
        want = header_abi_version()
        if l.snr_abi_version() != want:
            raise SnrError(f"libsupnerf_hip.so ABI version {l.snr_abi_version()} != include/supnerf_hip.h's {want}: rebuild "
                           "(python sup-nerf_amd/build.py --force)")
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        msg = ERRORS.get(rc, str(rc))
        if rc == -4:
            msg += ": " + lib().snr_last_hip_error().decode()
        raise SnrError(f"{what} failed: {msg}")
