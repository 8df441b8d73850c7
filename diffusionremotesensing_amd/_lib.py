"""ctypes binding of the C-ABI in include/drs_hip.h (libdrs_hip.so, built in-tree by
`__graft_entry__.build()` / `python -m diffusionremotesensing_amd.build`).

There is deliberately no fallback: if the shared library is missing or a call returns a
non-zero status, a RuntimeError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdrs_hip.so")

IMPL_DIRECT, IMPL_MFMA_F32, IMPL_MFMA_BF16X3, IMPL_MFMA_F16 = 0, 1, 2, 3
IMPL_BY_NAME = {"direct": IMPL_DIRECT, "mfma_f32": IMPL_MFMA_F32, "mfma_bf16x3": IMPL_MFMA_BF16X3,
                "mfma_f16": IMPL_MFMA_F16}
FWD_REUSE_COND = 1
PLAN_KEEP_ALL = 1
PLAN_TRAIN = 2
VARIANT_SUPERRES, VARIANT_SAR_TO_NDVI, VARIANT_GENERATION = 0, 1, 2


class UNetConfig(C.Structure):
    _fields_ = [("batch", C.c_int), ("lr_batch", C.c_int), ("image_channels", C.c_int), ("out_dim", C.c_int),
                ("height", C.c_int), ("width", C.c_int), ("magnification", C.c_int), ("impl", C.c_int),
                ("bn_eps", C.c_float), ("flags", C.c_int), ("variant", C.c_int),
                ("cond_channels", C.c_int), ("num_classes", C.c_int)]


# name -> (restype, argtypes); kept in one table so the symbol-export test can walk it
_P, _I, _L, _Z, _F = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float
SIGNATURES = {
    "drs_last_error": (C.c_char_p, []),
    "drs_abi_version": (_I, []),
    "drs_noise_images": (_I, [_P, _P, _P, _P, _I, _P, _I, _L, _P]),
    "drs_sampler_step": (_I, [_P, _P, _P, _I, _P, _P, _P, _I, _L, _P]),
    "drs_sampler_step_cfg": (_I, [_P, _P, _P, C.c_float, _P, _I, _P, _P, _P, _I, C.c_int64, _P]),
    "drs_adam_multi": (_I, [_P, _I, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, _P]),
    "drs_ema_multi": (_I, [_P, _I, C.c_int64, C.c_double, _I, _P]),
    "drs_downblur_scratch_bytes": (_Z, [_I, _I, _I, _I, _I, _I]),
    "drs_downblur_u8": (_I, [_P, _I, _I, _I, _I, _I, _I, C.c_float, _P, _P, _P, _Z, _P]),
    "drs_add_noise_clip_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "drs_aggregate_tiles": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "drs_conv2d_workspace_bytes": (_Z, [_I] * 11),
    "drs_conv2d_nchw": (_I, [_P, _P, _P, _P] + [_I] * 12 + [_P, _Z, _I, _P]),
    "drs_upconv_fused_workspace_bytes": (_Z, [_I] * 5),
    "drs_upconv_fused_nchw": (_I, [_P] * 9 + [_I] + [_P, _P] + [_I] * 5 + [_P, _Z, _P]),
    "drs_bicubic_upsample_nchw": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "drs_time_mlp": (_I, [_P] * 7 + [_I, _I, _I, _P]),
    "drs_unet_plan_create": (_I, [C.POINTER(_P), C.POINTER(UNetConfig)]),
    "drs_unet_plan_destroy": (None, [_P]),
    "drs_unet_num_params": (_I, [_P]),
    "drs_unet_param_name": (C.c_char_p, [_P, _I]),
    "drs_unet_param_numel": (_L, [_P, _I]),
    "drs_unet_packed_bytes": (_Z, [_P]),
    "drs_unet_workspace_bytes": (_Z, [_P]),
    "drs_unet_pack_weights": (_I, [_P, C.POINTER(_P), C.POINTER(_F), _P, _Z, _P]),
    "drs_unet_forward": (_I, [_P, _P, _P, _P, _P, _P, _P, _Z, _I, _P]),
    "drs_unet_forward_labels": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _Z, _I, _P]),
    "drs_unet_num_tensors": (_I, [_P]),
    "drs_unet_tensor_name": (C.c_char_p, [_P, _I]),
    "drs_unet_tensor_shape": (_I, [_P, _I] + [C.POINTER(_I)] * 4),
    "drs_unet_read_tensor": (_I, [_P, _I, _P, _P, _P]),
    "drs_unet_packed_bwd_bytes": (_Z, [_P]),
    "drs_unet_backward": (_I, [_P, _P, _P, _Z, _P, _P, _P, C.POINTER(_P), _P, _Z, _P]),
    "drs_unet_backward_labels": (_I, [_P, _P, _P, _Z, _P, _P, _P, _I, _P, C.POINTER(_P), _P, _Z, _P]),
    "drs_unet_check_faults": (_I, [_P, _P, _P]),
    "drs_unet_profile_enable": (_I, [_P, _I]),
    "drs_unet_profile_num_ops": (_I, [_P]),
    "drs_unet_profile_read": (_I, [_P, _I, C.c_char_p, _I, C.POINTER(_F), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "drs_unet_profile_num_launches": (_I, [_P]),
    "drs_unet_profile_launch": (_I, [_P, _I, C.c_char_p, _I, C.c_char_p, _I]),
}

_lib = None


def lib_path():
    """The shared object load() binds (DRS_LIB overrides the in-tree build: A/B experiments with builds of the same ABI)."""
    return os.environ.get("DRS_LIB", LIB_PATH)


def load():
    """Return the loaded library (cached).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP kernels are not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (needs hipcc, cross-compiles for gfx950 without a GPU). There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class RangeFault(RuntimeError):
    """DRS_ERR_RANGE from drs_unet_check_faults: an activation left fp16's range in the FL arithmetic; the plan has switched
    itself to the split-bf16 kernels, the forwards since the last check must be run again."""


def check(status, what):
    if status != 0:
        msg = load().drs_last_error()
        text = f"{what} failed with status {status}: {msg.decode() if msg else '?'}"
        if status == 6:
            raise RangeFault(text)
        raise RuntimeError(text)
