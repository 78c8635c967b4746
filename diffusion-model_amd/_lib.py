"""ctypes binding of libddpm_mi355x.so (declared in include/ddpm_mi355x.h).

There is deliberately no fallback: if the library is missing or a call fails, an exception is
raised. Device pointers are passed as integers (tensor.data_ptr()).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DDPM_LIB") or os.path.join(_HERE, "libddpm_mi355x.so")   # DDPM_LIB: diagnostic builds

P, I32, I64, F32, F64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double


class ConvDesc(C.Structure):
    _fields_ = [(n, I32) for n in ("batch", "in_h", "in_w", "c1", "ld1", "c2", "ld2", "cout", "ldo", "act", "ldr",
                                   "out_h", "out_w", "off_y", "off_x")]


class Epilogue(C.Structure):
    _fields_ = [("scale", P), ("shift", P), ("residual", P), ("temb", P)]


class EmbedBwdArgs(C.Structure):
    _fields_ = [("batch", I32), ("emb_dim", I32), ("n_labels", I32), ("n_classes", I32),
                ("emb", P), ("y", P), ("label_table", P),
                ("ch", I32 * 4), ("dtemb", P * 4), ("lw", P * 4), ("dw", P * 4), ("db", P * 4), ("dlw", P * 4), ("dlb", P * 4),
                ("dlabel_table", P)]


class EmbedArgs(C.Structure):
    _fields_ = [("batch", I32), ("emb_dim", I32), ("n_labels", I32), ("t_is_float", I32), ("n_classes", I32), ("uniform", I32),
                ("t", P), ("y", P), ("label_table", P),
                ("ch", I32 * 4), ("w", P * 4), ("b", P * 4), ("lw", P * 4), ("lb", P * 4),
                ("out", P * 4), ("emb_out", P)]


_SIGNATURES = {
    "ddpm_abi_version": [],
    "ddpm_schedule_betas": [I32, I32, F64, F64, P],
    "ddpm_schedule_tables": [I32, P, P, P],
    "ddpm_step_coefficients": [F32, F32, F32, P],
    "ddpm_reverse_step": [P, P, P, F32, P, F32, F32, F32, P, I64, P],
    "ddpm_philox_normal": [P, I64, C.c_uint64, C.c_uint64, P],
    "ddpm_final_conv_reverse_step": [P, I32, P, P, P, P, F32, I32, P, C.c_uint64, C.c_uint64, F32, F32, F32, P, P, I32, I32, I32, I32, I32, P],
    "ddpm_postprocess_images": [P, P, I64, P],
    "ddpm_make_grid": [P, I32, I32, I32, I32, I32, I32, F32, I32, F32, F32, P, P],
    "ddpm_add_noise": [P, P, P, P, I32, P, I32, I64, P],
    "ddpm_mse_partial_count": [],
    "ddpm_mse_fwd": [P, P, I64, P, P, P],
    "ddpm_mse_bwd": [P, P, P, I64, P, P],
    "ddpm_time_embed": [C.POINTER(EmbedArgs), P],
    "ddpm_sinusoidal_embedding": [P, I32, I32, I32, P, P],
    "ddpm_conv3x3_fwd": [C.POINTER(ConvDesc), P, P, P, C.POINTER(Epilogue), P, P],
    "ddpm_conv1x1_fwd": [C.POINTER(ConvDesc), P, P, P, C.POINTER(Epilogue), P, P],
    "ddpm_convT2x2_fwd": [C.POINTER(ConvDesc), P, P, C.POINTER(Epilogue), P, P],
    "ddpm_pack_conv3x3": [P, I32, I32, I32, P, P],
    "ddpm_pack_conv3x3_batch": [I32, P, P, P, P, P, P],
    "ddpm_pack_convT2x2": [P, I32, I32, P, P],
    "ddpm_pack_conv1x1": [P, I32, I32, I32, P, P],
    "ddpm_pack_qkv": [P, P, P, P, P, P, I32, I32, I32, P, P, P, P],
    "ddpm_conv3x3_first_fwd": [P, P, C.POINTER(Epilogue), I32, P, I32, I32, I32, I32, I32, I32, P],
    "ddpm_conv1x1_final_fwd": [P, I32, P, P, P, I32, I32, I32, I32, I32, P],
    "ddpm_pool_skip_fwd": [P, I32, P, P, I32, P, I32, P, I32, I32, I32, I32, P],
    "ddpm_add_temb": [P, I32, P, P, I32, I32, I32, I32, P],
    "ddpm_bn_stats": [P, I32, I64, I32, F32, F32, P, P, P, P, P, P],
    "ddpm_bn_stats_noalloc_max_channels": [],
    "ddpm_bn_stats_noalloc": [P, I32, I64, I32, F32, F32, P, P, P, P, P],
    "ddpm_bn_partial_sums": [P, I32, I64, I32, P, P, P],
    "ddpm_bn_sums_from_partials": [P, I32, I32, P, P],
    "ddpm_bn_finalize_stats": [P, I64, I32, F32, F32, P, P, P, P, P],
    "ddpm_bn_bwd_sums": [P, I32, P, I32, P, I32, P, P, P, P, I32, I64, I32, P, P, P],
    "ddpm_bn_bwd_apply": [P, I32, P, I32, P, I32, P, P, P, P, P, I64, I32, I64, I32, P, P, I32, P, I32, P, P],
    "ddpm_bn_apply_fwd": [P, I32, P, P, P, P, P, I32, P, I32, P, I32, I32, I32, I32, P],
    "ddpm_bn_bwd": [P, I32, P, I32, P, I32, P, P, P, P, I32, I64, I32, P, P, P, P, I32, P, I32, P, I32, P, P],
    "ddpm_colsum": [P, I32, I32, I64, I32, P, I32, P, P],
    "ddpm_conv3x3_wgrad": [P, I32, I32, P, I32, I32, I32, I32, I32, P, P, P],
    "ddpm_conv1x1_wgrad": [P, I32, I32, P, I32, I32, I64, P, P],
    "ddpm_pack_convT2x2_dgrad": [P, I32, I32, P, P],
    "ddpm_convT2x2_dgrad": [C.POINTER(ConvDesc), P, P, C.POINTER(Epilogue), P, P],
    "ddpm_convT2x2_wgrad": [P, I32, I32, I32, I32, I32, I32, P, I32, I32, I32, I32, I32, P, P, P],
    "ddpm_conv3x3_first_wgrad": [P, I32, P, P, P, I32, I32, I32, I32, I32, P],
    "ddpm_conv1x1_final_bwd": [P, P, I32, P, P, P, I32, P, P, I32, I32, I32, I32, I32, P],
    "ddpm_pool_skip_bwd": [P, I32, P, I32, P, P, I32, I32, I32, I32, I32, P],
    "ddpm_colsum_window": [P, I32, I32, I32, I32, I32, I32, I32, I32, I32, P, P, P],
    "ddpm_time_embed_bwd": [C.POINTER(EmbedBwdArgs), P],
    "ddpm_attention_bwd": [P, P, I32, P, P, P, P, P, P, P, I32, I32, I32, I32, P],
    "ddpm_attention_fwd": [P, P, I32, P, P, I32, P, P, P, I32, I32, I32, I32, P],
    "ddpm_self_attention_infer": [P, I32, P, P, P, P, P, I32, P, I32, I32, I32, I32, P],
    "ddpm_softmax_rows": [P, I64, I32, P, P],
    "ddpm_inception_split_scores": [P, I64, I32, I32, P, P],
    "ddpm_mean_cov": [P, I64, I32, P, P, P, P],
    "ddpm_adam_step": [I32, P, P, P, P, P, F64, F64, F64, F64, F64, I64, P],
    "ddpm_conv3x3_fwd_stats": [C.POINTER(ConvDesc), P, P, P, C.POINTER(Epilogue), P, P, C.POINTER(C.c_int), P],
    "ddpm_bn_partials": [P, I32, I64, I32, P, C.POINTER(C.c_int), P],
    "ddpm_bn_stats_from_partials": [P, I32, I64, I32, F32, F32, P, P, P, P, P],
    "ddpm_grad_amax": [P, I64, P, P],
    "ddpm_grad_scale": [P, I64, P, P, P],
    "ddpm_scale_tensors": [I32, P, P, P, P],
    "ddpm_conv_math": [],
    "ddpm_set_conv_math": [I32],
    "ddpm_range_flag_read": [P, I32, P],
}

_lib = None


class DdpmError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DdpmError(f"{LIB_PATH} is missing: build it with `python diffusion-model_amd/build.py` "
                            "(there is no CPU fallback for the HIP path)")
        handle = C.CDLL(LIB_PATH)
        handle.ddpm_last_error.restype = C.c_char_p
        handle.ddpm_last_error.argtypes = []
        handle.ddpm_reduce_workspace_doubles.restype = C.c_int64
        handle.ddpm_reduce_workspace_doubles.argtypes = [I32]
        handle.ddpm_packed_weight_floats.restype = C.c_int64
        handle.ddpm_packed_weight_floats.argtypes = [I32, I32, I32]
        handle.ddpm_mean_cov_workspace_doubles.restype = C.c_int64
        handle.ddpm_mean_cov_workspace_doubles.argtypes = [I32]
        for name, sig in _SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the header and library disagree
            fn.argtypes = sig
            fn.restype = C.c_int
        _lib = handle
    return _lib


def exported_symbols():
    return ["ddpm_last_error", "ddpm_reduce_workspace_doubles", "ddpm_packed_weight_floats", "ddpm_mean_cov_workspace_doubles"] + list(_SIGNATURES)


def packed_floats(taps, rows, cin):
    """Size, in floats, of the packed weight buffer of a (taps x rows x cin) contraction."""
    return int(lib().ddpm_packed_weight_floats(taps, rows, cin))


MATH_MODES = ("fp32", "bf16x3", "f16x2")
_mode = None      # the library's current mode as this module last read or set it (this module is its only writer)


def conv_math():
    """Arithmetic of the MFMA-fed kernels that the library is in right now: 'fp32', 'bf16x3' or 'f16x2' (it starts
    in the one DDPM_CONV_MATH names; set_conv_math / math_scope change it). Asked on every launch and cache lookup of
    the host mirror, hence answered from a copy kept here instead of a call into the library."""
    global _mode
    if _mode is None:
        _mode = MATH_MODES[int(lib().ddpm_conv_math())]
    return _mode


def set_conv_math(mode):
    """Switch the library's arithmetic; returns the previous mode. Packed weights belong to the mode they were
    packed in: callers key their caches on conv_math()."""
    global _mode
    prev = conv_math()
    if mode != prev:
        call("ddpm_set_conv_math", MATH_MODES.index(mode))
        _mode = mode
    return prev


class math_scope:
    """`with math_scope('bf16x3'):` runs the body in that arithmetic and restores the previous one (None = no change)."""

    def __init__(self, mode):
        self.mode, self.prev = mode, None

    def __enter__(self):
        if self.mode is not None:
            self.prev = set_conv_math(self.mode)
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            set_conv_math(self.prev)
        return False


def call(name, *args):
    """Invoke an int-returning entry point; raise DdpmError with the library's message on failure."""
    h = lib()
    rc = getattr(h, name)(*args)
    if rc != 0:
        raise DdpmError(f"{name} failed ({rc}): {h.ddpm_last_error().decode()}")
    return rc
