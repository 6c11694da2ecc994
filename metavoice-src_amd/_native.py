"""ctypes binding of libmetavoice_hip.so (C ABI: include/metavoice_hip.h).

There is deliberately NO fallback: if the HIP library has not been built (`python -c "import __graft_entry__ as g;
g.build()"` or `make -C metavoice-src_amd/csrc`) every entry point raises.  A CPU / eager-PyTorch path would void the
parity claims of this package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MV_HIP_LIB") or os.path.join(_HERE, "lib", "libmetavoice_hip.so")  # MV_HIP_LIB: another build (A/B experiments)

MV_OK = 0
MV_ERR_INVALID = -1
MV_ERR_HIP = -2
MV_ERR_STATE = -3
MV_ERR_MISSING_WEIGHT = -4
MV_ERR_PROMPT_TOO_LONG = -5

MV_BF16, MV_F32, MV_I32, MV_I8 = 0, 1, 2, 3
MV_W_BF16, MV_W_INT4_G128, MV_W_INT8 = 0, 1, 2


class mv_config(C.Structure):
    _fields_ = [
        ("n_layer", C.c_int32), ("n_head", C.c_int32), ("dim", C.c_int32), ("intermediate", C.c_int32),
        ("vocab", C.c_int32), ("block_size", C.c_int32), ("spk_dim", C.c_int32), ("norm_eps", C.c_float),
        ("max_utterances", C.c_int32), ("weight_format", C.c_int32),
    ]


class mv_stage2_config(C.Structure):
    _fields_ = [
        ("n_layer", C.c_int32), ("n_head", C.c_int32), ("n_embd", C.c_int32), ("hidden", C.c_int32), ("block_size", C.c_int32),
        ("spk_dim", C.c_int32), ("n_in", C.c_int32), ("n_out", C.c_int32), ("vocab_in", C.c_int32 * 4), ("vocab_out", C.c_int32 * 8),
        ("bias", C.c_int32), ("norm_type", C.c_int32), ("act", C.c_int32), ("causal", C.c_int32), ("norm_eps", C.c_float),
        ("max_batch", C.c_int32),
    ]


class mv_codec_config(C.Structure):
    _fields_ = [
        ("hidden", C.c_int32), ("num_filters", C.c_int32), ("n_ratios", C.c_int32), ("ratios", C.c_int32 * 8),
        ("kernel_size", C.c_int32), ("last_kernel_size", C.c_int32), ("residual_kernel_size", C.c_int32), ("compress", C.c_int32),
        ("lstm_layers", C.c_int32), ("n_q", C.c_int32), ("bins", C.c_int32), ("audio_channels", C.c_int32),
    ]


class mv_mbd_config(C.Structure):
    _fields_ = [("n_models", C.c_int32), ("chin", C.c_int32), ("depth", C.c_int32), ("channels", C.c_int32 * 8), ("kernel", C.c_int32),
                ("stride", C.c_int32), ("norm_groups", C.c_int32), ("res_blocks", C.c_int32), ("act", C.c_int32), ("codec_dim", C.c_int32),
                ("emb_all_layers", C.c_int32), ("bilstm", C.c_int32), ("num_steps", C.c_int32), ("n_eval", C.c_int32), ("proc_bands", C.c_int32),
                ("eq_bands", C.c_int32), ("precision", C.c_int32)]


class mv_spkenc_config(C.Structure):
    _fields_ = [("n_mels", C.c_int32), ("hidden", C.c_int32), ("n_layers", C.c_int32), ("emb", C.c_int32)]


class mv_sampling(C.Structure):
    _fields_ = [
        ("guidance_scale", C.c_float), ("one_minus_guidance", C.c_float), ("temperature", C.c_float),
        ("top_p", C.c_float), ("top_k", C.c_int32), ("seed", C.c_uint64),
    ]


# every symbol include/metavoice_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("mv_engine_create", C.c_int, [C.POINTER(mv_config), C.POINTER(_P)]),
    ("mv_engine_destroy", C.c_int, [_P]),
    ("mv_load_weight", C.c_int, [_P, C.c_char_p, _P, C.c_int32, C.c_int64, C.c_int64]),
    ("mv_finalize", C.c_int, [_P]),
    ("mv_model_bytes", C.c_int64, [_P]),
    ("mv_begin", C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_uint16),
                           C.POINTER(mv_sampling), C.c_int32, C.c_int32, C.c_int32]),
    ("mv_run", C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    ("mv_last_run_ms", C.c_float, [_P]),
    ("mv_decode_path", C.c_int32, [_P, C.POINTER(C.c_int32)]),
    ("mv_get_tokens", C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32)]),
    ("mv_set_logits_capture", C.c_int, [_P, C.c_int32, C.c_int32]),
    ("mv_get_captured_logits", C.c_int, [_P, C.POINTER(C.c_float), C.c_int32, C.c_int32]),
    ("mv_op_gemv", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_float, _P, _P]),
    ("mv_op_swiglu_gemv", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_float, _P]),
    ("mv_op_attention_decode", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P]),
    ("mv_op_attention_scratch_bytes", C.c_int64, [C.c_int32, C.c_int32]),
    ("mv_op_sample", C.c_int, [_P, C.c_int32, _P, C.POINTER(mv_sampling), C.c_uint32, _P, _P]),
    ("mv_op_quantize_int4", C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    ("mv_op_gemv_int4", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, C.c_float, _P, _P]),
    ("mv_op_gemm", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, C.c_int32, _P]),
    ("mv_op_attention_rows", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    ("mv_stage2_create", C.c_int, [C.POINTER(mv_stage2_config), C.POINTER(_P)]),
    ("mv_stage2_destroy", C.c_int, [_P]),
    ("mv_stage2_load_weight", C.c_int, [_P, C.c_char_p, _P, C.c_int32, C.c_int64, C.c_int64]),
    ("mv_stage2_finalize", C.c_int, [_P]),
    ("mv_stage2_forward", C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_uint16)]),
    ("mv_stage2_get_logits", C.c_int, [_P, C.c_int32, C.POINTER(C.c_float)]),
    ("mv_stage2_sample", C.c_int, [_P, C.c_float, C.c_int32, C.c_uint64, C.POINTER(C.c_int32)]),
    ("mv_codec_create", C.c_int, [C.POINTER(mv_codec_config), C.POINTER(_P)]),
    ("mv_codec_destroy", C.c_int, [_P]),
    ("mv_codec_load_weight", C.c_int, [_P, C.c_char_p, C.POINTER(C.c_float), C.c_int64]),
    ("mv_codec_finalize", C.c_int, [_P]),
    ("mv_codec_samples", C.c_int64, [_P, C.c_int32]),
    ("mv_codec_decode", C.c_int, [_P, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("mv_codec_decode_device", C.c_int, [_P, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("mv_mbd_create", C.c_int, [C.POINTER(mv_mbd_config), C.POINTER(_P)]),
    ("mv_mbd_destroy", C.c_int, [_P]),
    ("mv_mbd_load_weight", C.c_int, [_P, C.c_int32, C.c_char_p, C.POINTER(C.c_float), C.c_int64]),
    ("mv_mbd_set_schedule", C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                      C.c_float, C.c_float, C.c_float]),
    ("mv_mbd_set_processor", C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("mv_mbd_set_filterbank", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float)]),
    ("mv_mbd_finalize", C.c_int, [_P]),
    ("mv_mbd_clone", C.c_int, [_P, C.POINTER(_P)]),
    ("mv_mbd_generate", C.c_int, [_P, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_float), C.c_uint64,
                                  C.POINTER(C.c_float)]),
    ("mv_mbd_generate_device", C.c_int, [_P, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_uint64, C.POINTER(C.c_float)]),
    ("mv_mbd_unet_forward", C.c_int, [_P, C.c_int32, C.POINTER(C.c_float), C.c_int32, C.c_int32, C.POINTER(C.c_float), C.c_int32,
                                      C.POINTER(C.c_float)]),
    ("mv_mbd_unet_forward_sum", C.c_int, [_P, C.POINTER(C.c_float), C.c_int32, C.c_int32, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_float)]),
    ("mv_mbd_kernel_counts", C.c_int, [_P, C.POINTER(C.c_int64)]),
    ("mv_spkenc_create", C.c_int, [C.POINTER(mv_spkenc_config), C.POINTER(_P)]),
    ("mv_spkenc_destroy", C.c_int, [_P]),
    ("mv_spkenc_load_weight", C.c_int, [_P, C.c_char_p, C.POINTER(C.c_float), C.c_int64]),
    ("mv_spkenc_finalize", C.c_int, [_P]),
    ("mv_spkenc_embed", C.c_int, [_P, C.POINTER(C.c_float), C.c_int32, C.c_int32, C.POINTER(C.c_float)]),
    ("mv_audio_set_device", C.c_int, [C.c_int32]),
    ("mv_audio_resample", C.c_int, [C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_float), C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.POINTER(C.c_float)]),
    ("mv_audio_frame_rms", C.c_int, [C.POINTER(C.c_float), C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.c_int64]),
    ("mv_audio_mel", C.c_int, [C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32, C.c_int32, C.c_int32,
                               C.POINTER(C.c_float), C.c_int64]),
    ("mv_probe_kernel", C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_double)]),
    ("mv_tune", C.c_int, [_P, C.c_char_p, C.c_int32]),
    ("mv_debug_vectors", C.c_int, [_P, C.POINTER(C.c_uint16), C.c_int64]),
    ("mv_debug_stamps", C.c_int, [_P, C.POINTER(C.c_uint64), C.c_int64]),
    ("mv_last_error", C.c_char_p, []),
    ("mv_version", C.c_char_p, []),
]

_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the HIP library; raises NativeLibraryMissing if it was never built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing(
                f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                "(or __graft_entry__.build()).  This package has no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(handle, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(code: int) -> None:
    """Map a negative mv_status onto the exception the reference raises in the same situation."""
    if code >= 0:
        return
    msg = (lib().mv_last_error() or b"").decode()
    if code in (MV_ERR_INVALID, MV_ERR_PROMPT_TOO_LONG, MV_ERR_MISSING_WEIGHT):
        raise ValueError(msg)
    raise RuntimeError(msg)
