"""ctypes binding of libkokoro_hip.so (include/kokoro_hip.h).  Fails loudly when the library
is missing: there is no CPU or PyTorch fallback for the hot path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkokoro_hip.so")

KK_F32, KK_BF16, KK_I32, KK_F16 = 0, 1, 2, 3
NOISE_ZERO, NOISE_INJECTED, NOISE_PHILOX = 0, 1, 2
ACT_NONE, ACT_LRELU, ACT_GELU, ACT_SNAKE = 0, 1, 2, 3


class KKConfig(C.Structure):
    _fields_ = [
        ("n_token", C.c_int32), ("hidden_dim", C.c_int32), ("style_dim", C.c_int32), ("n_layer", C.c_int32),
        ("max_dur", C.c_int32), ("text_encoder_kernel_size", C.c_int32),
        ("plbert_hidden", C.c_int32), ("plbert_heads", C.c_int32), ("plbert_intermediate", C.c_int32),
        ("plbert_max_pos", C.c_int32), ("plbert_layers", C.c_int32), ("plbert_embedding", C.c_int32),
        ("decoder_hidden", C.c_int32), ("upsample_initial_channel", C.c_int32), ("n_upsamples", C.c_int32),
        ("upsample_rates", C.c_int32 * 4), ("upsample_kernel_sizes", C.c_int32 * 4),
        ("n_resblock_kernels", C.c_int32), ("resblock_kernel_sizes", C.c_int32 * 4),
        ("resblock_dilations", (C.c_int32 * 3) * 4),
        ("gen_istft_n_fft", C.c_int32), ("gen_istft_hop_size", C.c_int32), ("compute_dtype", C.c_int32),
    ]


class KKMimiConfig(C.Structure):
    _fields_ = [
        ("dim", C.c_int32), ("nq", C.c_int32), ("bins", C.c_int32), ("qdim", C.c_int32), ("num_heads", C.c_int32),
        ("num_layers", C.c_int32), ("dim_feedforward", C.c_int32), ("nfilters", C.c_int32),
        ("n_ratios", C.c_int32), ("ratios", C.c_int32 * 8),
        ("ksize", C.c_int32), ("residual_ksize", C.c_int32), ("last_ksize", C.c_int32), ("upsample_stride", C.c_int32),
        ("compress", C.c_int32), ("rope_base", C.c_float), ("compute_dtype", C.c_int32),
    ]


class KKLlamaArgs(C.Structure):
    _fields_ = [("num_layers", C.c_int32), ("num_heads", C.c_int32), ("num_kv_heads", C.c_int32), ("head_dim", C.c_int32),
                ("hidden", C.c_int32), ("intermediate", C.c_int32), ("rope_theta", C.c_float), ("rope_factor", C.c_float), ("rms_eps", C.c_float)]


class KKCsmConfig(C.Structure):
    _fields_ = [("text_vocab_size", C.c_int32), ("audio_vocab_size", C.c_int32), ("audio_num_codebooks", C.c_int32), ("max_seq_len", C.c_int32),
                ("backbone", KKLlamaArgs), ("decoder", KKLlamaArgs)]


# every symbol include/kokoro_hip.h declares: name -> (restype, argtypes)
_vp, _i, _f, _sz, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_uint64
SIGNATURES = {
    "kk_create": (_i, [C.POINTER(KKConfig), C.POINTER(_vp)]),
    "kk_destroy": (None, [_vp]),
    "kk_load_tensor": (_i, [_vp, C.c_char_p, _i, C.POINTER(C.c_int64), _i, _vp]),
    "kk_finalize": (_i, [_vp, _vp]),
    "kk_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "kk_context_create": (_i, [_vp, C.POINTER(_vp)]),
    "kk_context_destroy": (None, [_vp]),
    "kk_context_model": (_vp, [_vp]),
    "kk_context_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "kk_forward_text": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kk_forward_audio": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _u64, _vp, _sz, _vp, _vp]),
    "kk_forward": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _u64, _vp, _sz, _vp, _vp, _vp]),
    "kk_last_error": (C.c_char_p, []),
    "kk_abi_version": (_i, []),
    "kk_op_conv1d": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _f, _vp, _i, _f, _i, _vp, _i, _i, _vp, _i, _i]),
    "kk_op_conv1d_bf16": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _f, _vp, _i, _f, _i, _vp, _i, _i, _vp, _i]),
    "kk_op_conv1d_bf16_fused": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _f, _vp, _vp, _i, _f,
                                     _vp, _i, _vp, C.POINTER(C.c_int)]),
    "kk_op_adain": (_i, [_vp, _i, _vp, _i, _i, _vp, _i, _vp, _i, _i, _f, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _i, _i]),
    "kk_op_layernorm": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _f, _i, _f, _vp, _i, _i]),
    "kk_op_lstm": (_i, [_vp, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _i]),
    "kk_op_lstm_bf16": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i]),
    "kk_op_linear_rows": (_i, [_vp, _i, _vp, C.c_longlong, _i, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, C.c_longlong, _i]),
    "kk_op_attention": (_i, [_vp, _i, _vp, _i, _i, _vp, _i, _vp, _i, _i]),
    "kk_op_source_stft": (_i, [_vp, _i, _vp, _i, _vp, _vp, _f, _i, _vp, _u64, _vp, _vp, _vp, _i, _i]),
    "kk_op_istft_head": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _i, _i]),
    "kk_op_pack_head_w": (_i, [_vp, _vp, _vp]),
    "kk_op_conv_post_istft": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _f, _vp, _vp, _i]),
    "kk_debug_info": (_i, [_vp, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "kk_debug_fetch": (_i, [_vp, _vp, C.c_char_p, _vp]),
    "kk_debug_override": (_i, [_vp, C.c_char_p, _vp]),
    "kk_debug_clear": (None, [_vp]),
    "kk_debug_force_generic": (None, [_vp, _i]),
    "kk_op_pack_w_frag": (_i, [_vp, _vp, _vp, _i, _i, _i]),
    "kk_csm_create": (_i, [C.POINTER(KKCsmConfig), C.POINTER(_vp)]),
    "kk_csm_destroy": (None, [_vp]),
    "kk_csm_load_tensor": (_i, [_vp, C.c_char_p, C.POINTER(C.c_int64), _i, _vp]),
    "kk_csm_set_weight_dtype": (_i, [_vp, _i]),
    "kk_csm_finalize": (_i, [_vp, _vp]),
    "kk_csm_share": (_i, [_vp, C.POINTER(_vp)]),
    "kk_csm_setup_caches": (_i, [_vp, _i]),
    "kk_csm_reset_caches": (_i, [_vp]),
    "kk_csm_position": (_i, [_vp]),
    "kk_csm_set_padding": (_i, [_vp, _i, _vp]),
    "kk_csm_workspace_bytes": (_sz, [_vp, _i, _i]),
    "kk_csm_generate_frame": (_i, [_vp, _vp, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _sz, _vp]),
    "kk_csm_set_graph_mode": (_i, [_vp, _i]),
    "kk_csm_debug_logits": (_i, [_vp, _vp, _i, _vp]),
    "kk_csm_debug_skip": (_i, [_i]),
    "kk_csm_debug_timestamps": (_i, [_vp, _i]),
    "kk_op_csm_sample": (_i, [_vp, _i, _i, _vp, _f, _i, _vp, _vp]),
    "kk_mimi_create": (_i, [C.POINTER(KKMimiConfig), C.POINTER(_vp)]),
    "kk_mimi_destroy": (None, [_vp]),
    "kk_mimi_load_tensor": (_i, [_vp, C.c_char_p, C.POINTER(C.c_int64), _i, _vp]),
    "kk_mimi_finalize": (_i, [_vp, _vp]),
    "kk_mimi_samples_per_frame": (C.c_int64, [_vp]),
    "kk_mimi_workspace_bytes": (_sz, [_vp, _i, _i]),
    "kk_mimi_decode": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "kk_mimi_stream_create": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "kk_mimi_stream_create_chunked": (_i, [_vp, _i, _i, _i, _i, C.POINTER(_vp)]),
    "kk_mimi_stream_chunk_frames": (_i, [_vp]),
    "kk_mimi_stream_set_chunk": (_i, [_vp, _i]),
    "kk_mimi_stream_max_chunk_frames": (_i, [_vp]),
    "kk_mimi_encode_step": (_i, [_vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "kk_mimi_stream_destroy": (None, [_vp]),
    "kk_mimi_stream_reset": (_i, [_vp]),
    "kk_mimi_stream_frames": (_i, [_vp]),
    "kk_mimi_stream_set_context": (_i, [_vp, _i]),
    "kk_mimi_stream_workspace_bytes": (_sz, [_vp, _i]),
    "kk_mimi_decode_step": (_i, [_vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "kk_mimi_encode_frames": (_i, [_vp, _i]),
    "kk_mimi_encode_workspace_bytes": (_sz, [_vp, _i, _i]),
    "kk_mimi_encode": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "kk_mimi_debug_info": (_i, [_vp, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "kk_mimi_debug_fetch": (_i, [_vp, _vp, C.c_char_p, _vp]),
    "kk_debug_set_op_wfrag": (None, [_vp]),
    "kk_debug_set_op_variant": (None, [_i]),
    "kk_set_graph_mode": (_i, [_vp, _i]),
    "kk_set_quantization": (_i, [_vp, _i, _i]),
    "kk_quantized_layers": (_i, [_vp]),
    "kk_mxfp8_bytes": (_i, [_i, _i, C.POINTER(_sz), C.POINTER(_sz)]),
    "kk_mxfp8_pack_weight": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "kk_op_linear_mxfp8": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i]),
    "kk_profile_begin": (_i, [_vp, _i]),
    "kk_profile_end": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

_lib = None


class KokoroHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises KokoroHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("KK_HIP_LIB", LIB_PATH)  # instrumented side builds only (build.py --trace)
    if not os.path.exists(path):
        raise KokoroHipError(
            f"{path} not found: build it with `python mlx-audio_amd/build.py` (hipcc, gfx950). "
            "The Kokoro hot path has no CPU or PyTorch fallback."
        )
    # torch brings its own copy of the HIP runtime (torch/lib/libamdhip64.so).  If this library were loaded first it would bind to the system
    # copy under /opt/rocm, torch would then load its own, and the process would hold TWO HIP runtimes: the first hipMalloc of kk_finalize
    # fails (seen with `python __graft_entry__.py smoke`, where build() loads the library before anything imports torch).  Importing torch
    # first makes its runtime the one this library's libamdhip64 dependency resolves to.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.kk_abi_version() != 2:
        raise KokoroHipError("libkokoro_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().kk_last_error()
        raise KokoroHipError(f"{what}: {msg.decode() if msg else 'unknown error'}")
