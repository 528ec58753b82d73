"""ctypes binding of ``libs3enc.so`` (C ABI declared in ``include/s3enc.h``).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C s3prl_amd/csrc``.  There is no
fallback: if it is missing, or no gfx950 GPU is visible at ``s3enc_create`` time, we raise.
"""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libs3enc.so")

S3ENC_MAX_CONV = 16
S3ENC_MAX_RES = 4
F32, BF16, F16, F32X3, F16X2 = 0, 1, 2, 3, 4
STATUS_NONFINITE, STATUS_PENDING = 1, 1 << 30  # s3enc_forward_status bits (include/s3enc.h)
DTYPES = {"fp32": F32, "f32": F32, "float32": F32, "bf16": BF16, "bfloat16": BF16, "fp16": F16, "f16": F16,
          "float16": F16, "fp32x3": F32X3, "f32x3": F32X3, "bf16x3": F32X3,
          "fp16x2": F16X2, "f16x2": F16X2}
FAMILY = {"hubert": 0, "wav2vec2": 1, "wavlm": 2, "distiller": 3, "multires_hubert": 4}
SEL_HIDDEN, SEL_LAYER_OUT, SEL_FFN_OUT = 0, 1, 2
SELECTIONS = {None: SEL_HIDDEN, "hidden_states": SEL_HIDDEN, "fairseq_layers": SEL_LAYER_OUT,
              "fairseq_layers_before_residual": SEL_FFN_OUT}
ABI_VERSION = 7
EXCHANGE_COLLECTIVE, EXCHANGE_DIRECT, EXCHANGE_COPY = 0, 1, 2   # S3ENC_EXCHANGE_*
COPY_HANDLE_BYTES = 256                                         # S3ENC_COPY_HANDLE_BYTES


class S3Config(C.Structure):
    _fields_ = [
        ("family", C.c_int32), ("n_conv", C.c_int32), ("conv_dim", C.c_int32),
        ("conv_kernel", C.c_int32 * S3ENC_MAX_CONV), ("conv_stride", C.c_int32 * S3ENC_MAX_CONV),
        ("extractor_layer_norm", C.c_int32), ("conv_bias", C.c_int32), ("encoder_layers", C.c_int32),
        ("embed_dim", C.c_int32), ("ffn_dim", C.c_int32), ("heads", C.c_int32), ("layer_norm_first", C.c_int32),
        ("conv_pos", C.c_int32), ("conv_pos_groups", C.c_int32), ("normalize", C.c_int32), ("rel_pos", C.c_int32),
        ("num_buckets", C.c_int32), ("max_distance", C.c_int32), ("gru_rel_pos", C.c_int32),
        ("compute_dtype", C.c_int32), ("no_feature_layer_norm", C.c_int32), ("pos_conv_depth", C.c_int32), ("wav_norm_eps", C.c_float), ("pred_heads", C.c_int32),
        ("mr_pairs", C.c_int32), ("mr_ratios", C.c_int32 * (2 * (S3ENC_MAX_RES - 1))),
        ("mr_layers", C.c_int32 * (2 * S3ENC_MAX_RES - 1)), ("mr_kernel", C.c_int32), ("mr_plain", C.c_int32),
    ]


class S3ForwardOpts(C.Structure):
    _fields_ = [("selection", C.c_int32), ("out_dtype", C.c_int32), ("featurize", C.c_int32),
                ("feat_normalize", C.c_int32), ("feat_w", C.POINTER(C.c_float))]


class S3Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


class S3ProfileEntry(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


class S3FbankConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("num_mel_bins", C.c_int32), ("frame_length_ms", C.c_float),
                ("frame_shift_ms", C.c_float), ("preemphasis", C.c_float), ("delta_order", C.c_int32),
                ("delta_win_length", C.c_int32), ("use_cmvn", C.c_int32), ("cmvn_eps", C.c_float)]


# every symbol include/s3enc.h declares: (restype, argtypes)
_VP, _I32, _I64 = C.c_void_p, C.c_int32, C.c_int64
_PROTOS = {
    "s3enc_version": (C.c_int, []),
    "s3enc_last_error": (C.c_char_p, []),
    "s3enc_create": (C.c_int, [C.POINTER(S3Config), C.POINTER(S3Tensor), _I32, _I32, C.POINTER(_VP)]),
    "s3enc_destroy": (C.c_int, [_VP]),
    "s3enc_num_frames": (C.c_int, [_VP, _I64, C.POINTER(_I32)]),
    "s3enc_num_output_frames": (C.c_int, [_VP, _I64, C.POINTER(_I32)]),
    "s3enc_downsample_rate": (C.c_int, [_VP, C.POINTER(_I32)]),
    "s3enc_valid_frames": (C.c_int, [_VP, _I64, _I64, C.POINTER(_I32)]),
    "s3enc_forward": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_I64), _I32, _I64, _VP, _I64, _VP]),
    "s3enc_forward_status": (C.c_int, [_VP, _I32, C.POINTER(_I32)]),
    "s3enc_forward_ex": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_I64), _I32, _I64, C.POINTER(S3ForwardOpts), _VP, _I64, _VP]),
    "s3enc_num_states": (C.c_int, [_VP, _I32, C.POINTER(_I32)]),
    "s3enc_forward_padded": (C.c_int, [_VP, _VP, _I64, C.POINTER(_I64), _I32, _I64, _VP, _I64, _VP]),
    "s3enc_set_layer_events": (C.c_int, [_VP, C.POINTER(_VP), _I32]),
    "s3enc_profile_enable": (C.c_int, [_VP, _I32]),
    "s3enc_profile_reset": (C.c_int, [_VP]),
    "s3enc_profile_read": (C.c_int, [_VP, C.POINTER(S3ProfileEntry), _I32, C.POINTER(_I32)]),
    "s3enc_debug_tap": (C.c_int, [_VP, C.c_char_p, C.POINTER(C.c_float), _I64, C.POINTER(_I64)]),
    "s3enc_comm_version": (C.c_int, [C.POINTER(_I32)]),
    "s3enc_comm_unique_id": (C.c_int, [_VP]),
    "s3enc_comm_init_rank": (C.c_int, [_VP, _I32, _I32, _I32, C.POINTER(_VP)]),
    "s3enc_comm_info": (C.c_int, [_VP, C.POINTER(_I32), C.POINTER(_I32)]),
    "s3enc_comm_allgather_states": (C.c_int, [_VP, _VP, _I64, _VP, _I64, _I32, _I64, C.POINTER(_VP), _VP]),
    "s3enc_comm_exchange_states": (C.c_int, [_VP, _I32, _VP, _I64, _VP, _I64, _I32, _I64, C.POINTER(_VP), _VP]),
    "s3enc_comm_destroy": (C.c_int, [_VP]),
    "s3enc_comm_init_local": (C.c_int, [_I32, _I32, _I32, C.POINTER(_VP)]),
    "s3enc_comm_copy_export": (C.c_int, [_VP, _VP, _I64, _VP]),
    "s3enc_comm_copy_attach": (C.c_int, [_VP, _VP]),
    "s3enc_comm_copy_status": (C.c_int, [_VP, C.POINTER(_I32)]),
    "s3enc_comm_copy_release": (C.c_int, [_VP, _VP]),
    "s3enc_set_handle_tuning": (C.c_int, [_VP, C.c_char_p, _I32]),
    "s3enc_set_tuning": (C.c_int, [C.c_char_p, _I32]),
    "s3enc_op_gemm": (C.c_int, [_I32, _VP, _I64, _I64, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _VP, _VP, _VP, _VP,
                                _I64, _I64, _VP]),
    "s3enc_debug_occupy_cus": (C.c_int, [_I32, _I32, C.c_double, _VP]),
    "s3enc_debug_clock_sample": (C.c_int, [_VP, _VP]),
    "s3enc_op_layernorm": (C.c_int, [_I32, _VP, _VP, _VP, _I32, _I32, _I32, _VP, _VP, _VP]),
    "s3enc_op_attention": (C.c_int, [_I32, _VP, _VP, _VP, _I32, _I32, _I32, _VP, _I32, _VP, _VP]),
    "s3enc_op_conv0": (C.c_int, [_I32, C.POINTER(_VP), C.POINTER(_I64), _I32, _I64, _I32, _VP, _VP, _VP, _VP, _VP, _VP, _I32,
                                 _I32, _VP, _VP]),
    "s3enc_op_wavlm_gate": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _VP, _VP]),
    "s3enc_op_posconv": (C.c_int, [_I32, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _VP, _VP]),
    "s3enc_weighted_sum": (C.c_int, [_VP, _I64, _I32, C.POINTER(C.c_float), _I32, _I64, _I32, _VP, _VP]),
    "s3enc_weighted_sum_backward_scratch": (_I64, [_I64, _I32]),
    "s3enc_weighted_sum_backward": (C.c_int, [_VP, _I64, _I32, _I32, _I64, _I32, _VP, _VP, _VP, _VP]),
    "s3enc_fbank_num_frames": (C.c_int, [C.POINTER(S3FbankConfig), _I64, C.POINTER(_I32)]),
    "s3enc_fbank_forward": (C.c_int, [C.POINTER(S3FbankConfig), _VP, C.POINTER(_I64), _I32, _VP, _I64, _I32, _VP]),
}

_lib = None


class S3EncError(RuntimeError):
    pass


def load():
    """Load libs3enc.so (once).  ``import torch`` first so that the HIP runtime torch ships is the one
    the library binds to (same SONAME libamdhip64.so.7) and device pointers are interchangeable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise S3EncError(
            f"{LIB_PATH} not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C s3prl_amd/csrc`.  There is no CPU / PyTorch fallback for the encoder path."
        )
    try:
        import torch  # noqa: F401  (loads libamdhip64 first)
    except Exception:  # pragma: no cover - torch is only plumbing
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.s3enc_version() != ABI_VERSION:
        raise S3EncError(f"{LIB_PATH} has ABI version {lib.s3enc_version()}, this package needs {ABI_VERSION}: rebuild it")
    _lib = lib
    return lib


def check(rc: int, what: str = "libs3enc"):
    if rc != 0:
        msg = load().s3enc_last_error()
        raise S3EncError(f"{what}: {msg.decode() if msg else 'unknown error'}")


def make_config(cfg, dtype: str) -> S3Config:
    """EncoderConfig → C struct."""
    c = S3Config()
    c.family = FAMILY[cfg.family]
    if len(cfg.conv_layers) > S3ENC_MAX_CONV:
        raise S3EncError("too many conv layers")
    c.n_conv = len(cfg.conv_layers)
    c.conv_dim = cfg.conv_dim
    for i, (_, k, s) in enumerate(cfg.conv_layers):
        c.conv_kernel[i] = k
        c.conv_stride[i] = s
    c.extractor_layer_norm = int(cfg.extractor_mode == "layer_norm")
    c.conv_bias = int(cfg.conv_bias)
    c.encoder_layers = cfg.encoder_layers
    c.embed_dim = cfg.encoder_embed_dim
    c.ffn_dim = cfg.encoder_ffn_embed_dim
    c.heads = cfg.encoder_attention_heads
    c.layer_norm_first = int(cfg.layer_norm_first)
    c.conv_pos = cfg.conv_pos
    c.conv_pos_groups = cfg.conv_pos_groups
    c.normalize = int(cfg.normalize)
    c.rel_pos = int(cfg.family == "wavlm" and cfg.relative_position_embedding)
    c.num_buckets = cfg.num_buckets
    c.max_distance = cfg.max_distance
    c.gru_rel_pos = int(cfg.gru_rel_pos)
    if dtype not in DTYPES:
        raise S3EncError(f"unknown dtype {dtype!r}; use one of fp32 / bf16 / fp16")
    c.compute_dtype = DTYPES[dtype]
    c.no_feature_layer_norm = int(not cfg.feature_layer_norm)
    c.pos_conv_depth = int(cfg.pos_conv_depth)
    c.wav_norm_eps = float(cfg.wav_norm_eps)
    c.pred_heads = int(cfg.pred_heads)
    if cfg.family == "multires_hubert":
        pairs = cfg.rate_pairs
        if len(pairs) > S3ENC_MAX_RES - 1:
            raise S3EncError("too many resolutions")
        c.mr_pairs = len(pairs)
        for i, r in enumerate(cfg.label_rate_ratios):
            c.mr_ratios[i] = int(r)
        for i, n in enumerate(cfg.block_layers):
            c.mr_layers[i] = int(n)
        c.mr_kernel = int(cfg.conv_adapter_kernel)
        c.mr_plain = int(cfg.use_plain_updownsample)
    return c
