"""ctypes binding of libserhip.so (the C ABI in include/ser_hip.h).

There is exactly one compute backend.  If the library cannot be loaded the
import of this module raises: the product never falls back to PyTorch ops or to
the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- FIRST: torch brings its own libamdhip64; libserhip must bind to that copy, not load a second
#                              HIP runtime beside it (with the library loaded first, launches fail with "no ROCm-capable device")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SER_HIP_LIB") or os.path.join(_HERE, "lib", "libserhip.so")   # override: A/B builds (tools/)

MODE_BF16 = 1
MODE_FP32X = 2
MODE_FP16 = 3
MODE_FP16X = 4
MODE_FP16Q = 5
MODE_FP16M = 6
ACT_NONE = 0
ACT_GELU = 1
WS_LOGMEL = 1
WS_WAVE_FRAMES = 2
ABI_VERSION = 14

c_void_p, c_int, c_i64, c_float = C.c_void_p, C.c_int, C.c_int64, C.c_float


class GemmArgs(C.Structure):
    """Mirror of ``ser_gemm_args`` (include/ser_hip.h); field order is ABI."""
    _fields_ = [
        ("A", c_void_p), ("a_plane_stride", c_i64), ("a_rowoff", c_void_p), ("lda", c_i64),
        ("kc", C.c_int32), ("ldj", c_i64),
        ("W", c_void_p), ("w_plane_stride", c_i64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("groups", C.c_int32),
        ("a_group_stride", c_i64), ("w_group_stride", c_i64), ("c_group_stride", C.c_int32),
        ("mode", C.c_int32), ("bias", c_void_p), ("act", C.c_int32),
        ("residual", c_void_p), ("ldr", c_i64), ("res_row_mod", C.c_int32),
        ("out_f32", c_void_p), ("ldo_f32", c_i64),
        ("out_act", c_void_p), ("ldo_act", c_i64), ("out_plane_stride", c_i64),
        ("out_rowmap", c_void_p),
        ("ln_gamma", c_void_p), ("ln_beta", c_void_p), ("ln_eps", c_float), ("tile_cfg", C.c_int32),
        ("ln_stats_in", c_void_p), ("ln_groups", C.c_int32), ("ln_colsum", c_void_p),
        ("stat_out", c_void_p), ("stat_groups", C.c_int32), ("f32_col_begin", C.c_int32),
        ("col_scale", c_float), ("col_scale_end", C.c_int32),
        ("shift_in", c_void_p), ("shift_out", c_void_p), ("shift_const", c_float), ("out_mode", C.c_int32),
        ("ln_shift", c_void_p), ("mean_out", c_void_p), ("lnstat_out", c_void_p),
        ("a_scale", c_void_p), ("a_scale_ld", c_i64), ("w_scale", c_void_p), ("w_scale_ld", c_i64),
        ("out_scale", c_void_p), ("out_scale_ld", c_i64), ("range_flag", c_void_p),
    ]


class AttentionArgs(C.Structure):
    """Mirror of ``ser_attention_args``."""
    _fields_ = [
        ("qkv", c_void_p), ("ld", c_i64), ("plane_stride", c_i64),
        ("q_col", C.c_int32), ("k_col", C.c_int32), ("v_col", C.c_int32), ("B", C.c_int32),
        ("frame_offs", c_void_p), ("table", c_void_p), ("gate", c_void_p),
        ("max_frames", C.c_int32), ("table_T", C.c_int32),
        ("out", c_void_p), ("ldo", c_i64), ("out_plane_stride", c_i64),
        ("H", C.c_int32), ("dh", C.c_int32), ("scale", c_float), ("mode", C.c_int32),
        ("gate_col", C.c_int32), ("out_mode", C.c_int32),
        ("gru_const", c_void_p), ("key_lens", c_void_p),
        ("bias2d", c_void_p), ("bias2d_ld", c_i64),
        ("gate_x", c_void_p), ("gate_x_ld", c_i64), ("gate_x_plane_stride", c_i64),
        ("gate_stat", c_void_p), ("gate_w", c_void_p), ("gate_cb", c_void_p),
        ("gate_x_planes", C.c_int32), ("reserved1", C.c_int32), ("gate_w_plane_stride", c_i64),
        ("out_scale", c_void_p), ("out_scale_ld", c_i64),
    ]


class LayerNormArgs(C.Structure):
    """Mirror of ``ser_layernorm_args``."""
    _fields_ = [
        ("x", c_void_p), ("ldx", c_i64), ("g", c_void_p), ("b", c_void_p), ("eps", c_float), ("gelu", C.c_int32),
        ("out_f32", c_void_p), ("ldo_f32", c_i64), ("out_act", c_void_p), ("ldo_act", c_i64), ("out_plane_stride", c_i64),
        ("mode", C.c_int32), ("rows", C.c_int32), ("D", C.c_int32), ("reserved0", C.c_int32),
        ("range_flag", c_void_p),
    ]


class WaveFramesArgs(C.Structure):
    """Mirror of ``ser_wave_frames_args``."""
    _fields_ = [
        ("wav", c_void_p), ("sample_offs", c_void_p), ("frame_offs", c_void_p),
        ("B", C.c_int32), ("k", C.c_int32), ("stride", C.c_int32), ("mode", C.c_int32),
        ("out", c_void_p), ("out_plane_stride", c_i64), ("work", c_void_p),
        ("total_rows", C.c_int32), ("reserved0", C.c_int32),
        ("range_flag", c_void_p),
    ]


class RowCenterArgs(C.Structure):
    """Mirror of ``ser_row_center_args``."""
    _fields_ = [
        ("x", c_void_p), ("ldx", c_i64), ("out_act", c_void_p), ("ldo_act", c_i64), ("out_plane_stride", c_i64),
        ("stats", c_void_p), ("shift", c_void_p),
        ("stat_groups", C.c_int32), ("mode", C.c_int32), ("rows", C.c_int32), ("D", C.c_int32),
        ("out_scale", c_void_p), ("out_scale_ld", c_i64), ("range_flag", c_void_p),
    ]


class LogmelArgs(C.Structure):
    """Mirror of ``ser_logmel_args``."""
    _fields_ = [("wav", c_void_p), ("sample_offs", c_void_p), ("mel", c_void_p), ("out", c_void_p), ("work", c_void_p),
                ("B", C.c_int32), ("n_mels", C.c_int32)]


class PackActArgs(C.Structure):
    """Mirror of ``ser_pack_act_args``."""
    _fields_ = [("x", c_void_p), ("out", c_void_p), ("ldo", c_i64), ("out_plane_stride", c_i64),
                ("B", C.c_int32), ("C", C.c_int32), ("T", C.c_int32), ("halo", C.c_int32), ("mode", C.c_int32), ("reserved0", C.c_int32),
                ("range_flag", c_void_p)]


class _CmdUnion(C.Union):
    _fields_ = [("gemm", GemmArgs), ("attention", AttentionArgs), ("layernorm", LayerNormArgs), ("wave_frames", WaveFramesArgs),
                ("row_center", RowCenterArgs), ("logmel", LogmelArgs), ("pack_act", PackActArgs)]


class Cmd(C.Structure):
    """Mirror of ``ser_cmd``: one recorded launch of a command list (``ser_run``)."""
    _fields_ = [("op", C.c_int32), ("reserved0", C.c_int32), ("u", _CmdUnion)]


OP_GEMM, OP_ATTENTION, OP_LAYERNORM, OP_WAVE_FRAMES, OP_ROW_CENTER, OP_LOGMEL, OP_PACK_ACT = 1, 2, 3, 4, 5, 6, 7
STRUCT_MIRRORS = {"ser_gemm_args": GemmArgs, "ser_attention_args": AttentionArgs, "ser_layernorm_args": LayerNormArgs,
                  "ser_wave_frames_args": WaveFramesArgs, "ser_row_center_args": RowCenterArgs, "ser_logmel_args": LogmelArgs,
                  "ser_pack_act_args": PackActArgs, "ser_cmd": Cmd}

_SIGNATURES = {
    "ser_version": (c_int, []),
    "ser_last_error": (C.c_char_p, []),
    "ser_wave_norm": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "ser_wave_frames": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_i64, c_int, c_void_p,
                                c_int, c_void_p]),
    "ser_conv0_ln_gelu": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ser_gemm": (c_int, [C.POINTER(GemmArgs), c_void_p]),
    "ser_layernorm": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_float, c_int, c_void_p, c_i64, c_void_p, c_i64,
                              c_i64, c_int, c_int, c_int, c_void_p]),
    "ser_row_center": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_i64, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ser_row_center_v": (c_int, [c_void_p, c_void_p]),
    "ser_layernorm_v": (c_int, [c_void_p, c_void_p]),
    "ser_wave_frames_v": (c_int, [c_void_p, c_void_p]),
    "ser_pack_act_v": (c_int, [c_void_p, c_void_p]),
    "ser_pack_f16m": (c_int, [c_void_p, c_i64, c_int, c_int, c_void_p, c_i64, c_i64, c_void_p, c_i64, c_int, c_void_p, c_void_p]),
    "ser_wavlm_bias_table": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ser_wavlm_gate": (c_int, [c_void_p, c_i64, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                               c_int, c_void_p]),
    "ser_attention": (c_int, [c_void_p, c_i64, c_i64, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                              c_void_p, c_void_p, c_i64, c_i64, c_int, c_int, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    "ser_attention_v": (c_int, [c_void_p, c_void_p]),
    "ser_embed_ln": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                             c_i64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ser_logmel_init": (c_int, [c_void_p, c_int, c_void_p]),
    "ser_logmel_whisper": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "ser_pack_act": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_i64, c_i64, c_int, c_void_p]),
    "ser_mean4": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    "ser_split_bf16": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_i64, c_void_p]),
    "ser_embed_ln_masked": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_i64,
                                    c_int, c_int, c_int, c_int, c_void_p]),
    "ser_deberta_attention": (c_int, [c_void_p, c_i64, c_i64, c_int, c_int, c_int, c_void_p, c_void_p, c_i64, c_int, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ser_pack_rows": (c_int, [c_void_p, c_i64, c_int, c_int, c_int, c_int, c_void_p, c_i64, c_i64, c_int, c_void_p]),
    "ser_zero_padded_rows": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_i64, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ser_deberta_bias": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_int,
                                 c_float, c_void_p]),
    "ser_run": (c_int, [c_void_p, C.c_int32, c_void_p, c_void_p]),
    "ser_ragged_index": (c_int, [c_void_p, c_void_p, c_int, c_i64, c_i64, c_i64, c_void_p, c_i64, c_void_p]),
    "ser_workspace_bytes": (C.c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "ser_wav_read_f32": (c_i64, [C.c_char_p, c_void_p, c_i64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ser_pt_write_f32": (c_int, [C.c_char_p, c_void_p, c_i64, c_i64]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class SerHipError(RuntimeError):
    pass


def _load():
    if not os.path.isfile(LIB_PATH):
        raise SerHipError(
            f"{LIB_PATH} is missing: build it with `make -C interspeech_ser_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()').  There is no fallback backend.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    got = lib.ser_version()
    if got != ABI_VERSION:
        raise SerHipError(f"libserhip ABI version {got} != expected {ABI_VERSION}")
    return lib


lib = _load()


def check(rc: int, what: str = "") -> None:
    """Turn a non-zero launcher return into a Python exception; the drivers'
    per-utterance try/except then logs and skips (preprocess_speech.py:46,72-73)."""
    if rc != 0:
        msg = lib.ser_last_error()
        raise SerHipError(f"{what or 'libserhip'} failed (rc={rc}): {msg.decode() if msg else ''}")
