"""Encoder geometries for the SSL embedding-extraction path.

The reference never spells these numbers out: it calls
``AutoModel.from_pretrained(--ssl_type)`` (preprocess_speech.py:111-114,
preprocess_whisper.py:119-122) and the hub ``config.json`` supplies them.
SURVEY.md section 8a cross-checks the values below against parameter counts and
the ``feat1_dim`` entries of the reference's configs/*.json.

A geometry is a plain dataclass so that the CPU oracle (oracle/) can consume it
by attribute access without importing this package.
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Tuple

FAMILY_WAVLM = "wavlm"
FAMILY_WAV2VEC2 = "wav2vec2"
FAMILY_HUBERT = "hubert"
FAMILY_WHISPER = "whisper"
FAMILY_ROBERTA = "roberta"      # text side of the bimodal heads (next row 8f-1)
FAMILY_DEBERTA = "deberta"      # DeBERTa-v2/v3 variant of the text side (engine.DebertaEncoder, csrc/deberta.hip)

SPEECH_FAMILIES = (FAMILY_WAVLM, FAMILY_WAV2VEC2, FAMILY_HUBERT)


@dataclass(frozen=True)
class EncoderGeometry:
    family: str
    num_layers: int
    hidden: int
    heads: int
    ffn: int
    # wav2vec2-style convolutional waveform encoder (all three speech families)
    conv_dim: Tuple[int, ...] = (512,) * 7
    conv_kernel: Tuple[int, ...] = (10, 3, 3, 3, 3, 2, 2)
    conv_stride: Tuple[int, ...] = (5, 2, 2, 2, 2, 2, 2)
    conv_bias: bool = False
    feat_proj_layer_norm: bool = True      # HuBERT makes this optional
    pos_conv_kernel: int = 128
    pos_conv_groups: int = 16
    # WavLM gated relative position bias
    num_buckets: int = 320
    max_bucket_distance: int = 800
    layer_norm_eps: float = 1e-5
    # Whisper encoder
    n_mels: int = 128
    max_source_positions: int = 1500
    # RoBERTa text encoder
    vocab_size: int = 50265
    max_positions: int = 514
    pad_token_id: int = 1
    type_vocab_size: int = 1
    # DeBERTa-v2/v3 disentangled attention (log-bucketed relative positions, shared q/k projections for positions)
    position_buckets: int = 256
    # DeBERTa-v2 xlarge / xxlarge: ConvLayer after encoder layer 0 (HF modeling_deberta_v2.py ConvLayer; 0 = none, as in v3)
    text_conv_kernel: int = 0
    name: str = ""

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def num_hidden_states(self) -> int:
        return self.num_layers + 1

    def frames_for(self, num_samples: int) -> int:
        """Integer frame count of the conv stack: floor((L-k)/s)+1 per layer
        (HF modeling_wavlm.py:633-652).  Bit-exact gate of SURVEY 8a row a8."""
        n = int(num_samples)
        for k, s in zip(self.conv_kernel, self.conv_stride):
            n = (n - k) // s + 1
        return n

    def frame_chain(self, num_samples: int):
        out = []
        n = int(num_samples)
        for k, s in zip(self.conv_kernel, self.conv_stride):
            n = (n - k) // s + 1
            out.append(n)
        return out


WAVLM_LARGE = EncoderGeometry(
    family=FAMILY_WAVLM, num_layers=24, hidden=1024, heads=16, ffn=4096,
    conv_bias=False, name="microsoft/wavlm-large")

XLSR_2B = EncoderGeometry(
    family=FAMILY_WAV2VEC2, num_layers=48, hidden=1920, heads=16, ffn=7680,
    conv_bias=True, name="facebook/wav2vec2-xls-r-2b")

HUBERT_XLARGE = EncoderGeometry(
    family=FAMILY_HUBERT, num_layers=48, hidden=1280, heads=16, ffn=5120,
    conv_bias=True, feat_proj_layer_norm=True,
    name="facebook/hubert-xlarge-ls960-ft")

WHISPER_LARGE_V3 = EncoderGeometry(
    family=FAMILY_WHISPER, num_layers=32, hidden=1280, heads=20, ffn=5120,
    n_mels=128, max_source_positions=1500, name="openai/whisper-large-v3")

ROBERTA_LARGE = EncoderGeometry(
    family=FAMILY_ROBERTA, num_layers=24, hidden=1024, heads=16, ffn=4096, name="roberta-large")

# deberta-v3: DebertaV2Config(position_buckets=256, share_att_key, pos_att_type p2c|c2p, norm_rel_ebd layer_norm,
# position_biased_input False, type_vocab_size 0, layer_norm_eps 1e-7, pad_token_id 0, vocab 128100)
DEBERTA_V3_LARGE = EncoderGeometry(
    family=FAMILY_DEBERTA, num_layers=24, hidden=1024, heads=16, ffn=4096, vocab_size=128100, max_positions=512,
    pad_token_id=0, type_vocab_size=0, layer_norm_eps=1e-7, position_buckets=256, name="microsoft/deberta-v3-large")
DEBERTA_V3_BASE = EncoderGeometry(
    family=FAMILY_DEBERTA, num_layers=12, hidden=768, heads=12, ffn=3072, vocab_size=128100, max_positions=512,
    pad_token_id=0, type_vocab_size=0, layer_norm_eps=1e-7, position_buckets=256, name="microsoft/deberta-v3-base")

# deberta-v2-xlarge is what the reference's README runs preprocess_deroberta.py with (README.md:66): the v3 attention
# settings plus conv_kernel_size = 3, conv_act = "gelu" (a token-axis Conv1d of the embeddings added to layer 0's output)
DEBERTA_V2_XLARGE = EncoderGeometry(
    family=FAMILY_DEBERTA, num_layers=24, hidden=1536, heads=24, ffn=6144, vocab_size=128100, max_positions=512,
    pad_token_id=0, type_vocab_size=0, layer_norm_eps=1e-7, position_buckets=256, text_conv_kernel=3,
    name="microsoft/deberta-v2-xlarge")
DEBERTA_V2_XXLARGE = EncoderGeometry(
    family=FAMILY_DEBERTA, num_layers=48, hidden=1536, heads=24, ffn=6144, vocab_size=128100, max_positions=512,
    pad_token_id=0, type_vocab_size=0, layer_norm_eps=1e-7, position_buckets=256, text_conv_kernel=3,
    name="microsoft/deberta-v2-xxlarge")

_REGISTRY = {
    "microsoft/deberta-v3-large": DEBERTA_V3_LARGE,
    "microsoft/deberta-v2-xlarge": DEBERTA_V2_XLARGE,
    "microsoft/deberta-v2-xxlarge": DEBERTA_V2_XXLARGE,
    "microsoft/deberta-v3-base": DEBERTA_V3_BASE,
    "roberta-large": ROBERTA_LARGE,
    "FacebookAI/roberta-large": ROBERTA_LARGE,
    "microsoft/wavlm-large": WAVLM_LARGE,
    "wavlm-large": WAVLM_LARGE,                       # the reference's argparse default
    "facebook/wav2vec2-xls-r-2b": XLSR_2B,
    "facebook/hubert-xlarge-ls960-ft": HUBERT_XLARGE,
    "facebook/hubert-xlarge-ll60k": HUBERT_XLARGE,
    "openai/whisper-large-v3": WHISPER_LARGE_V3,
}


def tiny_geometry(family: str, *, hidden: int = 128, heads: int = 2, layers: int = 2,
                  ffn: int = 256, conv_dim: int = 64, pos_groups: int = 2, text_conv_kernel: int = 0) -> EncoderGeometry:
    """Small geometries with the real kernel/stride tuples; used by the parity
    fixtures under tests/golden.  ``hidden // heads`` selects
    the head-dim code path (64 WavLM/Whisper, 80 HuBERT-XL, 120 XLS-R-2B) and
    ``hidden // pos_groups`` the pos-conv group width (64 / 80 / 120 in the real models)."""
    if family == FAMILY_ROBERTA:
        return EncoderGeometry(family=family, num_layers=layers, hidden=hidden, heads=heads, ffn=ffn,
                               vocab_size=300, max_positions=90, name=f"tiny-{family}-d{hidden}h{heads}")
    if family == FAMILY_DEBERTA:
        # 16 buckets: relative distances beyond +-8 are log-bucketed already at 80 tokens, like +-128 at 512 in v3-large
        return EncoderGeometry(family=family, num_layers=layers, hidden=hidden, heads=heads, ffn=ffn, vocab_size=300,
                               max_positions=512, pad_token_id=0, type_vocab_size=0, layer_norm_eps=1e-7,
                               position_buckets=16, text_conv_kernel=text_conv_kernel,
                               name=f"tiny-{family}-d{hidden}h{heads}" + ("-conv" if text_conv_kernel else ""))
    if family == FAMILY_WHISPER:
        return EncoderGeometry(family=family, num_layers=layers, hidden=hidden, heads=heads,
                               ffn=ffn, n_mels=128, max_source_positions=1500,
                               name=f"tiny-{family}-d{hidden}h{heads}")
    return EncoderGeometry(
        family=family, num_layers=layers, hidden=hidden, heads=heads, ffn=ffn,
        conv_dim=(conv_dim,) * 7, conv_bias=(family != FAMILY_WAVLM),
        pos_conv_groups=pos_groups,
        name=f"tiny-{family}-d{hidden}h{heads}")


def geometry_for(ssl_type: str) -> EncoderGeometry:
    """Map ``--ssl_type`` to a geometry.  Unknown names raise ``OSError`` because
    that is what ``from_pretrained`` raises in the reference and what its driver
    catches (preprocess_speech.py:115-117)."""
    key = ssl_type.strip()
    if key in _REGISTRY:
        return _REGISTRY[key]
    low = key.lower()
    for name, geo in _REGISTRY.items():
        if low == name.lower() or low == name.split("/")[-1].lower():
            return geo
    raise OSError(f"No geometry registered for ssl_type '{ssl_type}'")


def geometry_from_config(cfg: dict, name: str = "") -> EncoderGeometry:
    """Geometry from a checkpoint's ``config.json`` -- what ``AutoModel.from_pretrained(--ssl_type)`` reads for ANY hub id or
    local snapshot (preprocess_speech.py:111-112, preprocess_whisper.py:119-120), so fine-tunes published under another name
    work without a registry entry.  Variants this build does not implement are refused with ``OSError``, the class the
    reference's driver reports as "No pretrained model found" (:115-117): GroupNorm feature extractors
    (``feat_extract_norm="group"``: wav2vec2-base, hubert-base / large-ll60k) and post-LayerNorm encoders
    (``do_stable_layer_norm=False``)."""
    mt = str(cfg.get("model_type", "")).lower()
    name = name or str(cfg.get("_name_or_path", "")) or mt
    if mt in (FAMILY_WAVLM, FAMILY_WAV2VEC2, FAMILY_HUBERT):
        if cfg.get("feat_extract_norm", "group") != "layer":
            raise OSError(f"{name}: feat_extract_norm='{cfg.get('feat_extract_norm', 'group')}' (GroupNorm over time) is not supported; "
                          "the path implements the layer-norm feature extractor of the *-large / xlarge / XLS-R checkpoints")
        if not cfg.get("do_stable_layer_norm", False):
            raise OSError(f"{name}: do_stable_layer_norm=False (post-LayerNorm encoder) is not supported")
        if mt == FAMILY_HUBERT and not cfg.get("feat_proj_layer_norm", True):
            raise OSError(f"{name}: feat_proj_layer_norm=False is not supported")
        conv_dim = tuple(int(c) for c in cfg.get("conv_dim", (512,) * 7))
        return EncoderGeometry(
            family=mt, num_layers=int(cfg["num_hidden_layers"]), hidden=int(cfg["hidden_size"]),
            heads=int(cfg["num_attention_heads"]), ffn=int(cfg["intermediate_size"]), conv_dim=conv_dim,
            conv_kernel=tuple(int(k) for k in cfg.get("conv_kernel", (10, 3, 3, 3, 3, 2, 2))),
            conv_stride=tuple(int(k) for k in cfg.get("conv_stride", (5, 2, 2, 2, 2, 2, 2))),
            conv_bias=bool(cfg.get("conv_bias", False)), feat_proj_layer_norm=bool(cfg.get("feat_proj_layer_norm", True)),
            pos_conv_kernel=int(cfg.get("num_conv_pos_embeddings", 128)), pos_conv_groups=int(cfg.get("num_conv_pos_embedding_groups", 16)),
            num_buckets=int(cfg.get("num_buckets", 320)), max_bucket_distance=int(cfg.get("max_bucket_distance", 800)),
            layer_norm_eps=float(cfg.get("layer_norm_eps", 1e-5)), name=name)
    if mt == FAMILY_WHISPER:
        return EncoderGeometry(
            family=FAMILY_WHISPER, num_layers=int(cfg["encoder_layers"]), hidden=int(cfg["d_model"]),
            heads=int(cfg["encoder_attention_heads"]), ffn=int(cfg["encoder_ffn_dim"]), n_mels=int(cfg.get("num_mel_bins", 80)),
            max_source_positions=int(cfg.get("max_source_positions", 1500)), name=name)
    if mt in ("roberta", "xlm-roberta"):
        return EncoderGeometry(
            family=FAMILY_ROBERTA, num_layers=int(cfg["num_hidden_layers"]), hidden=int(cfg["hidden_size"]),
            heads=int(cfg["num_attention_heads"]), ffn=int(cfg["intermediate_size"]), vocab_size=int(cfg["vocab_size"]),
            max_positions=int(cfg.get("max_position_embeddings", 514)), pad_token_id=int(cfg.get("pad_token_id", 1)),
            type_vocab_size=int(cfg.get("type_vocab_size", 1)), layer_norm_eps=float(cfg.get("layer_norm_eps", 1e-5)), name=name)
    if mt == "deberta-v2":
        pos_att = cfg.get("pos_att_type") or []
        pos_att = pos_att.split("|") if isinstance(pos_att, str) else list(pos_att)
        if not cfg.get("relative_attention", False) or sorted(pos_att) != ["c2p", "p2c"] or not cfg.get("share_att_key", False) \
                or cfg.get("position_biased_input", True) or str(cfg.get("norm_rel_ebd", "none")) != "layer_norm":
            raise OSError(f"{name}: only the deberta-v3 / v2-xlarge attention configuration is supported (relative_attention, "
                          "pos_att_type p2c|c2p, share_att_key, norm_rel_ebd layer_norm, no absolute positions)")
        return EncoderGeometry(
            family=FAMILY_DEBERTA, num_layers=int(cfg["num_hidden_layers"]), hidden=int(cfg["hidden_size"]),
            heads=int(cfg["num_attention_heads"]), ffn=int(cfg["intermediate_size"]), vocab_size=int(cfg["vocab_size"]),
            max_positions=int(cfg.get("max_position_embeddings", 512)), pad_token_id=int(cfg.get("pad_token_id", 0)),
            type_vocab_size=int(cfg.get("type_vocab_size", 0)), layer_norm_eps=float(cfg.get("layer_norm_eps", 1e-7)),
            position_buckets=int(cfg.get("position_buckets", 256)), text_conv_kernel=int(cfg.get("conv_kernel_size", 0) or 0), name=name)
    raise OSError(f"{name}: model_type '{mt}' is not an encoder this path implements "
                  "(wavlm / wav2vec2 / hubert / whisper / roberta / deberta-v2)")


def find_config_json(ssl_type: str, checkpoint: str = "") -> str:
    """Path of the ``config.json`` that belongs to the weights the driver will load: next to ``--checkpoint`` (a snapshot
    directory or a file inside one), else in the offline HF cache snapshot of ``--ssl_type``; "" when there is none."""
    import os
    cands = []
    if checkpoint:
        cands.append(os.path.join(checkpoint if os.path.isdir(checkpoint) else os.path.dirname(os.path.abspath(checkpoint)), "config.json"))
    elif os.path.isdir(ssl_type):
        cands.append(os.path.join(ssl_type, "config.json"))
    home = os.environ.get("HF_HOME", os.path.join(os.path.expanduser("~"), ".cache", "huggingface"))
    snap = os.path.join(home, "hub", "models--" + ssl_type.replace("/", "--"), "snapshots")
    if not checkpoint and os.path.isdir(snap):
        cands += [os.path.join(snap, rev, "config.json") for rev in sorted(os.listdir(snap))]
    for c in cands:
        if os.path.isfile(c):
            return c
    return ""


def resolve_geometry(ssl_type: str, checkpoint: str = "") -> EncoderGeometry:
    """``config.json`` of the checkpoint when there is one (any name), else the built-in table by ``--ssl_type``."""
    import json
    path = find_config_json(ssl_type, checkpoint)
    if path:
        try:
            with open(path, "r") as f:
                cfg = json.load(f)
        except (OSError, ValueError) as e:
            raise OSError(f"cannot read {path}: {e}")
        return geometry_from_config(cfg, name=ssl_type)
    return geometry_for(ssl_type)


def with_layers(geo: EncoderGeometry, layers: int) -> EncoderGeometry:
    return replace(geo, num_layers=layers)


# The four fixture geometries under tests/golden (head dims 64 / 120 / 80 / 64 and
# pos-conv group widths 64 / 120 / 80, i.e. every code path of the real models).
TINY_WAVLM = tiny_geometry(FAMILY_WAVLM, hidden=128, heads=2, ffn=256, pos_groups=2)
TINY_WAV2VEC2 = tiny_geometry(FAMILY_WAV2VEC2, hidden=960, heads=8, ffn=512, pos_groups=8)
TINY_HUBERT = tiny_geometry(FAMILY_HUBERT, hidden=320, heads=4, ffn=384, pos_groups=4)
TINY_WHISPER = tiny_geometry(FAMILY_WHISPER, hidden=128, heads=2, ffn=256)
TINY_ROBERTA = tiny_geometry(FAMILY_ROBERTA, hidden=128, heads=2, ffn=256)
TINY_DEBERTA = tiny_geometry(FAMILY_DEBERTA, hidden=128, heads=2, ffn=256)
TINY_DEBERTA_CONV = tiny_geometry(FAMILY_DEBERTA, hidden=128, heads=2, ffn=256, text_conv_kernel=3)      # deberta-v2-xlarge style
