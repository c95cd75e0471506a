"""Frozen-encoder weights: HuggingFace state-dict names in, plain fp32 tensors out.

The reference obtains weights with ``AutoModel.from_pretrained(--ssl_type)``
(preprocess_speech.py:112, preprocess_whisper.py:120).  There is no network in
the build or benchmark environment, so two sources are supported:

* ``load_checkpoint(path)``   -- a local ``*.safetensors`` / ``pytorch_model.bin``
  (file or HF snapshot directory) using the hub's parameter names, so a real
  checkpoint drops in unchanged;
* ``synthetic_state_dict(geo, seed)`` -- seeded random tensors of the same names
  and shapes (benchmarks, parity tests).  Scales are chosen so that every code
  path carries signal: non-unit LayerNorm affine, peaky attention, non-trivial
  relative-position bias and gate.
"""
from __future__ import annotations

import math
import os
from typing import Dict

import numpy as np
import torch

from .config import EncoderGeometry, FAMILY_DEBERTA, FAMILY_ROBERTA, FAMILY_WAVLM, FAMILY_WHISPER

StateDict = Dict[str, torch.Tensor]


class _Rng:
    """numpy PCG64 stream: unlike torch's CPU ``randn`` (whose vectorised fill depends on the
    host's SIMD width) it yields the same numbers in the build container and on the GPU box."""

    def __init__(self, seed: int, fast: bool = False):
        self.g = np.random.default_rng(int(seed))
        # fast: torch's vectorised CPU generator (several GB/s instead of numpy's ~0.3) for throughput-only runs of the multi-billion
        # parameter geometries (bench.py other_encoders); NOT stable across hosts, so never used where a digest or an oracle compares
        self.t = torch.Generator().manual_seed(int(seed)) if fast else None

    def normal(self, *shape, std=1.0, mean=0.0):
        if self.t is not None:
            return torch.randn(*shape, generator=self.t) * std + mean
        x = self.g.standard_normal(size=shape, dtype=np.float32)
        return torch.from_numpy(x) * std + mean


def _linear(sd, r, name, out_f, in_f, gain=0.7, bias=True):
    sd[name + ".weight"] = r.normal(out_f, in_f, std=gain / math.sqrt(in_f))
    if bias:
        sd[name + ".bias"] = r.normal(out_f, std=0.05)


def _layer_norm(sd, r, name, dim):
    sd[name + ".weight"] = r.normal(dim, std=0.1, mean=1.0)
    sd[name + ".bias"] = r.normal(dim, std=0.1)


def whisper_sinusoids(length: int, channels: int) -> torch.Tensor:
    """Whisper's frozen position table at init (HF modeling_whisper.py:55-64)."""
    inc = math.log(10000.0) / (channels // 2 - 1)
    inv = np.exp(-inc * np.arange(channels // 2, dtype=np.float64))
    t = np.arange(length, dtype=np.float64)[:, None] * inv[None, :]
    # float64 then one rounding: bit-identical on every host (fixture digests depend on it)
    return torch.from_numpy(np.concatenate([np.sin(t), np.cos(t)], axis=1).astype(np.float32))


def synthetic_state_dict(geo: EncoderGeometry, seed: int = 0, fast: bool = False) -> StateDict:
    r = _Rng(seed, fast)
    sd: StateDict = {}
    D, H, Fd, dh = geo.hidden, geo.heads, geo.ffn, geo.head_dim
    if geo.family == FAMILY_ROBERTA:
        sd["embeddings.word_embeddings.weight"] = r.normal(geo.vocab_size, D, std=0.5)
        sd["embeddings.position_embeddings.weight"] = r.normal(geo.max_positions, D, std=0.3)
        sd["embeddings.token_type_embeddings.weight"] = r.normal(geo.type_vocab_size, D, std=0.2)
        _layer_norm(sd, r, "embeddings.LayerNorm", D)
        for i in range(geo.num_layers):
            p = f"encoder.layer.{i}"
            _linear(sd, r, p + ".attention.self.query", D, D, gain=1.6)
            _linear(sd, r, p + ".attention.self.key", D, D, gain=1.6)
            _linear(sd, r, p + ".attention.self.value", D, D)
            _linear(sd, r, p + ".attention.output.dense", D, D)
            _layer_norm(sd, r, p + ".attention.output.LayerNorm", D)
            _linear(sd, r, p + ".intermediate.dense", Fd, D)
            _linear(sd, r, p + ".output.dense", D, Fd)
            _layer_norm(sd, r, p + ".output.LayerNorm", D)
        return sd
    if geo.family == FAMILY_DEBERTA:       # HF DebertaV2Model names (v3 settings: shared q/k position projections)
        sd["embeddings.word_embeddings.weight"] = r.normal(geo.vocab_size, D, std=0.5)
        _layer_norm(sd, r, "embeddings.LayerNorm", D)
        sd["encoder.rel_embeddings.weight"] = r.normal(2 * geo.position_buckets, D, std=0.4)
        _layer_norm(sd, r, "encoder.LayerNorm", D)
        if geo.text_conv_kernel:           # DebertaV2Encoder.conv (ConvLayer): Conv1d(D, D, k) over tokens + its LayerNorm
            k = geo.text_conv_kernel
            sd["encoder.conv.conv.weight"] = r.normal(D, D, k, std=0.7 / math.sqrt(D * k))
            sd["encoder.conv.conv.bias"] = r.normal(D, std=0.05)
            _layer_norm(sd, r, "encoder.conv.LayerNorm", D)
        for i in range(geo.num_layers):
            p = f"encoder.layer.{i}"
            _linear(sd, r, p + ".attention.self.query_proj", D, D, gain=1.6)
            _linear(sd, r, p + ".attention.self.key_proj", D, D, gain=1.6)
            _linear(sd, r, p + ".attention.self.value_proj", D, D)
            _linear(sd, r, p + ".attention.output.dense", D, D)
            _layer_norm(sd, r, p + ".attention.output.LayerNorm", D)
            _linear(sd, r, p + ".intermediate.dense", Fd, D)
            _linear(sd, r, p + ".output.dense", D, Fd)
            _layer_norm(sd, r, p + ".output.LayerNorm", D)
        return sd
    if geo.family == FAMILY_WHISPER:
        sd["encoder.conv1.weight"] = r.normal(D, geo.n_mels, 3, std=math.sqrt(2.0 / (geo.n_mels * 3)))
        sd["encoder.conv1.bias"] = r.normal(D, std=0.05)
        sd["encoder.conv2.weight"] = r.normal(D, D, 3, std=math.sqrt(2.0 / (D * 3)))
        sd["encoder.conv2.bias"] = r.normal(D, std=0.05)
        sd["encoder.embed_positions.weight"] = whisper_sinusoids(geo.max_source_positions, D)
        for i in range(geo.num_layers):
            p = f"encoder.layers.{i}"
            _layer_norm(sd, r, p + ".self_attn_layer_norm", D)
            _linear(sd, r, p + ".self_attn.q_proj", D, D, gain=1.6)
            _linear(sd, r, p + ".self_attn.k_proj", D, D, gain=1.6, bias=False)
            _linear(sd, r, p + ".self_attn.v_proj", D, D)
            _linear(sd, r, p + ".self_attn.out_proj", D, D)
            _layer_norm(sd, r, p + ".final_layer_norm", D)
            _linear(sd, r, p + ".fc1", Fd, D)
            _linear(sd, r, p + ".fc2", D, Fd)
        _layer_norm(sd, r, "encoder.layer_norm", D)
        return sd

    cin = 1
    for i, (c, k) in enumerate(zip(geo.conv_dim, geo.conv_kernel)):
        p = f"feature_extractor.conv_layers.{i}"
        sd[p + ".conv.weight"] = r.normal(c, cin, k, std=math.sqrt(2.0 / (cin * k)))
        if geo.conv_bias:
            sd[p + ".conv.bias"] = r.normal(c, std=0.05)
        _layer_norm(sd, r, p + ".layer_norm", c)
        cin = c
    if geo.feat_proj_layer_norm:
        _layer_norm(sd, r, "feature_projection.layer_norm", cin)
    _linear(sd, r, "feature_projection.projection", D, cin)

    cg = D // geo.pos_conv_groups
    v = r.normal(D, cg, geo.pos_conv_kernel, std=math.sqrt(2.0 / (cg * geo.pos_conv_kernel)))
    vnorm = np.sqrt((v.numpy().astype(np.float64) ** 2).sum(axis=(0, 1), keepdims=True)).astype(np.float32)
    g = torch.from_numpy(vnorm) * r.normal(1, 1, geo.pos_conv_kernel, std=0.1, mean=1.0)
    sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = g
    sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = v
    sd["encoder.pos_conv_embed.conv.bias"] = r.normal(D, std=0.05)
    _layer_norm(sd, r, "encoder.layer_norm", D)
    for i in range(geo.num_layers):
        p = f"encoder.layers.{i}"
        a = p + ".attention"
        _linear(sd, r, a + ".q_proj", D, D, gain=1.6)
        _linear(sd, r, a + ".k_proj", D, D, gain=1.6)
        _linear(sd, r, a + ".v_proj", D, D)
        _linear(sd, r, a + ".out_proj", D, D)
        if geo.family == FAMILY_WAVLM:
            sd[a + ".gru_rel_pos_const"] = r.normal(1, H, 1, 1, std=0.2, mean=1.0)
            sd[a + ".gru_rel_pos_linear.weight"] = r.normal(8, dh, std=0.3)
            sd[a + ".gru_rel_pos_linear.bias"] = r.normal(8, std=0.3)
            if i == 0:
                sd[a + ".rel_attn_embed.weight"] = r.normal(geo.num_buckets, H, std=1.0)
        _layer_norm(sd, r, p + ".layer_norm", D)
        _linear(sd, r, p + ".feed_forward.intermediate_dense", Fd, D)
        _linear(sd, r, p + ".feed_forward.output_dense", D, Fd)
        _layer_norm(sd, r, p + ".final_layer_norm", D)
    return sd


def apply_stress(sd: StateDict, geo: EncoderGeometry, kind: str) -> StateDict:
    """Deterministic edits of a synthetic wav2vec2-style state dict that reproduce what real checkpoints do to the
    residual stream and seeded Gaussian weights never do (fixtures ``tiny_*_outlier`` / ``tiny_*_rowmean``):

    * ``"outliers"``: two "massive activation" channels -- layer 0's feed-forward output writes +800 / -500 into them
      (1000x the other channels), their weight rows are 30x larger, and every later LayerNorm damps them with a
      small gamma, as trained models do;
    * ``"rowmean"``: a uniform offset on every channel (positional-conv bias +40, which passes its GELU unchanged; each
      feed-forward output bias +3; one attention LayerNorm bias x8), so that rows have |mean| >> std: the one-pass
      variance and the deferred-LayerNorm term ``acc - mean * colsum`` of csrc/gemm.hip cancel catastrophically
      unless the operands are stored shifted;
    * ``"sharp"``: q / k projections x4 (logits x16): near one-hot attention rows.
    """
    D = geo.hidden
    sd = {k: v.clone() for k, v in sd.items()}
    fc2 = "encoder.layers.{}.feed_forward.output_dense"
    if kind == "outliers":
        c1, c2 = 7, D - 5
        sd[fc2.format(0) + ".bias"][c1] += 800.0
        sd[fc2.format(0) + ".bias"][c2] -= 500.0
        sd[fc2.format(0) + ".weight"][c1] *= 30.0
        sd[fc2.format(0) + ".weight"][c2] *= 30.0
        for k in sd:
            if k.endswith("layer_norm.weight") and k.startswith("encoder."):
                sd[k][c1] = 0.05
                sd[k][c2] = -0.05
    elif kind == "rowmean":
        sd["encoder.pos_conv_embed.conv.bias"] += 40.0
        for i in range(geo.num_layers):
            sd[fc2.format(i) + ".bias"] += 3.0
        sd["encoder.layers.0.layer_norm.bias"] *= 8.0
    elif kind == "sharp":
        # sharp attention: query and key projections 4x larger -> logits 16x larger, softmax rows near one-hot -- what
        # trained checkpoints (and LoRA-scaled query projections) have and Gaussian weights do not.  A softmax weight moves
        # by (logit error) * ln 2, so this is the fixture that separates single-product q / k rounding from the fp32 reference.
        for i in range(geo.num_layers):
            for proj in ("q_proj", "k_proj"):
                for leaf in ("weight", "bias"):
                    sd[f"encoder.layers.{i}.attention.{proj}.{leaf}"] *= 4.0
    else:
        raise ValueError(f"unknown stress kind '{kind}'")
    return sd


_STRIP_PREFIXES = ("wavlm.", "wav2vec2.", "hubert.", "model.", "roberta.", "deberta.")


def normalize_names(sd: StateDict) -> StateDict:
    """Strip task-head wrappers (``wav2vec2.`` in *ForCTC checkpoints, ``model.``
    in WhisperForConditionalGeneration) and drop everything off the encoder path
    (decoder, lm_head, quantizer, masked_spec_embed)."""
    out: StateDict = {}
    for k, v in sd.items():
        for p in _STRIP_PREFIXES:
            if k.startswith(p):
                k = k[len(p):]
                break
        if k.startswith(("decoder.", "lm_head", "proj_out", "quantizer", "project_", "masked_spec_embed", "pooler.",
                         "embeddings.position_ids")):
            continue
        out[k] = v.detach().to(torch.float32).contiguous()
    return out


def merge_lora(sd: StateDict, lora_alpha: float = 16.0) -> StateDict:
    """Fold PEFT LoRA adapters into their base weights (next row 8f-4).  The reference fine-tunes WavLM with
    ``LoraConfig(r=8, lora_alpha=16, target_modules=['q_proj','v_proj'])`` inside a classifier wrapper and extracts
    with ``ssl_model.wavlm.model`` in eval mode (preprocessing/preprocess_speech_pretrained.py:108-177): dropout is
    inactive, so every adapted Linear computes ``x W^T + (alpha/r) x A^T B^T`` = a plain Linear with
    ``W + (alpha/r) B A``.  Keys: ``<wrapper>.base_model.model.<name>.base_layer.weight|bias``,
    ``<name>.lora_A.<adapter>.weight`` [r, in], ``<name>.lora_B.<adapter>.weight`` [out, r]; the classifier head
    and anything else off the encoder are dropped later by ``normalize_names``.  ``alpha`` is not stored in a
    state dict (it lives in LoraConfig), hence the argument; r is read from A's shape.  A state dict without
    adapter keys is returned unchanged."""
    if not any(".lora_A." in k for k in sd):
        return sd
    out: StateDict = {}
    adapters = {}
    for k, v in sd.items():
        k2 = k
        for wrap in ("wavlm.base_model.model.", "whisper.base_model.model.", "base_model.model."):
            if k2.startswith(wrap):
                k2 = k2[len(wrap):]
                break
        if ".lora_A." in k2 or ".lora_B." in k2:
            mod, rest = k2.split(".lora_", 1)
            adapters.setdefault(mod, {})[rest[0]] = v.detach().to(torch.float64)
        elif ".lora_dropout" in k2 or ".lora_embedding" in k2 or k2.startswith("classifier."):
            continue
        else:
            out[k2.replace(".base_layer.", ".")] = v
    for mod, ab in adapters.items():
        if "A" not in ab or "B" not in ab or mod + ".weight" not in out:
            raise OSError(f"incomplete LoRA adapter for '{mod}'")
        a, b = ab["A"], ab["B"]
        scale = float(lora_alpha) / a.shape[0]
        out[mod + ".weight"] = (out[mod + ".weight"].detach().to(torch.float64) + scale * (b @ a)).to(torch.float32)
    return out


def load_checkpoint(path: str, lora_alpha: float = 16.0) -> StateDict:
    """Read a local checkpoint file or directory.  Raises ``OSError`` when nothing
    loadable is found -- the error class the reference's driver reports as
    "No pretrained model found" (preprocess_speech.py:115-117)."""
    candidates = []
    if os.path.isdir(path):
        for fn in sorted(os.listdir(path)):
            if fn.endswith(".safetensors") or fn in ("pytorch_model.bin",):
                candidates.append(os.path.join(path, fn))
    elif os.path.isfile(path):
        candidates.append(path)
    if not candidates:
        raise OSError(f"no checkpoint (*.safetensors / pytorch_model.bin) under '{path}'")
    sd: StateDict = {}
    for fn in candidates:
        if fn.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd.update(load_file(fn, device="cpu"))
        else:
            sd.update(torch.load(fn, map_location="cpu", weights_only=True))
    return normalize_names(merge_lora(sd, lora_alpha))


def fold_wavlm_gate(w8: torch.Tensor, b8: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, heads: int, head_dim: int):
    """Load-time fold of WavLM's GRU gate (HF modeling_wavlm.py:167-180: Linear(dh -> 8) on LayerNorm1(x) per head, the eight
    outputs summed in two fours) for ser_attention's in-kernel gate (ser_attention_args.gate_x): with the RAW layer input x of a
    row, its mean mu and rstd,
        pre_j[h] = rstd * (x_h . wg[2h + j] - mu * sum(wg[2h + j])) + t[h, j],      j = 0 (first four), 1 (last four)
    Returns (wg [2H, dh] = gamma-folded summed weights, t [H, 2] = beta . w_j + b_j), both float64.  The column sums the kernel
    subtracts are taken by the caller from the ROUNDED operand planes it uploads, like every deferred-LayerNorm GEMM's colsum."""
    w8, b8 = w8.detach().double(), b8.detach().double()
    wab = torch.stack([w8[:4].sum(0), w8[4:].sum(0)], 0)                               # [2, dh]
    gam, bet = gamma.detach().double().view(heads, 1, head_dim), beta.detach().double().view(heads, 1, head_dim)
    wg = (gam * wab[None]).reshape(2 * heads, head_dim)
    t = (bet * wab[None]).sum(2) + torch.stack([b8[:4].sum(), b8[4:].sum()])[None]
    return wg, t


def state_dict_digest(sd: StateDict) -> str:
    """Order-independent checksum of a state dict (fixtures record it so a drift of
    the RNG stream between containers is detected instead of silently compared)."""
    import hashlib
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().contiguous().numpy().tobytes())
    return h.hexdigest()[:16]
