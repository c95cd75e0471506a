"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

fp32 PyTorch-CPU restatement of the arithmetic on the reference's SSL
embedding-extraction hot path.  Only tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py may import this module; the product package
(interspeech_ser_amd/) never does, and fails loudly without its HIP library.

The reference (AI-Unicamp/interspeech_ser) keeps no arithmetic of its own on
this path: preprocessing/preprocess_speech.py:47-71 and
preprocessing/preprocess_whisper.py:47-80 call the third-party packages
``transformers==4.47.1`` (benchmark/requirements.txt:35) and ``librosa``.  This
file restates the published algorithms of those HuggingFace modules, function by
function, from the call sites listed in SURVEY.md section 8a:

  a6  Wav2Vec2FeatureExtractor.zero_mean_unit_var_norm   -> zero_mean_unit_var
  a7  *FeatureEncoder (7 x Conv1d -> LayerNorm(C) -> GELU) -> conv_feature_encoder
  a8  _get_feat_extract_output_lengths                    -> geometry.frames_for
  a9  *FeatureProjection                                  -> feature_projection
  a10 *PositionalConvEmbedding + SamePad                  -> positional_conv
  a11/a12 *EncoderStableLayerNorm / *EncoderLayerStableLayerNorm -> speech_hidden_states
  a13 WavLMAttention (bucketed relative bias + GRU gate)  -> relative_buckets, wavlm_attention
  a14 Wav2Vec2Attention / HubertAttention                 -> plain_attention
  a15 *FeedForward                                        -> feed_forward
  a16 WhisperFeatureExtractor._torch_extract_fbank_features -> whisper_log_mel
  a17/a18 WhisperEncoder / WhisperEncoderLayer            -> whisper_hidden_states
  a19 layer selection, a20 Whisper crop                   -> select_state, whisper_crop_rows

PARITY PIN: the reference has no tests, golden vectors or fixtures for this path
("parity unpinned" by the reference itself).  This restatement is
pinned instead against outputs of the very HuggingFace classes the reference
calls, generated in the build container by oracle/make_golden.py and committed
under tests/golden/ (tests/test_oracle_golden.py replays them).

The reference runs batch = 1 per call (preprocess_speech.py:76-81), so every
function here takes ONE utterance; ragged batches in the product must equal
these per-utterance results.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
StateDict = Dict[str, Tensor]


# --------------------------------------------------------------------------- a6
def zero_mean_unit_var(wave: np.ndarray) -> np.ndarray:
    """(x - mean) / sqrt(var + 1e-7), population variance, numpy fp32
    (HF feature_extraction_wav2vec2.py:77-97; call site preprocess_speech.py:48)."""
    x = np.asarray(wave, dtype=np.float32)
    return ((x - x.mean()) / np.sqrt(x.var() + 1e-7)).astype(np.float32)


# ------------------------------------------------------------------------ a7-a10
def _ln(x: Tensor, sd: StateDict, prefix: str, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def conv_feature_encoder(geo, sd: StateDict, x: Tensor) -> Tensor:
    """[L] waveform -> [T, C] frames.  Each of the 7 layers is Conv1d (no padding)
    -> LayerNorm over channels (eps 1e-5, affine) -> exact-erf GELU
    (HF modeling_wavlm.py:696-720,747-782)."""
    h = x[None, None, :]
    for i, (k, s) in enumerate(zip(geo.conv_kernel, geo.conv_stride)):
        p = f"feature_extractor.conv_layers.{i}"
        h = F.conv1d(h, sd[p + ".conv.weight"], sd.get(p + ".conv.bias"), stride=s)
        h = _ln(h.transpose(1, 2), sd, p + ".layer_norm", 1e-5).transpose(1, 2)
        h = F.gelu(h)
    return h[0].transpose(0, 1).contiguous()


def feature_projection(geo, sd: StateDict, feats: Tensor) -> Tensor:
    """LayerNorm(C) -> Linear(C, D) (HF modeling_wavlm.py:93-105; HuBERT makes the
    LayerNorm conditional on feat_proj_layer_norm, modeling_hubert.py:216-231)."""
    h = feats
    if geo.feat_proj_layer_norm:
        h = _ln(h, sd, "feature_projection.layer_norm", geo.layer_norm_eps)
    return F.linear(h, sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"])


def pos_conv_weight(sd: StateDict) -> Tensor:
    """Fold weight-norm(dim=2): w = g * v / ||v||, norm over dims (0,1) per tap
    (HF modeling_wavlm.py:48-75).  Accepts both parametrization namings."""
    base = "encoder.pos_conv_embed.conv."
    if base + "parametrizations.weight.original0" in sd:
        g, v = sd[base + "parametrizations.weight.original0"], sd[base + "parametrizations.weight.original1"]
    elif base + "weight_g" in sd:
        g, v = sd[base + "weight_g"], sd[base + "weight_v"]
    else:
        return sd[base + "weight"]
    norm = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    return v * (g / norm)


def positional_conv(geo, sd: StateDict, h: Tensor) -> Tensor:
    """Grouped Conv1d(D, D, k, pad k/2, groups) -> drop the last frame when k is
    even (SamePad) -> GELU (HF modeling_wavlm.py:37-90)."""
    w = pos_conv_weight(sd)
    k = geo.pos_conv_kernel
    y = F.conv1d(h.transpose(0, 1)[None], w, sd["encoder.pos_conv_embed.conv.bias"],
                 padding=k // 2, groups=geo.pos_conv_groups)
    if k % 2 == 0:
        y = y[:, :, :-1]
    return F.gelu(y)[0].transpose(0, 1)


# --------------------------------------------------------------------------- a13
def relative_buckets(rel: Tensor, num_buckets: int = 320, max_distance: int = 800) -> Tensor:
    """Bucket index of relative position rel = key - query, computed the way
    HF modeling_wavlm.py:253-271 does it: float32 log, truncation toward zero."""
    nb = num_buckets // 2
    ret = (rel > 0).to(torch.long) * nb
    n = rel.abs()
    max_exact = nb // 2
    is_small = n < max_exact
    large = torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)
    large = (max_exact + large).to(torch.long)
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return ret + torch.where(is_small, n, large)


def relative_bias_table(geo, sd: StateDict, T: int) -> Tensor:
    """[H, 2T-1] table: column (key - query) + (T-1).  Only 2T-1 distinct relative
    distances exist, so bias[h,q,k] = table[h, k-q+T-1] (HF :243-251)."""
    rel = torch.arange(-(T - 1), T, dtype=torch.long)
    emb = sd["encoder.layers.0.attention.rel_attn_embed.weight"]          # [320, H]
    return emb[relative_buckets(rel, geo.num_buckets, geo.max_bucket_distance)].transpose(0, 1).contiguous()


def wavlm_gate(geo, sd: StateDict, prefix: str, x_ln: Tensor) -> Tensor:
    """[T, D] (already layer-normed) -> gate [H, T] = a*(b*const-1)+2 with
    (a, b) = sigmoid(sum4(Linear(dh -> 8)(x_head))) (HF :167-180)."""
    T = x_ln.shape[0]
    H, dh = geo.heads, geo.head_dim
    xh = x_ln.view(T, H, dh).permute(1, 0, 2)                              # [H, T, dh]
    p = F.linear(xh, sd[prefix + ".gru_rel_pos_linear.weight"], sd[prefix + ".gru_rel_pos_linear.bias"])
    p = p.view(H, T, 2, 4).sum(-1)
    a, b = torch.sigmoid(p).unbind(-1)
    const = sd[prefix + ".gru_rel_pos_const"].view(H, 1)
    return a * (b * const - 1.0) + 2.0


def _proj(sd: StateDict, name: str, x: Tensor, bias: bool = True) -> Tensor:
    """nn.Linear, or -- when the state dict carries ``<name>.lora_A.weight`` [r, in] / ``<name>.lora_B.weight`` [out, r]
    and ``lora_scale`` -- a PEFT LoRA Linear in eval mode, applied UN-MERGED as the wrapped module computes it:
    ``x W^T + b + (alpha / r) * (x A^T) B^T``  (peft.tuners.lora.Linear.forward; the reference wraps q_proj / v_proj with
    r = 8, alpha = 16: preprocessing/preprocess_speech_pretrained.py:119-130, extracts with ``ssl_model.wavlm.model``
    :170-172).  This is what pins the product's load-time merge (weights.merge_lora)."""
    y = F.linear(x, sd[name + ".weight"], sd[name + ".bias"] if bias else None)
    if name + ".lora_A.weight" in sd:
        y = y + float(sd["lora_scale"]) * F.linear(F.linear(x, sd[name + ".lora_A.weight"]), sd[name + ".lora_B.weight"])
    return y


def _heads(x: Tensor, H: int) -> Tensor:
    T, D = x.shape
    return x.view(T, H, D // H).permute(1, 0, 2)                           # [H, T, dh]


def wavlm_attention(geo, sd: StateDict, prefix: str, x_ln: Tensor, bias_table: Tensor) -> Tensor:
    """softmax(q k^T / sqrt(dh) + gate[q] * bias[q,k]) v, then out_proj
    (HF modeling_wavlm.py:147-241 via F.multi_head_attention_forward)."""
    T = x_ln.shape[0]
    H, dh = geo.heads, geo.head_dim
    q = _heads(_proj(sd, prefix + ".q_proj", x_ln), H)
    k = _heads(_proj(sd, prefix + ".k_proj", x_ln), H)
    v = _heads(_proj(sd, prefix + ".v_proj", x_ln), H)
    idx = (torch.arange(T)[None, :] - torch.arange(T)[:, None]) + (T - 1)  # [q, k]
    bias = bias_table[:, idx]                                               # [H, T, T]
    gate = wavlm_gate(geo, sd, prefix, x_ln)                                # [H, T]
    scores = torch.matmul(q, k.transpose(1, 2)) * (dh ** -0.5) + gate[:, :, None] * bias
    ctx = torch.matmul(torch.softmax(scores, dim=-1), v)                    # [H, T, dh]
    ctx = ctx.permute(1, 0, 2).reshape(T, H * dh)
    return F.linear(ctx, sd[prefix + ".out_proj.weight"], sd[prefix + ".out_proj.bias"])


# --------------------------------------------------------------------------- a14
def plain_attention(geo, sd: StateDict, prefix: str, x_ln: Tensor, *, k_bias: bool = True) -> Tensor:
    """softmax(q k^T * dh^-0.5) v, out_proj (HF modeling_wav2vec2.py:438-548;
    Whisper scales q before the product and has no k bias, modeling_whisper.py:279-357)."""
    T = x_ln.shape[0]
    H, dh = geo.heads, geo.head_dim
    q = _heads(_proj(sd, prefix + ".q_proj", x_ln), H)
    k = _heads(_proj(sd, prefix + ".k_proj", x_ln, bias=k_bias), H)
    v = _heads(_proj(sd, prefix + ".v_proj", x_ln), H)
    scores = torch.matmul(q * (dh ** -0.5), k.transpose(1, 2))
    ctx = torch.matmul(torch.softmax(scores, dim=-1), v)
    ctx = ctx.permute(1, 0, 2).reshape(T, H * dh)
    return F.linear(ctx, sd[prefix + ".out_proj.weight"], sd[prefix + ".out_proj.bias"])


# --------------------------------------------------------------------------- a15
def feed_forward(sd: StateDict, prefix_in: str, prefix_out: str, x: Tensor) -> Tensor:
    h = F.gelu(F.linear(x, sd[prefix_in + ".weight"], sd[prefix_in + ".bias"]))
    return F.linear(h, sd[prefix_out + ".weight"], sd[prefix_out + ".bias"])


# ----------------------------------------------------------------------- a11/a12
def speech_hidden_states(geo, sd: StateDict, input_values: Tensor) -> List[Tensor]:
    """One normalised waveform [L] -> list of L+1 hidden states [T, D]
    (what ``model(**inputs, output_hidden_states=True).hidden_states`` gives with
    the batch dim squeezed, preprocess_speech.py:50-67).

    Stable-LayerNorm semantics (HF modeling_wavlm.py:465-522): hs[0] = proj +
    posconv(proj) with NO LayerNorm; hs[i] = output of layer i-1; hs[L] is the
    last layer's output AFTER the encoder's final LayerNorm."""
    eps = geo.layer_norm_eps
    # the arithmetic type follows the weights: fp32 state dicts give the reference's fp32 forward (the pinned path); a state dict
    # cast to float64 gives the same formulas in double -- used by tests/depth_envelope.py to measure how far the fp32 reference
    # itself sits from exact arithmetic at full depth (the conditioning of a stress case), never as the parity target
    feats = conv_feature_encoder(geo, sd, input_values.to(sd["feature_projection.projection.weight"].dtype))
    h = feature_projection(geo, sd, feats)
    h = h + positional_conv(geo, sd, h)
    T = h.shape[0]
    table = relative_bias_table(geo, sd, T) if geo.family == "wavlm" else None
    states: List[Tensor] = []
    for i in range(geo.num_layers):
        states.append(h)
        p = f"encoder.layers.{i}"
        x_ln = _ln(h, sd, p + ".layer_norm", eps)
        if geo.family == "wavlm":
            a = wavlm_attention(geo, sd, p + ".attention", x_ln, table)
        else:
            a = plain_attention(geo, sd, p + ".attention", x_ln)
        h = h + a
        h = h + feed_forward(sd, p + ".feed_forward.intermediate_dense", p + ".feed_forward.output_dense",
                             _ln(h, sd, p + ".final_layer_norm", eps))
    states.append(_ln(h, sd, "encoder.layer_norm", eps))
    return states


# --------------------------------------------------------------------------- a16
def _hz_to_mel_slaney(f: np.ndarray) -> np.ndarray:
    f = np.asarray(f, dtype=np.float64)
    mel = 3.0 * f / 200.0
    log_region = f >= 1000.0
    mel = np.where(log_region, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * (27.0 / np.log(6.4)), mel)
    return mel


def _mel_to_hz_slaney(m: np.ndarray) -> np.ndarray:
    m = np.asarray(m, dtype=np.float64)
    f = 200.0 * m / 3.0
    log_region = m >= 15.0
    return np.where(log_region, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), f)


def whisper_mel_filters(n_mels: int = 128, n_fft: int = 400, sr: int = 16000) -> np.ndarray:
    """[201, n_mels] triangular filters, slaney scale + slaney (area) norm, 0-8 kHz,
    built in float64 and cast to fp32 (HF feature_extraction_whisper.py:95-103 ->
    audio_utils.mel_filter_bank)."""
    n_bins = 1 + n_fft // 2
    fft_freqs = np.linspace(0, sr // 2, n_bins)
    mel_pts = np.linspace(_hz_to_mel_slaney(0.0), _hz_to_mel_slaney(8000.0), n_mels + 2)
    hz_pts = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(hz_pts)
    slopes = hz_pts[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / fdiff[:-1]
    up = slopes[:, 2:] / fdiff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    fb *= (2.0 / (hz_pts[2:n_mels + 2] - hz_pts[:n_mels]))[None, :]
    return fb.astype(np.float32)


def whisper_log_mel(wave: np.ndarray, n_mels: int = 128) -> np.ndarray:
    """[L] fp32 -> [n_mels, 3000]: zero-pad / truncate to 480000 samples, reflect
    pad 200, frame 400 hop 160, periodic Hann, |rfft|^2, drop the last frame,
    mel^T @, log10(max(.,1e-10)), max(., max-8), (.+4)/4
    (HF feature_extraction_whisper.py:135-169, 300-307)."""
    n_samples = 480000
    x = np.zeros(n_samples, dtype=np.float32)
    w = np.asarray(wave, dtype=np.float32)[:n_samples]
    x[: w.shape[0]] = w
    xt = torch.from_numpy(x)
    padded = F.pad(xt[None, None], (200, 200), mode="reflect")[0, 0]
    frames = padded.unfold(0, 400, 160)                                     # [3001, 400]
    window = torch.hann_window(400, periodic=True, dtype=torch.float32)
    spec = torch.fft.rfft(frames * window, n=400, dim=-1)                   # [3001, 201]
    power = (spec.real ** 2 + spec.imag ** 2)[:-1].transpose(0, 1)          # [201, 3000]
    mel = torch.from_numpy(whisper_mel_filters(n_mels)).transpose(0, 1) @ power
    log_spec = torch.clamp(mel, min=1e-10).log10()
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    return ((log_spec + 4.0) / 4.0).numpy()


# ----------------------------------------------------------------------- a17/a18
def whisper_hidden_states(geo, sd: StateDict, input_features: Tensor) -> List[Tensor]:
    """[n_mels, 3000] -> list of L+1 states [1500, D]: gelu(conv1), gelu(conv2,
    stride 2), + embed_positions, L pre-LN layers, final LayerNorm replaces the
    last entry (HF modeling_whisper.py:592-647, 379-413)."""
    eps = geo.layer_norm_eps
    x = input_features.float()[None]
    x = F.gelu(F.conv1d(x, sd["encoder.conv1.weight"], sd["encoder.conv1.bias"], padding=1))
    x = F.gelu(F.conv1d(x, sd["encoder.conv2.weight"], sd["encoder.conv2.bias"], stride=2, padding=1))
    h = x[0].transpose(0, 1) + sd["encoder.embed_positions.weight"]
    states: List[Tensor] = []
    for i in range(geo.num_layers):
        states.append(h)
        p = f"encoder.layers.{i}"
        a = plain_attention(geo, sd, p + ".self_attn", _ln(h, sd, p + ".self_attn_layer_norm", eps), k_bias=False)
        h = h + a
        h = h + feed_forward(sd, p + ".fc1", p + ".fc2", _ln(h, sd, p + ".final_layer_norm", eps))
    states.append(_ln(h, sd, "encoder.layer_norm", eps))
    return states


# --------------------------------------------------------- next row 8f-1: RoBERTa text
def roberta_position_ids(input_ids: Tensor, pad_token_id: int) -> Tensor:
    """cumsum over non-pad tokens, offset by padding_idx; pads keep padding_idx
    (HF modeling_roberta.py:142-155)."""
    mask = (input_ids != pad_token_id).to(torch.long)
    return torch.cumsum(mask, dim=-1) * mask + pad_token_id


def roberta_hidden_states(geo, sd: StateDict, input_ids: Tensor, attention_mask: Tensor) -> List[Tensor]:
    """[T] token ids + [T] 0/1 mask -> L+1 states [T, D]: embeddings (word + position + token-type -> LayerNorm)
    then L post-LayerNorm BERT layers; padded KEYS are masked, padded QUERY rows are still computed and saved
    (preprocessing/preprocess_roberta.py:47-69 keeps all max_len rows; HF modeling_roberta.py:56-120, 158-420)."""
    eps = geo.layer_norm_eps
    H, dh = geo.heads, geo.head_dim
    pos = roberta_position_ids(input_ids, geo.pad_token_id)
    h = (sd["embeddings.word_embeddings.weight"][input_ids] + sd["embeddings.position_embeddings.weight"][pos]
         + sd["embeddings.token_type_embeddings.weight"][0])
    h = _ln(h, sd, "embeddings.LayerNorm", eps)
    T = h.shape[0]
    neg = torch.zeros(T)
    neg[attention_mask == 0] = float("-inf")
    states = [h]
    for i in range(geo.num_layers):
        p = f"encoder.layer.{i}"
        a = p + ".attention.self"
        q = _heads(F.linear(h, sd[a + ".query.weight"], sd[a + ".query.bias"]), H)
        k = _heads(F.linear(h, sd[a + ".key.weight"], sd[a + ".key.bias"]), H)
        v = _heads(F.linear(h, sd[a + ".value.weight"], sd[a + ".value.bias"]), H)
        scores = torch.matmul(q, k.transpose(1, 2)) * (dh ** -0.5) + neg[None, None, :]
        ctx = torch.matmul(torch.softmax(scores, dim=-1), v).permute(1, 0, 2).reshape(T, H * dh)
        h = _ln(h + F.linear(ctx, sd[p + ".attention.output.dense.weight"], sd[p + ".attention.output.dense.bias"]),
                sd, p + ".attention.output.LayerNorm", eps)
        f = F.linear(F.gelu(F.linear(h, sd[p + ".intermediate.dense.weight"], sd[p + ".intermediate.dense.bias"])),
                     sd[p + ".output.dense.weight"], sd[p + ".output.dense.bias"])
        h = _ln(h + f, sd, p + ".output.LayerNorm", eps)
        states.append(h)
    return states


# ------------------------------------------------- DeBERTa-v2/v3 variant of next row 8f-1 (oracle + fixtures only)
def deberta_log_bucket(rel: Tensor, bucket_size: int, max_position: int) -> Tensor:
    """Signed relative distance -> bucket (HF modeling_deberta_v2.py make_log_bucket_position): distances inside
    +-bucket_size/2 keep their value, larger ones are spaced logarithmically up to max_position."""
    mid = bucket_size // 2
    sign = torch.sign(rel)
    inside = (rel < mid) & (rel > -mid)
    a = torch.where(inside, torch.full_like(rel, mid - 1), rel.abs()).to(torch.float32)
    logp = torch.ceil(torch.log(a / mid) / math.log((max_position - 1) / mid) * (mid - 1)) + mid
    return torch.where(a <= mid, rel.to(torch.float32), logp * sign).to(torch.long)


def deberta_hidden_states(geo, sd: StateDict, input_ids: Tensor, attention_mask: Tensor) -> List[Tensor]:
    """[T] ids + [T] 0/1 mask -> L+1 states [T, D] of DebertaV2Model in its v3 configuration (what AutoModel builds for
    preprocessing/preprocess_deroberta.py:106-107): no absolute positions, no token types; embeddings = LayerNorm(word) * mask;
    disentangled attention = (content.content + content->position + position->content) / sqrt(3 dh) with the q/k
    projections shared between content and the LayerNorm-ed relative embeddings (share_att_key), relative distances
    log-bucketed; a (query, key) pair is attendable only if BOTH are real tokens, everything else is filled with the
    dtype minimum (so a padded query row is a uniform average over all T keys); post-LayerNorm BERT blocks.
    HF modeling_deberta_v2.py: DebertaV2Embeddings, DebertaV2Encoder.get_attention_mask / get_rel_embedding,
    DisentangledSelfAttention.forward / disentangled_attention_bias."""
    eps = geo.layer_norm_eps
    H, dh, T = geo.heads, geo.head_dim, input_ids.shape[0]
    span = geo.position_buckets
    m = attention_mask.to(torch.float32)
    h = _ln(sd["embeddings.word_embeddings.weight"][input_ids], sd, "embeddings.LayerNorm", eps) * m[:, None]
    rel_emb = _ln(sd["encoder.rel_embeddings.weight"][: 2 * span], sd, "encoder.LayerNorm", eps)
    idx = torch.arange(T)
    bucket = deberta_log_bucket(idx[:, None] - idx[None, :], span, geo.max_positions)      # [q, k], q - k
    c2p_idx = torch.clamp(bucket + span, 0, 2 * span - 1)                                   # [q, k]
    p2c_idx = torch.clamp(-bucket + span, 0, 2 * span - 1)                                  # indexed [k(row), q(col)] below
    allowed = (attention_mask[:, None] * attention_mask[None, :]).bool()
    scale = math.sqrt(dh * 3.0)
    states = [h]
    for i in range(geo.num_layers):
        p = f"encoder.layer.{i}"
        a = p + ".attention.self"
        wq, bq, wk, bk = sd[a + ".query_proj.weight"], sd[a + ".query_proj.bias"], sd[a + ".key_proj.weight"], sd[a + ".key_proj.bias"]
        q = _heads(F.linear(h, wq, bq), H)
        k = _heads(F.linear(h, wk, bk), H)
        v = _heads(F.linear(h, sd[a + ".value_proj.weight"], sd[a + ".value_proj.bias"]), H)
        pos_q = _heads(F.linear(rel_emb, wq, bq), H)                  # [H, 2 span, dh]
        pos_k = _heads(F.linear(rel_emb, wk, bk), H)
        scores = torch.matmul(q, k.transpose(1, 2) / scale)
        c2p = torch.matmul(q, pos_k.transpose(1, 2))                  # [H, T, 2 span]
        scores = scores + torch.gather(c2p, 2, c2p_idx[None].expand(H, T, T)) / scale
        p2c = torch.matmul(k, pos_q.transpose(1, 2))                  # [H, T(key), 2 span]
        scores = scores + torch.gather(p2c, 2, p2c_idx[None].expand(H, T, T)).transpose(1, 2) / scale
        scores = scores.masked_fill(~allowed[None], torch.finfo(torch.float32).min)
        ctx = torch.matmul(torch.softmax(scores, dim=-1), v).permute(1, 0, 2).reshape(T, H * dh)
        h = _ln(h + F.linear(ctx, sd[p + ".attention.output.dense.weight"], sd[p + ".attention.output.dense.bias"]),
                sd, p + ".attention.output.LayerNorm", eps)
        f = F.linear(F.gelu(F.linear(h, sd[p + ".intermediate.dense.weight"], sd[p + ".intermediate.dense.bias"])),
                     sd[p + ".output.dense.weight"], sd[p + ".output.dense.bias"])
        h = _ln(h + f, sd, p + ".output.LayerNorm", eps)
        if i == 0 and getattr(geo, "text_conv_kernel", 0):
            # DebertaV2Encoder: after layer 0, ConvLayer(embeddings, layer-0 output, mask) (HF modeling_deberta_v2.py ConvLayer;
            # deberta-v2-xlarge / xxlarge: conv_kernel_size 3, conv_act "gelu" -- the checkpoint the reference's README names for
            # preprocess_deroberta.py, README.md:66): Conv1d over the token axis of the EMBEDDING output, padded rows zeroed,
            # activation, + residual, LayerNorm, padded rows zeroed again.
            kk = geo.text_conv_kernel
            c = F.conv1d(states[0].transpose(0, 1)[None], sd["encoder.conv.conv.weight"], sd["encoder.conv.conv.bias"],
                         padding=(kk - 1) // 2)[0].transpose(0, 1)
            c = F.gelu(c * m[:, None])
            h = _ln(h + c, sd, "encoder.conv.LayerNorm", eps) * m[:, None]
        states.append(h)
    return states


# ----------------------------------------------------------------------- a19/a20
def select_state(states: Sequence[Tensor], layer_index: int, use_average: bool) -> Tensor:
    """``--use_average y`` -> mean of the last four states, else states[index]
    (preprocess_speech.py:52-67; preprocess_whisper.py:55-73)."""
    if use_average:
        return torch.stack(list(states[-4:])).mean(dim=0)
    return states[layer_index]


def whisper_crop_rows(num_samples: int, feat_dim: int) -> int:
    """The reference crops to min(ceil(len/320), feats.shape[1]); feats is [1500, D]
    after squeeze so the cap is the hidden size (preprocess_whisper.py:49-50,75-76)."""
    return min(int(math.ceil(num_samples / 320)), int(feat_dim))


# -------------------------------------------------------- whole-path conveniences
def extract_speech(geo, sd: StateDict, wave: np.ndarray, layer_index: int = 0, use_average: bool = False) -> Tensor:
    x = torch.from_numpy(zero_mean_unit_var(wave))
    return select_state(speech_hidden_states(geo, sd, x), layer_index, use_average)


def extract_whisper(geo, sd: StateDict, wave: np.ndarray, layer_index: int = -1, use_average: bool = False) -> Tensor:
    feats = torch.from_numpy(whisper_log_mel(wave, geo.n_mels))
    st = select_state(whisper_hidden_states(geo, sd, feats), layer_index, use_average)
    return st[: whisper_crop_rows(len(wave), st.shape[1])]
