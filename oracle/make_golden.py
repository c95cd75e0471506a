"""Generate tests/golden/*.npz -- run ONCE in the build container, outputs committed.

TEST INFRASTRUCTURE ONLY.  The reference's arithmetic lives in the third-party
HuggingFace ``transformers`` package (pinned 4.47.1 in the reference's
benchmark/requirements.txt:35; 5.15.0 is what this image ships).  This script
instantiates the exact classes the reference's ``AutoModel`` /
``AutoFeatureExtractor`` / ``AutoProcessor`` calls resolve to
(preprocess_speech.py:111-114, preprocess_whisper.py:119-122), feeds them seeded
synthetic weights and waveforms, drives them the way the reference does (batch of
one, ``output_hidden_states=True``) and stores inputs' seeds + expected outputs.
It also asserts that oracle/ssl_oracle.py agrees with those classes, which is
what pins the oracle.

    python oracle/make_golden.py            # writes tests/golden/

The GPU box never runs this file and never needs ``transformers``.
"""
from __future__ import annotations

import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from interspeech_ser_amd import config as C                     # noqa: E402
from interspeech_ser_amd.weights import apply_stress, synthetic_state_dict, state_dict_digest  # noqa: E402
from oracle import ssl_oracle as O                               # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def synth_wave(seed: int, n: int) -> np.ndarray:
    """0.1*N(0,1) + 220 Hz sine at 0.2 (SURVEY 8d config 0 recipe), fp32 in [-1, 1]."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / 16000.0
    x = 0.1 * rng.standard_normal(n) + 0.2 * np.sin(2 * np.pi * 220.0 * t)
    return np.clip(x, -1.0, 1.0).astype(np.float32)


def hf_speech_model(geo):
    import transformers as tf
    common = dict(hidden_size=geo.hidden, num_hidden_layers=geo.num_layers, num_attention_heads=geo.heads,
                  intermediate_size=geo.ffn, conv_dim=list(geo.conv_dim), conv_kernel=list(geo.conv_kernel),
                  conv_stride=list(geo.conv_stride), conv_bias=geo.conv_bias, feat_extract_norm="layer",
                  do_stable_layer_norm=True, num_conv_pos_embeddings=geo.pos_conv_kernel,
                  num_conv_pos_embedding_groups=geo.pos_conv_groups, layer_norm_eps=geo.layer_norm_eps,
                  hidden_act="gelu", feat_extract_activation="gelu", vocab_size=32)
    if geo.family == C.FAMILY_WAVLM:
        cfg = tf.WavLMConfig(num_buckets=geo.num_buckets, max_bucket_distance=geo.max_bucket_distance, **common)
        m = tf.WavLMModel(cfg)
    elif geo.family == C.FAMILY_WAV2VEC2:
        m = tf.Wav2Vec2Model(tf.Wav2Vec2Config(**common))
    else:
        m = tf.HubertModel(tf.HubertConfig(feat_proj_layer_norm=geo.feat_proj_layer_norm, **common))
    return m.eval()


def hf_whisper_model(geo):
    import transformers as tf
    cfg = tf.WhisperConfig(num_mel_bins=geo.n_mels, d_model=geo.hidden, encoder_layers=geo.num_layers,
                           encoder_attention_heads=geo.heads, encoder_ffn_dim=geo.ffn,
                           decoder_layers=1, decoder_attention_heads=geo.heads, decoder_ffn_dim=geo.ffn,
                           max_source_positions=geo.max_source_positions, vocab_size=64,
                           pad_token_id=0, bos_token_id=1, eos_token_id=2, decoder_start_token_id=1,
                           activation_function="gelu")
    return tf.WhisperModel(cfg).eval()


def load_into(model, sd, allow_missing_prefixes=()):
    res = model.load_state_dict(sd, strict=False)
    missing = [k for k in res.missing_keys
               if not k.startswith(allow_missing_prefixes) and "masked_spec_embed" not in k]
    assert not missing, f"missing keys: {missing[:8]}"
    assert not res.unexpected_keys, f"unexpected keys: {res.unexpected_keys[:8]}"


def speech_case(tag, geo, seed, lengths, stress=None):
    """``stress``: weights.apply_stress kind -- fixtures whose residual stream carries 1000x outlier channels or rows
    with |mean| >> std, the two things real checkpoints do and seeded Gaussian weights do not."""
    import transformers as tf
    sd = synthetic_state_dict(geo, seed)
    if stress:
        sd = apply_stress(sd, geo, stress)
    model = hf_speech_model(geo)
    load_into(model, sd)
    fe = tf.Wav2Vec2FeatureExtractor(feature_size=1, sampling_rate=16000, padding_value=0.0,
                                     do_normalize=True, return_attention_mask=True)
    rec = {"seed": seed, "digest": state_dict_digest(sd), "lengths": np.array(lengths, dtype=np.int64)}
    if stress:
        rec["stress"] = np.array(stress)
    worst = 0.0
    for j, n in enumerate(lengths):
        wave = synth_wave(1000 + 17 * j + seed, n)
        inputs = fe(wave, sampling_rate=16000, return_tensors="pt", padding=True)   # preprocess_speech.py:48
        with torch.no_grad():
            hs = model(**inputs, output_hidden_states=True).hidden_states            # preprocess_speech.py:50,66
        hs = [h.squeeze(0) for h in hs]
        ours_in = O.zero_mean_unit_var(wave)
        assert np.abs(ours_in - inputs["input_values"][0].numpy()).max() == 0.0
        ours = O.speech_hidden_states(geo, sd, torch.from_numpy(ours_in))
        assert len(ours) == len(hs) == geo.num_layers + 1
        assert ours[0].shape[0] == geo.frames_for(n) == hs[0].shape[0]
        for a, b in zip(ours, hs):
            worst = max(worst, float((a - b).abs().max() / max(1.0, float(b.abs().max()))))
        rec[f"wave_seed_{j}"] = np.array(1000 + 17 * j + seed)
        rec[f"states_{j}"] = torch.stack(hs).numpy().astype(np.float32)              # [L+1, T, D]
        rec[f"input_values_{j}"] = ours_in if n <= 20000 else ours_in[:64]
        if stress:
            st = torch.stack(hs)
            rec[f"row_mean_over_std_{j}"] = (st.mean(-1).abs() / st.std(-1)).amax(dim=1).numpy().astype(np.float32)
            rec[f"absmax_over_median_{j}"] = (st.abs().amax(dim=(1, 2)) / st.abs().flatten(1).median(dim=1).values).numpy().astype(np.float32)
    print(f"{tag}: oracle vs HF rel-max err {worst:.2e}")
    if stress:
        print(f"   residual stream: max |row mean|/std per state {rec['row_mean_over_std_0'].round(1).tolist()}, "
              f"max|x| / median|x| per state {rec['absmax_over_median_0'].round(0).tolist()}")
    assert worst < 2e-5, worst
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **rec)


def whisper_case(tag, geo, seed, lengths):
    import transformers as tf
    sd = synthetic_state_dict(geo, seed)
    model = hf_whisper_model(geo)
    load_into(model, sd, allow_missing_prefixes=("decoder.",))
    fe = tf.WhisperFeatureExtractor(feature_size=geo.n_mels)
    rec = {"seed": seed, "digest": state_dict_digest(sd), "lengths": np.array(lengths, dtype=np.int64)}
    worst = 0.0
    worst_mel = 0.0
    for j, n in enumerate(lengths):
        wave = synth_wave(2000 + 13 * j + seed, n)
        feats = fe(wave, sampling_rate=16000, return_tensors="pt")["input_features"]   # preprocess_whisper.py:48
        with torch.no_grad():
            hs = model.encoder(feats, output_hidden_states=True).hidden_states           # preprocess_whisper.py:57,71
        hs = [h.squeeze(0) for h in hs]
        mel = O.whisper_log_mel(wave, geo.n_mels)
        worst_mel = max(worst_mel, float(np.abs(mel - feats[0].numpy()).max()))
        ours = O.whisper_hidden_states(geo, sd, torch.from_numpy(mel))
        for a, b in zip(ours, hs):
            worst = max(worst, float((a - b).abs().max() / max(1.0, float(b.abs().max()))))
        rows = O.whisper_crop_rows(n, geo.hidden)
        rec[f"wave_seed_{j}"] = np.array(2000 + 13 * j + seed)
        rec[f"rows_{j}"] = np.array(rows)
        # keep the fixture small: every state, but only the rows the reference would save
        rec[f"states_{j}"] = torch.stack(hs)[:, :rows].numpy().astype(np.float32)
        rec[f"mel_probe_{j}"] = feats[0].numpy()[:, ::50].astype(np.float32)            # [n_mels, 60]
    print(f"{tag}: oracle vs HF rel-max err {worst:.2e}, log-mel abs err {worst_mel:.2e}")
    assert worst < 2e-5 and worst_mel < 1e-5, (worst, worst_mel)
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **rec)


def roberta_case(tag, geo, seed):
    import transformers as tf
    sd = synthetic_state_dict(geo, seed)
    cfg = tf.RobertaConfig(vocab_size=geo.vocab_size, hidden_size=geo.hidden, num_hidden_layers=geo.num_layers,
                           num_attention_heads=geo.heads, intermediate_size=geo.ffn, hidden_act="gelu",
                           max_position_embeddings=geo.max_positions, type_vocab_size=geo.type_vocab_size,
                           layer_norm_eps=geo.layer_norm_eps, pad_token_id=geo.pad_token_id, bos_token_id=0, eos_token_id=2)
    model = tf.RobertaModel(cfg, add_pooling_layer=False).eval()
    load_into(model, sd)
    max_len = 80
    rng = np.random.default_rng(seed)
    rec = {"seed": seed, "digest": state_dict_digest(sd), "max_len": np.array(max_len)}
    worst = 0.0
    for j, n in enumerate((80, 37, 5)):                       # full, padded, almost empty (tokenizer padding="max_length")
        ids = np.full(max_len, geo.pad_token_id, dtype=np.int64)
        ids[:n] = rng.integers(3, geo.vocab_size, n)
        ids[0], ids[n - 1] = 0, 2
        mask = (np.arange(max_len) < n).astype(np.int64)
        with torch.no_grad():
            out = model(input_ids=torch.from_numpy(ids)[None], attention_mask=torch.from_numpy(mask)[None],
                        output_hidden_states=True)             # preprocess_roberta.py:57,68
        hs = [h.squeeze(0) for h in out.hidden_states]
        assert torch.equal(hs[-1], out.last_hidden_state.squeeze(0))
        ours = O.roberta_hidden_states(geo, sd, torch.from_numpy(ids), torch.from_numpy(mask))
        for a, b in zip(ours, hs):
            worst = max(worst, float((a - b).abs().max() / max(1.0, float(b.abs().max()))))
        rec[f"ids_{j}"], rec[f"mask_{j}"] = ids, mask
        rec[f"states_{j}"] = torch.stack(hs).numpy().astype(np.float32)
    print(f"{tag}: oracle vs HF rel-max err {worst:.2e}")
    assert worst < 2e-5, worst
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **rec)


def deberta_case(tag, geo, seed):
    """(with geo.text_conv_kernel: the deberta-v2-xlarge variant, ConvLayer after encoder layer 0)"""
    """DeBERTa-v3-style fixture (HIP counterpart: engine.DebertaEncoder, tests/test_gpu_e2e.py::test_deberta_golden)."""
    import transformers as tf
    sd = synthetic_state_dict(geo, seed)
    cfg = tf.DebertaV2Config(vocab_size=geo.vocab_size, hidden_size=geo.hidden, num_hidden_layers=geo.num_layers,
                             num_attention_heads=geo.heads, intermediate_size=geo.ffn, hidden_act="gelu",
                             max_position_embeddings=geo.max_positions, type_vocab_size=0, layer_norm_eps=geo.layer_norm_eps,
                             pad_token_id=geo.pad_token_id, relative_attention=True, position_buckets=geo.position_buckets,
                             norm_rel_ebd="layer_norm", share_att_key=True, pos_att_type=["p2c", "c2p"],
                             position_biased_input=False, max_relative_positions=-1,
                             hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                             **({"conv_kernel_size": geo.text_conv_kernel, "conv_act": "gelu"} if geo.text_conv_kernel else {}))
    model = tf.DebertaV2Model(cfg).eval()
    load_into(model, sd)
    max_len = 80
    rng = np.random.default_rng(seed)
    rec = {"seed": seed, "digest": state_dict_digest(sd), "max_len": np.array(max_len)}
    worst = 0.0
    for j, n in enumerate((80, 37, 5)):
        ids = np.full(max_len, geo.pad_token_id, dtype=np.int64)
        ids[:n] = rng.integers(3, geo.vocab_size, n)
        ids[0], ids[n - 1] = 1, 2
        mask = (np.arange(max_len) < n).astype(np.int64)
        with torch.no_grad():
            out = model(input_ids=torch.from_numpy(ids)[None], attention_mask=torch.from_numpy(mask)[None],
                        output_hidden_states=True)             # preprocess_deroberta.py:57,68 (same calls as roberta)
        hs = [h.squeeze(0) for h in out.hidden_states]
        assert torch.equal(hs[-1], out.last_hidden_state.squeeze(0))
        ours = O.deberta_hidden_states(geo, sd, torch.from_numpy(ids), torch.from_numpy(mask))
        for a, b in zip(ours, hs):
            worst = max(worst, float((a - b).abs().max() / max(1.0, float(b.abs().max()))))
        rec[f"ids_{j}"], rec[f"mask_{j}"] = ids, mask
        rec[f"states_{j}"] = torch.stack(hs).numpy().astype(np.float32)
    print(f"{tag}: oracle vs HF rel-max err {worst:.2e}")
    assert worst < 2e-5, worst
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **rec)


def integer_tables():
    import transformers as tf
    geo = C.WAVLM_LARGE
    Ls = np.array([400, 401, 719, 720, 16000, 23457, 48000, 160000, 480000], dtype=np.int64)
    model = hf_speech_model(C.tiny_geometry(C.FAMILY_WAVLM))
    Ts = np.array([int(model._get_feat_extract_output_lengths(int(n))) for n in Ls], dtype=np.int64)
    assert all(geo.frames_for(int(n)) == int(t) for n, t in zip(Ls, Ts))
    assert all((int(n) - 400) // 320 + 1 == int(t) for n, t in zip(Ls, Ts))
    att = model.encoder.layers[0].attention
    rel = torch.arange(-1500, 1501, dtype=torch.long)
    buckets = att._relative_positions_bucket(rel[None, :])[0].numpy().astype(np.int32)
    assert np.array_equal(buckets, O.relative_buckets(rel).numpy())
    wl = np.array([1, 319, 320, 321, 16000, 160000, 409600, 409601, 480000, 500000], dtype=np.int64)
    rows = np.array([min(int(np.ceil(n / 320)), 1280) for n in wl], dtype=np.int64)   # preprocess_whisper.py:49-50,75
    assert all(O.whisper_crop_rows(int(n), 1280) == int(r) for n, r in zip(wl, rows))
    fe = tf.WhisperFeatureExtractor(feature_size=128)
    mel = np.asarray(fe.mel_filters, dtype=np.float32)
    assert np.abs(mel - O.whisper_mel_filters(128)).max() < 1e-7, np.abs(mel - O.whisper_mel_filters(128)).max()
    np.savez_compressed(os.path.join(OUT, "integer_tables.npz"), sample_counts=Ls, frame_counts=Ts,
                        rel=rel.numpy().astype(np.int32), buckets=buckets, whisper_len=wl, whisper_rows=rows,
                        mel_filters=mel)
    print("integer tables + mel filters ok")


def full_size_pins():
    """WavLM-large geometry, seed 0, one 3 s clip: per-state scalars + 32 probes
     -- too big to commit as tensors."""
    geo = C.WAVLM_LARGE
    sd = synthetic_state_dict(geo, 0)
    model = hf_speech_model(geo)
    load_into(model, sd)
    wave = synth_wave(7, 48000)
    x = torch.from_numpy(O.zero_mean_unit_var(wave))
    with torch.no_grad():
        hs = [h.squeeze(0) for h in model(input_values=x[None], output_hidden_states=True).hidden_states]
    ours = O.speech_hidden_states(geo, sd, x)
    worst = max(float((a - b).abs().max() / max(1.0, float(b.abs().max()))) for a, b in zip(ours, hs))
    print(f"wavlm-large full size: oracle vs HF rel-max err {worst:.2e}")
    assert worst < 5e-5
    T, D = hs[0].shape
    g = np.random.default_rng(5)
    pr, pc = g.integers(0, T, 32), g.integers(0, D, 32)
    np.savez_compressed(
        os.path.join(OUT, "wavlm_large_pins.npz"), digest=state_dict_digest(sd), wave_seed=np.array(7),
        num_samples=np.array(48000), shape=np.array([len(hs), T, D]), probe_rows=pr, probe_cols=pc,
        probes=np.stack([h.numpy()[pr, pc] for h in hs]).astype(np.float32),
        absmax=np.array([float(h.abs().max()) for h in hs], dtype=np.float32),
        mean=np.array([float(h.mean()) for h in hs], dtype=np.float32),
        l2=np.array([float(h.norm()) for h in hs], dtype=np.float32))


def stress_cases():
    speech_case("tiny_wavlm_outlier", C.TINY_WAVLM, 21, [16000, 9000], stress="outliers")
    speech_case("tiny_hubert_outlier", C.TINY_HUBERT, 22, [16000, 9000], stress="outliers")
    speech_case("tiny_wavlm_rowmean", C.TINY_WAVLM, 23, [16000, 9000], stress="rowmean")
    speech_case("tiny_hubert_rowmean", C.TINY_HUBERT, 24, [16000, 9000], stress="rowmean")
    sharp_cases()


def sharp_cases():
    """Sharp-attention fixtures (weights.apply_stress "sharp"): the envelope of the single-product numerics modes is set by the
    attention logits, so the pin comes from the HF classes, not from an oracle-vs-oracle comparison."""
    speech_case("tiny_wavlm_sharp", C.TINY_WAVLM, 25, [16000, 9000], stress="sharp")
    speech_case("tiny_hubert_sharp", C.TINY_HUBERT, 26, [16000, 9000], stress="sharp")
    speech_case("tiny_wav2vec2_sharp", C.TINY_WAV2VEC2, 27, [12000], stress="sharp")


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "deberta":
        deberta_case("tiny_deberta_d128h2", C.TINY_DEBERTA, 17)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "deberta_conv":
        deberta_case("tiny_deberta_conv_d128h2", C.TINY_DEBERTA_CONV, 19)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sharp":         # sharp-attention fixtures only
        sharp_cases()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "stress":        # outlier-stress fixtures only
        stress_cases()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "roberta":       # add the text fixture without touching the others
        roberta_case("tiny_roberta_d128h2", C.TINY_ROBERTA, 15)
        return
    integer_tables()
    ragged = [16000, 23457]
    speech_case("tiny_wavlm_d128h2", C.TINY_WAVLM, 11, ragged)
    speech_case("tiny_wav2vec2_d960h8", C.TINY_WAV2VEC2, 12, ragged)
    speech_case("tiny_hubert_d320h4", C.TINY_HUBERT, 13, ragged)
    whisper_case("tiny_whisper_d128h2", C.TINY_WHISPER, 14, [16000, 100000])
    roberta_case("tiny_roberta_d128h2", C.TINY_ROBERTA, 15)
    deberta_case("tiny_deberta_d128h2", C.TINY_DEBERTA, 17)
    deberta_case("tiny_deberta_conv_d128h2", C.TINY_DEBERTA_CONV, 19)
    stress_cases()
    full_size_pins()


if __name__ == "__main__":
    main()
