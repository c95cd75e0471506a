"""CPU: host-side logic of the drivers (rows a1-a5, a8, a19-a21 of SURVEY 8a)."""
import os
import wave

import numpy as np
import pytest
import torch

from interspeech_ser_amd import config as C
from interspeech_ser_amd import driver, frontend
from interspeech_ser_amd.dist import shard_files


def write_wav(path, x, sr=16000, ch=1):
    pcm = (np.clip(x, -1, 1) * 32767).astype("<i2")
    with wave.open(str(path), "wb") as wf:
        wf.setnchannels(ch)
        wf.setsampwidth(2)
        wf.setframerate(sr)
        wf.writeframes(pcm.tobytes())
    return pcm


def test_cli_surface_matches_reference():
    """Flag names and defaults of preprocess_speech.py:13-22 / preprocess_whisper.py:14-23."""
    for whisper in (False, True):
        a = driver.build_parser(whisper).parse_args([])
        assert (a.seed, a.ssl_type, a.save_path, a.wav_dir, a.num_workers, a.n_layer, a.use_average) == \
               (7, "wavlm-large", "./", "./", 4, -1, "n")


def test_geometry_registry():
    assert C.geometry_for("microsoft/wavlm-large").hidden == 1024
    assert C.geometry_for("wavlm-large") is C.WAVLM_LARGE
    assert C.geometry_for("facebook/wav2vec2-xls-r-2b").head_dim == 120
    assert C.geometry_for("facebook/hubert-xlarge-ls960-ft").head_dim == 80
    assert C.geometry_for("openai/whisper-large-v3").num_hidden_states == 33
    with pytest.raises(OSError):
        C.geometry_for("no/such-model")


def test_geometry_from_config_json(tmp_path):
    """a3: the reference builds its model from the checkpoint's own config.json (AutoModel.from_pretrained,
    preprocess_speech.py:111-112), so any name works.  HF's config classes give the field names: a geometry written out
    through them and read back through geometry_from_config is the same geometry; unsupported variants raise OSError (the class
    the driver reports as "No pretrained model found"); a snapshot directory under an unknown name resolves."""
    tf = pytest.importorskip("transformers")
    import dataclasses
    import json

    def speech_cfg(geo):
        common = dict(hidden_size=geo.hidden, num_hidden_layers=geo.num_layers, num_attention_heads=geo.heads,
                      intermediate_size=geo.ffn, conv_dim=list(geo.conv_dim), conv_kernel=list(geo.conv_kernel),
                      conv_stride=list(geo.conv_stride), conv_bias=geo.conv_bias, feat_extract_norm="layer", do_stable_layer_norm=True,
                      num_conv_pos_embeddings=geo.pos_conv_kernel, num_conv_pos_embedding_groups=geo.pos_conv_groups,
                      layer_norm_eps=geo.layer_norm_eps)
        if geo.family == C.FAMILY_WAVLM:
            return tf.WavLMConfig(num_buckets=geo.num_buckets, max_bucket_distance=geo.max_bucket_distance, **common)
        if geo.family == C.FAMILY_WAV2VEC2:
            return tf.Wav2Vec2Config(**common)
        return tf.HubertConfig(feat_proj_layer_norm=True, **common)

    def same(a, b):
        return all(getattr(a, f.name) == getattr(b, f.name) for f in dataclasses.fields(a) if f.name != "name")

    for geo in (C.WAVLM_LARGE, C.XLSR_2B, C.HUBERT_XLARGE, C.TINY_WAVLM, C.TINY_WAV2VEC2, C.TINY_HUBERT):
        assert same(C.geometry_from_config(speech_cfg(geo).to_dict()), geo), geo.name
    g = C.WHISPER_LARGE_V3
    wcfg = tf.WhisperConfig(num_mel_bins=g.n_mels, d_model=g.hidden, encoder_layers=g.num_layers, encoder_attention_heads=g.heads,
                            encoder_ffn_dim=g.ffn, max_source_positions=g.max_source_positions)
    assert same(C.geometry_from_config(wcfg.to_dict()), g)
    g = C.ROBERTA_LARGE
    rcfg = tf.RobertaConfig(vocab_size=g.vocab_size, hidden_size=g.hidden, num_hidden_layers=g.num_layers, num_attention_heads=g.heads,
                            intermediate_size=g.ffn, max_position_embeddings=g.max_positions, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1)
    assert same(C.geometry_from_config(rcfg.to_dict()), g)
    for g in (C.DEBERTA_V3_LARGE, C.DEBERTA_V2_XLARGE):
        dcfg = tf.DebertaV2Config(vocab_size=g.vocab_size, hidden_size=g.hidden, num_hidden_layers=g.num_layers, num_attention_heads=g.heads,
                                  intermediate_size=g.ffn, max_position_embeddings=512, type_vocab_size=0, layer_norm_eps=1e-7, pad_token_id=0,
                                  relative_attention=True, position_buckets=256, norm_rel_ebd="layer_norm", share_att_key=True,
                                  pos_att_type=["p2c", "c2p"], position_biased_input=False,
                                  **({"conv_kernel_size": 3, "conv_act": "gelu"} if g.text_conv_kernel else {}))
        assert same(C.geometry_from_config(dcfg.to_dict()), g), g.name
    # refused variants
    for bad in (dict(feat_extract_norm="group"), dict(do_stable_layer_norm=False)):
        d = speech_cfg(C.WAVLM_LARGE).to_dict()
        d.update(bad)
        with pytest.raises(OSError):
            C.geometry_from_config(d, name="some/fine-tune")
    with pytest.raises(OSError):
        C.geometry_from_config({"model_type": "bert"})
    # a local snapshot under a name the table does not know: directory as --ssl_type, or unknown name + --checkpoint
    snap = tmp_path / "my-wavlm-finetune"
    snap.mkdir()
    (snap / "config.json").write_text(json.dumps(speech_cfg(C.TINY_WAVLM).to_dict()))
    assert same(C.resolve_geometry(str(snap)), C.TINY_WAVLM)
    assert same(C.resolve_geometry("acme/wavlm-large-ser-v7", str(snap)), C.TINY_WAVLM)
    assert same(C.resolve_geometry("acme/wavlm-large-ser-v7", str(snap / "model.safetensors")), C.TINY_WAVLM)
    with pytest.raises(OSError):
        C.resolve_geometry("acme/wavlm-large-ser-v7")                     # no config.json, not in the table
    assert C.resolve_geometry("microsoft/wavlm-large") is C.WAVLM_LARGE      # table fallback


def test_text_driver_tokenizer_call_with_local_files(tmp_path):
    """The text drivers' default tokenizer path (driver.hf_tokenize_fn = the reference's
    ``RobertaTokenizer.from_pretrained(...)(text, padding="max_length", truncation=True, max_length=L)``,
    preprocess_roberta.py:45-54) with vocabulary files on disk and no network: [n, L] ids, right padding with <pad> = 1,
    truncation at L; a missing directory is the OSError the driver reports as "No pretrained model found"."""
    pytest.importorskip("transformers")
    from conftest import write_tiny_roberta_tokenizer
    d = write_tiny_roberta_tokenizer(str(tmp_path / "tok"))
    fn = driver.hf_tokenize_fn(d, 16)
    ids, mask = fn(["the cat and the hat", "hm", "x " * 40])
    assert tuple(ids.shape) == tuple(mask.shape) == (3, 16)
    assert int(ids[0, 0]) == 0 and int(ids[1, 3]) == 2 and ids[1, 4:].eq(1).all() and mask[1].tolist() == [1] * 4 + [0] * 12
    assert mask[2].all() and int(ids[2, -1]) == 2                        # truncated to 16 with </s> kept
    assert int(ids.max()) < 300                                          # fits the tiny fixture geometry's embedding table
    with pytest.raises(OSError):
        driver.hf_tokenize_fn(str(tmp_path / "nowhere"), 16)


def test_deberta_v2_xlarge_geometry_is_registered():
    """The checkpoint the reference's README names for preprocess_deroberta.py (README.md:66): 24 x 1536, 24 heads of 64,
    ConvLayer with kernel 3; 884.6 M parameters with HF's DebertaV2Model at these values."""
    g = C.geometry_for("microsoft/deberta-v2-xlarge")
    assert (g.family, g.num_layers, g.hidden, g.heads, g.head_dim, g.ffn, g.text_conv_kernel) == (C.FAMILY_DEBERTA, 24, 1536, 24, 64, 6144, 3)
    assert C.geometry_for("microsoft/deberta-v2-xxlarge").num_layers == 48
    assert C.geometry_for("microsoft/deberta-v3-large").text_conv_kernel == 0


def test_frame_arithmetic():
    g = C.WAVLM_LARGE
    assert g.frame_chain(160000) == [31999, 15999, 7999, 3999, 1999, 999, 499]
    assert g.frames_for(48000) == 149 and g.frames_for(400) == 1 and g.frames_for(399) == 0


def test_wav_decode_matches_soundfile_convention(tmp_path):
    rng = np.random.default_rng(0)
    x = 0.3 * rng.standard_normal(4000)
    pcm = write_wav(tmp_path / "a.wav", x)
    y = frontend.load_wav_16k(str(tmp_path / "a.wav"))
    assert y.dtype == np.float32 and np.array_equal(y, pcm.astype(np.float32) / 32768.0)
    st = np.stack([x, -0.5 * x], axis=1).reshape(-1)
    write_wav(tmp_path / "s.wav", st, ch=2)
    ys = frontend.load_wav_16k(str(tmp_path / "s.wav"))
    assert ys.shape == (4000,)
    write_wav(tmp_path / "r.wav", x, sr=8000)
    with pytest.raises(frontend.UnsupportedAudio):
        frontend.load_wav_16k(str(tmp_path / "r.wav"))


def test_feature_file_contract(tmp_path):
    """<save_path>/<basename>.pt, bare 2-D float32 CPU tensor loadable with torch.load
    (bin/train_cat_bimodal_lazy_1head.py:220-228 in the reference)."""
    p = frontend.feature_path(str(tmp_path), "/data/wavs/MSP-PODCAST_0001_0008.wav")
    assert p == os.path.join(str(tmp_path), "MSP-PODCAST_0001_0008.pt")
    feats = torch.arange(12, dtype=torch.float32).view(4, 3)
    frontend.save_feature(feats[:3], p)
    back = torch.load(p)
    assert back.dtype == torch.float32 and back.ndim == 2 and back.device.type == "cpu"
    assert torch.equal(back, feats[:3]) and back.untyped_storage().nbytes() == 3 * 3 * 4


def test_whisper_crop_and_mel(golden_dir):
    g = np.load(os.path.join(golden_dir, "integer_tables.npz"))
    for n, r in zip(g["whisper_len"], g["whisper_rows"]):
        assert frontend.whisper_saved_rows(int(n), 1280) == int(r)
    assert frontend.whisper_saved_rows(480000, 1280) == 1280            # the reference's cap is the hidden size
    assert np.abs(frontend.whisper_mel_filters(128) - g["mel_filters"]).max() < 1e-7


def test_layer_index_rules():
    assert driver.resolve_layer_index(-1, 25) == 24
    assert driver.resolve_layer_index(0, 25) == 0
    with pytest.raises(IndexError):
        driver.resolve_layer_index(25, 25)


def test_sharding_is_a_balanced_partition():
    files = [f"u{i:03d}.wav" for i in range(37)]
    sizes = [1000 + (i * 7919) % 5000 for i in range(37)]
    shards = [shard_files(files, sizes, r, 4) for r in range(4)]
    assert sorted(sum(shards, [])) == sorted(files)
    tot = [sum(sizes[files.index(f)] for f in s) for s in shards]
    assert max(tot) - min(tot) <= max(sizes)
    for s in shards:
        sz = [sizes[files.index(f)] for f in s]
        assert sz == sorted(sz, reverse=True)
    assert driver.make_batches(shards[0], 4)[0] == shards[0][:4]


def test_driver_without_gpu_logs_and_exits_zero(tmp_path, capsys):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    wav_dir, out = tmp_path / "w", tmp_path / "o"
    wav_dir.mkdir()
    write_wav(wav_dir / "a.wav", np.zeros(1600))
    rc = driver.run_speech(["--ssl_type", "microsoft/wavlm-large", "--wav_dir", str(wav_dir), "--save_path", str(out)])
    text = capsys.readouterr().out
    assert rc == 0
    assert "Using average = False" in text and "1 file are going to be processed..." in text
    assert f"Save path = {out} created. It has 0 files in it." in text
    assert "no CPU path" in text and not list(out.iterdir())


def test_synthetic_weights_have_hf_names_and_shapes():
    from interspeech_ser_amd.weights import normalize_names, synthetic_state_dict
    sd = synthetic_state_dict(C.TINY_WAVLM, 1)
    assert sd["encoder.layers.0.attention.rel_attn_embed.weight"].shape == (320, 2)
    assert "encoder.layers.1.attention.rel_attn_embed.weight" not in sd
    assert sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"].shape == (128, 64, 128)
    assert "feature_extractor.conv_layers.0.conv.bias" not in sd
    sd2 = synthetic_state_dict(C.TINY_HUBERT, 1)
    assert "feature_extractor.conv_layers.0.conv.bias" in sd2
    wrapped = {"wav2vec2." + k: v for k, v in sd2.items()}
    wrapped["lm_head.weight"] = torch.zeros(2, 2)
    assert sorted(normalize_names(wrapped)) == sorted(sd2)


def test_checkpoint_round_trip_with_hub_prefixes(tmp_path):
    """A hub-style checkpoint (task-head prefix, extra heads, safetensors) loads to the bare encoder names."""
    from safetensors.torch import save_file
    from interspeech_ser_amd.weights import load_checkpoint, state_dict_digest, synthetic_state_dict
    sd = synthetic_state_dict(C.TINY_WAVLM, 3)
    hub = {"wavlm." + k: v.clone() for k, v in sd.items()}
    hub["wavlm.masked_spec_embed"] = torch.zeros(128)
    hub["lm_head.weight"] = torch.zeros(4, 128)
    d = tmp_path / "snap"
    d.mkdir()
    save_file(hub, str(d / "model.safetensors"))
    back = load_checkpoint(str(d))
    assert sorted(back) == sorted(sd) and state_dict_digest(back) == state_dict_digest(sd)
    with pytest.raises(OSError):
        load_checkpoint(str(tmp_path / "nothing_here"))


def test_lora_adapters_are_merged_at_load(tmp_path):
    """Next row 8f-4: a PEFT-wrapped fine-tuned checkpoint (preprocess_speech_pretrained.py:108-177: r=8, alpha=16 on
    q_proj / v_proj, classifier head on top) loads as plain HF names with W + (alpha/r) B A -- the algebraic identity
    of an eval-mode LoRA Linear -- and a checkpoint without adapters is untouched."""
    import torch
    from interspeech_ser_amd import config as C
    from interspeech_ser_amd.weights import load_checkpoint, merge_lora, synthetic_state_dict
    geo = C.TINY_WAVLM
    base = synthetic_state_dict(geo, 9)
    g = torch.Generator().manual_seed(4)
    wrapped, want = {}, dict(base)
    for k, v in base.items():
        mod = k.rsplit(".", 1)[0]
        if mod.endswith(("q_proj", "v_proj")):
            wrapped[f"wavlm.base_model.model.{mod}.base_layer.{k.rsplit('.', 1)[1]}"] = v
            if k.endswith(".weight"):
                a, b = torch.randn(8, v.shape[1], generator=g), torch.randn(v.shape[0], 8, generator=g) * 0.1
                wrapped[f"wavlm.base_model.model.{mod}.lora_A.default.weight"] = a
                wrapped[f"wavlm.base_model.model.{mod}.lora_B.default.weight"] = b
                want[k] = (v.double() + 2.0 * (b.double() @ a.double())).float()
        else:
            wrapped["wavlm.base_model.model." + k] = v
    wrapped["classifier.0.weight"] = torch.randn(512, geo.hidden, generator=g)
    path = tmp_path / "whisper_lora_ser.pt"
    torch.save(wrapped, path)
    got = load_checkpoint(str(path))
    assert set(got) == set(want)
    for k in want:
        assert torch.equal(got[k], want[k]), k
    x = torch.randn(5, geo.hidden, generator=g)                      # the identity itself, on one adapted Linear
    k = "encoder.layers.0.attention.q_proj"
    a, b = wrapped[f"wavlm.base_model.model.{k}.lora_A.default.weight"], wrapped[f"wavlm.base_model.model.{k}.lora_B.default.weight"]
    lora_out = x @ base[k + ".weight"].T + (16 / 8) * (x @ a.T) @ b.T
    assert torch.allclose(x @ got[k + ".weight"].T, lora_out, atol=1e-4)
    assert merge_lora(base) is base
    half = load_checkpoint(str(path), lora_alpha=8.0)                # alpha is configuration, not stored in the file
    assert torch.allclose(half[k + ".weight"], (base[k + ".weight"] + want[k + ".weight"]) / 2, atol=1e-6)


def test_opt_in_resampler_keeps_pitch_and_length(tmp_path):
    """Next row 8f-3 (opt-in, parity unpinned): a 44.1 kHz file is refused by default and, with resample=True, comes out
    at 16 kHz with the right length and the same tone."""
    import wave
    import pytest
    from interspeech_ser_amd import frontend
    sr, secs, f0 = 44100, 1.0, 440.0
    t = np.arange(int(sr * secs)) / sr
    pcm = (0.5 * np.sin(2 * np.pi * f0 * t) * 32767).astype("<i2")
    with wave.open(str(tmp_path / "hi.wav"), "wb") as wf:
        wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(sr); wf.writeframes(pcm.tobytes())
    with pytest.raises(frontend.UnsupportedAudio):
        frontend.load_wav_16k(str(tmp_path / "hi.wav"))
    y = frontend.load_wav_16k(str(tmp_path / "hi.wav"), resample=True)
    assert y.dtype == np.float32 and abs(len(y) - 16000) <= 1
    spec = np.abs(np.fft.rfft(y * np.hanning(len(y))))
    assert abs(np.argmax(spec) * 16000.0 / len(y) - f0) < 2.0
    ref = 0.5 * np.sin(2 * np.pi * f0 * np.arange(len(y)) / 16000.0)
    assert float(np.abs(y[200:-200] - ref[200:-200]).max()) < 2e-3


def _python_decoder(path):
    """frontend.load_wav_16k with the native reader switched off: the pure-Python statement of the same decode."""
    saved = frontend._native_wav
    frontend._native_wav = lambda p, pinned=False: None
    try:
        return frontend.load_wav_16k(path)
    finally:
        frontend._native_wav = saved


@pytest.mark.parametrize("width,ch", [(1, 1), (2, 1), (2, 2), (3, 1), (4, 1), (2, 3), (4, 2)])
def test_native_wav_reader_equals_python_decoder(tmp_path, width, ch):
    """ser_wav_read_f32 (C, off the GIL) against the Python decoder, bit for bit: 8/16/24/32-bit PCM, mono and
    multi-channel (soundfile's integer / 2^(bits-1) scaling, numpy's float32 channel mean)."""
    import ctypes
    from interspeech_ser_amd._lib import lib
    rng = np.random.default_rng(width * 10 + ch)
    n = 3001
    raw = rng.integers(0, 256, size=n * ch * width, dtype=np.uint8).tobytes()
    p = str(tmp_path / "x.wav")
    with wave.open(p, "wb") as wf:
        wf.setnchannels(ch)
        wf.setsampwidth(width)
        wf.setframerate(16000)
        wf.writeframes(raw)
    sr, nch = ctypes.c_int32(), ctypes.c_int32()
    frames = lib.ser_wav_read_f32(os.fsencode(p), None, 0, ctypes.byref(sr), ctypes.byref(nch))
    assert (frames, sr.value, nch.value) == (n, 16000, ch)
    got = frontend.load_wav_16k(p)
    want = _python_decoder(p)
    assert got.dtype == np.float32 and got.shape == (n,) and np.array_equal(got, want)
    assert np.abs(got).max() <= 1.0


def test_native_wav_reader_float_files_and_errors(tmp_path):
    from scipy.io import wavfile
    from interspeech_ser_amd._lib import lib
    x = (0.5 * np.random.default_rng(1).standard_normal(2000)).astype(np.float32)
    p = str(tmp_path / "f.wav")
    wavfile.write(p, 16000, x)                                           # IEEE float WAVE (format tag 3)
    assert np.array_equal(frontend.load_wav_16k(p), x)
    junk = tmp_path / "junk.wav"
    junk.write_bytes(b"not a wave file at all")
    assert lib.ser_wav_read_f32(os.fsencode(str(junk)), None, 0, None, None) < 0
    assert b"RIFF" in lib.ser_last_error()
    assert lib.ser_wav_read_f32(os.fsencode(str(tmp_path / "missing.wav")), None, 0, None, None) < 0
    with pytest.raises(Exception):
        frontend.load_wav_16k(str(junk))                                 # the Python path gives the verdict, as before
    buf = np.empty(10, dtype=np.float32)                                 # too small a buffer is an error, not a truncation
    assert lib.ser_wav_read_f32(os.fsencode(p), buf.ctypes.data, 10, None, None) < 0


@pytest.mark.parametrize("rows,cols", [(0, 8), (1, 4), (3, 5), (17, 33), (149, 1024), (499, 1280)])
def test_native_pt_writer_is_a_torch_archive(tmp_path, rows, cols):
    """ser_pt_write_f32 writes what torch.save(tensor) would: torch.load (both weights_only settings) returns the same
    bare float32 CPU tensor, every zip member's CRC-32 (PCLMULQDQ path and table tail) verifies, and the payload is
    64-byte aligned like torch's own archives."""
    import zipfile
    t = torch.randn(rows, cols)
    p = str(tmp_path / "MSP-PODCAST_0001_0008.pt")
    frontend.save_feature(t, p)
    for wo in (True, False):
        back = torch.load(p, weights_only=wo)
        assert back.dtype == torch.float32 and back.device.type == "cpu" and back.shape == t.shape and torch.equal(back, t)
    z = zipfile.ZipFile(p)
    assert z.testzip() is None
    names = z.namelist()
    assert names == [f"MSP-PODCAST_0001_0008/{m}" for m in ("data.pkl", "byteorder", "data/0", "version")]
    info = z.getinfo("MSP-PODCAST_0001_0008/data/0")
    raw = open(p, "rb").read()
    name_len, extra_len = np.frombuffer(raw[info.header_offset + 26: info.header_offset + 30], dtype="<u2")   # local header
    payload = info.header_offset + 30 + int(name_len) + int(extra_len)
    assert payload % 64 == 0 and info.file_size == rows * cols * 4
    assert np.array_equal(np.frombuffer(raw[payload: payload + rows * cols * 4], dtype="<f4"), t.numpy().reshape(-1))
    # same pickle program as torch.save's (modulo integer opcode widths): rebuilds through torch._utils._rebuild_tensor_v2
    import pickletools
    ops = [op.name for op, _, _ in pickletools.genops(z.read("MSP-PODCAST_0001_0008/data.pkl"))]
    torch.save(t, str(tmp_path / "ref.pt"))
    ref_ops = [op.name for op, _, _ in pickletools.genops(zipfile.ZipFile(str(tmp_path / "ref.pt")).read("ref/data.pkl"))]
    norm = lambda o: ["INT" if x in ("BININT", "BININT1", "BININT2") else x for x in o]      # noqa: E731
    assert rows == 0 or norm(ops) == norm(ref_ops)        # (torch pickles an empty tensor's integers differently)


def test_pt_writer_threads_do_not_interfere(tmp_path):
    from concurrent.futures import ThreadPoolExecutor
    ts = [torch.randn(50 + i, 64) for i in range(32)]
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(lambda it: frontend.save_feature(it[1], str(tmp_path / f"u{it[0]}.pt")), enumerate(ts)))
    for i, t in enumerate(ts):
        assert torch.equal(torch.load(str(tmp_path / f"u{i}.pt")), t)


def test_driver_save_format_npy_and_bad_layer_report(tmp_path, capsys):
    """The driver around a stubbed model call (no GPU): --save_format npy writes <name>.npy holding the same [T, D] float32
    array, and an out-of-range --n_layer is reported per file the way ``hidden_states[N]`` fails in the reference, not raised."""
    class Stub:
        pipelined = False

        def __init__(self, args, whisper, device):
            self.geo = C.TINY_WAVLM
            self.weight_source = "stub"

        def extract(self, waves, layer_index):
            return [torch.full((self.geo.frames_for(len(w)), 4), float(layer_index)) for w in waves]

    wav_dir = tmp_path / "wav"
    wav_dir.mkdir()
    rng = np.random.default_rng(0)
    for i, n in enumerate((4000, 6000, 9000)):
        write_wav(wav_dir / f"u{i}.wav", 0.1 * rng.standard_normal(n))
    out = tmp_path / "npy"
    assert driver._run(["--wav_dir", str(wav_dir), "--save_path", str(out), "--save_format", "npy", "--use_n_layer", "--n_layer", "1"],
                       whisper=False, extractor_factory=Stub) == 0
    assert sorted(os.listdir(out)) == ["u0.npy", "u1.npy", "u2.npy"]
    a = np.load(out / "u1.npy")
    assert a.dtype == np.float32 and a.shape == (C.TINY_WAVLM.frames_for(6000), 4) and float(a[0, 0]) == 1.0
    capsys.readouterr()
    out2 = tmp_path / "bad"
    assert driver._run(["--wav_dir", str(wav_dir), "--save_path", str(out2), "--use_n_layer", "--n_layer", "7"], whisper=False,
                       extractor_factory=Stub) == 0
    log = capsys.readouterr().out
    assert log.count("tuple index out of range") == 3 and os.listdir(out2) == []
    assert "SER_RUN " in log


def test_default_layer_rule_is_the_reference_rule(tmp_path, capsys):
    """a19: the reference's speech script ignores --n_layer and writes hidden_states[N], N = files found in --save_path at
    start-up (preprocess_speech.py:41,67): hidden_states[0] on a fresh directory (README.md:71), [3] on a re-run over three
    outputs.  The whisper script honours --n_layer (preprocess_whisper.py:71).  --use_n_layer is the additive override."""
    class Stub:
        pipelined = False

        def __init__(self, args, whisper, device):
            self.geo = C.with_layers(C.TINY_WAVLM, 6) if not whisper else C.TINY_WHISPER
            self.weight_source = "stub"

        def extract(self, waves, layer_index):
            return [torch.full((3, 4), float(layer_index)) for _ in waves]

    wav_dir = tmp_path / "wav"
    wav_dir.mkdir()
    rng = np.random.default_rng(0)
    for i in range(3):
        write_wav(wav_dir / f"u{i}.wav", 0.1 * rng.standard_normal(4000))
    out = tmp_path / "pt"
    assert driver._run(["--wav_dir", str(wav_dir), "--save_path", str(out), "--n_layer", "5"], whisper=False, extractor_factory=Stub) == 0
    log = capsys.readouterr().out
    assert "Layer rule: hidden_states[0]" in log and "preprocess_speech.py:41,67" in log
    assert float(torch.load(out / "u1.pt")[0, 0]) == 0.0
    assert driver._run(["--wav_dir", str(wav_dir), "--save_path", str(out)], whisper=False, extractor_factory=Stub) == 0
    assert "Layer rule: hidden_states[3]" in capsys.readouterr().out and float(torch.load(out / "u1.pt")[0, 0]) == 3.0
    assert driver._run(["--wav_dir", str(wav_dir), "--save_path", str(out), "--use_n_layer", "--n_layer", "5"], whisper=False,
                       extractor_factory=Stub) == 0
    assert "Layer rule: hidden_states[5] (--n_layer)" in capsys.readouterr().out and float(torch.load(out / "u2.pt")[0, 0]) == 5.0
    outw = tmp_path / "ptw"
    assert driver._run(["--wav_dir", str(wav_dir), "--save_path", str(outw), "--n_layer", "1"], whisper=True, extractor_factory=Stub) == 0
    assert "Layer rule: hidden_states[1] (--n_layer)" in capsys.readouterr().out and float(torch.load(outw / "u0.pt")[0, 0]) == 1.0


def test_layer_rule_and_resume_do_not_mix_layers_silently(tmp_path, capsys):
    """The reference's rule (layer = files found at start-up) against this build's resume features: partial outputs a
    killed run left (``*.pt.tmp``) are removed and not counted; a non-empty directory gets a WARNING naming the layer; and
    --skip_existing -- which by construction resumes into a non-empty directory -- is refused unless the layer is chosen explicitly."""
    class Stub:
        pipelined = False

        def __init__(self, args, whisper, device):
            self.geo = C.with_layers(C.TINY_WAVLM, 6)
            self.weight_source = "stub"

        def extract(self, waves, layer_index):
            return [torch.full((3, 4), float(layer_index)) for _ in waves]

    wav_dir = tmp_path / "wav"
    wav_dir.mkdir()
    rng = np.random.default_rng(0)
    for i in range(3):
        write_wav(wav_dir / f"u{i}.wav", 0.1 * rng.standard_normal(4000))
    out = tmp_path / "pt"
    out.mkdir()
    (out / "u0.pt.tmp").write_bytes(b"half a file")                          # earlier versions' name: stale at once
    (out / "u1.npy.tmp").write_bytes(b"half a file")
    dead = 1
    while True:                                                             # a pid nobody has: its partial output is stale
        dead += 7919
        try:
            os.kill(dead, 0)
        except ProcessLookupError:
            break
        except OSError:
            continue
    (out / f"u2.pt.{dead}.tmp").write_bytes(b"half a file")
    two_min_ago = __import__("time").time() - 120.0
    os.utime(out / f"u2.pt.{dead}.tmp", (two_min_ago, two_min_ago))
    fresh = out / f"u3.pt.{dead}.tmp"                                       # same dead pid, written this very minute: a writer on ANOTHER host
    fresh.write_bytes(b"half a file")                                       # (its pid means nothing here) -- kept
    # ... and one whose writer is ALIVE (round 5, ADVICE r4: a second job sharing --save_path must not lose its files in progress)
    live = out / f"other_job.pt.{os.getppid()}.tmp"
    live.write_bytes(b"being written by somebody else")
    assert driver._run(["--wav_dir", str(wav_dir), "--save_path", str(out)], whisper=False, extractor_factory=Stub) == 0
    log = capsys.readouterr().out
    assert "It has 0 files in it" in log and "Layer rule: hidden_states[0]" in log and "WARNING" not in log
    assert log.count("Removed stale partial output") == 3 and live.exists() and fresh.exists()
    live.unlink()
    fresh.unlink()
    assert sorted(os.listdir(out)) == ["u0.pt", "u1.pt", "u2.pt"] and float(torch.load(out / "u1.pt")[0, 0]) == 0.0
    os.remove(out / "u2.pt")                                                # "a killed run": two of three outputs exist
    assert driver._run(["--wav_dir", str(wav_dir), "--save_path", str(out), "--skip_existing"], whisper=False, extractor_factory=Stub) == 0
    log = capsys.readouterr().out
    assert "Error: --skip_existing" in log and "hidden_states[2]" in log and not (out / "u2.pt").exists()
    assert driver._run(["--wav_dir", str(wav_dir), "--save_path", str(out), "--skip_existing", "--use_n_layer", "--n_layer", "0"],
                       whisper=False, extractor_factory=Stub) == 0
    capsys.readouterr()
    assert [float(torch.load(out / f"u{i}.pt")[0, 0]) for i in range(3)] == [0.0, 0.0, 0.0]      # one layer in the directory
    assert driver._run(["--wav_dir", str(wav_dir), "--save_path", str(out)], whisper=False, extractor_factory=Stub) == 0
    log = capsys.readouterr().out
    assert "WARNING: --save_path already holds 3 files, so this run writes hidden_states[3]" in log


def test_driver_on_an_empty_directory(tmp_path, capsys):
    """Nothing to do is not an error: the reference prints its header lines and exits 0 (preprocess_speech.py:87-124)."""
    class Stub:
        pipelined = False

        def __init__(self, args, whisper, device):
            self.geo = C.TINY_WAVLM
            self.weight_source = "stub"

        def extract(self, waves, layer_index):
            raise AssertionError("no file, no model call")

    wav_dir = tmp_path / "wav"
    wav_dir.mkdir()
    assert driver._run(["--wav_dir", str(wav_dir), "--save_path", str(tmp_path / "out")], whisper=False, extractor_factory=Stub) == 0
    log = capsys.readouterr().out
    assert "0 file are going to be processed..." in log and "0 utterances" in log
    assert os.listdir(tmp_path / "out") == []


def test_native_wav_reader_extensible_header_and_extra_chunks(tmp_path):
    """WAVE_FORMAT_EXTENSIBLE (format tag 0xFFFE, sub-format PCM) with a LIST chunk before ``fmt `` and an odd-sized chunk before
    ``data`` -- what many recorders write and Python's ``wave`` module refuses: the native reader decodes it like plain PCM16."""
    import struct
    pcm = (np.random.default_rng(3).integers(-20000, 20000, size=777)).astype("<i2")
    fmt = struct.pack("<HHIIHHHHIH14s", 0xFFFE, 1, 16000, 32000, 2, 16, 22, 16, 4, 1, b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71")
    chunks = b"LIST" + struct.pack("<I", 4) + b"INFO" + b"fmt " + struct.pack("<I", len(fmt)) + fmt \
        + b"junk" + struct.pack("<I", 3) + b"abc\x00" + b"data" + struct.pack("<I", pcm.nbytes) + pcm.tobytes()
    p = tmp_path / "ext.wav"
    p.write_bytes(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)
    got = frontend.load_wav_16k(str(p))
    assert got.dtype == np.float32 and np.array_equal(got, pcm.astype(np.float32) / 32768.0)


def test_pretrained_entry_points_demand_a_checkpoint():
    """preprocessing/preprocess_{speech,whisper}_pretrained.py: the reference hard-codes the fine-tuned checkpoint's path
    (preprocess_speech_pretrained.py:170, preprocess_whisper_pretrained.py:183); here it is an argument, and leaving it out
    stops the script before any model is built -- with the reference's closing line and its exit status 0."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for script in ("preprocess_speech_pretrained.py", "preprocess_whisper_pretrained.py"):
        r = subprocess.run([sys.executable, os.path.join(root, "preprocessing", script), "--ssl_type", "anything",
                            "--wav_dir", "/nonexistent", "--save_path", "/nonexistent"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (script, r.stderr[-400:])
        assert "--checkpoint" in r.stdout and "Something went wrong" in r.stdout, (script, r.stdout)


def test_wavlm_gate_fold_matches_the_oracle_gate():
    """weights.fold_wavlm_gate (what ser_attention's in-kernel gate multiplies, ser_attention_args.gate_x) against the oracle's statement of
    HF modeling_wavlm.py:167-180 on LayerNorm1(x): the closed form  rstd (x . wg - mu sum(wg)) + t  reproduces both pre-activation sums."""
    import torch
    import torch.nn.functional as F
    from interspeech_ser_amd import config as C
    from interspeech_ser_amd.weights import fold_wavlm_gate, synthetic_state_dict
    from oracle import ssl_oracle as O
    geo = C.TINY_WAVLM
    sd = {k: v.double() for k, v in synthetic_state_dict(geo, 11).items()}
    H, dh, D = geo.heads, geo.head_dim, geo.hidden
    a = "encoder.layers.1.attention"
    ln = "encoder.layers.1.layer_norm"
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(37, D, generator=g) * 3.0 + torch.randn(37, 1, generator=g) * 5.0).double()      # rows with their own offsets
    wg, t = fold_wavlm_gate(sd[a + ".gru_rel_pos_linear.weight"], sd[a + ".gru_rel_pos_linear.bias"], sd[ln + ".weight"], sd[ln + ".bias"], H, dh)
    assert wg.shape == (2 * H, dh) and t.shape == (H, 2)
    shift = x.mean(1, keepdim=True) + 0.7                          # the operand copy is stored relative to SOME per-row shift: the form is invariant
    xs = x - shift
    mu = xs.mean(1, keepdim=True)
    rstd = torch.rsqrt(xs.var(1, unbiased=False, keepdim=True) + geo.layer_norm_eps)
    dots = torch.einsum("thd,hjd->thj", xs.view(-1, H, dh), wg.view(H, 2, dh))
    pre = rstd[:, :, None] * (dots - mu[:, :, None] * wg.view(H, 2, dh).sum(2)[None]) + t[None]
    aa, bb = torch.sigmoid(pre[..., 0]), torch.sigmoid(pre[..., 1])
    gate = aa * (bb * sd[a + ".gru_rel_pos_const"].view(1, H) - 1.0) + 2.0                      # [T, H]
    x_ln = F.layer_norm(x, (D,), sd[ln + ".weight"], sd[ln + ".bias"], geo.layer_norm_eps)
    ref = O.wavlm_gate(geo, sd, a, x_ln)                                                       # [H, T]
    assert (gate.T - ref).abs().max() < 1e-10


@pytest.mark.parametrize("slots", [2, 3])
def test_pipelined_driver_loop_with_a_stubbed_model(tmp_path, capsys, slots):
    """The pipelined form of the driver loop (submit / collect / hold over SLOTS slots, the form every GPU run takes) around a stub:
    every file is written once with ITS batch's features, at most SLOTS - 1 batches are in flight when a new one is prepared, a slot is
    never refilled before the batch that used it was collected, and a batch whose collect fails is retried per utterance."""
    events = []

    class Stub:
        pipelined = True
        SLOTS = slots

        def __init__(self, args, whisper, device):
            self.geo = C.TINY_WAVLM
            self.weight_source = "stub"
            self.busy = {}
            self.n = 0

        def submit(self, waves, layer_index, slot):
            assert slot not in self.busy, "slot refilled before its batch was collected"
            assert len(self.busy) <= self.SLOTS - 1
            self.n += 1
            self.busy[slot] = self.n
            events.append(("submit", self.n, slot))
            return dict(slot=slot, batch=self.n, lengths=[len(w) for w in waves])

        def collect(self, ticket):
            assert self.busy.pop(ticket["slot"]) == ticket["batch"]
            events.append(("collect", ticket["batch"], ticket["slot"]))
            if ticket["batch"] == 2:
                raise RuntimeError("device lost this batch")
            return [torch.full((self.geo.frames_for(n), 4), float(ticket["batch"])) for n in ticket["lengths"]]

        def hold(self, slot, futures):
            for f in futures:
                f.result()

        def extract(self, waves, layer_index):                   # the per-utterance retry of the failed batch
            return [torch.full((self.geo.frames_for(len(w)), 4), -1.0) for w in waves]

    wav_dir = tmp_path / "wav"
    wav_dir.mkdir()
    rng = np.random.default_rng(0)
    for i in range(10):
        write_wav(wav_dir / f"u{i}.wav", 0.1 * rng.standard_normal(4000 + 100 * i))
    out = tmp_path / "out"
    assert driver._run(["--wav_dir", str(wav_dir), "--save_path", str(out), "--batch_size", "2", "--use_n_layer", "--n_layer", "0"],
                       whisper=False, extractor_factory=Stub) == 0
    assert sorted(os.listdir(out)) == sorted(f"u{i}.pt" for i in range(10))
    vals = [float(torch.load(str(out / f"u{i}.pt"))[0, 0]) for i in range(10)]          # two files per batch (the driver orders files by length)
    assert sorted(vals) == [-1.0, -1.0, 1.0, 1.0, 3.0, 3.0, 4.0, 4.0, 5.0, 5.0]        # batch 2 came back through the per-utterance retry
    for i in range(10):
        assert torch.load(str(out / f"u{i}.pt")).shape == (C.TINY_WAVLM.frames_for(4000 + 100 * i), 4)
    order = [e for e in events if e[0] == "collect"]
    assert [e[1] for e in order] == [1, 2, 3, 4, 5]                                    # collected oldest first
    first_collect = events.index(("collect", 1, 0))
    assert [e[0] for e in events[:first_collect]].count("submit") == slots              # SLOTS batches submitted before the first wait
    assert "10 utterances" in capsys.readouterr().out
