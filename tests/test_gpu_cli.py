"""-m gpu: the drivers end to end (BASELINE.json configs[0]: 8 synthetic 3 s wavs -> 8 .pt of
shape [149, 1024]) and the Whisper driver's crop rule, through files on disk."""
import os
import wave

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def write_wav(path, x):
    pcm = (np.clip(x, -1, 1) * 32767).astype("<i2")
    with wave.open(str(path), "wb") as wf:
        wf.setnchannels(1)
        wf.setsampwidth(2)
        wf.setframerate(16000)
        wf.writeframes(pcm.tobytes())
    return pcm.astype(np.float32) / 32768.0


def synth(seed, n):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    return 0.1 * rng.standard_normal(n) + 0.2 * np.sin(2 * np.pi * 220 * t)


def test_speech_driver_config0(tmp_path, capsys):
    from interspeech_ser_amd import config as C
    from interspeech_ser_amd import driver
    from interspeech_ser_amd.weights import synthetic_state_dict
    from oracle import ssl_oracle as O
    wav_dir, out = tmp_path / "wavs", tmp_path / "feats"
    wav_dir.mkdir()
    waves = {}
    for i in range(8):
        n = 48000 if i < 6 else 48000 - 777 * i            # two ragged ones
        waves[f"syn_{i:04d}"] = write_wav(wav_dir / f"syn_{i:04d}.wav", synth(7 + i, n))
    (wav_dir / "broken.wav").write_bytes(b"not a wav file")
    write_wav(wav_dir / "syn_0003b_tiny.wav", synth(99, 300))       # decodes, but is shorter than the receptive field:
    # the README recipe (README.md:71) on a fresh directory: the reference's rule gives hidden_states[0] (0 files found at
    # start-up, preprocess_speech.py:41,67) whatever --n_layer says -- and the forward stops after that state
    rc = driver.run_speech(["--ssl_type", "microsoft/wavlm-large", "--wav_dir", str(wav_dir), "--save_path", str(out),
                            "--synthetic_weights", "--n_layer", "12", "--batch_size", "5"])
    text = capsys.readouterr().out                                    # its batch falls back to one-by-one, neighbours survive
    assert rc == 0
    assert "Layer rule: hidden_states[0]" in text and "reference's rule" in text
    assert "Using device = cuda" in text and "10 file are going to be processed..." in text
    assert "Failed to process" in text and "broken.wav" in text      # logged and skipped, like the reference
    assert "syn_0003b_tiny.wav" in text and "receptive field" in text
    os.remove(wav_dir / "syn_0003b_tiny.wav")
    files = sorted(os.listdir(out))
    assert files == [f"syn_{i:04d}.pt" for i in range(8)]
    geo = C.WAVLM_LARGE
    sd = synthetic_state_dict(geo, 7)                                  # --seed default
    for name in ("syn_0000", "syn_0007"):
        got = torch.load(out / f"{name}.pt")
        w = waves[name]
        assert got.dtype == torch.float32 and got.device.type == "cpu"
        assert tuple(got.shape) == (geo.frames_for(len(w)), 1024)
        ref = O.extract_speech(geo, sd, w, layer_index=0)
        assert float((got - ref).abs().max() / max(1.0, float(ref.abs().max()))) < 1e-3
    assert tuple(torch.load(out / "syn_0000.pt").shape) == (149, 1024)

    # the reference's rule on a re-run: 8 files now in save_path -> hidden_states[8]
    rc = driver.run_speech(["--ssl_type", "microsoft/wavlm-large", "--wav_dir", str(wav_dir), "--save_path", str(out),
                            "--synthetic_weights", "--batch_size", "8"])
    assert rc == 0 and "Layer rule: hidden_states[8]" in capsys.readouterr().out
    got = torch.load(out / "syn_0001.pt")
    with torch.no_grad():
        ref = O.extract_speech(geo, sd, waves["syn_0001"], layer_index=8)
    assert float((got - ref).abs().max() / max(1.0, float(ref.abs().max()))) < 1e-3

    # --use_average y through the two-slot pipeline (mean of the last four states, preprocess_speech.py:52-63)
    out_avg = tmp_path / "feats_avg"
    rc = driver.run_speech(["--ssl_type", "microsoft/wavlm-large", "--wav_dir", str(wav_dir), "--save_path", str(out_avg),
                            "--synthetic_weights", "--use_average", "y", "--batch_size", "3"])
    assert rc == 0 and len(os.listdir(out_avg)) == 8
    for name in ("syn_0002", "syn_0006"):
        got = torch.load(out_avg / f"{name}.pt")
        with torch.no_grad():
            ref = O.extract_speech(geo, sd, waves[name], use_average=True)
        assert got.shape == ref.shape
        assert float((got - ref).abs().max() / max(1.0, float(ref.abs().max()))) < 1e-3


def test_unknown_hub_name_resolves_through_the_snapshot_config_json(tmp_path, capsys, monkeypatch):
    """a3: ``AutoModel.from_pretrained(--ssl_type)`` works for any hub id because the snapshot carries its own config.json
    (preprocess_speech.py:111-112).  A fine-tune under a name the built-in table does not know -- offline HF cache layout
    ``$HF_HOME/hub/models--acme--wavlm-ser-v7/snapshots/<rev>/{config.json, model.safetensors}``, keys with the ``wavlm.`` prefix a
    task-head checkpoint has -- runs through the speech driver with the README's command line (no --checkpoint, no
    --synthetic_weights), and so does the same directory given as --ssl_type; a GroupNorm variant is refused with the reference's line."""
    import json
    from safetensors.torch import save_file
    from interspeech_ser_amd import config as C
    from interspeech_ser_amd import driver
    from interspeech_ser_amd.frontend import load_wav_16k
    from interspeech_ser_amd.weights import synthetic_state_dict
    from oracle import ssl_oracle as O
    geo = C.TINY_WAVLM
    sd = synthetic_state_dict(geo, 77)
    cfg = {"model_type": "wavlm", "hidden_size": geo.hidden, "num_hidden_layers": geo.num_layers, "num_attention_heads": geo.heads,
           "intermediate_size": geo.ffn, "conv_dim": list(geo.conv_dim), "conv_kernel": list(geo.conv_kernel),
           "conv_stride": list(geo.conv_stride), "conv_bias": geo.conv_bias, "feat_extract_norm": "layer", "do_stable_layer_norm": True,
           "num_conv_pos_embeddings": geo.pos_conv_kernel, "num_conv_pos_embedding_groups": geo.pos_conv_groups,
           "num_buckets": geo.num_buckets, "max_bucket_distance": geo.max_bucket_distance, "layer_norm_eps": geo.layer_norm_eps}
    snap = tmp_path / "hf" / "hub" / "models--acme--wavlm-ser-v7" / "snapshots" / "0123abcd"
    snap.mkdir(parents=True)
    (snap / "config.json").write_text(json.dumps(cfg))
    save_file({"wavlm." + k: v.contiguous() for k, v in sd.items()}, str(snap / "model.safetensors"))
    monkeypatch.setenv("HF_HOME", str(tmp_path / "hf"))
    wav_dir = tmp_path / "wav"
    wav_dir.mkdir()
    write_wav(wav_dir / "a.wav", synth(5, 16000))
    write_wav(wav_dir / "b.wav", synth(6, 9000))
    for tag, ssl_type in (("cache", "acme/wavlm-ser-v7"), ("dir", str(snap))):
        out = tmp_path / f"pt_{tag}"
        rc = driver.run_speech(["--ssl_type", ssl_type, "--wav_dir", str(wav_dir), "--save_path", str(out), "--mode", "fp32x"])
        text = capsys.readouterr().out
        assert rc == 0 and "No pretrained model found" not in text, text
        assert sorted(os.listdir(out)) == ["a.pt", "b.pt"]
        x = load_wav_16k(str(wav_dir / "b.wav"))
        ref = O.extract_speech(geo, sd, x, layer_index=0)                 # fresh directory -> hidden_states[0] (the reference's rule)
        got = torch.load(out / "b.pt")
        assert got.shape == ref.shape and float((got - ref).abs().max() / max(1.0, float(ref.abs().max()))) < 1e-3
    cfg["feat_extract_norm"] = "group"
    (snap / "config.json").write_text(json.dumps(cfg))
    rc = driver.run_speech(["--ssl_type", "acme/wavlm-ser-v7", "--wav_dir", str(wav_dir), "--save_path", str(tmp_path / "pt_bad")])
    text = capsys.readouterr().out
    assert rc == 0 and "No pretrained model found with the name acme/wavlm-ser-v7" in text and "GroupNorm" in text


def test_early_exit_states_are_bit_equal_to_the_full_forward():
    """``last_state`` = N launches only what hidden_states[0..N] need (the reference keeps one state per utterance,
    preprocess_speech.py:67 / preprocess_whisper.py:71): the states it leaves are bit-equal to the full forward's, through
    the recorded command list (partial replay) and launch by launch, for the speech and the Whisper encoder."""
    from interspeech_ser_amd import config as C
    from interspeech_ser_amd.engine import SpeechEncoder, WhisperEncoder
    from interspeech_ser_amd.weights import synthetic_state_dict
    for geo, cls in ((C.TINY_WAVLM, SpeechEncoder), (C.TINY_HUBERT, SpeechEncoder), (C.TINY_WHISPER, WhisperEncoder)):
        sd = synthetic_state_dict(geo, 3)
        waves = [synth(5, 9000).astype(np.float32), synth(6, 16000).astype(np.float32)]
        lens = [len(w) for w in waves]
        for mode in ("f16a", "bf16"):
            enc = cls(geo, sd, "cuda:0", mode=mode)
            full = enc.forward(enc.upload(waves), lens)
            torch.cuda.synchronize()
            ref = full.states.clone()
            L = geo.num_layers
            for use_tape in (True, False):
                enc.use_tape = use_tape
                for n in range(L + 1):
                    full.states.fill_(float("nan"))
                    hs = enc.forward(enc.upload(waves), lens, last_state=n)
                    torch.cuda.synchronize()
                    assert hs.computed == n + 1 if n < L else hs.computed == L + 1
                    assert torch.equal(hs.states[: n + 1], ref[: n + 1]), (geo.name, mode, use_tape, n)
                    if n < L:
                        assert torch.isnan(hs.states[n + 1:]).all()                   # nothing beyond state n was launched
                        with pytest.raises(IndexError):
                            hs.utterance(0, n + 1)
                        with pytest.raises(IndexError):
                            hs.utterance(0, -1)
            with pytest.raises(IndexError):
                enc.forward(enc.upload(waves), lens, last_state=L + 1)


def test_whisper_driver_crop(tmp_path, capsys):
    from interspeech_ser_amd import config as C
    from interspeech_ser_amd import driver
    wav_dir, out = tmp_path / "wavs", tmp_path / "feats"
    wav_dir.mkdir()
    write_wav(wav_dir / "short.wav", synth(1, 16000))
    write_wav(wav_dir / "long.wav", synth(2, 16000 * 29))
    rc = driver.run_whisper(["--ssl_type", "openai/whisper-large-v3", "--wav_dir", str(wav_dir), "--save_path", str(out),
                             "--synthetic_weights", "--mode", "bf16"])
    assert rc == 0, capsys.readouterr().out
    a, b = torch.load(out / "short.pt"), torch.load(out / "long.pt")
    assert tuple(a.shape) == (50, 1280)                                # ceil(16000/320)
    assert tuple(b.shape) == (1280, 1280)                              # capped by the hidden size (reference quirk)
    assert torch.isfinite(a).all() and torch.isfinite(b).all()


def test_config4_plumbing_speech_plus_text_into_the_head(tmp_path, capsys):
    """BASELINE.json configs[4] in miniature: HuBERT-xlarge speech features + RoBERTa-large text features for the same
    utterances, written by the two drivers, read back the way the bimodal head's dataset / collate / forward do."""
    import pandas as pd
    from interspeech_ser_amd import driver
    from oracle import fusion_head as H          # the reference head, restated and pinned (tests/test_consumer_contract.py)
    wav_dir, d1, d2 = tmp_path / "Audios", tmp_path / "hubert", tmp_path / "roberta"
    wav_dir.mkdir()
    names = [f"MSP_{i:03d}.wav" for i in range(4)]
    for i, n in enumerate(names):
        write_wav(wav_dir / n, synth(30 + i, 16000 * (1 + i % 3) + 123 * i))
    df = pd.DataFrame({"FileName": names, "transcription": ["yes", "no not really", "what a lovely day it is today", "hm"]})
    csv = tmp_path / "labels.csv"
    df.to_csv(csv, index=False)

    def tok(texts, max_len=80):
        ids = torch.full((len(texts), max_len), 1, dtype=torch.int64)
        mask = torch.zeros((len(texts), max_len), dtype=torch.int64)
        for i, t in enumerate(texts):
            toks = [0] + [3 + (hash(w) % 40000) for w in t.split()] + [2]
            ids[i, : len(toks)] = torch.tensor(toks)
            mask[i, : len(toks)] = 1
        return ids, mask

    assert driver.run_speech(["--ssl_type", "facebook/hubert-xlarge-ll60k", "--wav_dir", str(wav_dir), "--save_path", str(d1),
                              "--synthetic_weights", "--mode", "bf16", "--use_n_layer", "--n_layer", "-1", "--batch_size", "3"]) == 0
    assert driver.run_roberta(["--roberta_type", "roberta-large", "--df_path", str(csv), "--save_path", str(d2),
                               "--synthetic_weights", "--mode", "bf16"], tokenize=tok) == 0
    assert sorted(os.listdir(d1)) == sorted(os.listdir(d2)) == [n.replace(".wav", ".pt") for n in names]
    labels = np.eye(8, dtype=np.float32)[[0, 3, 5, 7]]
    batch = H.collate_fn([H.dataset_item(n, str(d1), str(d2), lab) for n, lab in zip(names, labels)])
    assert batch["feat1"].shape[0] == 4 and batch["feat1"].shape[2] == 1280 and batch["feat2"].shape == (4, 80, 1024)
    # the reference head at its real size (Linear(1280,512) / Linear(1024,512) -> biGRU(512) -> cross-MHA(1024) -> 8), seeded
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fusion_head_pins.npz"))
    shapes = {str(k): tuple(int(x) for x in str(sh).split(",")) for k, sh in zip(g["keys"], g["shapes"])}
    head = H.MultiModalEmotionClassifier(1280, 1024, 512, 8, 0.5).eval()
    head.load_state_dict(H.seeded_head_weights(shapes, int(g["seed_weights"])), strict=True)
    with torch.no_grad():
        logits = head(batch["feat1"], batch["feat2"])
    assert logits.shape == (4, 8) and torch.isfinite(logits).all()
    loss = torch.nn.CrossEntropyLoss()(logits, batch["label"].max(dim=1)[1].long())
    assert torch.isfinite(loss)


def test_config4_extract_train_eval_end_to_end(tmp_path, capsys):
    """BASELINE.json configs[4] closed end to end: 64 synthetic clips + transcripts -> preprocess_speech (HuBERT-xlarge) and
    preprocess_roberta (RoBERTa-large) write the two lazy dirs -> the train counterpart (interspeech_ser_amd/head.py,
    bin/train_cat_bimodal_lazy_1head.py here) runs 2 epochs from a reference-style config JSON and saves the best-F1
    ``multimodal_ser.pt`` -> the eval counterpart writes results/dev.csv.  The checkpoint is a checkpoint of the REFERENCE head:
    it loads strictly into the restated class that tests/golden/fusion_head_pins.npz pins to the reference's own definition and
    gives the same logits as the product model (/root/reference does not exist on this box; the CPU suite loads the same kind
    of checkpoint into the reference's class itself)."""
    import json
    import pandas as pd
    from interspeech_ser_amd import driver
    from interspeech_ser_amd import head as HD
    from oracle import fusion_head as H
    wav_dir, d1, d2 = tmp_path / "Audios", tmp_path / "hubert_xlarge", tmp_path / "roberta_large"
    wav_dir.mkdir()
    rng = np.random.default_rng(99)
    names = [f"MSP-PODCAST_{i:04d}.wav" for i in range(64)]
    cls = rng.integers(0, 8, 64)
    words = ["angry", "sad", "happy", "wow", "scared", "yuck", "pff", "okay"]
    for i, n in enumerate(names):
        x = synth(100 + i, int(rng.integers(16000, 48000)))
        write_wav(wav_dir / n, x * (0.5 + 0.2 * cls[i]))                 # a (weak) class cue in the level
    lab = pd.DataFrame(np.eye(8, dtype=np.float32)[cls], columns=HD.CLASSES)
    lab.insert(0, "FileName", names)
    lab["Split_Set"] = ["Train" if i % 4 else "Development" for i in range(64)]
    lab.to_csv(tmp_path / "processed_labels.csv", index=False)
    pd.DataFrame({"FileName": names, "transcription": [" ".join([words[c]] * (1 + i % 5)) for i, c in enumerate(cls)]}).to_csv(
        tmp_path / "transcripts.csv", index=False)

    def tok(texts, max_len=80):
        ids = torch.full((len(texts), max_len), 1, dtype=torch.int64)
        mask = torch.zeros((len(texts), max_len), dtype=torch.int64)
        for i, t in enumerate(texts):
            toks = [0] + [3 + 7 * words.index(w) for w in t.split()] + [2]
            ids[i, : len(toks)] = torch.tensor(toks)
            mask[i, : len(toks)] = 1
        return ids, mask

    assert driver.run_speech(["--ssl_type", "facebook/hubert-xlarge-ll60k", "--wav_dir", str(wav_dir), "--save_path", str(d1),
                              "--synthetic_weights", "--mode", "bf16", "--use_n_layer", "--n_layer", "-1"]) == 0
    assert driver.run_roberta(["--roberta_type", "roberta-large", "--df_path", str(tmp_path / "transcripts.csv"), "--save_path", str(d2),
                               "--synthetic_weights", "--mode", "bf16"], tokenize=tok) == 0
    assert len(os.listdir(d1)) == len(os.listdir(d2)) == 64
    cfg = {"wav_dir": str(wav_dir), "txt_dir": str(tmp_path / "transcripts.csv"), "lazy_dir1": str(d1), "lazy_dir2": str(d2),
           "label_path": str(tmp_path / "processed_labels.csv"), "feat1_dim": 1280, "feat2_dim": 1024, "use_balanced_batch": False,
           "use_focalloss": False, "epochs": 2, "lr": 1e-4, "model_path": str(tmp_path / "experiments" / "head1"), "batch_size": 16,
           "accum_step": 1}
    cfg_path = tmp_path / "config_cat_bimodal_lazy_lr1e4_hubertxlarge_roberta_head1.json"
    cfg_path.write_text(json.dumps(cfg))
    capsys.readouterr()
    assert HD.main(["--config_path", str(cfg_path), "--seed", "7"]) == 0                       # bin/train_cat_bimodal_lazy_1head.py
    assert HD.main(["--config_path", str(cfg_path), "--seed", "7"], evaluate_only=True) == 0   # bin/eval_cat_bimodal_lazy_1head.py
    ck = os.path.join(cfg["model_path"], "multimodal_ser.pt")
    sd = torch.load(ck, map_location="cpu")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fusion_head_pins.npz"))
    assert list(sd.keys()) == [str(k) for k in g["keys"]]                                      # the reference's 42 keys, in order
    assert [tuple(v.shape) for v in sd.values()] == [tuple(int(x) for x in str(sh).split(",")) for sh in g["shapes"]]
    rows = list(__import__("csv").reader(open(os.path.join(cfg["model_path"], "results", "dev.csv"))))
    assert len(rows) == 17 and rows[0][:2] == ["Filename", "Prediction"]
    batch = HD.collate_fn([HD.MultiLabelAudioDataset(names[:4], np.eye(8)[cls[:4]], str(d1), str(d2))[i] for i in range(4)])
    mine, pinned = HD.MultiModalEmotionClassifier(1280, 1024).eval(), H.MultiModalEmotionClassifier(1280, 1024, 512, 8, 0.5).eval()
    mine.load_state_dict(sd, strict=True)
    pinned.load_state_dict(sd, strict=True)
    with torch.no_grad():
        a, b = mine(batch["feat1"], batch["feat2"]), pinned(batch["feat1"], batch["feat2"])
    assert torch.isfinite(a).all() and float((a - b).abs().max()) < 1e-4
    # the CSV's logits are the evaluation run's (GPU) logits of the same checkpoint
    dev_names = [n for i, n in enumerate(names) if i % 4 == 0]
    assert [r[0] for r in rows[1:]] == dev_names
    assert all(r[1] in HD.CLASS_LETTERS and len(r) == 10 and all(np.isfinite(float(x)) for x in r[2:]) for r in rows[1:])


@pytest.mark.parametrize("family,mode", [("wavlm", "f16x"), ("hubert", "f16x"), ("wavlm", "fp32x"), ("hubert", "fp32x"), ("wavlm", "f16a"), ("hubert", "f16a"),
                                         ("wavlm", "f16q"), ("hubert", "f16q"), ("wavlm", "f16"), ("wavlm", "f16m"), ("hubert", "f16m"), ("wavlm", "f16mf"), ("hubert", "f16mf")])
def test_lora_checkpoint_through_the_driver_matches_unmerged_oracle(tmp_path, capsys, family, mode):
    """Next row 8f-4 end to end (preprocessing/preprocess_speech_pretrained.py:108-177): a PEFT-wrapped checkpoint --
    ``wavlm.base_model.model.*`` names, ``q_proj`` / ``v_proj`` split into ``base_layer`` + ``lora_A`` / ``lora_B`` (r = 8,
    alpha = 16), a classifier head beside the encoder -- goes through ``run_speech --checkpoint``; the HIP path sees
    merged weights (weights.merge_lora), the oracle applies the adapters UN-MERGED (x W^T + 2 (x A^T) B^T) like the
    wrapped module does.  Adapters are large on purpose (the merge changes the states by far more than the tolerance)."""
    from safetensors.torch import save_file
    from interspeech_ser_amd import config as C
    from interspeech_ser_amd import driver
    from interspeech_ser_amd.weights import synthetic_state_dict
    from oracle import ssl_oracle as O
    geo = C.TINY_WAVLM if family == "wavlm" else C.TINY_HUBERT
    base = synthetic_state_dict(geo, 41)
    g = torch.Generator().manual_seed(42)
    r, alpha, D = 8, 16.0, geo.hidden
    ckpt, ref_sd = {}, dict(base)
    ref_sd["lora_scale"] = torch.tensor(alpha / r)
    for k, v in base.items():
        mod, _, leaf = k.rpartition(".")
        if mod.endswith((".q_proj", ".v_proj")):
            ckpt[f"wavlm.base_model.model.{mod}.base_layer.{leaf}"] = v
            if leaf == "weight":
                A = torch.randn(r, D, generator=g) * 0.3
                B = torch.randn(D, r, generator=g) * 0.3
                ckpt[f"wavlm.base_model.model.{mod}.lora_A.default.weight"] = A
                ckpt[f"wavlm.base_model.model.{mod}.lora_B.default.weight"] = B
                ref_sd[mod + ".lora_A.weight"], ref_sd[mod + ".lora_B.weight"] = A, B
        else:
            ckpt["wavlm.base_model.model." + k] = v
    ckpt["classifier.0.weight"], ckpt["classifier.0.bias"] = torch.zeros(512, D), torch.zeros(512)     # the fine-tuning head
    ckpt["classifier.3.weight"], ckpt["classifier.3.bias"] = torch.zeros(8, 512), torch.zeros(8)
    ck = tmp_path / "lora_ser.safetensors"
    save_file({k: v.contiguous() for k, v in ckpt.items()}, str(ck))
    wav_dir, out = tmp_path / "wav", tmp_path / "pt"
    wav_dir.mkdir()
    waves = {"a.wav": synth(61, 16000), "b.wav": synth(62, 23457)}
    for n, w in waves.items():
        write_wav(wav_dir / n, w)
    # register the tiny geometry under a name the driver can resolve
    C._REGISTRY["tiny-lora-test"] = geo
    try:
        rc = driver.run_speech(["--ssl_type", "tiny-lora-test", "--wav_dir", str(wav_dir), "--save_path", str(out),
                                "--checkpoint", str(ck), "--mode", mode, "--use_n_layer", "--n_layer", "-1", "--lora_alpha", "16"])
    finally:
        C._REGISTRY.pop("tiny-lora-test")
    assert rc == 0, capsys.readouterr().out
    from interspeech_ser_amd.frontend import load_wav_16k
    worst = changed = 0.0
    for n in waves:
        x = load_wav_16k(str(wav_dir / n))
        with torch.no_grad():
            ref = O.speech_hidden_states(geo, ref_sd, torch.from_numpy(O.zero_mean_unit_var(x)))[-1]
            plain = O.speech_hidden_states(geo, base, torch.from_numpy(O.zero_mean_unit_var(x)))[-1]
        got = torch.load(out / n.replace(".wav", ".pt"))
        assert got.shape == ref.shape
        worst = max(worst, float((got - ref).abs().max() / max(1.0, float(ref.abs().max()))))
        changed = max(changed, float((plain - ref).abs().max()))
    assert changed > 0.1, changed            # the adapters matter ...
    # ... and the merged HIP path equals the un-merged reference arithmetic.  fp32x and f16a: the parity gate (1e-3).  f16: these
    # adapters make the query projection ~4x larger than the base weights, i.e. attention logits ~4x larger, and single-product
    # operand rounding (2^-12 relative per operand) moves a softmax weight by (logit error) * ln 2: measured 2.9e-3 here against
    # 6-8e-4 on the unadapted geometries.  f16q (fp32-grade logit path only) measures 1.8e-3 / 4.0e-3: roundings of v, P, the
    # context rows and the output projection are amplified by the NEXT layer's softmax just the same (oracle/numerics_whatif.py)
    # -- the reason f16a (whole attention block on the 3-product split) exists.
    print(f"LoRA {family} {mode}: worst rel err {worst:.3e}")
    assert worst < (5e-3 if mode in ("f16", "f16q") else 1e-3), worst


def test_fp16_range_guard_fails_the_files_instead_of_clipping(tmp_path, capsys):
    """The default numerics mode keeps operand copies on fp16 planes (range 65 504).  A checkpoint whose residual stream leaves that
    range (here: a feed-forward output bias of 1e5 in one channel, 100x the "massive activation" of the outlier fixtures) must not
    yield silently clipped features: the kernels report every value they round to an fp16 plane into the slot's guard word (round 5; round 4
    watched max|hidden state| on sampled batches), the driver reads it with every batch and reports every file of such a batch as
    ``Failed to process ...`` with the way out (--mode fp32x: bf16 planes, fp32 range), where the same checkpoint extracts fine."""
    from safetensors.torch import save_file
    from interspeech_ser_amd import config as C
    from interspeech_ser_amd import driver
    from interspeech_ser_amd.weights import apply_stress, synthetic_state_dict
    geo = C.TINY_WAVLM
    sd = apply_stress(synthetic_state_dict(geo, 41), geo, "outliers")
    sd["encoder.layers.0.feed_forward.output_dense.bias"][7] += 1.0e5
    ck = tmp_path / "huge.safetensors"
    save_file({k: v.contiguous() for k, v in sd.items()}, str(ck))
    wav_dir = tmp_path / "wav"
    wav_dir.mkdir()
    for i, n in enumerate((16000, 9000, 12000)):
        write_wav(wav_dir / f"u{i}.wav", synth(70 + i, n))
    C._REGISTRY["tiny-range-test"] = geo
    try:
        out = tmp_path / "pt_f16x"
        assert driver.run_speech(["--ssl_type", "tiny-range-test", "--wav_dir", str(wav_dir), "--save_path", str(out), "--checkpoint", str(ck),
                                  "--use_n_layer", "--n_layer", "-1"]) == 0                      # default mode: f16mf
        log = capsys.readouterr().out
        assert log.count("Failed to process") == 3 and "fp16 operand range" in log and "--mode fp32x" in log, log
        assert os.listdir(out) == []
        out2 = tmp_path / "pt_fp32x"
        assert driver.run_speech(["--ssl_type", "tiny-range-test", "--wav_dir", str(wav_dir), "--save_path", str(out2), "--checkpoint", str(ck),
                                  "--use_n_layer", "--n_layer", "1", "--mode", "fp32x"]) == 0
        log = capsys.readouterr().out
        assert "Failed to process" not in log and sorted(os.listdir(out2)) == ["u0.pt", "u1.pt", "u2.pt"]
        t = torch.load(out2 / "u0.pt")
        assert float(t.abs().max()) > 9.0e4                                                     # the value the fp16 planes could not hold
    finally:
        C._REGISTRY.pop("tiny-range-test")


def test_fp16_range_guard_sees_the_feed_forward_intermediate(tmp_path, capsys):
    """Round 5: the guard is a device word that EVERY kernel rounding to an fp16 operand plane reports into (ser_gemm_args.range_flag and
    the row kernels'), read back with every batch.  A checkpoint whose FC1 / GELU output leaves the fp16 range while every hidden state
    stays far below it -- intermediate_dense scaled up by 3e4, output_dense down by the same factor, so the residual stream is the
    unscaled one -- passed round 4's watch (max |hidden state| on sampled batches) and wrote features computed from saturated operands.
    Now all files of such a batch fail with the --mode fp32x hint in the fp16-plane modes (f16mf default, f16x, f16m), and extract in fp32x;
    the same checkpoint scaled by 1e4 (beyond half the range only) extracts with the one-line warning.
    Contract: preprocessing/preprocess_speech.py:46,72-73 -- a bad file is a printed failure, never silent garbage."""
    from safetensors.torch import save_file
    from interspeech_ser_amd import config as C
    from interspeech_ser_amd import driver
    from interspeech_ser_amd.weights import synthetic_state_dict
    geo = C.TINY_WAVLM
    wav_dir = tmp_path / "wav"
    wav_dir.mkdir()
    for i, n in enumerate((16000, 9000, 12000)):
        write_wav(wav_dir / f"u{i}.wav", synth(80 + i, n))

    def checkpoint(factor, name):
        sd = synthetic_state_dict(geo, 43)
        for i in range(geo.num_layers):
            p = f"encoder.layers.{i}.feed_forward"
            sd[p + ".intermediate_dense.weight"] *= factor
            sd[p + ".intermediate_dense.bias"] *= factor
            sd[p + ".output_dense.weight"] /= factor
        ck = tmp_path / name
        save_file({k: v.contiguous() for k, v in sd.items()}, str(ck))
        return ck

    C._REGISTRY["tiny-range-test"] = geo
    try:
        ck = checkpoint(3.0e4, "ffn_huge.safetensors")
        for mode in ("f16mf", "f16x", "f16m"):
            out = tmp_path / f"pt_{mode}"
            assert driver.run_speech(["--ssl_type", "tiny-range-test", "--wav_dir", str(wav_dir), "--save_path", str(out), "--checkpoint", str(ck),
                                      "--use_n_layer", "--n_layer", "-1", "--mode", mode]) == 0
            log = capsys.readouterr().out
            assert log.count("Failed to process") == 3 and "fp16 operand range" in log and "--mode fp32x" in log, log
            assert os.listdir(out) == []
        out2 = tmp_path / "pt_fp32x"
        assert driver.run_speech(["--ssl_type", "tiny-range-test", "--wav_dir", str(wav_dir), "--save_path", str(out2), "--checkpoint", str(ck),
                                  "--use_n_layer", "--n_layer", "-1", "--mode", "fp32x"]) == 0
        log = capsys.readouterr().out
        assert "Failed to process" not in log and sorted(os.listdir(out2)) == ["u0.pt", "u1.pt", "u2.pt"]
        t = torch.load(out2 / "u0.pt")
        assert float(t.abs().max()) < 100.0                      # ... while every hidden state is small: round 4's watch saw nothing
        # GELU of the scaled pre-activations: ~1e4 x the plain ones (a few units) -> beyond half the range, inside it: warning, files written
        ck2 = checkpoint(4.0e3, "ffn_large.safetensors")
        out3 = tmp_path / "pt_warn"
        assert driver.run_speech(["--ssl_type", "tiny-range-test", "--wav_dir", str(wav_dir), "--save_path", str(out3), "--checkpoint", str(ck2),
                                  "--use_n_layer", "--n_layer", "-1"]) == 0
        log = capsys.readouterr().out
        assert "Failed to process" not in log and sorted(os.listdir(out3)) == ["u0.pt", "u1.pt", "u2.pt"], log
        if "within a factor 2 of the fp16 operand range" in log:
            assert log.count("WARNING: values within a factor 2") == 1
    finally:
        C._REGISTRY.pop("tiny-range-test")


def test_whisper_lora_checkpoint_through_the_driver(tmp_path, capsys):
    """preprocessing/preprocess_whisper_pretrained.py:115-190: a PEFT-wrapped Whisper (``whisper.base_model.model.*``, adapters on
    q_proj / v_proj of encoder AND decoder, a classifier head) saved with torch.save as a ``.pt`` state dict, extracted with the
    encoder only.  The HIP path merges the encoder's adapters at load; the oracle applies them un-merged."""
    from interspeech_ser_amd import config as C
    from interspeech_ser_amd import driver
    from interspeech_ser_amd.frontend import load_wav_16k, whisper_saved_rows
    from interspeech_ser_amd.weights import synthetic_state_dict
    from oracle import ssl_oracle as O
    geo = C.TINY_WHISPER
    base = synthetic_state_dict(geo, 51)
    g = torch.Generator().manual_seed(52)
    r, alpha, D = 8, 16.0, geo.hidden
    ckpt, ref_sd = {}, dict(base)
    ref_sd["lora_scale"] = torch.tensor(alpha / r)
    for k, v in base.items():
        mod, _, leaf = k.rpartition(".")
        if mod.endswith((".q_proj", ".v_proj")):
            ckpt[f"whisper.base_model.model.{mod}.base_layer.{leaf}"] = v
            if leaf == "weight":
                A, B = torch.randn(r, D, generator=g) * 0.2, torch.randn(D, r, generator=g) * 0.2
                ckpt[f"whisper.base_model.model.{mod}.lora_A.default.weight"] = A
                ckpt[f"whisper.base_model.model.{mod}.lora_B.default.weight"] = B
                ref_sd[mod + ".lora_A.weight"], ref_sd[mod + ".lora_B.weight"] = A, B
        else:
            ckpt["whisper.base_model.model." + k] = v
    # what else such a checkpoint carries: an adapted decoder layer and the fine-tuning head
    dq = "whisper.base_model.model.decoder.layers.0.self_attn.q_proj"
    ckpt[dq + ".base_layer.weight"], ckpt[dq + ".base_layer.bias"] = torch.zeros(D, D), torch.zeros(D)
    ckpt[dq + ".lora_A.default.weight"], ckpt[dq + ".lora_B.default.weight"] = torch.zeros(r, D), torch.zeros(D, r)
    ckpt["classifier.0.weight"], ckpt["classifier.0.bias"] = torch.zeros(512, D), torch.zeros(512)
    ck = tmp_path / "whisper_lora_ser.pt"
    torch.save(ckpt, str(ck))
    wav_dir, out = tmp_path / "wav", tmp_path / "pt"
    wav_dir.mkdir()
    waves = {"a.wav": synth(71, 16000), "b.wav": synth(72, 40000)}
    for n, w in waves.items():
        write_wav(wav_dir / n, w)
    C._REGISTRY["tiny-whisper-lora-test"] = geo
    try:
        rc = driver.run_whisper(["--ssl_type", "tiny-whisper-lora-test", "--wav_dir", str(wav_dir), "--save_path", str(out),
                                 "--checkpoint", str(ck), "--mode", "fp32x", "--n_layer", "-1"])
    finally:
        C._REGISTRY.pop("tiny-whisper-lora-test")
    assert rc == 0, capsys.readouterr().out
    worst = changed = 0.0
    for n in waves:
        x = load_wav_16k(str(wav_dir / n))
        mel = torch.from_numpy(O.whisper_log_mel(x, geo.n_mels))
        with torch.no_grad():
            ref = O.whisper_hidden_states(geo, ref_sd, mel)[-1]
            plain = O.whisper_hidden_states(geo, base, mel)[-1]
        rows = whisper_saved_rows(len(x), geo.hidden)
        got = torch.load(out / n.replace(".wav", ".pt"))
        assert got.shape == (rows, geo.hidden)
        worst = max(worst, float((got - ref[:rows]).abs().max() / max(1.0, float(ref.abs().max()))))
        changed = max(changed, float((plain - ref).abs().max()))
    assert changed > 0.05, changed
    assert worst < 1e-3, worst
