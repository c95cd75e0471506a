"""CPU: the on-disk features are consumable by the reference's downstream head exactly as it reads them
(SURVEY 8f-2).  The consumer below is a compact restatement -- test infrastructure, not product -- of what
bin/train_cat_bimodal_lazy_1head.py does with the files: `MultiLabelAudioDataset.__getitem__` (:220-234: file name =
wav name with .wav -> .pt under each lazy dir, bare `torch.load`), `collate_fn` (:181-207: `pad_sequence(batch_first=True)`
over the per-utterance [T, D] tensors) and `MultiModalEmotionClassifier.forward` (:236-334: per-modality Linear + LayerNorm,
bidirectional GRU, single-head cross attention both ways, softmax attention pooling, LayerNorm, 2-layer classifier -> 8)."""
import os

import numpy as np
import torch
from torch import nn
from torch.nn.utils.rnn import pad_sequence


class _Head(nn.Module):
    def __init__(self, d_speech, d_text, h=32, classes=8):
        super().__init__()
        self.proj = nn.ModuleList([nn.Linear(d_speech, h), nn.Linear(d_text, h)])
        self.norm = nn.ModuleList([nn.LayerNorm(h), nn.LayerNorm(h)])
        self.gru = nn.ModuleList([nn.GRU(h, h, batch_first=True, bidirectional=True) for _ in range(2)])
        self.cross = nn.ModuleList([nn.MultiheadAttention(2 * h, 1, batch_first=True) for _ in range(2)])
        self.pool = nn.ModuleList([nn.Linear(2 * h, 1), nn.Linear(2 * h, 1)])
        self.out_norm = nn.LayerNorm(4 * h)
        self.classifier = nn.Sequential(nn.Linear(4 * h, h), nn.ReLU(), nn.Linear(h, classes))

    def forward(self, speech, text):
        hid = [self.gru[i](self.norm[i](self.proj[i](x)))[0] for i, x in enumerate((speech, text))]
        att = [self.cross[0](hid[0], hid[1], hid[1])[0], self.cross[1](hid[1], hid[0], hid[0])[0]]
        pooled = []
        for i in range(2):
            f = hid[i] + att[i]
            w = torch.softmax(self.pool[i](f), dim=1)
            pooled.append((f * w).sum(dim=1))
        return self.classifier(self.out_norm(torch.cat(pooled, dim=-1)))


def _item(wav_name, lazy1, lazy2, label):
    f1 = torch.load(os.path.join(lazy1, wav_name.replace(".wav", ".pt")))
    f2 = torch.load(os.path.join(lazy2, wav_name.replace(".wav", ".pt")))
    return {"feat1": f1, "feat2": f2, "label": torch.tensor(label, dtype=torch.float)}


def _collate(batch):
    return {"feat1": pad_sequence([b["feat1"] for b in batch], batch_first=True),
            "feat2": pad_sequence([b["feat2"] for b in batch], batch_first=True),
            "label": torch.stack([b["label"] for b in batch])}


def test_saved_features_feed_the_downstream_head(tmp_path):
    from interspeech_ser_amd.frontend import feature_path, save_feature
    speech_dir, text_dir = tmp_path / "wavlm", tmp_path / "roberta"
    speech_dir.mkdir(); text_dir.mkdir()
    rng = np.random.default_rng(0)
    names, frames = [f"MSP-PODCAST_{i:04d}.wav" for i in range(5)], [149, 499, 37, 250, 1]
    for name, t in zip(names, frames):
        # what the speech driver writes: [T, 1024] fp32 for <wav_dir>/<name>; what the text driver writes: [80, 1024]
        save_feature(torch.from_numpy(rng.standard_normal((t, 1024)).astype(np.float32)), feature_path(str(speech_dir), "/corpus/Audios/" + name))
        save_feature(torch.from_numpy(rng.standard_normal((80, 1024)).astype(np.float32)), feature_path(str(text_dir), name))
    labels = np.eye(8, dtype=np.float32)[rng.integers(0, 8, size=5)]
    batch = _collate([_item(n, str(speech_dir), str(text_dir), lab) for n, lab in zip(names, labels)])
    assert batch["feat1"].shape == (5, 499, 1024) and batch["feat2"].shape == (5, 80, 1024)
    assert batch["feat1"].dtype == torch.float32 and batch["feat1"].device.type == "cpu"
    assert torch.equal(batch["feat1"][4, 1:], torch.zeros(498, 1024))            # the one-frame utterance is padded, not broken
    torch.manual_seed(0)
    with torch.no_grad():
        logits = _Head(1024, 1024).eval()(batch["feat1"], batch["feat2"])
    assert logits.shape == (5, 8) and torch.isfinite(logits).all()
    loss = nn.CrossEntropyLoss()(logits, batch["label"].argmax(dim=1))
    assert torch.isfinite(loss)
