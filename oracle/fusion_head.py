"""CONSUMER OF THE PATH -- TEST INFRASTRUCTURE ONLY.

Restatement of what the reference's bimodal fusion head does with the feature files the extraction path writes
(bin/train_cat_bimodal_lazy_1head.py), module names, dimensions and arithmetic unchanged, so that "the downstream heads
consume the output unchanged" is an executable statement:

  * ``dataset_item``       MultiLabelAudioDataset.__getitem__ (:220-234): <lazy dir>/<wav name with .wav -> .pt>,
                           bare ``torch.load`` (no map_location), label as a float tensor;
  * ``collate_fn``         :181-207: ``pad_sequence(batch_first=True)`` over the per-utterance [T, D] tensors;
  * ``MultiModalEmotionClassifier``  :236-334: per-modality Linear(feat_dim, 512) -> LayerNorm -> bidirectional GRU(512)
                           -> single-head cross attention both ways (embed 1024, dropout inactive in eval) -> residual ->
                           softmax attention pooling -> concat -> LayerNorm(2048) -> Linear(2048, 512) -> ReLU ->
                           Dropout -> Linear(512, 8).

PARITY PIN: oracle/make_head_golden.py executes the reference's own class definition (taken from its source file with
``ast``; the script around it cannot be imported -- it trains at import time and needs packages this image lacks) on
seeded weights and a fixed synthetic batch and commits tests/golden/fusion_head_pins.npz: the state-dict keys + shapes
and the logits.  tests/test_consumer_contract.py replays them against this file.  Training / evaluation of the head is
out of scope; only the product's drivers write the files it reads.
"""
from __future__ import annotations

import os
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.utils.rnn import pad_sequence


def dataset_item(wav_name: str, lazy_path1: str, lazy_path2: str, label: Sequence[float]) -> Dict[str, torch.Tensor]:
    feat1 = torch.load(os.path.join(lazy_path1, wav_name.replace(".wav", ".pt")))
    feat2 = torch.load(os.path.join(lazy_path2, wav_name.replace(".wav", ".pt")))
    return {"feat1": feat1, "feat2": feat2, "label": torch.tensor(label, dtype=torch.float)}


def collate_fn(batch: List[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    return {"feat1": pad_sequence([b["feat1"] for b in batch], batch_first=True),
            "feat2": pad_sequence([b["feat2"] for b in batch], batch_first=True),
            "label": torch.stack([b["label"] for b in batch])}


class MultiModalEmotionClassifier(nn.Module):
    def __init__(self, features1_dim=1024, features2_dim=768, fusion_hidden_dim=512, num_emotions=8, dropout=0.5):
        super().__init__()
        h = fusion_hidden_dim
        self.speech_projection = nn.Linear(features1_dim, h)
        self.text_projection = nn.Linear(features2_dim, h)
        self.speech_norm = nn.LayerNorm(h)
        self.text_norm = nn.LayerNorm(h)
        self.speech_gru = nn.GRU(h, h, batch_first=True, bidirectional=True)
        self.text_gru = nn.GRU(h, h, batch_first=True, bidirectional=True)
        self.speech_attention = nn.MultiheadAttention(2 * h, 1, dropout=dropout, batch_first=True)
        self.text_attention = nn.MultiheadAttention(2 * h, 1, dropout=dropout, batch_first=True)
        self.speech_attn = nn.Linear(2 * h, 1)
        self.text_attn = nn.Linear(2 * h, 1)
        self.classifier = nn.Sequential(nn.Linear(4 * h, h), nn.ReLU(), nn.Dropout(dropout), nn.Linear(h, num_emotions))
        self.layer_norm = nn.LayerNorm(4 * h)

    @staticmethod
    def attention_pool(features: torch.Tensor, attention_layer: nn.Module) -> torch.Tensor:
        weights = F.softmax(attention_layer(features), dim=1)          # over the (padded) sequence axis, as the reference does
        return (features * weights).sum(dim=1)

    def forward(self, features1: torch.Tensor, features2: torch.Tensor) -> torch.Tensor:
        speech = self.speech_norm(self.speech_projection(features1))
        text = self.text_norm(self.text_projection(features2))
        speech_hidden, _ = self.speech_gru(speech)
        text_hidden, _ = self.text_gru(text)
        speech_attended, _ = self.speech_attention(speech_hidden, text_hidden, text_hidden)
        text_attended, _ = self.text_attention(text_hidden, speech_hidden, speech_hidden)
        pooled = torch.cat([self.attention_pool(speech_hidden + speech_attended, self.speech_attn),
                            self.attention_pool(text_hidden + text_attended, self.text_attn)], dim=-1)
        return self.classifier(self.layer_norm(pooled))


def seeded_head_weights(shapes: Dict[str, Sequence[int]], seed: int) -> Dict[str, torch.Tensor]:
    """Deterministic weights for a head with the given state-dict layout (numpy PCG64: same numbers on every host).
    Too many parameters (16 M) to commit as a fixture, so fixture and test both regenerate them from the seed."""
    g = np.random.default_rng(seed)
    out = {}
    for k in sorted(shapes):
        shape = tuple(int(s) for s in shapes[k])
        x = g.standard_normal(size=shape, dtype=np.float32)
        if len(shape) >= 2:
            x *= 1.0 / np.sqrt(shape[-1])
        elif "norm" in k and k.endswith("weight"):
            x = 1.0 + 0.1 * x
        else:
            x *= 0.05
        out[k] = torch.from_numpy(x)
    return out


def synthetic_batch(feat1_dim: int, feat2_dim: int, seed: int):
    """Fixed ragged batch in the shapes the drivers write: speech [T, feat1_dim] with T in {149, 499, 37}, text [80, feat2_dim]."""
    g = np.random.default_rng(seed)
    items = []
    for t in (149, 499, 37):
        items.append({"feat1": torch.from_numpy(g.standard_normal((t, feat1_dim), dtype=np.float32)),
                      "feat2": torch.from_numpy(g.standard_normal((80, feat2_dim), dtype=np.float32)),
                      "label": torch.zeros(8)})
    return collate_fn(items)
