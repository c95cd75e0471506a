"""CPU: the on-disk features are consumable by the reference's downstream bimodal head exactly as it reads them
(SURVEY 8f-2).  The consumer is oracle/fusion_head.py -- test infrastructure, not product: the reference's
``MultiLabelAudioDataset.__getitem__`` / ``collate_fn`` / ``MultiModalEmotionClassifier``
(bin/train_cat_bimodal_lazy_1head.py:181-334) restated with its module names and real dimensions, and PINNED to the
reference's own class definition by tests/golden/fusion_head_pins.npz (oracle/make_head_golden.py): identical
state-dict keys and shapes, identical logits on seeded weights and a fixed batch."""
import os

import numpy as np
import torch
from torch import nn

from oracle import fusion_head as H


def _pins(golden_dir):
    g = np.load(os.path.join(golden_dir, "fusion_head_pins.npz"))
    shapes = {str(k): tuple(int(x) for x in str(s).split(",")) for k, s in zip(g["keys"], g["shapes"])}
    return g, shapes


def test_restated_head_has_the_reference_state_dict_layout(golden_dir):
    g, shapes = _pins(golden_dir)
    head = H.MultiModalEmotionClassifier(int(g["feat1_dim"]), int(g["feat2_dim"]), 512, 8, 0.5)
    sd = head.state_dict()
    assert list(sd.keys()) == list(shapes.keys())                       # a checkpoint of the reference head loads by name
    assert all(tuple(sd[k].shape) == shapes[k] for k in shapes)
    assert sd["speech_projection.weight"].shape == (512, 1280) and sd["text_projection.weight"].shape == (512, 1024)
    assert sd["speech_attention.in_proj_weight"].shape == (3 * 1024, 1024) and sd["layer_norm.weight"].shape == (2048,)
    assert sd["classifier.3.weight"].shape == (8, 512)


def test_restated_head_reproduces_the_reference_logits(golden_dir):
    g, shapes = _pins(golden_dir)
    weights = H.seeded_head_weights(shapes, int(g["seed_weights"]))
    assert abs(sum(float(v.double().sum()) for v in weights.values()) - float(g["weight_digest"])) < 1e-6
    head = H.MultiModalEmotionClassifier(int(g["feat1_dim"]), int(g["feat2_dim"]), 512, 8, 0.5).eval()
    head.load_state_dict(weights, strict=True)
    batch = H.synthetic_batch(int(g["feat1_dim"]), int(g["feat2_dim"]), int(g["seed_batch"]))
    assert batch["feat1"].shape == (3, 499, 1280) and batch["feat2"].shape == (3, 80, 1024)
    with torch.no_grad():
        logits = head(batch["feat1"], batch["feat2"])
    assert logits.shape == (3, 8)
    assert float((logits - torch.from_numpy(g["logits"])).abs().max()) < 1e-4


def test_saved_features_feed_the_downstream_head(tmp_path):
    """Files written by the product's writer (frontend.save_feature -> libserhip's ser_pt_write_f32) go through the
    reference's dataset item -> collate -> classifier -> loss chain."""
    from interspeech_ser_amd.frontend import feature_path, save_feature
    speech_dir, text_dir = tmp_path / "hubert", tmp_path / "roberta"
    speech_dir.mkdir()
    text_dir.mkdir()
    rng = np.random.default_rng(0)
    names, frames = [f"MSP-PODCAST_{i:04d}.wav" for i in range(5)], [149, 499, 37, 250, 1]
    for name, t in zip(names, frames):
        # what the speech driver writes: [T, 1280] fp32 for <wav_dir>/<name>; what the text driver writes: [80, 1024]
        save_feature(torch.from_numpy(rng.standard_normal((t, 1280)).astype(np.float32)), feature_path(str(speech_dir), "/corpus/Audios/" + name))
        save_feature(torch.from_numpy(rng.standard_normal((80, 1024)).astype(np.float32)), feature_path(str(text_dir), name))
    labels = np.eye(8, dtype=np.float32)[rng.integers(0, 8, size=5)]
    batch = H.collate_fn([H.dataset_item(n, str(speech_dir), str(text_dir), lab) for n, lab in zip(names, labels)])
    assert batch["feat1"].shape == (5, 499, 1280) and batch["feat2"].shape == (5, 80, 1024)
    assert batch["feat1"].dtype == torch.float32 and batch["feat1"].device.type == "cpu"
    assert torch.equal(batch["feat1"][4, 1:], torch.zeros(498, 1280))            # the one-frame utterance is padded, not broken
    torch.manual_seed(0)
    with torch.no_grad():
        logits = H.MultiModalEmotionClassifier(1280, 1024).eval()(batch["feat1"], batch["feat2"])
    assert logits.shape == (5, 8) and torch.isfinite(logits).all()
    y = batch["label"].max(dim=1)[1].long()                                       # the reference's target (:400)
    loss = nn.CrossEntropyLoss()(logits, y)
    assert torch.isfinite(loss)
