"""The consumer of the extraction path: the reference's bimodal fusion head, its training and its evaluation loop
(next row 8f-2 / BASELINE configs[4]: "HuBERT-xlarge + RoBERTa-large bimodal extract ... feeding
train_cat_bimodal_lazy_1head.py end-to-end").

Counterpart of /root/reference/bin/train_cat_bimodal_lazy_1head.py and bin/eval_cat_bimodal_lazy_1head.py as importable
functions (the reference scripts run at import time and pull ``benchmark.utils`` -> librosa / parselmouth, neither of which
this path needs): same config keys, same label / text CSV handling, same dataset item (``<lazy dir>/<wav name>.pt`` through a
bare ``torch.load``), same ``pad_sequence`` collate, same module names (the 42 state-dict keys of the reference's
``multimodal_ser.pt``), same optimiser / schedule / losses / model-selection rule, same ``results/dev.csv``.

Host code on PyTorch-ROCm (autograd, MIOpen's GRU): the head is 12 M parameters and trains in minutes; the kernels of this
repository are on the extraction side, which writes the files this module reads.
"""
from __future__ import annotations

import csv
import json
import logging
import math
import os
import random
import time
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.utils.rnn import pad_sequence
from torch.utils.data import DataLoader, Dataset, WeightedRandomSampler

CLASSES = ["Angry", "Sad", "Happy", "Surprise", "Fear", "Disgust", "Contempt", "Neutral"]      # train script :147
CLASS_LETTERS = ["A", "S", "H", "U", "F", "D", "C", "N"]                                        # eval script :129


def set_deterministic(seed: int = 42) -> None:
    """train script :45-65"""
    os.environ["PYTHONHASHSEED"] = str(seed)
    torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = True
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    print(f"Random seed set to: {seed}")


class CosineAnnealingScheduler(torch.optim.lr_scheduler._LRScheduler):
    """closed-form cosine schedule stepped once per epoch (train script :25-43)"""

    def __init__(self, optimizer, T_max, eta_min=0.0, last_epoch=-1):
        self.T_max, self.eta_min = T_max, eta_min
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        return [self.eta_min + (base - self.eta_min) * (1 + math.cos(math.pi * self.last_epoch / self.T_max)) / 2
                for base in self.base_lrs]


class FocalLoss(nn.Module):
    """src/losses/loss.py:7-32 with the arguments the train script uses (alpha = 1, gamma = 2, mean)"""

    def __init__(self, alpha=1.0, gamma=2.0, reduction="mean", dynamic_alpha=False):
        super().__init__()
        self.alpha, self.gamma, self.reduction, self.dynamic_alpha = alpha, gamma, reduction, dynamic_alpha

    def forward(self, preds, targets):
        probs = torch.softmax(preds, dim=1)
        pt = probs[torch.arange(targets.size(0)), targets]
        ce = -torch.log(pt + 1e-8)
        alpha = (1 - pt) if self.dynamic_alpha else self.alpha
        fl = alpha * (1 - pt) ** self.gamma * ce
        return fl.mean() if self.reduction == "mean" else (fl.sum() if self.reduction == "sum" else fl)


def ce_weight_category(pred, lab, weights):
    """benchmark/utils/loss_manager.py:85-87 (``lab``: class indices in training, the float label rows in validation)"""
    return nn.CrossEntropyLoss(weight=weights)(pred, lab)


def collate_fn(batch: List[Dict]) -> Dict:
    """train script :181-207 (+ the ``utt`` list of the eval script :138-160)"""
    out = {"feat1": pad_sequence([b["feat1"] for b in batch], batch_first=True),
           "feat2": pad_sequence([b["feat2"] for b in batch], batch_first=True),
           "label": torch.stack([b["label"] for b in batch])}
    if "utt" in batch[0]:
        out["utt"] = [b["utt"] for b in batch]
    return out


class MultiLabelAudioDataset(Dataset):
    """train script :209-234: one item = the two feature files the extraction drivers wrote for a wav name"""

    def __init__(self, wav_files, labels, lazy_path1, lazy_path2, with_utt: bool = False):
        self.wav_paths, self.labels = list(wav_files), labels
        self.lazy_path1, self.lazy_path2 = lazy_path1, lazy_path2
        self.with_utt = with_utt
        self.verbose_one = True

    def __len__(self):
        return len(self.wav_paths)

    def __getitem__(self, idx):
        name = self.wav_paths[idx].replace(".wav", ".pt")
        f1, f2 = os.path.join(self.lazy_path1, name), os.path.join(self.lazy_path2, name)
        if self.verbose_one:
            print(f1, f2)
            self.verbose_one = False
        item = {"feat1": torch.load(f1), "feat2": torch.load(f2),
                "label": torch.tensor(self.labels[idx], dtype=torch.float)}
        if self.with_utt:
            item["utt"] = self.wav_paths[idx]
        return item


class MultiModalEmotionClassifier(nn.Module):
    """train script :236-334.  Attribute names are the state-dict keys of the reference's checkpoints."""

    def __init__(self, features1_dim=1024, features2_dim=768, fusion_hidden_dim=512, num_emotions=8, dropout=0.5):
        super().__init__()
        h = fusion_hidden_dim
        self.speech_projection = nn.Linear(features1_dim, h)
        self.text_projection = nn.Linear(features2_dim, h)
        self.speech_norm = nn.LayerNorm(h)
        self.text_norm = nn.LayerNorm(h)
        self.speech_gru = nn.GRU(h, h, batch_first=True, bidirectional=True)
        self.text_gru = nn.GRU(h, h, batch_first=True, bidirectional=True)
        self.speech_attention = nn.MultiheadAttention(h * 2, 1, dropout=dropout, batch_first=True)
        self.text_attention = nn.MultiheadAttention(h * 2, 1, dropout=dropout, batch_first=True)
        self.speech_attn = nn.Linear(h * 2, 1)
        self.text_attn = nn.Linear(h * 2, 1)
        self.classifier = nn.Sequential(nn.Linear(h * 4, h), nn.ReLU(), nn.Dropout(dropout), nn.Linear(h, num_emotions))
        self.layer_norm = nn.LayerNorm(h * 4)

    @staticmethod
    def attention_pool(features, attention_layer):
        w = F.softmax(attention_layer(features), dim=1)              # [batch, seq, 1]; padded frames take part, as in the reference
        return (features * w).sum(dim=1)

    def forward(self, features1, features2):
        speech = self.speech_norm(self.speech_projection(features1))
        text = self.text_norm(self.text_projection(features2))
        speech_hidden, _ = self.speech_gru(speech)
        text_hidden, _ = self.text_gru(text)
        speech_att, _ = self.speech_attention(speech_hidden, text_hidden, text_hidden)
        text_att, _ = self.text_attention(text_hidden, speech_hidden, speech_hidden)
        speech_pooled = self.attention_pool(speech_hidden + speech_att, self.speech_attn)
        text_pooled = self.attention_pool(text_hidden + text_att, self.text_attn)
        return self.classifier(self.layer_norm(torch.cat([speech_pooled, text_pooled], dim=-1)))


def macro_f1(labels: Sequence[int], preds: Sequence[int]) -> float:
    """sklearn.metrics.f1_score(labels, preds, average='macro'): mean F1 over the classes present in labels or preds
    (a class with no true and no predicted sample is left out; 0/0 counts as 0)."""
    labels, preds = np.asarray(labels, dtype=np.int64), np.asarray(preds, dtype=np.int64)
    scores = []
    for c in np.union1d(labels, preds):
        tp = float(np.sum((preds == c) & (labels == c)))
        fp = float(np.sum((preds == c) & (labels != c)))
        fn = float(np.sum((preds != c) & (labels == c)))
        scores.append(0.0 if 2 * tp + fp + fn == 0 else 2 * tp / (2 * tp + fp + fn))
    return float(np.mean(scores)) if scores else 0.0


def _class_weights(df, device) -> torch.Tensor:
    """total / (n_classes * frequency), 0 for an absent class (train script :150-161)"""
    freq = df[CLASSES].sum().to_dict()
    total = len(df)
    return torch.tensor([total / (len(CLASSES) * freq[c]) if freq[c] != 0 else 0 for c in CLASSES], device=device, dtype=torch.float)


def _logger(model_path: str) -> logging.Logger:
    log = logging.getLogger(f"ser_head.{model_path}.{time.time()}")
    log.setLevel(logging.INFO)
    log.propagate = False
    fmt = logging.Formatter("%(asctime)s - %(levelname)s - %(message)s")
    for h in (logging.FileHandler(os.path.join(model_path, "%s-%d.log" % ("loggingtxt", time.time()))), logging.StreamHandler()):
        h.setFormatter(fmt)
        log.addHandler(h)
    return log


def _frames(config: Dict):
    import pandas as pd
    label_df, text_df = pd.read_csv(config["label_path"]), pd.read_csv(config["txt_dir"])
    return label_df.merge(text_df, on="FileName", how="left")


def _device(name: Optional[str]) -> torch.device:
    return torch.device(name) if name else torch.device("cuda" if torch.cuda.is_available() else "cpu")


def _model(config: Dict, device) -> MultiModalEmotionClassifier:
    return MultiModalEmotionClassifier(features1_dim=config["feat1_dim"], features2_dim=config["feat2_dim"],
                                       fusion_hidden_dim=512, num_emotions=8, dropout=0.5).to(device)


def _validate(model, loader, device):
    """one pass over the Development split (train script :447-476, eval script :310-341)"""
    model.eval()
    logits_all, labels_all, preds, gold, utts = [], [], [], [], []
    for batch in loader:
        x1, x2 = batch["feat1"].to(device), batch["feat2"].to(device)
        labels = batch["label"].to(device)
        with torch.no_grad():
            logits = model(x1, x2)
        logits_all.append(logits)
        labels_all.append(labels)
        preds.extend(torch.argmax(logits, dim=1).cpu().numpy())
        gold.extend(batch["label"].max(dim=1)[1].numpy())
        utts.extend(batch.get("utt", []))
    return torch.cat(logits_all, 0), torch.cat(labels_all, 0), preds, gold, utts


def train(config: Dict, seed: int = 7, device: Optional[str] = None) -> Dict:
    """bin/train_cat_bimodal_lazy_1head.py as a function.  Returns {"best_f1", "best_epoch", "model_file", "history"}."""
    set_deterministic(seed)
    dev = _device(device)
    batch_size, accum = config["batch_size"], config["accum_step"]
    assert accum > 0 and batch_size % accum == 0
    epochs, lr, model_path = config["epochs"], config["lr"], config["model_path"]
    os.makedirs(model_path, exist_ok=True)
    balanced = bool(config.get("use_balanced_batch", False))
    focal = bool(config.get("use_focalloss", False))
    log = _logger(model_path)
    log.info(f"Starting an Lazy OwnSermodel wavlm-based experiment in model path = {model_path}")
    log.info(f"Using LR = {lr} Epochs = {epochs} Batch size = {batch_size} Accum steps = {accum}")
    log.info(f"Using balanced batch = {balanced}")
    log.info(f"Using focalloss = {focal}")

    df = _frames(config)
    train_df, val_df = df[df["Split_Set"] == "Train"], df[df["Split_Set"] == "Development"]
    w_train, w_val = _class_weights(train_df, dev), _class_weights(val_df, dev)
    log.info(f"Class weights: {w_train}")
    train_ds = MultiLabelAudioDataset(train_df["FileName"].tolist(), train_df[CLASSES].values, config["lazy_dir1"], config["lazy_dir2"])
    val_ds = MultiLabelAudioDataset(val_df["FileName"].tolist(), val_df[CLASSES].values, config["lazy_dir1"], config["lazy_dir2"])
    if balanced:                                                       # train script :340-361
        log.info("Using balanced batch. Computing sample weights...")
        freq = train_df[CLASSES].sum().to_dict()
        cw = {c: 1 / f if f != 0 else 0 for c, f in freq.items()}
        factor = len(cw) / sum(cw.values())
        cw = {c: w * factor for c, w in cw.items()}
        sample_w = [cw[train_df[CLASSES].iloc[i].idxmax()] for i in range(len(train_df))]
        sampler = WeightedRandomSampler(weights=sample_w, num_samples=len(train_ds), replacement=True)
        train_loader = DataLoader(train_ds, batch_size=batch_size, sampler=sampler, collate_fn=collate_fn)
    else:
        train_loader = DataLoader(train_ds, batch_size=batch_size, shuffle=True, collate_fn=collate_fn)
    val_loader = DataLoader(val_ds, batch_size=batch_size, collate_fn=collate_fn)

    model = _model(config, dev)
    optimizer = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=1e-6)
    scheduler = CosineAnnealingScheduler(optimizer, T_max=epochs, eta_min=1e-6)
    focal_loss = FocalLoss(alpha=1, gamma=2, reduction="mean", dynamic_alpha=False)
    best = {"best_f1": 0.0, "best_epoch": 0, "best_loss": 1e10, "model_file": os.path.join(model_path, "multimodal_ser.pt"), "history": []}
    log.info("Starting training...")
    for epoch in range(epochs):
        print("Epoch: ", epoch)
        model.train()
        n_batches = len(train_loader)
        for cnt, batch in enumerate(train_loader):
            x1, x2 = batch["feat1"].to(dev), batch["feat2"].to(dev)
            y = batch["label"].max(dim=1)[1].to(dev).long()
            optimizer.zero_grad()                                      # per batch, as the reference does (:413)
            logits = model(x1, x2)
            loss = ce_weight_category(logits, y, None if balanced else w_train)
            total = (focal_loss(logits, y) if focal else loss) / accum
            total.backward()
            if (cnt + 1) % accum == 0 or (cnt + 1) == n_batches:
                optimizer.step()
            if (cnt + 2) % 200 == 0:
                log.info(f"Epoch ({epoch + 1}/{epochs})| step = {cnt + 1}: loss = {loss} current lr = {scheduler.get_last_lr()[0]}")
        scheduler.step()
        logits_all, labels_all, preds, gold, _ = _validate(model, val_loader, dev)
        dev_loss = ce_weight_category(logits_all, labels_all, w_val)
        f1 = macro_f1(gold, preds)
        log.info(f"|VALIDATION| Epoch ({epoch + 1}/{epochs}): eval_loss = {dev_loss} eval f1 = {f1}")
        best["history"].append({"epoch": epoch + 1, "eval_loss": float(dev_loss), "eval_f1": f1})
        if best["best_f1"] < f1:
            log.info(f"New best model at epoch {epoch + 1}")
            best.update(best_f1=f1, best_epoch=epoch, best_loss=float(dev_loss))
            print("Save", epoch)
            print("Loss", float(dev_loss))
            torch.save(model.state_dict(), best["model_file"])
    for h in list(log.handlers):
        h.close()
        log.removeHandler(h)
    return best


def evaluate(config: Dict, seed: int = 7, device: Optional[str] = None) -> Dict:
    """bin/eval_cat_bimodal_lazy_1head.py as a function: Development split through ``multimodal_ser.pt``, macro-F1,
    ``<model_path>/results/dev.csv`` (Filename, Prediction letter, the 8 logits as class_i_prob)."""
    set_deterministic(seed)
    dev = _device(device)
    model_path = config["model_path"]
    os.makedirs(model_path, exist_ok=True)
    log = _logger(model_path)
    df = _frames(config)
    val_df = df[df["Split_Set"] == "Development"]
    val_ds = MultiLabelAudioDataset(val_df["FileName"].tolist(), val_df[CLASSES].values, config["lazy_dir1"], config["lazy_dir2"], with_utt=True)
    val_loader = DataLoader(val_ds, batch_size=config["batch_size"], collate_fn=collate_fn)
    model = _model(config, dev)
    model.load_state_dict(torch.load(os.path.join(model_path, "multimodal_ser.pt"), map_location=dev), strict=False)
    log.info("Starting evaluation...")
    logits_all, labels_all, preds, gold, utts = _validate(model, val_loader, dev)
    loss = ce_weight_category(logits_all, labels_all, None)
    f1 = macro_f1(gold, preds)
    log.info(f"|Metrics| eval_loss = {loss} eval f1 = {f1}")
    os.makedirs(os.path.join(model_path, "results"), exist_ok=True)
    csv_file = os.path.join(model_path, "results", "dev.csv")
    with open(csv_file, mode="w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Filename", "Prediction"] + [f"class_{i}_prob" for i in range(len(CLASSES))])
        for row, utt in zip(logits_all.cpu().numpy(), utts):
            w.writerow([utt, CLASS_LETTERS[int(np.argmax(row))]] + [f"{p:.4f}" for p in row.flatten()])
    for h in list(log.handlers):
        h.close()
        log.removeHandler(h)
    return {"eval_loss": float(loss), "eval_f1": f1, "csv": csv_file, "n": len(utts)}


def main(argv: Optional[Sequence[str]] = None, evaluate_only: bool = False) -> int:
    import argparse
    p = argparse.ArgumentParser()
    p.add_argument("--seed", type=int, default=7)
    p.add_argument("--config_path", type=str, default="./configs/config_cat.json")
    args = p.parse_args(argv)
    with open(args.config_path, "r") as f:
        config = json.load(f)
    (evaluate if evaluate_only else train)(config, seed=args.seed)
    return 0
