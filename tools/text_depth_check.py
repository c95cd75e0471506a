"""Text encoders at full depth under sharp attention (GPU box): roberta-large (24 post-LN layers) and deberta-v3-large with the query / key
projections x F, 4 texts of 80 / 61 / 23 / 5 valid tokens padded to 80, all L + 1 states of every row against the CPU oracle -- the margin of
the text drivers' modes (default fp32x; f16x) at the depth they ship at, like tests/depth_envelope.py for the speech encoders.
    python tools/text_depth_check.py [F ...]      (default factors 1 2)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import config as C
from interspeech_ser_amd.engine import build_encoder
from interspeech_ser_amd.weights import synthetic_state_dict
from oracle import ssl_oracle as O
rel = lambda a, b: float((a.double() - b.double()).abs().max() / max(1.0, float(b.abs().max())))
factors = [float(x) for x in sys.argv[1:]] or [1.0, 2.0]
T, lens = 80, [80, 61, 23, 5]
for name in ("roberta-large", "microsoft/deberta-v3-large"):
    geo = C.geometry_for(name)
    base = synthetic_state_dict(geo, 0)
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(3, geo.vocab_size, (len(lens), T), generator=g)
    mask = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).long()
    ids[mask == 0] = geo.pad_token_id
    for f in factors:
        sd = {k: v.clone() for k, v in base.items()}
        for k in sd:
            if ("attention.self.query" in k or "attention.self.key" in k or "query_proj" in k or "key_proj" in k) and "layer." in k:
                sd[k] *= f
        oracle = O.roberta_hidden_states if geo.family == "roberta" else O.deberta_hidden_states
        with torch.no_grad():
            ref = [oracle(geo, sd, ids[b], mask[b]) for b in range(len(lens))]
        row = []
        for mode in ("f16x", "fp32x", "bf16"):
            enc = build_encoder(geo, sd, "cuda:0", mode)
            hs = enc.forward(ids, mask)
            torch.cuda.synchronize()
            per = [max(rel(hs.utterance(b, l).cpu(), ref[b][l]) for b in range(len(lens))) for l in range(geo.num_layers + 1)]
            row.append(f"{mode} worst {max(per):.2e} (states 0/8/16/24: " + " ".join(f"{per[i]:.1e}" for i in (0, 8, 16, 24)) + ")")
            del enc
            torch.cuda.empty_cache()
        print(f"{name} ({geo.num_layers} layers), q, k x {f:g}: " + "; ".join(row), flush=True)
