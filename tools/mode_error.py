"""bf16 mode against fp32x mode on the full-size geometry (GPU box): per-state error of the headline numerics mode
relative to the parity-grade one (which tests pin to the fp32 oracle within 1e-3)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import config as C
from interspeech_ser_amd.engine import build_encoder
from interspeech_ser_amd.weights import synthetic_state_dict
geo = C.geometry_for(sys.argv[1] if len(sys.argv) > 1 else "microsoft/wavlm-large")
sd = synthetic_state_dict(geo, 0)
rng = np.random.default_rng(3)
t = np.arange(160000) / 16000.0
waves = [(0.1 * rng.standard_normal(n) + 0.2 * np.sin(2 * np.pi * 220 * t[:n])).astype(np.float32) for n in (160000, 73211)]
lens = [len(w) for w in waves]
out = {}
for mode in ("fp32x", "bf16"):
    enc = build_encoder(geo, sd, "cuda:0", mode)
    hs = enc.forward(enc.upload(waves), lens)
    torch.cuda.synchronize()
    out[mode] = hs.states.double().cpu()
    del enc
a, b = out["fp32x"], out["bf16"]
for l in (0, 1, 6, 12, 18, geo.num_layers - 1, geo.num_layers):
    d = (a[l] - b[l]).abs()
    scale = a[l].abs().max().clamp_min(1.0)
    cos = torch.nn.functional.cosine_similarity(a[l], b[l], dim=1)
    print(f"state {l:2d}: max|d|/max|x| {float(d.max()/scale):.2e}  rms(d)/rms(x) {float(d.pow(2).mean().sqrt()/a[l].pow(2).mean().sqrt()):.2e}  min row cosine {float(cos.min()):.6f}")
