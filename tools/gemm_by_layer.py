"""Per-launch table of one forward (GPU box): every ser_gemm of a WavLM-large group of 8 x 10 s, eager, HIP events."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import config as C
from interspeech_ser_amd.engine import build_encoder
from interspeech_ser_amd.weights import synthetic_state_dict
geo = C.geometry_for(sys.argv[1] if len(sys.argv) > 1 else "microsoft/wavlm-large")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mode = sys.argv[3] if len(sys.argv) > 3 else "bf16"
enc = build_encoder(geo, synthetic_state_dict(geo, 0), "cuda:0", mode)
rng = np.random.default_rng(1)
waves = [(0.1 * rng.standard_normal(160000)).astype(np.float32) for _ in range(B)]
lengths = [160000] * B
dev = enc.upload(waves)
for _ in range(2): enc.forward(dev, lengths)
torch.cuda.synchronize()
acc = None
for rep in range(5):
    enc.gemm_trace = []
    enc.forward(dev, lengths); torch.cuda.synchronize()
    t = [(e0.elapsed_time(e1) * 1e3, fl) for e0, e1, fl, *_ in enc.gemm_trace]
    acc = t if acc is None else [(min(a[0], b[0]), a[1]) for a, b in zip(acc, t)]
enc.gemm_trace = None
tot = sum(a[0] for a in acc)
print(f"{len(acc)} GEMM launches, {tot/1e3:.3f} ms (best of 5 each)")
for i, (us, fl) in enumerate(acc[:12]):
    print(f"launch {i:3d}: {us:8.1f} us  {fl/1e9:8.2f} GF  {fl/us/1e6:7.1f} TF/s  ({100*us/tot:4.1f} %)")
lay = acc[9:13]
print("layer 0 GEMMs (qkv, out, fc1, fc2):", " ".join(f"{us:.1f}us/{fl/us/1e6:.0f}TF" for us, fl in lay))
print("conv stack + proj + pos-conv:", f"{sum(a[0] for a in acc[:9])/1e3:.3f} ms of {tot/1e3:.3f}")
