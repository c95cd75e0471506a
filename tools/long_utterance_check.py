"""WavLM utterances of 1-5 minutes through the tiny fixture geometry against the CPU oracle (GPU box): beyond ~2 min the attention
kernel reads the relative-position table from global memory instead of an LDS window (csrc/attention.hip, GB form) -- no length limit."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from interspeech_ser_amd import config as C
from interspeech_ser_amd.engine import SpeechEncoder
from interspeech_ser_amd.weights import synthetic_state_dict
from oracle import ssl_oracle as O
geo = C.TINY_WAVLM
sd = synthetic_state_dict(geo, 1)
rng = np.random.default_rng(0)
for mode in ("f16x", "fp32x", "bf16"):
    enc = SpeechEncoder(geo, sd, "cuda:0", mode=mode)
    for secs in (60, 110, 130, 300):
        w = (0.1 * rng.standard_normal(16000 * secs)).astype(np.float32)
        hs = enc.forward(enc.upload([w, w[:50000]]), [len(w), 50000]); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            hs = enc.forward(enc.upload([w, w[:50000]]), [len(w), 50000])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        ref = O.speech_hidden_states(geo, sd, torch.from_numpy(O.zero_mean_unit_var(w)))
        err = max(float((hs.utterance(0, l).cpu() - r).abs().max() / max(1.0, float(r.abs().max()))) for l, r in enumerate(ref))
        print(f"{mode} {secs:4d} s: frames {hs.frames(0)}, max rel err {err:.2e}, forward {dt * 1e3:.1f} ms", flush=True)
