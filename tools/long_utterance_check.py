import numpy as np, torch, sys
sys.path.insert(0, ".")
from interspeech_ser_amd import config as C
from interspeech_ser_amd.engine import SpeechEncoder
from interspeech_ser_amd.weights import synthetic_state_dict
from oracle import ssl_oracle as O
geo = C.TINY_WAVLM
sd = synthetic_state_dict(geo, 1)
enc = SpeechEncoder(geo, sd, "cuda:0", mode="fp32x")
rng = np.random.default_rng(0)
for secs in (60, 110, 130):
    w = (0.1 * rng.standard_normal(16000 * secs)).astype(np.float32)
    try:
        hs = enc.forward(enc.upload([w, w[:50000]]), [len(w), 50000]); torch.cuda.synchronize()
        ref = O.speech_hidden_states(geo, sd, torch.from_numpy(O.zero_mean_unit_var(w)))
        err = max(float((hs.utterance(0, l).cpu() - r).abs().max() / max(1.0, float(r.abs().max()))) for l, r in enumerate(ref))
        print(secs, "s: frames", hs.frames(0), "max rel err", f"{err:.2e}")
    except Exception as e:
        print(secs, "s:", type(e).__name__, str(e)[:160])
