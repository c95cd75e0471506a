"""Randomised ragged-batch shake-out (GPU box, a few minutes): tiny geometries of the three speech families, random batch sizes
and utterance lengths through ONE long-lived encoder per (family, mode) -- arenas grow and shrink, command lists are patched --
checking (1) batched == batch-of-one bit for bit on a sampled utterance, (2) the command-list path == the launch-by-launch path,
(3) f16x / fp32x / f16a / f16q / f16 / bf16 within their tolerances of the CPU oracle on a sampled utterance."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import config as C
from interspeech_ser_amd.engine import SpeechEncoder
from interspeech_ser_amd.weights import synthetic_state_dict
from oracle import ssl_oracle as O
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
TOL = {"f16x": 1e-3, "fp32x": 1e-3, "f16a": 1e-3, "f16q": 1e-3, "f16": 1e-3, "bf16": 3e-2}
MODES = tuple(TOL)
fams = [("wavlm", C.TINY_WAVLM), ("hubert", C.TINY_HUBERT), ("wav2vec2", C.TINY_WAV2VEC2)]
encs = {}
t_end = time.time() + budget
n = 0
worst = {m: 0.0 for m in TOL}
while time.time() < t_end:
    fam, geo = fams[int(rng.integers(len(fams)))]
    mode = MODES[int(rng.integers(len(MODES)))]
    key = (fam, mode)
    if key not in encs:
        sd = synthetic_state_dict(geo, 100 + len(encs))
        e1, e2 = SpeechEncoder(geo, sd, "cuda:0", mode), SpeechEncoder(geo, sd, "cuda:0", mode)
        e2.use_tape = False
        encs[key] = (sd, e1, e2)
    sd, taped, eager = encs[key]
    B = int(rng.integers(1, 20))
    kind = rng.integers(4)
    lens = [int(x) for x in (rng.integers(400, 2000, B) if kind == 0 else rng.integers(400, 60000, B) if kind == 1
                             else rng.integers(100000, 200000, B) if kind == 2 else rng.choice([400, 719, 720, 401, 16000, 33333], B))]
    waves = [(0.1 * rng.standard_normal(L) + 0.2 * np.sin(np.arange(L) * 0.05)).astype(np.float32) for L in lens]
    a = taped.forward(taped.upload(waves), lens)
    b = eager.forward(eager.upload(waves), lens)
    torch.cuda.synchronize()
    assert a.frame_offs == b.frame_offs and torch.equal(a.states, b.states), ("tape != eager", key, lens)
    if rng.integers(4) == 0:                                  # early exit: the states it leaves equal the full forward's, bit for bit
        nstop = int(rng.integers(geo.num_layers + 1))
        full = a.states.clone()
        part = taped.forward(taped.upload(waves), lens, last_state=nstop)
        torch.cuda.synchronize()
        assert torch.equal(part.states[: nstop + 1], full[: nstop + 1]), ("early exit", key, lens, nstop)
        a = taped.forward(taped.upload(waves), lens)
        torch.cuda.synchronize()
    assert bool(torch.isfinite(a.states).all()), ("non-finite", key, lens)
    j = int(rng.integers(B))
    keep = [a.utterance(j, l).clone() for l in range(len(a))]
    one = taped.forward(taped.upload([waves[j]]), [lens[j]])
    torch.cuda.synchronize()
    for l in range(len(one)):
        assert torch.equal(one.utterance(0, l), keep[l]), ("batched != single", key, lens, j, l)
    with torch.no_grad():
        ref = O.speech_hidden_states(geo, sd, torch.from_numpy(O.zero_mean_unit_var(waves[j])))
    err = max(float((k.cpu() - r).abs().max() / max(1.0, float(r.abs().max()))) for k, r in zip(keep, ref))
    assert err < TOL[mode], ("oracle", key, lens, j, err)
    worst[mode] = max(worst[mode], err)
    n += 1
print(f"{n} random ragged batches ok; worst vs oracle: " + ", ".join(f"{m} {v:.2e}" for m, v in worst.items()))
