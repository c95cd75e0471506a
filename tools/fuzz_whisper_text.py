"""Randomised shake-out of the Whisper and text encoders (GPU box): random batch sizes / clip lengths / token counts through
long-lived tiny encoders; batched == batch-of-one bit for bit, command list == launch by launch (Whisper), fp32x within 1e-3 and
bf16 within 3e-2 of the CPU oracle."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import config as C
from interspeech_ser_amd.engine import build_encoder
from interspeech_ser_amd.frontend import whisper_saved_rows
from interspeech_ser_amd.weights import synthetic_state_dict
from oracle import ssl_oracle as O
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
TOL = {"fp32x": 1e-3, "f16": 1e-3, "f16a": 1e-3, "f16q": 1e-3, "f16x": 1e-3, "bf16": 3e-2}
rel = lambda a, b: float((a - b).abs().max() / max(1.0, float(b.abs().max())))
encs, worst, n = {}, {}, 0
t_end = time.time() + budget
while time.time() < t_end:
    fam = ("whisper", "roberta", "deberta")[int(rng.integers(3))]
    mode = ("fp32x", "bf16", "f16x", "f16", "f16a", "f16q")[int(rng.integers(6 if fam == "whisper" else 3))]
    geo = {"whisper": C.TINY_WHISPER, "roberta": C.TINY_ROBERTA, "deberta": C.TINY_DEBERTA}[fam]
    key = (fam, mode)
    if key not in encs:
        sd = synthetic_state_dict(geo, 200 + len(encs))
        encs[key] = (sd, build_encoder(geo, sd, "cuda:0", mode))
    sd, enc = encs[key]
    if fam == "whisper":
        B = int(rng.integers(1, 5))
        lens = [int(x) for x in rng.integers(1000, 500000, B)]
        waves = [(0.1 * rng.standard_normal(L)).astype(np.float32) for L in lens]
        hs = enc.forward(enc.upload(waves), lens)
        torch.cuda.synchronize()
        j = int(rng.integers(B))
        keep = hs.states[:, hs.frame_offs[j]:hs.frame_offs[j + 1]].clone()
        enc.use_tape = False
        one = enc.forward(enc.upload([waves[j]]), [lens[j]])
        enc.use_tape = True
        torch.cuda.synchronize()
        assert torch.equal(one.states, keep), ("whisper batched/tape != single/eager", mode, lens, j)
        rows = whisper_saved_rows(lens[j], geo.hidden)
        with torch.no_grad():
            ref = O.whisper_hidden_states(geo, sd, torch.from_numpy(O.whisper_log_mel(waves[j], geo.n_mels)))
        err = max(rel(keep[l][:rows].cpu(), r[:rows]) for l, r in enumerate(ref))
    else:
        B, T = int(rng.integers(1, 9)), int(rng.choice([16, 80, 80, 130] if fam == "deberta" else [16, 80, 80, 87]))
        lens = rng.integers(1, T + 1, B)
        ids = torch.from_numpy(rng.integers(3, geo.vocab_size, (B, T)))
        mask = (torch.arange(T)[None, :] < torch.from_numpy(lens)[:, None]).to(torch.int64)
        ids = torch.where(mask.bool(), ids, torch.full_like(ids, geo.pad_token_id))
        hs = enc.forward(ids, mask)
        torch.cuda.synchronize()
        j = int(rng.integers(B))
        keep = hs.states[:, hs.frame_offs[j]:hs.frame_offs[j + 1]].clone()
        one = enc.forward(ids[j:j + 1], mask[j:j + 1])
        torch.cuda.synchronize()
        assert torch.equal(one.states, keep), (fam + " batched != single", mode, T, lens.tolist(), j)
        with torch.no_grad():
            fn = O.roberta_hidden_states if fam == "roberta" else O.deberta_hidden_states
            ref = fn(geo, sd, ids[j], mask[j])
        err = max(rel(keep[l].cpu(), r) for l, r in enumerate(ref))
    assert err < TOL[mode], (fam, mode, err)
    worst[key] = max(worst.get(key, 0.0), err)
    n += 1
print(f"{n} random batches ok; worst vs oracle:", {f"{k[0]}/{k[1]}": f"{v:.2e}" for k, v in sorted(worst.items())})
