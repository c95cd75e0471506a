"""Parity of the numerics modes at FULL depth under the stress weights (-m gpu; helper + report generator: tests/depth_envelope.py).

Every stress fixture under tests/golden/ has 2-3 layers; the reference runs 24 (WavLM-large) and 48 (HuBERT-xlarge) in fp32
(preprocessing/preprocess_speech.py:50,66,111-114) and every rounding inside an attention block is amplified by the later layers'
softmax.  Measured in round 4 (profiles/r04_depth_envelope.txt, error form of test_gpu_e2e.py, all L + 1 states, one 10 s utterance
and a ragged 3 s + 10 s pair against oracle.ssl_oracle.speech_hidden_states):

  * the fp32 REFERENCE itself is only as good as its conditioning: against the same formulas in float64 it sits at 9e-7 on Gaussian
    weights, 2.7e-5 with the q / k projections x2 (logits x4), 5e-4 at x2.5 and 0.22 at x4 (the tiny fixtures' "sharp" setting is
    chaotic at depth: no implementation, the reference on another BLAS included, reproduces it).  The gate cases here are the ones
    where the reference is well defined (<= 5e-5): sharp x2, outlier channels, row-mean offsets, LoRA-scaled q / v;
  * "f16x" (3 products everywhere on fp16 hi + lo planes, the drivers' default since round 4): <= 1.0e-4 on every gate case;
  * "fp32x" (bf16 hi + lo): <= 8e-4 (sharp x2 7.9e-4 / 4.8e-4, LoRA 7.4e-4 / 6.9e-4): inside the gate, little margin;
  * "f16a" (round 3's default: single-product fp16 feed-forward): 3.7e-3 / 2.9e-3 under sharp attention -- the feed-forward's rounding,
    benign in 2-3 layers, is amplified by 24-48: it LEAVES the gate, which is why it is no longer the default; <= 6e-4 elsewhere;
  * "f16mf" (round 5, the drivers' default: f16m's operand format on the feed-forward pair of every layer and on the packed projection from a
    third of the depth on -- the early packed projections, whose rounding every later softmax amplifies, keep f16x's 22 bits; 1.15 x f16x's throughput): WavLM-large 1.4e-5 plain, 1.75e-4 sharp x2, 9.1e-5 LoRA, 4.9e-6 outliers;
    HuBERT-xlarge sharp x2 1.2e-4; Whisper-large-v3 sharp x2 1.7e-4 -- inside fp32x's on EVERY case by 2-7 x and inside the default's 4x margin.
  * "f16m" (round 5: packed projection / FC1 / FC2 as fp16 main product + block-scaled e4m3 cross terms, ~2^-15 operands, 1.1 x f16x's
    throughput; profiles/r05_depth_envelope_f16m*.txt): WavLM-large 1.7e-5 plain, 4.8e-4 sharp x2, 4.2e-4 LoRA, 6e-6 outliers -- inside
    fp32x's on each; HuBERT-xlarge sharp x2 3.0e-4 (fp32x 4.8e-4); Whisper-large-v3 sharp x2 5.7e-4 (fp32x 3.2e-4: the one case where it is
    the worse of the two) -- inside the gate everywhere, so it ships as the documented FAST tolerance-grade mode and "f16x" stays the default.
"""
import pytest

pytestmark = pytest.mark.gpu

GATE = 1e-3
# The suite's time budget (round-4 verdict 7c: <= 550 s on a driver box; the CPU oracle and the weight generation are most of a case): every gate
# case runs the default ("f16mf") and "f16x"; "fp32x" only on WavLM sharp x2, "f16m" on WavLM sharp x2 / LoRA and Whisper.  The report generator
# (python tests/depth_envelope.py -> profiles/r04_depth_envelope*.txt, r05_depth_envelope_f16m*.txt, r05_depth_envelope_f16mf.txt) holds the rest,
# measured there and stable since: WavLM row means, HuBERT LoRA / row means / outliers, XLS-R-2B, every "f16a" row, fp32x and f16m on the
# other cases.  The weights stay the host-stable numpy
# stream: with torch's generator (seconds faster) the LoRA case lands on a worse-conditioned draw (fp32x 1.06e-3, f16m 7.6e-4, f16x 1.5e-4
# against 7.4e-4 / 4.2e-4 / 1.0e-4 here) -- the envelope is a property of the checkpoint as much as of the mode, which is why the default
# keeps a 4x margin.
CASES = [("wavlm", "sharp2"), ("wavlm", "lora"), ("wavlm", "outliers"), ("hubert", "sharp2")]


@pytest.mark.parametrize("model,kind", CASES)
def test_full_depth_stress_envelope(model, kind):
    import depth_envelope as DE
    modes = {("wavlm", "sharp2"): ("f16x", "fp32x", "f16m", "f16mf"), ("wavlm", "lora"): ("f16x", "f16m", "f16mf")}.get((model, kind), ("f16x", "f16mf"))
    res = DE.envelope(model, kind, modes)
    worst = {k: max(v) for k, v in res.items()}
    print(f"{DE.MODELS[model]} stress={kind}: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
    for mode in modes:                                   # the parity-grade modes hold north_star's 1e-3 at the depth they ship at
        assert worst[mode] < GATE, (model, kind, mode, worst)
        if kind == "outliers":                           # ... and on the ordinary channels' own scale beside the 800-sized ones
            assert worst[mode + ":ordinary"] < GATE, worst
    assert worst["f16x"] < 2.5e-4, worst                 # round 4's default keeps a 4x margin (measured <= 1.0e-4)
    # "f16mf", the default (f16m's operand format on FC1 / FC2 only, round 5): measured 1.75e-4 / 9.1e-5 / 2.8e-6 / 1.2e-4 on these cases -- 4x too
    assert worst["f16mf"] < 2.5e-4, worst
    if kind != "outliers":                               # (there all sit at the shared fp16x stem's 2.8e-6)
        assert worst["f16x"] <= worst["f16mf"] * 1.25, worst  # 22-bit operands everywhere vs ~15-bit ones in the feed-forward (XLS-R LoRA: 3.9e-5 / 3.5e-5)
    if "f16m" in worst:
        assert worst["f16mf"] <= worst["f16m"] * 1.05, worst  # ... which are never worse than ~15-bit ones in the packed projection too
    if "fp32x" in worst:
        assert worst["f16mf"] <= worst["fp32x"] * 1.05, worst
        assert worst["f16x"] <= worst["fp32x"] * 1.05, worst   # ... nor than 16-bit ones at the same cost
        assert worst["f16m"] <= worst["fp32x"] * 1.05, worst   # measured: f16m inside fp32x's envelope on the wav2vec2-style encoders


def test_full_depth_whisper_sharp_attention():
    """The Whisper-large-v3 encoder (32 layers, 1 500 frames; preprocessing/preprocess_whisper.py:48-76) under the same stress: q / k
    projections x 2, a full 30 s window alone and a ragged 7.3 s + 30 s pair, the rows the driver saves of all 33 states against
    oracle.whisper_hidden_states on oracle.whisper_log_mel.  Measured (profiles/r04_depth_envelope_whisper.txt, r05_depth_envelope_f16m_hubert_whisper.txt):
    f16x 5.6e-5, fp32x 3.2e-4, f16m 5.7e-4 -- its widest case (1 500 keys per softmax) and the one where it is worse than fp32x; f16a 4.8e-3: outside."""
    import depth_envelope as DE
    res = DE.whisper_envelope("sharp2", ("f16x", "f16m", "f16mf"))
    worst = {k: max(v) for k, v in res.items()}
    print("openai/whisper-large-v3 stress=sharp2: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
    assert worst["f16m"] < GATE, worst
    assert worst["f16x"] < 2.5e-4 and worst["f16x"] <= worst["f16m"] * 1.05, worst
    assert worst["f16mf"] < 2.5e-4 and worst["f16mf"] < 3.23e-4, worst     # measured 1.73e-4: inside fp32x's 3.2e-4 here, where f16m (5.7e-4) is not
