"""ANALYSIS / TEST INFRASTRUCTURE ONLY (never imported by the product).  CPU what-if for the conv stem (no GPU): which operand of the stem GEMMs needs its second plane?  fp64 arithmetic with fp16
rounding of the conv / projection / positional-conv ACTIVATION operand and / or WEIGHT operand, followed by the encoder
layers of oracle/numerics_whatif.py either exact or with the "f16a" rounding set.  Today every parity mode runs the stem on the
3-product split (both operands two planes); a 2-product form (one operand single-plane) would cut a third of its MFMA work.

    python oracle/numerics_whatif_stem.py [tiny_wavlm|wavlm_large]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from numerics_whatif import layers, r16                                   # noqa: E402
from interspeech_ser_amd import config as C                              # noqa: E402
from interspeech_ser_amd.weights import synthetic_state_dict             # noqa: E402
from oracle import ssl_oracle as O                                        # noqa: E402


def stem(geo, sd, x, rx, rw, first=1, last=99):
    """rx / rw: round the activation / weight operand of stem GEMM i (0 = conv0 ... 6 = conv6, 7 = projection, 8 = pos conv)
    for first <= i <= last"""
    X = lambda i, t: r16(t) if rx and first <= i <= last else t          # noqa: E731
    W = lambda i, t: r16(t) if rw and first <= i <= last else t          # noqa: E731
    sd = {k: v.double() for k, v in sd.items()}
    h = x.double()[None, None, :]
    for i, (k, s) in enumerate(zip(geo.conv_kernel, geo.conv_stride)):
        p = f"feature_extractor.conv_layers.{i}"
        h = F.conv1d(X(i, h), W(i, sd[p + ".conv.weight"]), sd.get(p + ".conv.bias"), stride=s)
        h = F.layer_norm(h.transpose(1, 2), (h.shape[1],), sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"], 1e-5).transpose(1, 2)
        h = F.gelu(h)
    f = h[0].transpose(0, 1)
    f = F.layer_norm(f, (f.shape[-1],), sd["feature_projection.layer_norm.weight"], sd["feature_projection.layer_norm.bias"], geo.layer_norm_eps)
    pr = F.linear(X(7, f), W(7, sd["feature_projection.projection.weight"]), sd["feature_projection.projection.bias"])
    w = O.pos_conv_weight(sd)
    k = geo.pos_conv_kernel
    y = F.conv1d(X(8, pr).transpose(0, 1)[None], W(8, w), sd["encoder.pos_conv_embed.conv.bias"], padding=k // 2, groups=geo.pos_conv_groups)
    if k % 2 == 0:
        y = y[:, :, :-1]
    return pr + F.gelu(y)[0].transpose(0, 1)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "tiny_wavlm"
    geo = {"tiny_wavlm": C.TINY_WAVLM, "tiny_hubert": C.TINY_HUBERT, "wavlm_large": C.WAVLM_LARGE}[which]
    sd = synthetic_state_dict(geo, 0)
    rng = np.random.default_rng(3)
    n = 48000 if which == "wavlm_large" else 16000
    t = np.arange(n) / 16000.0
    wave = (0.1 * rng.standard_normal(n) + 0.2 * np.sin(2 * np.pi * 220 * t)).astype(np.float32)
    f16a = ("h", "w_1", "ffn", "w_2")
    with torch.no_grad():
        x = torch.from_numpy(O.zero_mean_unit_var(wave))
        exact = layers(geo, sd, stem(geo, sd, x, False, False), ())
        print(f"{geo.name or which}: {exact[0].shape[0]} frames; error of all states vs the all-exact run")
        rows = [("stem exact, layers f16a", False, False, 1, 99, f16a),
                ("stem activations one plane (weights two), layers exact", True, False, 0, 99, ()),
                ("stem weights one plane (activations two), layers exact", False, True, 0, 99, ()),
                ("stem both one plane, layers exact", True, True, 0, 99, ()),
                ("conv 1-6 weights one plane only, layers exact", False, True, 1, 6, ()),
                ("conv 1-6 activations one plane only, layers exact", True, False, 1, 6, ()),
                ("conv 1 weights one plane only, layers exact", False, True, 1, 1, ()),
                ("conv 1 activations one plane only, layers exact", True, False, 1, 1, ()),
                ("stem weights one plane, layers f16a", False, True, 0, 99, f16a),
                ("stem activations one plane, layers f16a", True, False, 0, 99, f16a)]
        for name, rx, rw, a, b, lset in rows:
            got = layers(geo, sd, stem(geo, sd, x, rx, rw, a, b), lset)
            err = max(float((g - e).abs().max() / max(1.0, float(e.abs().max()))) for g, e in zip(got, exact))
            print(f"  {name:62s} {err:.2e}")


if __name__ == "__main__":
    main()
