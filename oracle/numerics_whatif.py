"""ANALYSIS / TEST INFRASTRUCTURE ONLY (never imported by the product).  CPU what-if for the numerics modes (no GPU): the encoder layers in fp64 with fp16 OPERAND ROUNDING switched on per GEMM
site, so that the error of a candidate mode is known before a kernel is written (DESIGN.md section 4 records the outcomes).

    python oracle/numerics_whatif.py [tiny_wavlm|tiny_hubert|tiny_wav2vec2|wavlm_large] [plain|sharp|lora]

Sites (what the HIP path rounds to one fp16 plane in the "f16" mode; "exact" = fp16 hi + lo planes, 3 products):
    x_qk, w_qk   operands of the q / k (+ gate) columns of the packed projection      q_k    the stored q, k (operands of S = K Q^T)
    x_v,  w_v    operands of the v columns                                             v, p   the stored v, the probabilities P
    ctx, w_o     operands of the output projection                                     h, w_1 operands of FC1
    ffn, w_2     operands of FC2
The stem (conv stack, projection, positional conv) is exact here (it runs the fp32x split in every parity mode).
Error form of the tests: max|a - b| / max(1, max|b|) per hidden state, worst state, against the all-exact fp64 run."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import config as C                              # noqa: E402
from interspeech_ser_amd.weights import apply_stress, synthetic_state_dict  # noqa: E402
from oracle import ssl_oracle as O                                        # noqa: E402  (this file lives under oracle/: checker-side code)

ALL = ("x_qk", "w_qk", "q_k", "x_v", "w_v", "v", "p", "ctx", "w_o", "h", "w_1", "ffn", "w_2")


def r16(t):
    return t.to(torch.float16).double()


def layers(geo, sd, h0, rounded):
    R = lambda site, t: r16(t) if site in rounded else t                  # noqa: E731
    eps, H, dh, D = geo.layer_norm_eps, geo.heads, geo.head_dim, geo.hidden
    sd = {k: v.double() for k, v in sd.items()}
    h = h0.double()
    T = h.shape[0]
    table = O.relative_bias_table(geo, {k: v.float() for k, v in sd.items()}, T).double() if geo.family == "wavlm" else None
    states = []

    def deferred_ln_linear(x, lnp, W, b, sx, sw):
        """what the deferred-LayerNorm GEMM computes: rounded centred rows times rounded gamma-folded weights"""
        mu = x.mean(-1, keepdim=True)
        var = x.var(-1, unbiased=False, keepdim=True)
        g, be = sd[lnp + ".weight"], sd[lnp + ".bias"]
        xc = R(sx, x - mu)
        Wg = R(sw, W * g[None, :])
        return (xc @ Wg.T) * torch.rsqrt(var + eps) + (W @ be + (b if b is not None else 0.0))

    for i in range(geo.num_layers):
        states.append(h)
        p = f"encoder.layers.{i}"
        a = p + ".attention"
        Wq, Wk, Wv = (sd[a + f".{n}_proj.weight"] for n in "qkv")
        bq, bk, bv = (sd[a + f".{n}_proj.bias"] for n in "qkv")
        for n, Wn in (("q", Wq), ("v", Wv)):                                # LoRA adapters, merged like weights.merge_lora does
            if a + f".{n}_proj.lora_A.weight" in sd:
                delta = float(sd["lora_scale"]) * sd[a + f".{n}_proj.lora_B.weight"] @ sd[a + f".{n}_proj.lora_A.weight"]
                if n == "q":
                    Wq = Wq + delta
                else:
                    Wv = Wv + delta
        q = deferred_ln_linear(h, p + ".layer_norm", Wq, bq, "x_qk", "w_qk") * dh ** -0.5
        k = deferred_ln_linear(h, p + ".layer_norm", Wk, bk, "x_qk", "w_qk")
        v = deferred_ln_linear(h, p + ".layer_norm", Wv, bv, "x_v", "w_v")
        qh, kh, vh = (O._heads(t, H) for t in (R("q_k", q), R("q_k", k), R("v", v)))
        scores = qh @ kh.transpose(1, 2)
        if geo.family == "wavlm":
            x_ln = F.layer_norm(h, (D,), sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"], eps)
            gate = O.wavlm_gate(geo, sd, a, x_ln)
            idx = (torch.arange(T)[None, :] - torch.arange(T)[:, None]) + (T - 1)
            scores = scores + gate[:, :, None] * table[:, idx]
        P = torch.softmax(scores, dim=-1)
        ctx = (R("p", P) @ vh) / 1.0
        ctx = ctx.permute(1, 0, 2).reshape(T, D)
        h = h + R("ctx", ctx) @ R("w_o", sd[a + ".out_proj.weight"]).T + sd[a + ".out_proj.bias"]
        f = F.gelu(deferred_ln_linear(h, p + ".final_layer_norm", sd[p + ".feed_forward.intermediate_dense.weight"],
                                      sd[p + ".feed_forward.intermediate_dense.bias"], "h", "w_1"))
        h = h + R("ffn", f) @ R("w_2", sd[p + ".feed_forward.output_dense.weight"]).T + sd[p + ".feed_forward.output_dense.bias"]
    states.append(F.layer_norm(h, (D,), sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], eps))
    return states


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "tiny_wavlm"
    stress = sys.argv[2] if len(sys.argv) > 2 else "plain"
    geo = {"tiny_wavlm": C.TINY_WAVLM, "tiny_hubert": C.TINY_HUBERT, "tiny_wav2vec2": C.TINY_WAV2VEC2, "wavlm_large": C.WAVLM_LARGE}[which]
    sd = synthetic_state_dict(geo, int(os.environ.get("WHATIF_SEED", "25")))
    if stress == "sharp":
        sd = apply_stress(sd, geo, "sharp")
    if stress == "lora":                                                   # the adapters of tests/test_gpu_cli.py's LoRA test
        g = torch.Generator().manual_seed(42)
        sd["lora_scale"] = torch.tensor(2.0)
        for key in list(sd):
            mod, _, leaf = key.rpartition(".")
            if mod.endswith((".q_proj", ".v_proj")) and leaf == "weight":
                sd[mod + ".lora_A.weight"] = torch.randn(8, geo.hidden, generator=g) * 0.3
                sd[mod + ".lora_B.weight"] = torch.randn(geo.hidden, 8, generator=g) * 0.3
    rng = np.random.default_rng(int(os.environ.get("WHATIF_WAVE_SEED", "3")))
    n = int(os.environ.get("WHATIF_SAMPLES", "48000" if which == "wavlm_large" else "16000"))
    t = np.arange(n) / 16000.0
    wave = (0.1 * rng.standard_normal(n) + 0.2 * np.sin(2 * np.pi * 220 * t)).astype(np.float32)
    with torch.no_grad():
        x = torch.from_numpy(O.zero_mean_unit_var(wave))
        base = {k: v for k, v in sd.items() if "lora" not in k}
        feats = O.conv_feature_encoder(geo, base, x)
        h0 = O.feature_projection(geo, base, feats)
        h0 = h0 + O.positional_conv(geo, base, h0)
        exact = layers(geo, sd, h0, ())
        variants = {
            "f16   (every site one fp16 plane)": ALL,
            "f16q  (q/k projection + QK^T exact)": tuple(s for s in ALL if s not in ("x_qk", "w_qk", "q_k")),
            "      + v projection, P, V exact (whole attention core)": tuple(s for s in ALL if s not in ("x_qk", "w_qk", "q_k", "x_v", "w_v", "v", "p")),
            "      + output projection exact": ("h", "w_1", "ffn", "w_2"),
            "      only the feed-forward exact (attention f16)": tuple(s for s in ALL if s not in ("h", "w_1", "ffn", "w_2")),
            "      f16q + feed-forward exact": ("x_v", "w_v", "v", "p", "ctx", "w_o"),
            "      f16q + FC2 exact": tuple(s for s in ALL if s not in ("x_qk", "w_qk", "q_k", "ffn", "w_2")),
            "      f16q, activations exact everywhere (weights one plane)": ("w_v", "w_o", "w_1", "w_2"),
            "      f16q, weights exact everywhere (activations one plane)": ("x_v", "v", "p", "ctx", "h", "ffn"),
        }
        print(f"{geo.name or which} [{stress}], {h0.shape[0]} frames, {geo.num_layers} layers")
        for name, rounded in variants.items():
            got = layers(geo, sd, h0, rounded)
            err = max(float((a - b).abs().max() / max(1.0, float(b.abs().max()))) for a, b in zip(got, exact))
            print(f"  {name:70s} {err:.2e}")


if __name__ == "__main__":
    main()
