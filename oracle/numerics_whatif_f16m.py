"""ANALYSIS / TEST INFRASTRUCTURE ONLY (never imported by the product).  CPU what-if for the "f16m" numerics mode (no GPU), at FULL depth:
the encoder layers in fp64 with the operand roundings of a candidate GEMM arithmetic switched on, before any kernel is written.

    python oracle/numerics_whatif_f16m.py [wavlm|hubert] [plain|sharp2|lora|outliers] [seconds]

Candidate arithmetics of  y = x W^T  (x, W in fp64; every one accumulates exactly -- only OPERAND formats differ):
    f16      one fp16 plane per operand                                   (the "f16" mode's layers)
    bf16x3   bf16 hi + lo planes, hi*hi + lo*hi + hi*lo                    ("fp32x")
    f16x     fp16 hi + lo planes, three products                           ("f16x", the drivers' default)
    f16m8    fp16 hi*hi  +  e4m3(lo_x)*e4m3(hi_w) + e4m3(hi_x)*e4m3(lo_w)  with one power-of-two scale per 32 consecutive k of a row
             (the OCP MX block format the gfx950 v_mfma_scale_f32_16x16x128_f8f6f4 instruction consumes)
    f16m6    the same with e2m3 (fp6) cross-term planes
The attention core (q, k, v, P, context rows) keeps fp16 hi + lo planes in the f16m variants, as the HIP path's ser_attention does.
A fifth argument lists variants as arith@site+site...: sites q, k, v (qkv = all three), out, fc1, fc2 take the arithmetic, the others stay f16x;
"pv1" puts P and V of the context product on ONE fp16 plane each.  Round 5, WavLM-large, 5 s, worst state (sharp x2 / LoRA): f16m8 everywhere
5.2e-4 / 4.7e-4; @qkv 4.3e-4 / 4.4e-4; @fc1+fc2 1.4e-4 / 2.7e-5 (= the "f16mf" mode, the drivers' default); @fc1+fc2+out 1.9e-4 / 1.7e-4;
@fc1+fc2+v 1.8e-4 / 1.5e-4; @fc1+fc2+q 3.2e-4 / 3.0e-4; @fc1+fc2+k 3.5e-4 / 3.4e-4; @fc1+fc2+pv1 3.5e-3 / 3.5e-3; f16x 2.1e-5 / 1.6e-5; bf16x3 1.9e-4 / 2.4e-4.
Sites may carry a first layer (qkv>=8) and groups may be chained (f16m8@fc1+fc2+qkv>=8;f16@fc1>=22+fc2>=22): qkv>=8 1.44e-4 / 3.7e-5, qkv>=6
1.67e-4 / 3.6e-5, qkv>=4 1.75e-4 / 4.4e-5, qkv>=2 2.4e-4 / 1.2e-4, qkv>=8+out>=8 1.44e-4 / 4.1e-5; single fp16 products in the feed-forward of the
last 2 / 4 layers on top of qkv>=8: 1.6e-4 / 4.1e-5 and 1.9e-4 / 3.8e-5.
`... stem [model] [seconds]`: the conv stem's layers 1-6 in each format behind exact layers -- last hidden state under sharp x2 (3 s): f16x 6.6e-5,
bf16x3 6.2e-4, f16m8 1.4e-3 (outside the gate: every stem error passes through all 24 softmax layers -- the stem stays on 22-bit operands).
Error form of the tests: max|a - b| / max(1, max|b|) per hidden state, worst state, against the all-exact fp64 run.
Follows HF modeling_wavlm.py:147-241,288-295,355-373 (reference call site preprocessing/preprocess_speech.py:50,66) through oracle.ssl_oracle."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from interspeech_ser_amd import config as C                              # noqa: E402
from oracle import ssl_oracle as O                                        # noqa: E402  (checker-side code)


def r16(t):
    return t.to(torch.float16).double()


def rbf(t):
    return t.to(torch.bfloat16).double()


def minifloat(t, mbits, emin, vmax):
    """round-to-nearest-even onto a sign + exponent + mbits-mantissa grid with subnormals below 2^emin, saturating at vmax"""
    a = t.abs().clamp(max=vmax)
    e = torch.floor(torch.log2(a.clamp(min=1e-300))).clamp(min=emin)
    q = torch.pow(2.0, e - mbits)
    return torch.sign(t) * torch.round(a / q) * q                          # torch.round: half to even


def mx_quant(t, kind):
    """[rows, K] -> block-scaled (32 consecutive k per block, scale = the smallest power of two that brings the block into range) minifloat"""
    mbits, emin, vmax = {"e4m3": (3, -6, 448.0), "e2m3": (3, 0, 7.5)}[kind]
    R, K = t.shape
    b = t.reshape(R, K // 32, 32)
    amax = b.abs().amax(-1, keepdim=True)
    code = torch.ceil(torch.log2((amax / vmax).clamp(min=2.0 ** -126)))
    s = torch.pow(2.0, code)
    return (minifloat(b / s, mbits, emin, vmax) * s).reshape(R, K)


def mm(x, W, arith):
    if arith == "exact":
        return x @ W.T
    if arith == "f16":
        return r16(x) @ r16(W).T
    if arith == "bf16x3":
        xh, wh = rbf(x), rbf(W)
        xl, wl = rbf(x - xh), rbf(W - wh)
        return xh @ wh.T + xl @ wh.T + xh @ wl.T
    xh, wh = r16(x), r16(W)
    if arith == "f16x":
        xl, wl = r16(x - xh), r16(W - wh)
        return xh @ wh.T + xl @ wh.T + xh @ wl.T
    kind = {"f16m8": "e4m3", "f16m6": "e2m3"}[arith]
    xl, wl = x - xh, W - wh
    return xh @ wh.T + mx_quant(xl, kind) @ mx_quant(W, kind).T + mx_quant(x, kind) @ mx_quant(wl, kind).T


def layers(geo, sd, h0, arith, sites=("qkv", "out", "fc1", "fc2")):
    """sites: which GEMMs run `arith`; the others and the attention core run fp16 hi + lo planes ('f16x')."""
    # site names: "qkv" (= "q", "k", "v" together), "out", "fc1", "fc2"; "pv1": P and V of the context product on ONE fp16 plane each
    # a site may carry a first layer, "qkv>=12": the arithmetic from that layer on (errors injected late pass through fewer softmax layers)
    # `arith` may also be a list of (arithmetic, sites) groups -- "f16m8@fc1+fc2+qkv>=8;f16@fc1>=20+fc2>=20": the LAST group that names a
    # site at the current layer wins
    groups = arith if isinstance(arith, list) else [(arith, sites)]
    arith = groups[0][0]
    parsed = []
    for ar, ss in groups:
        f = {}
        for e in ss:
            name, _, n = e.partition(">=")
            f[name] = int(n) if n else 0
        parsed.append((ar, f))
    first = parsed[0][1]
    layer_now = [0]

    def A(site):
        for ar, f in reversed(parsed):
            for name in (site, "qkv" if site in ("q", "k", "v") else site):
                if name in f and layer_now[0] >= f[name]:
                    return ar
        return "f16x"
    eps, H, dh, D = geo.layer_norm_eps, geo.heads, geo.head_dim, geo.hidden
    sd = {k: v.double() for k, v in sd.items()}
    h = h0.double()
    T = h.shape[0]
    table = O.relative_bias_table(geo, {k: v.float() for k, v in sd.items()}, T).double() if geo.family == "wavlm" else None
    core = (lambda t: t) if arith == "exact" else (lambda t: r16(t) + r16(t - r16(t)))
    states = []

    def deferred_ln_linear(x, lnp, W, b, site):
        mu = x.mean(-1, keepdim=True)
        var = x.var(-1, unbiased=False, keepdim=True)
        g, be = sd[lnp + ".weight"], sd[lnp + ".bias"]
        return mm(x - mu, W * g[None, :], A(site)) * torch.rsqrt(var + eps) + (W @ be + (b if b is not None else 0.0))

    for i in range(geo.num_layers):
        layer_now[0] = i
        states.append(h)
        p = f"encoder.layers.{i}"
        a = p + ".attention"
        Wq, Wk, Wv = (sd[a + f".{n}_proj.weight"] for n in "qkv")
        bq, bk, bv = (sd[a + f".{n}_proj.bias"] for n in "qkv")
        for n in ("q", "v"):                                                # LoRA adapters, merged like weights.merge_lora does
            if a + f".{n}_proj.lora_A.weight" in sd:
                delta = float(sd["lora_scale"]) * sd[a + f".{n}_proj.lora_B.weight"] @ sd[a + f".{n}_proj.lora_A.weight"]
                if n == "q":
                    Wq = Wq + delta
                else:
                    Wv = Wv + delta
        q = deferred_ln_linear(h, p + ".layer_norm", Wq, bq, "q") * dh ** -0.5
        k = deferred_ln_linear(h, p + ".layer_norm", Wk, bk, "k")
        v = deferred_ln_linear(h, p + ".layer_norm", Wv, bv, "v")
        pv1 = "pv1" in first and i >= first["pv1"] and arith != "exact"
        qh, kh = (O._heads(core(t), H) for t in (q, k))
        vh = O._heads(r16(v) if pv1 else core(v), H)
        scores = qh @ kh.transpose(1, 2)
        if geo.family == "wavlm":
            x_ln = F.layer_norm(h, (D,), sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"], eps)
            gate = O.wavlm_gate(geo, sd, a, x_ln)
            idx = (torch.arange(T)[None, :] - torch.arange(T)[:, None]) + (T - 1)
            scores = scores + gate[:, :, None] * table[:, idx]
        P = torch.softmax(scores, dim=-1)
        ctx = ((r16(P) if pv1 else core(P)) @ vh).permute(1, 0, 2).reshape(T, D)
        h = h + mm(ctx, sd[a + ".out_proj.weight"], A("out")) + sd[a + ".out_proj.bias"]
        f = F.gelu(deferred_ln_linear(h, p + ".final_layer_norm", sd[p + ".feed_forward.intermediate_dense.weight"],
                                      sd[p + ".feed_forward.intermediate_dense.bias"], "fc1"))
        h = h + mm(f, sd[p + ".feed_forward.output_dense.weight"], A("fc2")) + sd[p + ".feed_forward.output_dense.bias"]
    states.append(F.layer_norm(h, (D,), sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], eps))
    return states


def stem(geo, sd, x, arith, from_layer=1):
    """The conv feature encoder in fp64 with the operand roundings of `arith` on conv layers >= from_layer (layer 0, K = 10 taps, and the others
    stay f16x): each layer as windows [T_out, k * C_in] (frame-major, channel-minor: the order the HIP path's implicit GEMM reads) times
    W [C_out, k * C_in]^T, then LayerNorm(C) + exact GELU.  Follows oracle.ssl_oracle.conv_feature_encoder (HF modeling_wavlm.py:696-782)."""
    sd = {k: v.double() for k, v in sd.items() if k.startswith("feature_extractor")}
    h = x.double()[:, None]                                                  # [L, 1]
    for i, (k, st) in enumerate(zip(geo.conv_kernel, geo.conv_stride)):
        p = f"feature_extractor.conv_layers.{i}"
        W = sd[p + ".conv.weight"]                                           # [C_out, C_in, k]
        T_out = (h.shape[0] - k) // st + 1
        idx = (torch.arange(T_out)[:, None] * st + torch.arange(k)[None, :])  # [T_out, k]
        win = h[idx].reshape(T_out, -1)                                      # [T_out, k * C_in]
        Wm = W.permute(0, 2, 1).reshape(W.shape[0], -1)                      # [C_out, k * C_in]
        a = arith if (i >= from_layer and arith != "exact") else ("exact" if arith == "exact" else "f16x")
        if a in ("f16m8", "f16m6") and win.shape[1] % 32:
            a = "f16x"
        y = mm(win, Wm, a)
        if p + ".conv.bias" in sd:
            y = y + sd[p + ".conv.bias"]
        y = F.layer_norm(y, (y.shape[1],), sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"], 1e-5)
        h = F.gelu(y)
    return h


def main_stem():
    """python oracle/numerics_whatif_f16m.py stem [wavlm] [seconds]: error of the conv stem's output (and of hidden_states[L] behind exact layers
    fed with it) when conv layers 1..6 multiply in f16m's format instead of f16x's."""
    import depth_envelope as DE
    model = sys.argv[2] if len(sys.argv) > 2 else "wavlm"
    seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
    geo = C.geometry_for(DE.MODELS[model])
    for kind in ("plain", "sharp2"):
        ref_sd, _ = DE.case_state_dicts(geo, kind)
        base = {k: v for k, v in ref_sd.items() if "lora" not in k}
        with torch.no_grad():
            x = torch.from_numpy(O.zero_mean_unit_var(DE.clip(101, seconds)))
            feats = {a: stem(geo, base, x, a) for a in ("exact", "f16x", "bf16x3", "f16m8")}
            print(f"{DE.MODELS[model]} [{kind}] conv stem, {feats['exact'].shape[0]} frames: error of the stem output / of the last hidden state behind exact layers")
            outs = {}
            for a, f in feats.items():
                h0 = O.feature_projection(geo, base, f.float()).double()
                h0 = h0 + O.positional_conv(geo, base, h0.float()).double()
                outs[a] = layers(geo, ref_sd, h0, "exact")
            for a in ("f16x", "bf16x3", "f16m8"):
                e0 = float((feats[a] - feats["exact"]).abs().max() / max(1.0, float(feats["exact"].abs().max())))
                eL = max(float((g - b).abs().max() / max(1.0, float(b.abs().max()))) for g, b in zip(outs[a], outs["exact"]))
                print(f"  stem layers 1-6 in {a:7s}: stem output {e0:.2e}   worst hidden state {eL:.2e}")
                sys.stdout.flush()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "stem":
        return main_stem()
    import depth_envelope as DE
    model = sys.argv[1] if len(sys.argv) > 1 else "wavlm"
    kind = sys.argv[2] if len(sys.argv) > 2 else "sharp2"
    seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
    geo = C.geometry_for(DE.MODELS[model])
    ref_sd, _ = DE.case_state_dicts(geo, kind)
    wave = DE.clip(101, seconds)
    with torch.no_grad():
        x = torch.from_numpy(O.zero_mean_unit_var(wave))
        base = {k: v for k, v in ref_sd.items() if "lora" not in k}
        feats = O.conv_feature_encoder(geo, base, x)
        h0 = O.feature_projection(geo, base, feats)
        h0 = h0 + O.positional_conv(geo, base, h0)
        exact = layers(geo, ref_sd, h0, "exact")
        print(f"{DE.MODELS[model]} [{kind}], {h0.shape[0]} frames, {geo.num_layers} layers; error of the worst state against exact fp64 layers")
        variants = [("f16x", None), ("bf16x3", None), ("f16m8", None), ("f16m6", None), ("f16", ("fc1", "fc2")), ("f16", None)]
        if len(sys.argv) > 4:
            # "f16m8@fc1+fc2": the arithmetic on those GEMMs only, fp16 hi + lo planes (f16x) on the others
            variants = []
            for v in sys.argv[4].split(","):
                if ";" in v:
                    variants.append(([(g.split("@")[0], tuple(g.split("@")[1].split("+"))) for g in v.split(";")], "groups"))
                else:
                    variants.append((v.split("@")[0], tuple(v.split("@")[1].split("+")) if "@" in v else None))
        for arith, sites in variants:
            got = layers(geo, ref_sd, h0, arith) if sites is None else layers(geo, ref_sd, h0, arith, sites)
            per = [float((a - b).abs().max() / max(1.0, float(b.abs().max()))) for a, b in zip(got, exact)]
            L = len(per) - 1
            tag = (";".join(a + "@" + "+".join(ss) for a, ss in arith) if isinstance(arith, list) else arith + ("" if sites is None else " on " + "+".join(sites)))
            print(f"  {tag:22s} state {L // 2}: {per[L // 2]:.2e}   state {L}: {per[L]:.2e}   worst {max(per):.2e}")
            sys.stdout.flush()


if __name__ == "__main__":
    main()
