"""Kernel-level tests of SER_MODE_FP16X (fp16 hi + lo planes, 3 products on the f16 MFMA) and SER_MODE_FP16Q (ser_attention
with only q, k split): the attention block of the "f16a" numerics mode and the logit path of "f16q" -- packed projection,
S = K Q^T, P V, output projection -- against fp64 statements, through the C ABI.  -m gpu.  (Reference arithmetic is fp32:
preprocessing/preprocess_speech.py:50,66.)"""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from test_gpu_kernels import DEV, attention_reference, run_gemm, stream

pytestmark = pytest.mark.gpu

FP16, FP16X, FP16Q = 3, 4, 5


@pytest.fixture(scope="module")
def L():
    from interspeech_ser_amd import _lib
    assert torch.cuda.is_available()
    return _lib


def split_h(x):
    """fp32 CPU -> (hi, lo) fp16 planes the way ser_common.h split_h does"""
    hi = x.to(torch.float16)
    lo = (x - hi.float()).to(torch.float16)
    return torch.stack([hi, lo]).contiguous().to(DEV)


def planes_value(t):
    return t.double().sum(0)


@pytest.mark.parametrize("mode", [2, 4])
@pytest.mark.parametrize("M,N,K", [(3000, 3104, 192), (2700, 3592, 64), (3992, 3104, 320)])
def test_gemm_two_plane_256x256_tile_integer_exact(L, mode, M, N, K):
    """>= 150 tiles of 256 x 256: FP32X / FP16X launches take the 256 x 256 two-plane tile (64 x 128 wave tiles, weight fragments
    in two halves) -- ragged M / N edges, 2..10 K tiles of 32, with bias."""
    assert ((M + 255) // 256) * ((N + 255) // 256) >= 150
    from test_gpu_kernels import to_act
    g = torch.Generator().manual_seed(M + N)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    W = torch.randint(-3, 4, (N, K), generator=g).float() + (torch.arange(N)[:, None] % 3).float()
    bias = torch.randint(-4, 5, (N,), generator=g).float()
    ref = A.double() @ W.double().T + bias.double()
    a, w = (split_h(A), split_h(W)) if mode == 4 else (to_act(A, 2), to_act(W, 2))
    out, _ = run_gemm(L, a, w, M, N, K, mode, bias=bias.to(DEV))
    assert torch.equal(out.cpu().double(), ref)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 192), (515, 392, 1024), (2561, 1288, 192), (3000, 1160, 320)])
def test_gemm_fp16x_integer_exact(L, M, N, K):
    """Small integers are exact in fp16: any slip in the two-plane stage layout or the fragment order shows as a wrong
    integer.  The last two shapes have >= 100 tiles of 256 x 128 and take the BK = 32 / 3-stage tile, the others the 128 x 128 one."""
    g = torch.Generator().manual_seed(M + N)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    W = torch.randint(-3, 4, (N, K), generator=g).float() + (torch.arange(N)[:, None] % 3).float()
    bias = torch.randint(-4, 5, (N,), generator=g).float()
    ref = A.double() @ W.double().T + bias.double()
    out, _ = run_gemm(L, split_h(A), split_h(W), M, N, K, FP16X, bias=bias.to(DEV))
    assert torch.equal(out.cpu().double(), ref)


@pytest.mark.parametrize("wscale,bound", [(1.0, 3e-6), (0.02, 3e-6), (1e-3, 4e-5)])
def test_gemm_fp16x_is_fp32_grade(L, wscale, bound):
    """3 products on fp16 hi + lo planes against fp64, and against the single-product FP16 launch on the hi planes alone.
    wscale = 0.02 is the size of real projection weights: their lo plane is made of fp16 SUBNORMALS (|w - fp16(w)| < 6e-5),
    which the f16 MFMA must not flush (measured 1.1e-6).  At 1e-3 the whole lo plane sits on the subnormal grid and the pair
    keeps an absolute 2^-25 per element: 2e-5 relative to such weights, still 15x below the single product."""
    M, N, K = 391, 264, 1024
    g = torch.Generator().manual_seed(3)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * wscale
    ref = A.double() @ W.double().T
    a4, w4 = split_h(A), split_h(W)
    out4, act4 = run_gemm(L, a4, w4, M, N, K, FP16X, want_act=False)
    out3, _ = run_gemm(L, a4[:1].contiguous(), w4[:1].contiguous(), M, N, K, FP16)
    scale = ref.abs().max().item()
    e4 = (out4.cpu().double() - ref).abs().max().item() / scale
    e3 = (out3.cpu().double() - ref).abs().max().item() / scale
    print(f"wscale {wscale}: FP16X {e4:.2e}, FP16 (hi planes only) {e3:.2e}")
    assert e4 < bound, e4                     # fp32 accumulation noise of a 1024-long sum
    assert e3 > 10 * e4, (e3, e4)             # the lo planes are doing the work


@pytest.mark.parametrize("cfg", [1, 2, 3])
def test_gemm_fp16_writes_fp16x_planes(L, cfg):
    """FC2 of the "f16q" mode (ser_gemm_args.out_mode): a single-product FP16 GEMM whose operand copy is written as fp16
    hi + lo planes for the next layer's 3-product q / k projection -- with the shifted-copy and row-partial outputs on."""
    M, N, K = 700, 520, 256
    g = torch.Generator().manual_seed(cfg)
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    res = torch.randn(M, N, generator=g) * 3
    G = (N + 63) // 64
    G += G & 1
    stat = torch.zeros(M, G, 2, device=DEV)
    sh_in = (torch.randn(M, generator=g) * 0.1).to(DEV)
    sh_out = torch.zeros(M, device=DEV)
    ga = L.GemmArgs()
    a3, w3 = A.to(torch.float16)[None].contiguous().to(DEV), W.to(torch.float16)[None].contiguous().to(DEV)
    out = torch.zeros((M, N), device=DEV)
    oact = torch.zeros((2, M, N), dtype=torch.float16, device=DEV)
    resd = res.to(DEV)
    ga.A, ga.lda, ga.W, ga.M, ga.N, ga.K, ga.groups, ga.mode, ga.out_mode = a3.data_ptr(), K, w3.data_ptr(), M, N, K, 1, FP16, FP16X
    ga.residual, ga.ldr, ga.out_f32, ga.ldo_f32 = resd.data_ptr(), N, out.data_ptr(), N
    ga.out_act, ga.ldo_act, ga.out_plane_stride = oact.data_ptr(), N, M * N
    ga.stat_out, ga.stat_groups, ga.shift_in, ga.shift_out, ga.shift_const = stat.data_ptr(), G, sh_in.data_ptr(), sh_out.data_ptr(), 0.25
    ga.tile_cfg = cfg
    L.check(L.lib.ser_gemm(C.byref(ga), stream()), "ser_gemm")
    torch.cuda.synchronize()
    v = out.cpu().double() - sh_out.cpu().double()[:, None]                     # what the copy holds: v - c[m]
    assert torch.equal(sh_out.cpu(), (sh_in.cpu() + 0.25))
    hi = oact[0].cpu()
    assert torch.equal(hi, v.float().to(torch.float16))                         # plane 0 IS the FP16 copy
    err = (planes_value(oact.cpu()) - v).abs().max().item()
    assert err < 4e-6 * max(1.0, v.abs().max().item()), err                     # hi + lo: ~22 bits
    # refused where no dense tile exists
    ga.ln_gamma = ga.ln_beta = out.data_ptr()
    assert L.lib.ser_gemm(C.byref(ga), stream()) < 0


def test_row_center_fp16x(L):
    M, D = 77, 320
    g = torch.Generator().manual_seed(9)
    x = torch.randn(M, D, generator=g) * 2 + 5
    xa = torch.zeros((2, M, D), dtype=torch.float16, device=DEV)
    st = torch.zeros((M, 2, 2), device=DEV)
    sh = torch.zeros(M, device=DEV)
    L.check(L.lib.ser_row_center(x.to(DEV).data_ptr(), D, xa.data_ptr(), D, M * D, st.data_ptr(), 2, sh.data_ptr(), FP16X, M, D, stream()))
    torch.cuda.synchronize()
    cen = x.double() - sh.cpu().double()[:, None]
    assert (planes_value(xa.cpu()) - cen).abs().max() < 2e-6
    assert torch.equal(xa[0].cpu(), cen.float().to(torch.float16))


def _attention_case(L, dh, H, bias, sharp, mode, Ts=(70, 129, 5, 200), ramp=0.0):
    """[q | k | gate | v] layout of the "f16q" mode; q pre-scaled by dh^-0.5 * log2(e).  Returns (max abs error vs fp64, reference)."""
    Ts = list(Ts)
    D, M, Tmax = H * dh, sum(Ts), max(Ts)
    gpad = 8 if bias else 0
    ld = 3 * D + gpad
    g = torch.Generator().manual_seed(dh + H + int(sharp))
    qkv = torch.randn(M, ld, generator=g)
    qkv[:, : 2 * D] *= sharp ** 0.5                       # logits grow with `sharp`: near one-hot softmax rows at 8-16
    if ramp:                                              # scores that climb with the key position (see test_attention_stale_running_maximum)
        u = torch.ones(dh) / dh ** 0.5
        pos = torch.cat([torch.arange(T) / max(T - 1, 1) for T in Ts])
        for h in range(H):
            qkv[:, h * dh:(h + 1) * dh] += 3.0 * u
            qkv[:, D + h * dh: D + (h + 1) * dh] += (ramp * pos)[:, None] * u
    c = dh ** -0.5 * 1.4426950408889634
    pre = qkv.clone()
    pre[:, :D] *= c
    planes = split_h(pre)                                 # both planes everywhere; FP16 reads plane 0 only
    if mode == FP16:
        val = planes[0].cpu().double()
    else:
        val = planes_value(planes.cpu())
        if mode == FP16Q:
            val[:, 2 * D + gpad:] = planes[0].cpu().double()[:, 2 * D + gpad:]  # FP16Q reads v from plane 0 only
    q_all, k_all, v_all = val[:, :D] / c, val[:, D:2 * D], val[:, 2 * D + gpad:]
    table = cst = None
    gate = None
    if bias:
        table = torch.randn(H, 2 * Tmax - 1, generator=g)
        cst = torch.randn(H, generator=g) + 1.0
        pg = val[:, 2 * D: 2 * D + 2 * H].view(M, H, 2)
        gate = torch.sigmoid(pg[..., 0]) * (torch.sigmoid(pg[..., 1]) * cst.double()[None, :] - 1.0) + 2.0
    offs = np.concatenate([[0], np.cumsum(Ts)])
    ref = torch.empty(M, D, dtype=torch.float64)
    for b, T in enumerate(Ts):
        sl = slice(offs[b], offs[b + 1])
        q, k, v = (t[sl].view(T, H, dh).permute(1, 0, 2) for t in (q_all, k_all, v_all))
        tb = table[:, Tmax - 1 - (T - 1): Tmax - 1 + T].double() if bias else None
        o = attention_reference(q, k, v, dh ** -0.5, tb, gate[sl] if bias else None)
        ref[sl] = o.permute(1, 0, 2).reshape(T, D)
    out = torch.zeros(2 if mode == FP16X else 1, M, D, dtype=torch.float16, device=DEV)
    foffs = torch.tensor(offs, dtype=torch.int32, device=DEV)
    td, cd = (table.to(DEV), cst.to(DEV)) if bias else (None, None)
    L.check(L.lib.ser_attention(planes.data_ptr(), ld, M * ld, 0, D, 2 * D + gpad, foffs.data_ptr(), len(Ts), Tmax,
                                td.data_ptr() if bias else None, Tmax if bias else 0, None, out.data_ptr(), D, M * D, H, dh,
                                -1.0, mode, 2 * D, cd.data_ptr() if bias else None, None, None, 0, stream()), "ser_attention")
    torch.cuda.synchronize()
    return (planes_value(out.cpu()) - ref).abs().max().item(), ref


@pytest.mark.parametrize("dh,H,bias", [(64, 2, True), (64, 3, False), (80, 2, False), (120, 2, False), (128, 1, False)])
def test_attention_fp16q(L, dh, H, bias):
    """3-product S = K Q^T from fp16 hi + lo planes of q and k, single-product P V: every head-dim / bias path, ragged batch."""
    err, ref = _attention_case(L, dh, H, bias, 1.0, FP16Q)
    assert err < 3e-3, err                                # P and the output are fp16 (2^-11 relative); |v| up to ~4


@pytest.mark.parametrize("dh,H,bias", [(64, 2, True), (64, 3, False), (80, 2, False), (120, 2, False), (128, 1, False)])
def test_attention_fp16x(L, dh, H, bias):
    """everything split ("f16a"): S, P V on 3 products, the context rows written as fp16 hi + lo planes -- fp32-grade"""
    err, ref = _attention_case(L, dh, H, bias, 1.0, FP16X)
    assert err < 2e-5, err


@pytest.mark.parametrize("mode,bound", [(FP16Q, 3e-3), (FP16X, 2e-5)])
@pytest.mark.parametrize("dh,bias", [(64, True), (128, False)])
def test_attention_two_plane_stale_running_maximum(L, mode, bound, dh, bias):
    """the stale-maximum branch of the bias-table launches (csrc/attention.hip, LAZY) and the exact-maximum rescale of the others on the
    two-plane forms: 11 key tiles of climbing scores"""
    err, ref = _attention_case(L, dh, 2, bias, 1.0, mode, Ts=(700, 65, 1, 130), ramp=100.0)
    assert err < bound, err


def test_gemm_fp16x_writes_one_plane(L):
    """output projection of "f16a": 3-product GEMM on two-plane operands, one-plane FP16 copy for FC1 (ser_gemm_args.out_mode)"""
    M, N, K = 300, 200, 192
    g = torch.Generator().manual_seed(8)
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    out, oact = run_gemm(L, split_h(A), split_h(W), M, N, K, FP16X, out_mode=FP16)
    ref = A.double() @ W.double().T
    assert (out.cpu().double() - ref).abs().max() < 2e-6
    assert oact.dtype == torch.float16 and oact.shape[0] == 1 and torch.equal(oact[0].cpu(), out.cpu().to(torch.float16))


def test_attention_fp16x_keeps_sharp_logits(L):
    """Near one-hot softmax rows (q.k scaled x16): here single-product rounding of q and k -- each 2^-12 relative, logits of
    +-100 -- moves softmax weights by percents.  Both kernels get IDENTICAL inputs (the reference is built from what each
    reads), so the difference is purely S = K Q^T in 3 products against 1... which is exact either way for fp16 inputs:
    the point of this test is that the FP16X reference keeps the lo planes, i.e. the kernel really consumes them."""
    e4, ref4 = _attention_case(L, 64, 2, True, 16.0, FP16Q)
    e3, ref3 = _attention_case(L, 64, 2, True, 16.0, FP16)
    moved = (ref4 - ref3).abs().max().item()              # what dropping the lo planes of q, k does to the OUTPUT
    print(f"sharp attention: FP16X err {e4:.2e}, FP16 err {e3:.2e}, lo planes move the result by {moved:.2e}")
    assert e4 < 3e-3 and e3 < 3e-3, (e4, e3)
    assert moved > 5 * e4, (moved, e4)                    # the lo planes matter at this sharpness, and FP16X follows them
