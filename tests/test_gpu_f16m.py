"""SER_MODE_FP16M ("f16m" numerics mode, round 5): fp16 main product + block-scaled e4m3 cross terms on v_mfma_scale_f32_16x16x128_f8f6f4.
Kernel-level parity through the C ABI (-m gpu): the packing kernels against the host restatement (tests/f16m_ref.py) bit for bit, the three
product paths of ser_gemm each on data that makes them EXACT, the mixed product against the float64 statement of what the planes hold, the
mode's accuracy against exact arithmetic, the FP16M output copy of the epilogues, and ser_row_center.
Reference arithmetic: fp32 Linear layers, HF modeling_wavlm.py:133-136,288-294 behind preprocessing/preprocess_speech.py:50,66."""
import ctypes as C

import numpy as np
import pytest
import torch

import f16m_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FP16, FP16X, FP16M = 3, 4, 6


@pytest.fixture(scope="module")
def L():
    from interspeech_ser_amd import _lib
    assert torch.cuda.is_available()
    return _lib


def stream():
    return torch.cuda.current_stream().cuda_stream


def device_pack(L, x: torch.Tensor, weight: bool, flag=None):
    rows, cols = x.shape
    xd = x.to(DEV).contiguous()
    out = torch.zeros((2, rows, cols), dtype=torch.float16, device=DEV)
    sc = torch.zeros((cols // 64, rows), dtype=torch.int32, device=DEV)
    L.check(L.lib.ser_pack_f16m(xd.data_ptr(), cols, rows, cols, out.data_ptr(), cols, rows * cols, sc.data_ptr(), rows, int(weight),
                                None if flag is None else flag.data_ptr(), stream()), "ser_pack_f16m")
    torch.cuda.synchronize()
    return out, sc


def planes_of(out: torch.Tensor, sc: torch.Tensor):
    """device tensors -> (hi fp16 [R, C], x8 uint8 [R, 2C], scales int32 [tiles, R]) on the CPU"""
    o = out.cpu()
    return o[0], o[1].contiguous().view(torch.uint8).reshape(o.shape[1], 2 * o.shape[2]), sc.cpu()


def gemm_m(L, A, As, W, Ws, M, N, K, *, mode=FP16M, want_act=False, out_mode=0, tile_cfg=0, bias=None, act=0, flag=None, a_planes=2):
    g = L.GemmArgs()
    g.A, g.a_plane_stride, g.lda = A.data_ptr(), A.shape[1] * A.shape[2], A.shape[2]
    g.W, g.w_plane_stride = W.data_ptr(), W.shape[1] * W.shape[2]
    g.M, g.N, g.K, g.groups, g.mode, g.tile_cfg, g.act = M, N, K, 1, mode, tile_cfg, act
    if As is not None:
        g.a_scale, g.a_scale_ld = As.data_ptr(), As.shape[1]
    if Ws is not None:
        g.w_scale, g.w_scale_ld = Ws.data_ptr(), Ws.shape[1]
    g.bias = None if bias is None else bias.data_ptr()
    out = torch.full((M, N), float("nan"), device=DEV)
    g.out_f32, g.ldo_f32 = out.data_ptr(), N
    oa = osc = None
    if want_act:
        om = out_mode or mode
        oa = torch.zeros((2, M, N), dtype=torch.float16, device=DEV)
        g.out_act, g.ldo_act, g.out_plane_stride, g.out_mode = oa.data_ptr(), N, M * N, out_mode
        if om == FP16M:
            osc = torch.zeros((N // 64, M), dtype=torch.int32, device=DEV)
            g.out_scale, g.out_scale_ld = osc.data_ptr(), M
    if flag is not None:
        g.range_flag = flag.data_ptr()
    L.check(L.lib.ser_gemm(C.byref(g), stream()), "ser_gemm")
    torch.cuda.synchronize()
    return out, oa, osc


def rand_mat(rows, cols, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(rows, cols, generator=g) * scale


@pytest.mark.parametrize("weight", [False, True])
@pytest.mark.parametrize("rows,cols", [(5, 64), (130, 1024), (77, 4096), (300, 192)])
def test_pack_matches_host_restatement(L, weight, rows, cols):
    """ser_pack_f16m = tests/f16m_ref.pack bit for bit: fp16 hi plane, e4m3 bytes in their [P | Q] tile layout, E8M0 scale words."""
    x = rand_mat(rows, cols, rows + cols, 3.0)
    x[0, :32] *= 1e-3                       # a block of small values (its own scale), a block of zeros, one large outlier
    x[min(1, rows - 1), 32:64] = 0.0
    x[rows - 1, cols - 1] = 6.0e4
    out, sc = device_pack(L, x, weight)
    hi, x8, s = planes_of(out, sc)
    ref = R.pack(x, weight)
    assert torch.equal(hi.view(torch.int16), ref["hi"].view(torch.int16))
    assert torch.equal(s, ref["scales"])
    assert torch.equal(x8, ref["x8"])
    # ... and the planes decode to the values the format promises: hi + P = x to e4m3's 4 significant bits of the residual, Q = x to 4 bits
    h, P, Q = R.decode(hi, x8, s)
    big, res = (P, Q) if weight else (Q, P)
    xd = x.double()
    blockmax = xd.abs().reshape(rows, cols // 32, 32).amax(-1, keepdim=True).expand(-1, -1, 32).reshape(rows, cols)
    assert float(((big - xd).abs() / blockmax.clamp(min=1e-30)).max()) <= 2.0 ** -4      # half an ulp at the top binade + scale slack
    assert float(((h + res - xd).abs() / blockmax.clamp(min=1e-30)).max()) <= 2.0 ** -14


@pytest.mark.parametrize("cfg", [0, 1, 2, 3])
@pytest.mark.parametrize("M,N,K", [(300, 200, 192), (515, 392, 1024), (129, 128, 64), (1030, 264, 640)])
def test_gemm_f16m_three_paths_exact(L, cfg, M, N, K):
    """Each of the three products of the mode on data that makes it exact (any layout / swizzle / scale-pairing slip is a wrong number):
    (1) integers x integers: only hi x hi contributes; (2) values below fp16's smallest subnormal x integers: the fp16 plane of A is zero and
    the result is P_a P_w = x_lo w_8 alone; (3) integers x such tiny values: Q_a Q_w = x_8 w_lo alone.  W asymmetric on purpose."""
    g = torch.Generator().manual_seed(M * 7 + N + cfg)
    ints_a = torch.randint(-7, 8, (M, K), generator=g).float()
    ints_w = torch.randint(-7, 8, (N, K), generator=g).float() + (torch.arange(N)[:, None] % 3).float()
    # per-row / per-column magnitudes differ by powers of two, so the block scales differ between rows and between K tiles
    tiny_a = ints_a * 2.0 ** -30 * (2.0 ** -(torch.arange(M)[:, None] % 5)).float() * (2.0 ** -(torch.arange(K)[None, :] // 64 % 3)).float()
    tiny_w = ints_w * 2.0 ** -31 * (2.0 ** -(torch.arange(N)[:, None] % 4)).float()
    for A, W in ((ints_a, ints_w), (tiny_a, ints_w), (ints_a, tiny_w)):
        Ad, As = device_pack(L, A, False)
        Wd, Ws = device_pack(L, W, True)
        out, _, _ = gemm_m(L, Ad, As, Wd, Ws, M, N, K, tile_cfg=cfg)
        ref = A.double() @ W.double().T
        assert torch.equal(out.cpu().double(), ref), float((out.cpu().double() - ref).abs().max())


@pytest.mark.parametrize("cfg", [1, 2, 3])
def test_gemm_f16m_mixed_product_and_accuracy(L, cfg):
    """Gaussian operands: (a) the kernel equals the float64 statement of what its planes hold -- hi hi + P P + Q Q -- to fp32 accumulation
    noise; (b) against EXACT arithmetic the mode sits between a single fp16 product (2.9e-4 rms) and the fp16 hi + lo split (5e-7):
    measured 1.0e-5 rms of the result's rms (oracle/numerics_whatif_f16m.py)."""
    M, N, K = 520, 384, 1024
    A, W = rand_mat(M, K, 11), rand_mat(N, K, 12, 0.05)
    Ad, As = device_pack(L, A, False)
    Wd, Ws = device_pack(L, W, True)
    out, _, _ = gemm_m(L, Ad, As, Wd, Ws, M, N, K, tile_cfg=cfg)
    got = out.cpu().double()
    stated = R.product(R.decode(*planes_of(Ad, As)), R.decode(*planes_of(Wd, Ws)))
    exact = A.double() @ W.double().T
    rms = float(exact.pow(2).mean().sqrt())
    assert float((got - stated).abs().max()) <= 3e-6 * float(stated.abs().max())
    err = float((got - exact).pow(2).mean().sqrt()) / rms
    single = float((A.half().double() @ W.half().double().T - exact).pow(2).mean().sqrt()) / rms
    print(f"f16m rms error {err:.2e} of rms (single fp16 product {single:.2e})")
    assert err < 2.5e-5 and err < single / 10


def test_gemm_f16m_writes_f16m_and_f16x_copies(L):
    """The epilogue's FP16M output copy (hi plane, [P | Q] bytes gathered to 16-byte stores by the 4 x 4 lane transpose, scale words) equals the
    host restatement of packing the fp32 output; with out_mode = FP16X it writes fp16 hi + lo planes (the packed projection of "f16m")."""
    M, N, K = 300, 256, 192
    A, W = rand_mat(M, K, 21), rand_mat(N, K, 22, 0.3)
    bias = rand_mat(1, N, 23)[0].to(DEV)
    Ad, As = device_pack(L, A, False)
    Wd, Ws = device_pack(L, W, True)
    for cfg in (1, 2, 3):
        out, oa, osc = gemm_m(L, Ad, As, Wd, Ws, M, N, K, want_act=True, bias=bias, tile_cfg=cfg)
        ref = R.pack(out.cpu(), False)
        hi, x8, s = planes_of(oa, osc)
        assert torch.equal(hi.view(torch.int16), ref["hi"].view(torch.int16))
        assert torch.equal(s, ref["scales"]), cfg
        assert torch.equal(x8, ref["x8"]), cfg
        out2, oa2, _ = gemm_m(L, Ad, As, Wd, Ws, M, N, K, want_act=True, out_mode=FP16X, bias=bias, tile_cfg=cfg)
        assert torch.equal(out2, out)
        v = out.cpu()
        h = v.half()
        assert torch.equal(oa2[0].cpu().view(torch.int16), h.view(torch.int16))
        assert torch.equal(oa2[1].cpu().view(torch.int16), (v - h.float()).half().view(torch.int16))


def test_gemm_fp16x_writes_f16m_copy(L):
    """Output projection of "f16m": a 3-product FP16X launch (A = the attention kernel's fp16 hi + lo context rows) whose out_act is FP16M."""
    M, N, K = 520, 256, 128
    A, W = rand_mat(M, K, 31), rand_mat(N, K, 32, 0.2)

    def x2(t):
        h = t.half()
        return torch.stack([h, (t - h.float()).half()]).contiguous().to(DEV)
    out, oa, osc = gemm_m(L, x2(A), None, x2(W), None, M, N, K, mode=FP16X, want_act=True, out_mode=FP16M)
    ref = R.pack(out.cpu(), False)
    hi, x8, s = planes_of(oa, osc)
    assert torch.equal(hi.view(torch.int16), ref["hi"].view(torch.int16)) and torch.equal(s, ref["scales"]) and torch.equal(x8, ref["x8"])
    assert float((out.cpu().double() - A.double() @ W.double().T).abs().max()) < 2e-5


def test_row_center_f16m(L):
    """ser_row_center's FP16M copy of hidden_states[0] (layer 0's packed projection reads it): packing of x - row mean."""
    rows, D = 131, 1024
    x = rand_mat(rows, D, 41, 2.0) + 7.0
    xd = x.to(DEV)
    oa = torch.zeros((2, rows, D), dtype=torch.float16, device=DEV)
    osc = torch.zeros((D // 64, rows), dtype=torch.int32, device=DEV)
    stats = torch.zeros((rows, 2, 2), device=DEV)
    shift = torch.zeros(rows, device=DEV)
    a = L.RowCenterArgs()
    a.x, a.ldx, a.out_act, a.ldo_act, a.out_plane_stride = xd.data_ptr(), D, oa.data_ptr(), D, rows * D
    a.stats, a.shift, a.stat_groups, a.mode, a.rows, a.D = stats.data_ptr(), shift.data_ptr(), 2, FP16M, rows, D
    a.out_scale, a.out_scale_ld = osc.data_ptr(), rows
    L.check(L.lib.ser_row_center_v(C.byref(a), stream()), "ser_row_center")
    torch.cuda.synchronize()
    centred = (xd - shift[:, None]).cpu()            # the kernel's own mean: the copy must be the packing of exactly these values
    assert float((shift.cpu() - x.mean(1)).abs().max()) < 1e-5
    ref = R.pack(centred, False)
    hi, x8, s = planes_of(oa, osc)
    assert torch.equal(hi.view(torch.int16), ref["hi"].view(torch.int16)) and torch.equal(s, ref["scales"]) and torch.equal(x8, ref["x8"])


def test_gemm_f16m_rejects_what_it_does_not_take(L):
    A, W = rand_mat(64, 64, 1), rand_mat(64, 64, 2)
    Ad, As = device_pack(L, A, False)
    Wd, Ws = device_pack(L, W, True)
    with pytest.raises(L.SerHipError):
        gemm_m(L, Ad, None, Wd, Ws, 64, 64, 64)                     # no block scales


@pytest.mark.parametrize("bias", [False, True])
def test_attention_writes_f16m_context_rows(L, bias):
    """ser_attention_v with out_mode = SER_MODE_FP16M (ABI 14; mode FP16X, head dim 64): plane 0 of the context rows is bit for bit the fp16 hi
    plane of the FP16X launch, plane 1 + the scale words decode (tests/f16m_ref.py) to P = v - hi and Q = v within e4m3's half ulp of each
    32-column block's largest element, for ragged utterances incl. one shorter than a query block, with and without the WavLM bias table."""
    import f16m_ref as R
    H, dh = 4, 64
    D, Ts = H * dh, [70, 129, 5, 200]
    M, Tmax = sum(Ts), max(Ts)
    g = torch.Generator().manual_seed(5 + bias)
    qkv = torch.randn(M, 3 * D, generator=g)
    qkv[:, 2 * D:] *= torch.logspace(-3, 2, D)[None, :]                    # value columns over five decades: the block scales matter
    hi = qkv.to(torch.float16)
    qa = torch.stack([hi, (qkv - hi.float()).to(torch.float16)]).contiguous().to(DEV)
    offs = np.concatenate([[0], np.cumsum(Ts)])
    foffs = torch.tensor(offs, dtype=torch.int32, device=DEV)
    table = torch.randn(H, 2 * Tmax - 1, generator=g).to(DEV) if bias else None
    gate = (torch.rand(M, H, generator=g) * 2).to(DEV) if bias else None

    def run(out_m):
        out = torch.zeros(2, M, D, dtype=torch.float16, device=DEV)
        sc = torch.zeros(D // 64, M, dtype=torch.int32, device=DEV)
        a = L.AttentionArgs()
        a.qkv, a.ld, a.plane_stride = qa.data_ptr(), 3 * D, M * 3 * D
        a.q_col, a.k_col, a.v_col, a.B = 0, D, 2 * D, len(Ts)
        a.frame_offs, a.max_frames = foffs.data_ptr(), Tmax
        if bias:
            a.table, a.table_T, a.gate = table.data_ptr(), Tmax, gate.data_ptr()
        a.out, a.ldo, a.out_plane_stride = out.data_ptr(), D, M * D
        a.H, a.dh, a.scale, a.mode = H, dh, -1.0, 4                        # FP16X takes a pre-scaled q
        if out_m:
            a.out_mode, a.out_scale, a.out_scale_ld = FP16M, sc.data_ptr(), M
        rc = L.lib.ser_attention_v(C.byref(a), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return rc, out, sc

    rc, ox, _ = run(False)
    assert rc == 0
    rc, om, sc = run(True)
    assert rc == 0, L.lib.ser_last_error()
    assert torch.equal(om[0], ox[0])                                       # the fp16 copy
    v = ox[0].double().cpu() + ox[1].double().cpu()                        # the context rows to ~2^-22
    x8 = om[1].contiguous().view(torch.uint8).reshape(M, 2 * D).cpu()
    hi64, P, Q = R.decode(om[0].cpu(), x8, sc.cpu().view(torch.int32))
    lo = v - hi64
    bl = lambda t: t.abs().reshape(M, D // 32, 32).amax(-1, keepdim=True).expand(-1, -1, 32).reshape(M, D)   # noqa: E731
    assert ((P - lo).abs() <= bl(lo) * 2.0 ** -3 + 1e-12).all()            # e4m3: 3 mantissa bits, power-of-two scale up to 2x the block max
    assert ((Q - v).abs() <= bl(v) * 2.0 ** -3 + 1e-12).all()
    assert float((Q - v).abs().max() / v.abs().max()) > 1e-4               # ... and it IS an 8-bit copy
    # rejected: a head dim that is not one 64-column tile
    a = L.AttentionArgs()
    a.qkv, a.frame_offs, a.out, a.B, a.H, a.dh, a.max_frames, a.mode, a.out_mode, a.ld, a.ldo = qa.data_ptr(), foffs.data_ptr(), om.data_ptr(), 1, 4, 32, 8, 4, FP16M, 3 * D, D
    a.scale, a.out_scale, a.out_scale_ld = -1.0, sc.data_ptr(), M
    assert L.lib.ser_attention_v(C.byref(a), None) < 0
