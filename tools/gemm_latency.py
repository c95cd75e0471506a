"""Serial latency of ONE ser_gemm tile (GPU): 1-tile launches at K = 64 isolate launch + prologue + epilogue chain."""
import ctypes as C
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import _lib as L
DEV = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
def timeit(g, n=20):
    for _ in range(3): L.check(L.lib.ser_gemm(C.byref(g), st))
    ts = []
    for r in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): L.check(L.lib.ser_gemm(C.byref(g), st))
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return np.median(ts)
for M, N, cfg in ((128, 128, 1), (256, 256, 3), (3992, 1024, 1), (3992, 4096, 3), (3992, 4096, 1)):
    for K in (64, 128, 256):
        row = []
        for name, res, f32, act, gelu in (("bf16", 0, 0, 1, 0), ("f32", 0, 1, 0, 0), ("f32+res", 1, 1, 0, 0), ("f32+bf16+res", 1, 1, 1, 0)):
            A = torch.randn(1, M, K, device=DEV).to(torch.bfloat16)
            W = (torch.randn(1, N, K, device=DEV) / K ** 0.5).to(torch.bfloat16)
            bias = torch.randn(N, device=DEV); R = torch.randn(M, N, device=DEV)
            of = torch.empty(M, N, device=DEV); oa = torch.empty(1, M, N, dtype=torch.bfloat16, device=DEV)
            g = L.GemmArgs()
            g.A, g.a_plane_stride, g.lda = A.data_ptr(), M * K, K
            g.W, g.w_plane_stride = W.data_ptr(), N * K
            g.M, g.N, g.K, g.groups, g.mode = M, N, K, 1, 1
            g.bias, g.act = bias.data_ptr(), gelu
            if res: g.residual, g.ldr = R.data_ptr(), N
            if f32: g.out_f32, g.ldo_f32 = of.data_ptr(), N
            if act: g.out_act, g.ldo_act, g.out_plane_stride = oa.data_ptr(), N, M * N
            g.tile_cfg = cfg
            row.append(f"{name}:{timeit(g):6.1f}")
        print(f"M={M:5d} N={N:5d} cfg{cfg} K={K:4d}  " + "  ".join(row), flush=True)
