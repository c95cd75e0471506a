"""Fixed cost vs main-loop slope of ser_gemm per tile config (GPU box): K sweep at the bench's group shapes."""
import ctypes as C
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import _lib as L
DEV = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
def timeit(g, n=10):
    for _ in range(3): L.check(L.lib.ser_gemm(C.byref(g), st))
    ts = []
    for r in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): L.check(L.lib.ser_gemm(C.byref(g), st))
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return np.median(ts)
M = int(os.environ.get("M", 3992))
for name, N, cfg, res, f32, act, gelu in (("fc1", 4096, 3, 0, 0, 1, 1), ("fc1-nogelu", 4096, 3, 0, 0, 1, 0), ("fc1", 4096, 1, 0, 0, 1, 1),
                                           ("qkv", 3104, 3, 0, 0, 1, 0), ("out/fc2", 1024, 1, 1, 1, 1, 0), ("out/fc2-f32only", 1024, 1, 1, 1, 0, 0)):
    row = []
    for K in (64, 256, 512, 1024, 2048, 4096):
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
        row.append((K, timeit(g)))
    (k0, t0), (k1, t1) = row[3], row[5]
    slope = (t1 - t0) / (k1 - k0)            # us per K element
    print(f"{name:16s} N={N} cfg{cfg} " + " ".join(f"K{k}:{t:6.1f}" for k, t in row) +
          f" | slope {2.0*M*N/slope/1e6:7.1f} TF/s  intercept(K=1024 fit) {t0 - slope*k0:5.1f} us", flush=True)
