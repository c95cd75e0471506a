"""Per-K-tile time of the ser_gemm main loop for each tile config: (t(K=8192)-t(K=4096))/64."""
import ctypes as C
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import _lib as L
DEV = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
def timeit(g, n=10):
    for _ in range(2): L.check(L.lib.ser_gemm(C.byref(g), st))
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): L.check(L.lib.ser_gemm(C.byref(g), st))
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return np.median(ts)
M = 7984
for N in (1024, 4096):
    for cfg in (1, 2, 3):
        t = {}
        for K in (4096, 8192):
            A = torch.randn(1, M, K, device=DEV).to(torch.bfloat16)
            W = (torch.randn(1, N, K, device=DEV) / K ** 0.5).to(torch.bfloat16)
            oa = torch.empty(1, M, N, dtype=torch.bfloat16, device=DEV)
            g = L.GemmArgs()
            g.A, g.a_plane_stride, g.lda = A.data_ptr(), M * K, K
            g.W, g.w_plane_stride = W.data_ptr(), N * K
            g.M, g.N, g.K, g.groups, g.mode = M, N, K, 1, 1
            g.out_act, g.ldo_act, g.out_plane_stride = oa.data_ptr(), N, M * N
            g.tile_cfg = cfg
            t[K] = timeit(g)
        per = (t[8192] - t[4096]) / 64
        fl = 2.0 * M * N * 64
        print(f"N={N} cfg{cfg}: t4096={t[4096]:7.1f}us t8192={t[8192]:7.1f}us per-ktile={per:6.3f}us  main-loop {fl/per/1e6:7.1f} TF/s", flush=True)
