"""Time ser_gemm on the WavLM-large shapes under every block-tile configuration (GPU box).
Interleaved rounds in one process (cdna guide rule 24); prints median TF/s per (shape, cfg)."""
import ctypes as C
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import _lib as L

DEV = "cuda:0"
SHAPES = [  # name, M, N, K, gelu, residual, act_out
    ("qkv", 3992, 3104, 1024, 0, 0, 1), ("out", 3992, 1024, 1024, 0, 1, 0), ("fc1", 3992, 4096, 1024, 1, 0, 1),
    ("fc2", 3992, 1024, 4096, 0, 1, 0), ("qkv16", 7984, 3104, 1024, 0, 0, 1), ("fc1_16", 7984, 4096, 1024, 1, 0, 1),
    ("out16", 7984, 1024, 1024, 0, 1, 0), ("fc2_16", 7984, 1024, 4096, 0, 1, 0),
]
CFGS = tuple(int(c) for c in os.environ.get("SWEEP_CFGS", "1,2,3").split(","))
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
planes = 2 if mode == 2 else 1
rounds = 7
for name, M, N, K, gelu, res, act_out in SHAPES:
    A = torch.randn(planes, M, K, device=DEV).to(torch.bfloat16)
    W = (torch.randn(planes, N, K, device=DEV) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV)
    R = torch.randn(M, N, device=DEV)
    of = torch.empty(M, N, device=DEV)
    oa = torch.empty(planes, M, N, dtype=torch.bfloat16, device=DEV)
    res_ms = {}
    for cfg in CFGS:
        g = L.GemmArgs()
        g.A, g.a_plane_stride, g.lda = A.data_ptr(), M * K, K
        g.W, g.w_plane_stride = W.data_ptr(), N * K
        g.M, g.N, g.K, g.groups, g.mode = M, N, K, 1, mode
        g.bias, g.act = bias.data_ptr(), gelu
        if res:
            g.residual, g.ldr = R.data_ptr(), N
        if act_out:
            g.out_act, g.ldo_act, g.out_plane_stride = oa.data_ptr(), N, M * N
        else:
            g.out_f32, g.ldo_f32 = of.data_ptr(), N
        g.tile_cfg = cfg
        res_ms[cfg] = (g, [])
    st = torch.cuda.current_stream().cuda_stream
    for r in range(rounds + 1):
        for cfg, (g, ts) in res_ms.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                L.check(L.lib.ser_gemm(C.byref(g), st))
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts.append(e0.elapsed_time(e1) / 5)
    fl = 2.0 * M * N * K * (3 if mode == 2 else 1)
    print(name, M, N, K, " ".join(f"cfg{c}: {np.median(ts)*1e3:7.1f}us {fl/np.median(ts)/1e9:7.1f}TF(mfma)" for c, (g, ts) in res_ms.items()), flush=True)
