"""ser_attention per numerics mode at the step's shapes (GPU box): 8 x 499 frames (one utterance group of the headline), 16 heads
of 64, pre-scaled q, WavLM bias through the fused gate columns -- BF16 / FP16 (1 product), FP16Q (3-product S, 1-product PV),
FP16X / FP32X (3 products everywhere); and 8 x 1500 frames without bias (Whisper's shape, 20 heads).
    python tools/attn_modes_bench.py            (SER_HIP_LIB=<other build> for an A/B on the same box)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import _lib as L
DEV = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
NAMES = {1: "BF16", 3: "FP16", 5: "FP16Q", 4: "FP16X", 2: "FP32X"}
for (B, T, H, dh, bias) in ((8, 499, 16, 64, True), (16, 499, 16, 64, True), (8, 1500, 20, 64, False), (8, 499, 16, 80, False), (8, 499, 16, 120, False), (8, 1500, 20, 64, False)):
    D = H * dh
    M = B * T
    ld = 3 * D + 32
    for mode in (1, 3, 5, 4, 2):
        dt = torch.bfloat16 if mode in (1, 2) else torch.float16
        planes = 1 if mode in (1, 3) else 2
        oplanes = 2 if mode in (2, 4) else 1
        qkv = (torch.randn(planes, M, ld, device=DEV) * (1.0 if planes == 1 else 1.0)).to(dt)
        if planes == 2:
            qkv[1] *= 2.0 ** -10
        out = torch.empty(oplanes, M, D, dtype=dt, device=DEV)
        offs = torch.arange(0, M + 1, T, dtype=torch.int32, device=DEV)
        table = torch.randn(H, 2 * T - 1, device=DEV)
        cst = torch.randn(H, device=DEV)

        def run():
            L.check(L.lib.ser_attention(qkv.data_ptr(), ld, M * ld, 0, D, 2 * D, offs.data_ptr(), B, T,
                                        table.data_ptr() if bias else None, T if bias else 0, None, out.data_ptr(), D, M * D, H, dh,
                                        -1.0, mode, 3 * D, cst.data_ptr() if bias else None, None, None, 0, st))
        for _ in range(3):
            run()
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        fl = 4.0 * B * H * T * T * dh
        print(f"B={B:2d} T={T:4d} H={H} dh={dh} bias={int(bias)} {NAMES[mode]:6s}: {np.median(ts):7.1f} us  ({fl / np.median(ts) / 1e6:6.1f} TF/s algorithmic)", flush=True)
