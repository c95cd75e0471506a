"""Calibration (GPU box): what the vendor's fused attention (torch SDPA -> its flash / memory-efficient back ends on ROCm) takes on the
shapes ser_attention runs, bf16, one launch at a time, kernel names from the profiler beside it.  Not used by the product: it separates
"the shape" (8 key tiles per block, launch + prologue) from "the kernel".
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/vendor_attention_calibration.py"""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import _lib as L
DEV = "cuda:0"; st = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return float(np.median(ts))


for (B, T, H, dh, bias) in ((16, 499, 16, 64, True), (16, 499, 16, 64, False), (8, 499, 16, 80, False), (8, 1500, 20, 64, False), (16, 512, 16, 64, False)):
    D, M = H * dh, B * T
    fl = 4.0 * B * H * T * T * dh
    q, k, v = (torch.randn(B, H, T, dh, device=DEV, dtype=torch.bfloat16) for _ in range(3))
    mask = torch.randn(B, H, T, T, device=DEV, dtype=torch.bfloat16) if bias else None
    rows = []
    for name, ctx in (("flash", "FLASH_ATTENTION"), ("mem-efficient", "EFFICIENT_ATTENTION"), ("math", "MATH")):
        from torch.nn.attention import sdpa_kernel, SDPBackend
        try:
            with sdpa_kernel(getattr(SDPBackend, ctx)):
                us = timed(lambda: F.scaled_dot_product_attention(q, k, v, attn_mask=mask), reps=10 if name == "math" else 20)
            rows.append(f"sdpa {name:14s} {us:8.1f} us ({fl / us / 1e6:6.1f} TF/s)")
        except Exception as e:                                       # back end refuses the case (e.g. flash with a mask)
            rows.append(f"sdpa {name:14s} refused: {str(e).splitlines()[0][:70]}")
    ld = 3 * D + 32
    qkv = torch.randn(1, M, ld, device=DEV).to(torch.bfloat16)
    out = torch.empty(1, M, D, dtype=torch.bfloat16, device=DEV)
    offs = torch.arange(0, M + 1, T, dtype=torch.int32, device=DEV)
    table, cst = torch.randn(H, 2 * T - 1, device=DEV), torch.randn(H, device=DEV)
    us = timed(lambda: L.check(L.lib.ser_attention(qkv.data_ptr(), ld, M * ld, 0, D, 2 * D, offs.data_ptr(), B, T,
                                                   table.data_ptr() if bias else None, T if bias else 0, None, out.data_ptr(), D, M * D, H, dh,
                                                   -1.0, 1, 3 * D, cst.data_ptr() if bias else None, None, None, 0, st)))
    rows.append(f"ser_attention       {us:8.1f} us ({fl / us / 1e6:6.1f} TF/s)" + (" [gated relative-position bias from the table]" if bias else ""))
    print(f"B={B} T={T} H={H} dh={dh} " + ("dense additive mask [B,H,T,T] for sdpa" if bias else "no bias"), flush=True)
    for r in rows:
        print("    " + r, flush=True)
